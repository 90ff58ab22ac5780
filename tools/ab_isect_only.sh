#!/bin/bash
# the traversal microbench per library build
for lib in "$@"; do
  echo "== $lib"
  RT_LIB=$PWD/$lib timeout -k 10 300 python -u tools/bench_intersect.py 2>&1 | grep -v amdgpu.ids | grep -v "^scene"
done
