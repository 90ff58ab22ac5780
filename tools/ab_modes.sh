#!/bin/bash
# the other integrators per library build
for lib in "$@"; do
  echo "== $lib"
  RT_LIB=$PWD/$lib timeout -k 10 400 python -u tools/bench_modes.py 2>&1 | grep -v amdgpu.ids
done
