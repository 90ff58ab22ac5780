#!/bin/bash
# A/B of library builds on the traversal microbench + the frame bench
for lib in "$@"; do
  echo "== $lib"
  RT_LIB=$PWD/$lib timeout -k 10 300 python -u tools/bench_intersect.py 2>&1 | grep -v amdgpu.ids | sed 's/device_bytes.*collapse_nodes/.. collapse_nodes/'
done
bash tools/ab.sh "$@"
