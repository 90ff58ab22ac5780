#!/bin/bash
# first look at a new walk: structure + parity tests, then the traversal microbench and the frame bench (+ A/B libraries given as arguments)
out=$PWD/gpurun_out/first; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_bvh_build_gpu.py tests/test_intersect_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1
rc=$?; tail -15 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -u tools/bench_intersect.py > $out/isect.log 2>&1 || { tail -5 $out/isect.log; exit 1; }
grep -v amdgpu.ids $out/isect.log
bash tools/ab.sh raytracer_amd/librtamd.so "$@"
