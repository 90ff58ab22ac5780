#!/bin/bash
# Round-end measurement on the GPU box: full GPU test suite, the default bench line, the rocprofv3 kernel-trace of the
# same command, and the PMC passes.  Everything lands under gpurun_out/final/.
set -e
out=$PWD/gpurun_out/final; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/pytest_gpu.log 2>&1
tail -2 $out/pytest_gpu.log
timeout -k 10 400 python -u bench.py > $out/bench.json 2> $out/bench.err
echo "bench done"
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/prof -o vm --output-format csv -- python3 $repo/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/prof.err
echo "kernel trace done"
cd $repo
bash tools/pmc.sh final/pmc
