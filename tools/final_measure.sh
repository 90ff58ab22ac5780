#!/bin/bash
# Round-end measurement on the GPU box, one coherent set of artefacts from ONE tree: the bench line in the driver's shape, the
# rocprofv3 kernel-trace of the same command, and the PMC passes of the same batch shape.  Everything lands under
# gpurun_out/final/; tools/collect_profiles.sh copies the judged summaries into profiles/.
# usage: tools/final_measure.sh [skip-tests]
set -e
out=$PWD/gpurun_out/final; mkdir -p $out
if [ "$1" != "skip-tests" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest_gpu.log 2>&1
  tail -2 $out/pytest_gpu.log
fi
repo=$PWD
bash tools/pmc.sh final/pmc
# the PMC summary must be in place before the bench lines are printed: bench.py reads its ceilings from it
cp $out/pmc/summary.json profiles/r03_pmc.json
timeout -k 10 400 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/prof -o vm --output-format csv -- python3 $repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-workloads > $out/bench_under_rocprof.json 2> $out/prof.err
echo "kernel trace done"
