#!/bin/bash
# Copy the judged summaries of tools/final_measure.sh from gpurun_out/final (scratch) into profiles/ (tracked).
set -e
src=gpurun_out/final; dst=profiles; tag=${1:-r03}
cp $src/prof/vm_kernel_stats.csv $dst/${tag}_bench_kernel_stats.csv
cp $src/bench_under_rocprof.json $dst/${tag}_bench_under_rocprof.json
cp $src/bench.json $dst/${tag}_bench.json
cp $src/pmc/summary.json $dst/${tag}_pmc.json
rm -rf $dst/${tag}_pmc; mkdir -p $dst/${tag}_pmc
for g in sqa sqb ta fetch write tcc; do cp $src/pmc/$g/vm_counter_collection.csv $dst/${tag}_pmc/pmc_$g.csv; done
[ -f $src/pytest_gpu.log ] && tail -2 $src/pytest_gpu.log > $dst/${tag}_pytest_gpu.txt
echo collected
