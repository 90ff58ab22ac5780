#!/bin/bash
# PMC passes over bench.py in the driver's batch shape (--steps 20 --warmup 5; 2 repeats, no CPU leg, no one-stream pass):
# one rocprofv3 run per counter group, CSVs under gpurun_out/<tag>/<group>.
# usage: tools/pmc.sh <tag> [RT_LIB path]
set -e
tag=${1:-pmc}; lib=${2:-raytracer_amd/librtamd.so}
export RT_LIB=$PWD/$lib
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -s KILL 170 rocprofv3 --pmc "$@" -d $out/$name -o vm --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-unshared-pass --no-workloads > $out/$name.log 2>&1; echo "pass $name done"; }
run sqa SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sqb SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU
run ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py --bench-line $out/tcc.log $out/sqa $out/sqb $out/ta $out/fetch $out/write $out/tcc > $out/summary.json
