#!/bin/bash
# PMC passes over bench.py (default steps, no warm-up, no CPU baseline): one rocprofv3 run per counter group, CSVs under gpurun_out/pmc/<group>.
# usage: tools/pmc.sh <tag> [RT_LIB path]
set -e
tag=${1:-pmc}; lib=${2:-raytracer_amd/librtamd.so}
export RT_LIB=$PWD/$lib
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -s KILL 170 rocprofv3 --pmc "$@" -d $out/$name -o vm --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --warmup 0 > $out/$name.log 2>&1; echo "pass $name done"; }
run sqa SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sqb SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU
run sqc SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVES SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_TCP_LATENCY_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py $out/sqa $out/sqb $out/sqc $out/ta $out/ta2 $out/fetch $out/write $out/tcc > $out/summary.json
