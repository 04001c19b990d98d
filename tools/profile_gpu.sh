#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <tag>   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extras"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
# PMC passes (counters only; never combined with tracing).  SQ block has 8 slots, TCC 4 (FETCH_SIZE 3, WRITE_SIZE 2).
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -50
ls -la $OUT
# memory-path counters of every conv kernel in the step (the 5x5 + shortcut instantiation gets the treatment the 3x3 kernel got)
for set in "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 5 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_mem_$name -- $BENCH > $OUT/pmc_mem_$name.log 2>&1
done
