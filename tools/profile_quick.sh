#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel-trace stats of a short bench.py run, summarised; with PMC=1 in the environment also
# two SQ counter passes (counters only, never combined with tracing).   tools/profile_quick.sh <tag> [bench args]
set -u
TAG=${1:-q}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extras $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
if [ "${PMC:-0}" = "1" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
fi
cd - > /dev/null
python3 tools/summarize_profile.py $OUT $OUT/summary | head -45
