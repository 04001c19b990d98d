#!/bin/bash
# Run ON THE GPU BOX (via gpurun): PMC passes of the weight-stationary convolution prototype next to the product launch (tools/ws_probe.py).
set -u
OUT=$PWD/gpurun_out/prof_ws
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/tools/ws_probe.py 1024 64 once"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_mem_tcp -- $CMD > $OUT/pmc_mem_tcp.log 2>&1
cd $OLDPWD
python3 tools/summarize_profile.py gpurun_out/prof_ws gpurun_out/ws_pmc > /dev/null 2>&1
cat gpurun_out/ws_pmc.txt
