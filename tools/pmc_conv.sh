#!/bin/bash
# Run ON THE GPU BOX (each pass under its own timeout: a counter set the hardware cannot take makes rocprofv3 abort and then hang):
# PMC passes (counters only) over the f16x3 3x3 64->64 conv microbench -> gpurun_out/pmc_conv/
set -u
OUT=$PWD/gpurun_out/pmc_conv
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/tools/conv_x6_bench.py h2 256,64,64,64,64,3"
cd /tmp
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_h2_kernel" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc): print("%-40s %16.1f  (%d launches)" % (k, acc[k][0] / acc[k][1], acc[k][1]))
PY
