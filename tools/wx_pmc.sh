#!/bin/bash
# PMC passes over the Winograd-x ablation builds (run ON THE GPU BOX): LDS conflicts, MFMA busy, waits
set -u
OUT=$PWD/gpurun_out/pmc_wx
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/tools/wx_ablate.py 1024"
cd /tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_h2" not in k: continue
        k = k[k.index("conv_h2"):][:60]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]): print("    %-32s %16.1f  (%d launches)" % (c, acc[k][c][0] / acc[k][c][1], acc[k][c][1]))
PY
