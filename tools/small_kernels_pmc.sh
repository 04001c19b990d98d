#!/bin/bash
# Run ON THE GPU BOX: PMC passes (counters only) over a short bench.py run, summarised for the rbfuse32 kernels -> gpurun_out/pmc_rbfuse32/   (rbfuse32, chain16, stems, post-processing: the kernels next to the big convolutions)
set -u
OUT=$PWD/gpurun_out/pmc_small
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $PWD/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extras"
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done
cd - > /dev/null
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("rbfuse32", "branch16", "att16", "qt_rest16", "q3_rb64", "stem_mfma", "postprocess_kernel", "conv_h2_kernel<3, 3, 2")): continue
        k = k.replace("pmp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]): print("    %-36s %16.1f  (%d launches)" % (c, acc[k][c][0] / acc[k][c][1], acc[k][c][1]))
PY
