#!/bin/bash
# One-off validation of SURVEY.md 8(f) row N4 ON THE GPU BOX (run through gpurun from the repo root): the reference's patched
# VTM encoder with tools/vtm_build/pmp_hook.cpp predicting the partition maps IN-PROCESS through libpmp_hip.so must produce the
# bitstream the stock patched encoder produces from the text files of the Python driver.  Needs the binaries that
# tools/vtm_build builds in the build container, staged under tools/vtm_build/_bin/ (git-ignored: build products of the
# reference's sources never enter the history), and the encoder cfg next to them; take tools/vtm_build/_bin/ out of .gpurunignore for
# that one call (it is listed there so that the 13 MB of binaries do not travel with every other gpurun).  Output: gpurun_out/n4/summary.txt
set -u
ROOT=$PWD
BIN=$ROOT/tools/vtm_build/_bin
OUT=$ROOT/gpurun_out/n4
WORK=$(mktemp -d)
mkdir -p $OUT $WORK/in $WORK/cfg $WORK/models
cd $WORK
python3 - <<PY
import os, sys
sys.path.insert(0, "$ROOT")
import numpy as np
from pmp_vvc_tip2023_amd import synth, weights as W
Wd, H, F = 384, 192, 3
y, u, v = synth.recipe_r_frames(F, H, Wd, 11)
with open("in/Synth_384x192_30.yuv", "wb") as f:
    for i in range(F):
        f.write(y[i].tobytes()); f.write(u[i].tobytes()); f.write(v[i].tobytes())
open("in/table.txt", "w").write("Synth,Synth_384x192_30.yuv,%d,%d,%d,30\n#end!!!!\n" % (Wd, H, F))
open("cfg/Synth.cfg", "w").write("InputFile : in/Synth_384x192_30.yuv\nInputBitDepth : 8\nFrameRate : 30\nFrameSkip : 0\nSourceWidth : %d\nSourceHeight : %d\nFramesToBeEncoded : %d\nLevel : 4\n" % (Wd, H, F))
qp = 32
for comp in ("Luma", "Chroma"):   # real QT nets; the MTT nets' documented synthetic weights saved under the reference's file names
    wq, _ = W.load_net_weights(comp + "_Q", qp)
    W.save_pmpw("models/%s_Q_%d.pmpw" % (comp, qp), comp + "_Q", qp, wq, "weights/")
    W.save_pmpw("models/%s_BD_%d.pmpw" % (comp, qp), comp + "_MSBD", qp, synth.synth_msbd_weights(comp, qp), "synthetic(seed=%d)" % qp)
PY
# 1. the product's driver -> text files (the reference's hand-over)
PYTHONPATH=$ROOT python3 -m pmp_vvc_tip2023_amd.inference_qbd --jobID j --inputDir $WORK --outDir $WORK/out --seqTable in/table.txt --cfgDir $WORK/cfg \
    --modelDir $WORK/models --ssRatio 1 --seqNum 1 --qps 32 > $OUT/driver.log 2>&1 || { echo "driver failed" > $OUT/summary.txt; tail -5 $OUT/driver.log; exit 1; }
mkdir -p PartitionMat && cp out/j/PartitionMat/*.txt PartitionMat/
ENC="-c cfg/Synth.cfg -c $BIN/encoder_intra_vtm.cfg -f 3 -ts 1 -q 32 --SEIDecodedPictureHash=1"
( time $BIN/EncoderApp $ENC -b text.bin -o text.yuv ) > $OUT/enc_text.log 2>&1; RC1=$?
# 2. no files at all: the hook predicts in-process
mv PartitionMat PartitionMat_text
( time PMP_MODEL_DIR=$WORK/models $BIN/EncoderAppHook $ENC -b hook.bin -o hook.yuv ) > $OUT/enc_hook.log 2>&1; RC2=$?
$BIN/DecoderApp -b hook.bin -o dec.yuv > $OUT/dec.log 2>&1; RC3=$?
{
  echo "N4 in-process hook on $(python3 -c 'import torch;print(torch.cuda.get_device_name(0))' 2>/dev/null)"
  echo "encoder (text files from the driver): rc=$RC1   encoder (in-process hook): rc=$RC2   decoder: rc=$RC3"
  grep "pmp_hook" $OUT/enc_hook.log
  if cmp -s text.bin hook.bin; then echo "bitstreams IDENTICAL ($(stat -c %s text.bin) bytes)"; else echo "bitstreams DIFFER"; fi
  grep -c "(OK)" $OUT/dec.log | sed 's/^/decoded pictures with matching MD5: /'
  grep "Total Time" $OUT/enc_text.log $OUT/enc_hook.log
} > $OUT/summary.txt
cat $OUT/summary.txt
