"""End-to-end driver timing on a synthetic sequence (run on the GPU box): YUV file -> PartitionMat files for Luma+Chroma x 4 QPs.
Reports wall time per stage from the driver's own Time_Sta log plus the total, i.e. the whole-job rate including file I/O, H2D/D2H
and text emission (bench.py measures the device-resident hot path only).
Usage: python tools/driver_bench.py [W H FRAMES]      default 1920 1080 8"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pmp_vvc_tip2023_amd import inference_qbd as D, synth

W, H, F = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 8)
tmp = tempfile.mkdtemp(prefix="pmp_drv_")
inp, out, cfg = (os.path.join(tmp, d) for d in ("in", "out", "cfg"))
for d in (inp, cfg):
    os.makedirs(d)
name, fn = "Synth", "Synth_%dx%d_30.yuv" % (W, H)
y, u, v = synth.recipe_r_frames(F, H, W, 3)
with open(os.path.join(inp, fn), "wb") as f:
    for i in range(F):
        f.write(y[i].tobytes()); f.write(u[i].tobytes()); f.write(v[i].tobytes())
open(os.path.join(inp, "table.txt"), "w").write("%s,%s,%d,%d,%d,30\n#end!!!!\n" % (name, fn, W, H, F))
open(os.path.join(cfg, name + ".cfg"), "w").write("InputFile : %s\nInputBitDepth : 8\n" % fn)
args = ["--jobID", "b", "--inputDir", inp, "--outDir", out, "--seqTable", "table.txt", "--cfgDir", cfg, "--ssRatio", "1",
        "--startSeqID", "0", "--seqNum", "1", "--allowSyntheticMTT"]
D.main(args)            # warm-up (weights, workspace, first-touch)
t0 = time.time()
D.main(args)
dt = time.time() - t0
blocks = (W // 64) * (H // 64) * F
rows = [[float(x) for x in r.rstrip(",").split(",")] for r in open(os.path.join(out, "b", "Time_Sta_0_1.txt")).read().strip().split("\n")]
blk = rows[0][0]; net = sum(r[1] + r[2] for r in rows); post = sum(r[3] + r[4] for r in rows)
size = sum(os.path.getsize(os.path.join(out, "b", "PartitionMat", f)) for f in os.listdir(os.path.join(out, "b", "PartitionMat")))
print("driver end to end: %dx%d, %d frames = %d blocks x 8 (component, QP) passes: %.2f s total  (read+upload+cut %.2f s, "
      "inference+post-processing %.2f s, gather+emission hand-off %.2f s); %.0f blocks/s per pass-block, %.1f frames/s for all 8 "
      "files, %.1f MB of text" % (W, H, F, blocks, dt, blk, net, post, blocks * 8 / dt, F / dt, size / 1e6))
