"""End-to-end driver timing on a synthetic sequence (run on the GPU box): YUV file -> PartitionMat files for Luma+Chroma x 4 QPs,
i.e. the whole-job rate including file I/O, H2D, D2H, text formatting and file writes (bench.py measures the device-resident hot
path only).  Prints the main thread's wall time per stage (inference_qbd.Stages), the pure-GPU time of the same passes for
comparison, and the N-rank critical path those stages project to with sharded emission.

Usage: python tools/driver_bench.py [W H FRAMES] [--emit sharded|gather] [--seqs K] [--abOverlap 1] [--noManifestExps 1]      default 1920 1080 8"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pmp_vvc_tip2023_amd import inference_qbd as D, synth

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = {sys.argv[i]: sys.argv[i + 1] for i in range(1, len(sys.argv) - 1) if sys.argv[i].startswith("--")}
W, H, F = (int(v) for v in pos[:3]) if len(pos) >= 3 else (1920, 1080, 8)
emit_mode, nseq = opt.get("--emit", "sharded"), int(opt.get("--seqs", "1"))
tmp = tempfile.mkdtemp(prefix="pmp_drv_")
inp, out, cfg = (os.path.join(tmp, d) for d in ("in", "out", "cfg"))
for d in (inp, cfg):
    os.makedirs(d)
with open(os.path.join(inp, "table.txt"), "w") as tf:
    for k in range(nseq):
        name, fn = "Synth%d" % k, "Synth%d_%dx%d_30.yuv" % (k, W, H)
        y, u, v = synth.recipe_r_frames(F, H, W, 3 + k)
        with open(os.path.join(inp, fn), "wb") as f:
            for i in range(F):
                f.write(y[i].tobytes()); f.write(u[i].tobytes()); f.write(v[i].tobytes())
        tf.write("%s,%s,%d,%d,%d,30\n" % (name, fn, W, H, F))
        open(os.path.join(cfg, name + ".cfg"), "w").write("InputFile : %s\nInputBitDepth : 8\n" % fn)
    tf.write("#end!!!!\n")
# a CTU_Models-style directory as a production run has it: the shipped QT nets and - standing in for the trained *_BD_* files the
# reference checkout lacks - the documented synthetic MTT nets, all as .pmpw files, so that loading goes through the library's reader
import shutil
from pmp_vvc_tip2023_amd import weights as Wt
models = os.path.join(tmp, "CTU_Models")
os.makedirs(models)
for comp in ("Luma", "Chroma"):
    for qp in (22, 27, 32, 37):
        shutil.copy(os.path.join(Wt.default_weight_dir(), "%s_Q_%d.pmpw" % (comp, qp)), models)
        Wt.save_pmpw(os.path.join(models, "%s_BD_%d.pmpw" % (comp, qp)), comp + "_MSBD", qp, synth.synth_msbd_weights(comp, qp), source="synthetic(seed=%d)" % qp)
# ... and, as a converted model directory has them (tools/calibrate_pmpw.py), the MTT files carry their activation-scale exponents, so
# that loading them runs no calibration pass (--noManifestExps 1: leave them out, every job calibrates its eight nets while loading)
if opt.get("--noManifestExps") != "1":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import calibrate_pmpw
    calibrate_pmpw.calibrate_dir(models, 0, log=lambda *a: None)
args = ["--jobID", "b", "--inputDir", inp, "--outDir", out, "--seqTable", "table.txt", "--cfgDir", cfg, "--ssRatio", "1",
        "--startSeqID", "0", "--seqNum", str(nseq), "--modelDir", models, "--emit", emit_mode]
D.main(args)            # warm-up (weights, workspace, first-touch, page cache)
t0 = time.time()
D.main(args)
dt = time.time() - t0
st = D.LAST_STAGES
blocks = (W // 64) * (H // 64) * F * nseq
size = sum(os.path.getsize(os.path.join(out, "b", "PartitionMat", f)) for f in os.listdir(os.path.join(out, "b", "PartitionMat")))

# the same passes with nothing else going on: what the GPU alone needs (device-resident blocks, no emission)
import torch
from pmp_vvc_tip2023_amd import engine
eng = engine.Engine(0, allow_synthetic_mtt=True)
n = (W // 64) * (H // 64) * F
by, bu, bv = synth.recipe_r_blocks(min(n, 4096), 5)
reps = -(-n // by.shape[0])
d_y = torch.from_numpy(np.concatenate([by] * reps)[:n]).cuda(); d_u = torch.from_numpy(np.concatenate([bu] * reps)[:n]).cuda()
d_v = torch.from_numpy(np.concatenate([bv] * reps)[:n]).cuda()
rec = torch.empty((n, 1344), dtype=torch.uint8, device="cuda")
t_gpu = 0.0
for comp in ("Luma", "Chroma"):
    for qp in (22, 27, 32, 37):
        eng.load(comp, qp)
        pu, pv = (d_u.data_ptr(), d_v.data_ptr()) if comp == "Chroma" else (None, None)
        eng.infer_postprocess_records_device(comp, qp, d_y.data_ptr(), pu, pv, n, rec.data_ptr()); eng.synchronize()
        t1 = time.perf_counter()
        eng.infer_postprocess_records_device(comp, qp, d_y.data_ptr(), pu, pv, n, rec.data_ptr()); eng.synchronize()
        t_gpu += time.perf_counter() - t1
t_gpu *= nseq
eng.close()

print("driver end to end (%s emission): %dx%d, %d frames x %d sequence(s) = %d blocks x 8 (component, QP) passes: %.3f s total; "
      "%.0f pass-blocks/s, %.2f frames/s for all 8 files, %.1f MB of text" % (emit_mode, W, H, F, nseq, blocks, dt, blocks * 8 / dt, F * nseq / dt, size / 1e6))
print("  the same passes alone on the GPU: %.3f s  ->  the job runs at %.0f %% of its own kernels" % (t_gpu, 100 * t_gpu / dt))
print("  main-thread stages (s): " + "  ".join("%s %.3f" % (k, st.t[k]) for k in st.NAMES if st.t[k] > 5e-4))
print("  reader thread file I/O %.3f s (hidden behind the passes except for read_wait)" % st.prefetch_read)
acc = sum(st.t.values())
print("  accounted %.3f s of %.3f s" % (acc, dt))
serial = {k: st.t[k] for k in ("setup", "weights", "read_wait", "h2d_cut", "first_enqueue", "d2h", "emit_start", "emit_finish", "gather", "drain", "teardown")}
host = sum(serial.values())
print("  host-side serial share: %.3f s = %.1f %% of the job; per pass %.1f ms = %.2f of one GPU pass (%.1f ms)"
      % (host, 100 * host / dt, host / st.passes * 1e3, host / max(t_gpu, 1e-9), t_gpu / st.passes * 1e3))
# N ranks, sharded emission: every rank does 1/N of the rows (passes, copies, formatting, writes) but ALL of the per-job work: context,
# the first pass's weights (exposed), the other passes' weights and the launches (hidden behind its GPU passes while those last longer)
w_first = st.t["weights"] / max(st.passes, 1)
exposed = st.t["setup"] + st.t["teardown"] + st.t["first_enqueue"] + w_first + st.passes * 0.5e-3      # ~0.5 ms per size all-reduce
hidden = st.t["weights"] - w_first + st.t["enqueue"]
rows = sum(st.t[k] for k in ("read_wait", "h2d_cut", "gpu_wait", "d2h", "emit_start", "emit_finish", "drain"))
for N in (2, 4, 8):
    if emit_mode == "sharded":
        crit = exposed + max(rows / N, hidden)
    else:                          # rank 0 formats and writes everything: only the passes shrink
        crit = st.t["gpu_wait"] / N + (dt - st.t["gpu_wait"])
    print("  projected critical path at %d ranks: %.3f s  (x%.2f)%s" % (N, crit, dt / crit, "  [weight loading is the longer leg]" if emit_mode == "sharded" and hidden > rows / N else ""))

# --abOverlap 1: --overlap against the driver's default (off since round 6), alternating, same process and box
if opt.get("--abOverlap") == "1":
    res = {"on": [], "off": []}
    for rep in range(4):
        for tag, extra in (("on", ["--overlap"]), ("off", [])):
            t1 = time.time()
            D.main(args + extra)
            res[tag].append(time.time() - t1)
    on, off = min(res["on"]), min(res["off"])
    print("  overlap A/B, whole job (best of 4, alternating): on %.3f s, off %.3f s -> %+.2f %%  (all runs on: %s | off: %s)"
          % (on, off, 100 * (on - off) / off, " ".join("%.3f" % t for t in res["on"]), " ".join("%.3f" % t for t in res["off"])))
