"""Winograd-x form of the 3x3 64->64 convolutions (conv_f16x3_wx.hip, measurement library: make -C tools/abl) against the direct form: one layer
on random data (error vs the exact fp32 kernel, time), the whole nets against the oracle WITH the parity assertions a product kernel
has to pass (logits within 1e-3, split flags bit-exact on the device logits, range guard repairs a saturating net through it), and the
full luma / chroma step.  Run on the GPU box; exits non-zero if a check fails."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abl_lib  # tools/abl_lib.py: the measurement library lives in tools/abl/
from pmp_vvc_tip2023_amd import _lib, engine, synth, weights as W
from oracle import nets_torch as O, postproc as P
_lib.load(abl_lib.ensure())

eng = engine.Engine(0, allow_synthetic_mtt=True)
eng.set_precision("f16x3")
lib = eng.lib
for shape in ((256, 64, 64, 64, 64, 3), (1024, 64, 64, 64, 64, 3), (1024, 32, 32, 64, 64, 3), (512, 16, 16, 64, 64, 3), (64, 48, 32, 64, 64, 3)):
    n, h, w, ci, co, k = shape
    fl = 2.0 * n * h * w * co * ci * k * k
    for on in (0, 1, 0, 1):
        assert lib.pmp_debug_set_winograd(eng.h, on) == 0
        a, b, d, r = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        eng._ck(lib.pmp_debug_conv_bench(eng.h, n, h, w, ci, co, k, 10, C.byref(a), C.byref(b), C.byref(d), C.byref(r)))
        print("n%d %dx%d: %s %.3f ms (%.0f TF)  max|diff vs fp32| %.2e (max|ref| %.1f)" % (n, h, w, "winograd-x" if on else "direct    ", b.value, fl / b.value / 1e9, d.value, r.value), flush=True)
y, u, v = synth.recipe_r_blocks(16, 1)
for comp in ("Luma", "Chroma"):
    luma = comp == "Luma"
    wq, _ = W.load_net_weights(comp + "_Q", 22)
    wb, _ = W.load_net_weights(comp + "_MSBD", 22, allow_synthetic=True)
    x = O.luma_input(y) if luma else O.chroma_input(y, u, v)
    oq, obt, od = O.infer_qbd(wq, wb, x, luma)
    for on in (0, 1):
        lib.pmp_debug_set_winograd(eng.h, on)
        qt, bt, dire = eng.inference_pre_QBD(comp, 22, y, u, v)
        print("%s %s: max |logit - oracle| qt %.2e bt %.2e dire %.2e  saturated %s" % (comp, "winograd-x" if on else "direct    ", np.abs(qt - oq).max(), np.abs(bt - obt).max(),
                                                                             np.abs(dire - od).max(), eng.saturated()), flush=True)
# ---- the parity checks of a product kernel (they were a GPU test while the form was in the product library)
P.build()
TOL = 1e-3
y40, u40, v40 = synth.recipe_r_blocks(40, 321)
for comp in ("Luma", "Chroma"):
    luma = comp == "Luma"
    wq, _ = W.load_net_weights(comp + "_Q", 27)
    wb, _ = W.load_net_weights(comp + "_MSBD", 27, allow_synthetic=True)
    x = O.luma_input(y40) if luma else O.chroma_input(y40, u40, v40)
    oq, obt, od = O.infer_qbd(wq, wb, x, luma)
    eng._ck(lib.pmp_debug_set_winograd(eng.h, 0))
    direct = eng.infer_postprocess(comp, 27, y40, u40, v40, want_logits=True)
    eng._ck(lib.pmp_debug_set_winograd(eng.h, 1))
    hor, ver, q8, d8, qt, bt, dire = eng.infer_postprocess(comp, 27, y40, u40, v40, want_logits=True)
    err = max(np.abs(qt - oq).max(), np.abs(bt - obt).max(), np.abs(dire - od).max())
    assert err < TOL, "%s logits off by %g in the Winograd form" % (comp, err)
    assert not np.array_equal(bt, direct[5]) and np.abs(bt - direct[5]).max() < 2e-4      # another kernel, the same answer
    oh, ov, oq8, od8 = P.seq_post_process(qt, bt, dire, comp, 1, 64 * 40, 64, None)
    assert np.array_equal(hor, oh) and np.array_equal(ver, ov) and np.array_equal(q8, oq8.astype(np.uint8)) and np.array_equal(d8, od8)
    assert not eng.saturated()
    print("%s QP27, 40 blocks, Winograd form: max |logit - oracle| %.2e, split flags bit-exact on the device logits" % (comp, err), flush=True)
# range guard through the Winograd kernels: trunk activations 2^17 x the usual ones, logits unchanged (tests/test_gpu_parity.py: _range_stress_weights)
K = 2.0 ** 17
ws = dict(synth.synth_msbd_weights("Luma", 22))
for k in ("conv_b1_1", "conv_b1_2", "conv_b1_3"):
    ws[k + ".weight"] = (ws[k + ".weight"] * K).astype(np.float32); ws[k + ".bias"] = (ws[k + ".bias"] * K).astype(np.float32)
for t in ("trunk_B1.0", "trunk_B2.0", "trunk_B3.0"):
    for k in (".left.0.weight", ".shortcut.0.weight"):
        ws[t + k] = (ws[t + k] / K).astype(np.float32)
yb = np.ascontiguousarray(y40[:6])
wq, _ = W.load_net_weights("Luma_Q", 22)
oq, obt, od = O.infer_qbd(wq, ws, O.luma_input(yb), True)
eng.load("Luma", 22)
eng.load_pretrain_model("Luma_MSBD", 22, ws)
qt, bt, dire = eng.inference_pre_QBD("Luma", 22, yb)
assert eng.saturated() and eng.saturation_reruns() == 1
assert max(np.abs(qt - oq).max(), np.abs(bt - obt).max(), np.abs(dire - od).max()) < TOL
eng.clear_saturation()
eng.load_pretrain_model("Luma_MSBD", 22, synth.synth_msbd_weights("Luma", 22))
print("range guard: a saturating net is noticed through the Winograd kernels (NaN-aware flag) and repaired on bf16x6", flush=True)
dev = torch.device("cuda:0")
n = 4096
y, u, v = synth.recipe_r_blocks(n, 1)
d_y, d_u, d_v = (torch.from_numpy(a).to(dev) for a in (y, u, v))
rec = torch.empty((n, 1344), dtype=torch.uint8, device=dev)
for comp in ("Luma", "Chroma"):
    chroma = comp == "Chroma"
    for rnd in range(2):
        for on in (0, 1):
            lib.pmp_debug_set_winograd(eng.h, on)
            def step():
                eng.infer_postprocess_records_device(comp, 22, d_y.data_ptr(), d_u.data_ptr() if chroma else None, d_v.data_ptr() if chroma else None, n, rec.data_ptr())
            step(); eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                step()
            eng.synchronize()
            dt = (time.perf_counter() - t0) / 5
            print("%s step, %d blocks, %s: %.2f ms = %.0f blocks/s" % (comp, n, "winograd-x" if on else "direct    ", dt * 1e3, n / dt), flush=True)
eng.close()
