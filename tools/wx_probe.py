"""Winograd-x form of the 3x3 64->64 convolutions (conv_f16x3_wx.hip) against the direct form: one layer on random data (error vs the
exact fp32 kernel, time), the whole nets against the oracle, and the full luma / chroma step.  Run on the GPU box."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pmp_vvc_tip2023_amd import engine, synth, weights as W
from oracle import nets_torch as O

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
