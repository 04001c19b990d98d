"""The A/B forms of the f16x3 Cout = 64 convolution (measurement library libpmp_hip_abl.so, `make -C tools/abl`)
against the shipped form: logits of the whole luma net bit for bit (same K order, same accumulators), and the 3x3 Cout = 64
kernel forms on shapes the nets never launch against the exact fp32 kernel.  Run on the GPU box:

    python tools/variants_agree.py

A regression check of the notebook, not a parity test: the product library ships none of these forms.  Since round 4
the notebook forms of the three convolution kernel files are separate sources (tools/abl/), so the first check is that the measurement
library's DEFAULT form still computes what the product library computes, bit for bit, on every datapath."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import abl_lib  # tools/abl_lib.py: the measurement library lives in tools/abl/
from pmp_vvc_tip2023_amd import _lib, engine

g1 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g1_qt.npz"))
y = np.concatenate([g1["block_y"]] * 8)              # 128 blocks: several tiles per persistent workgroup at 64x64
u = np.concatenate([g1["block_u"]] * 8)
v = np.concatenate([g1["block_v"]] * 8)
bad = 0
product = {}
_lib.load(_lib.LIB_PATH)
eng = engine.Engine(0, allow_synthetic_mtt=True)
for prec in ("f16x3", "bf16x6", "fp32"):
    eng.set_precision(prec)
    for comp in ("Luma", "Chroma"):
        product[(prec, comp)] = eng.inference_pre_QBD(comp, 22, y, u, v)
eng.close()
_lib._lib = None                                     # both builds in one process (ctypes loads them RTLD_LOCAL): forget the first handle
_lib.load(abl_lib.ensure())
eng = engine.Engine(0, allow_synthetic_mtt=True)
assert b"abl" in eng.lib.pmp_version()
for prec in ("f16x3", "bf16x6", "fp32"):
    eng.set_precision(prec)
    for comp in ("Luma", "Chroma"):
        got = eng.inference_pre_QBD(comp, 22, y, u, v)
        same = all(np.array_equal(a, b) for a, b in zip(product[(prec, comp)], got))
        bad += not same
        print("product library vs measurement library, default forms, %s %s: logits %s" % (prec, comp, "bit-identical" if same else "DIFFER"), flush=True)
eng.set_precision("f16x3")
ref = eng.inference_pre_QBD("Luma", 22, y)
for variant in (1, 3, 4, 5, 6, 7, 8, 9):
    assert eng.lib.pmp_debug_set_conv_variant(variant) == 0
    got = eng.inference_pre_QBD("Luma", 22, y)
    eng.lib.pmp_debug_set_conv_variant(2)
    same = all(np.array_equal(a, b) for a, b in zip(ref, got))
    bad += not same
    print("variant %d: logits %s" % (variant, "bit-identical" if same else "DIFFER"), flush=True)
for shape in ((8, 32, 32, 48, 64, 3), (8, 32, 32, 16, 64, 3), (4, 48, 32, 64, 64, 3)):
    n, h, w, ci, co, k = shape
    for variant in (2, 3, 7, 8):
        eng.lib.pmp_debug_set_conv_variant(variant)
        a, b, d, r = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        eng._ck(eng.lib.pmp_debug_conv_bench(eng.h, n, h, w, ci, co, k, 1, C.byref(a), C.byref(b), C.byref(d), C.byref(r)))
        eng.lib.pmp_debug_set_conv_variant(2)
        ok = d.value < 1e-4 * max(1.0, r.value)
        bad += not ok
        print("shape %s variant %d: max |split - fp32| %.2e of %.2e %s" % (shape, variant, d.value, r.value, "ok" if ok else "OFF"), flush=True)
eng.close()
sys.exit(1 if bad else 0)
