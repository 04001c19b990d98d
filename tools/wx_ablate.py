"""Timing-only ablations of the Winograd-x kernel (conv_f16x3_wx.hip) - needs libpmp_hip_abl.so (make -C tools/abl).  Run on the GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abl_lib  # tools/abl_lib.py: the measurement library lives in tools/abl/
from pmp_vvc_tip2023_amd import _lib, engine
_lib.load(abl_lib.ensure())
eng = engine.Engine(0, allow_synthetic_mtt=True)
eng.set_precision("f16x3")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fl = 2.0 * n * 64 * 64 * 64 * 64 * 9
names = {0: "full kernel", 1: "no staging after the prologue", 2: "weights loaded once", 4: "no epilogue", 7: "1+2+4", 12: "no staging at all, no epilogue",
         14: "no staging at all, weights once, no epilogue", 30: "... and no barriers: MFMAs + LDS reads", 32: "full, second workgroup of a CU delayed", 16: "full, no barriers", 64: "no B path (rows 16, 17)", 80: "no B path, no barriers",
         84: "no B path, no barriers, no epilogue", 128: "full, with in-kernel stamps"}
for rnd in range(2):
    eng.lib.pmp_debug_set_winograd(eng.h, 0); eng.lib.pmp_debug_set_conv_variant(2)
    a, b, d, r = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    eng._ck(eng.lib.pmp_debug_conv_bench(eng.h, n, 64, 64, 64, 64, 3, 10, C.byref(a), C.byref(b), C.byref(d), C.byref(r)))
    print("direct (3 workgroups per CU)        %.3f ms  %.0f TF" % (b.value, fl / b.value / 1e9), flush=True)
    eng.lib.pmp_debug_set_winograd(eng.h, 1)
    for v in (0, 16, 64, 80, 84, 4, 12, 30, 128):
        eng.lib.pmp_debug_set_conv_variant(200 + v if v else 2)
        eng._ck(eng.lib.pmp_debug_conv_bench(eng.h, n, 64, 64, 64, 64, 3, 10, C.byref(a), C.byref(b), C.byref(d), C.byref(r)))
        print("winograd-x %-32s %.3f ms  %.0f TF (algorithmic)" % (names[v], b.value, fl / b.value / 1e9), flush=True)
