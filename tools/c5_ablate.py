"""Timing-only ablations of the direct 5x5 64->64 kernel (conv_f16x3.hip, two-workgroup form) beside the shipped three-workgroup form -
needs libpmp_hip_abl.so (make -C tools/abl).  Run on the GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abl_lib  # tools/abl_lib.py: the measurement library lives in tools/abl/
from pmp_vvc_tip2023_amd import _lib, engine
_lib.load(abl_lib.ensure())
eng = engine.Engine(0, allow_synthetic_mtt=True)
eng.set_precision("f16x3")
names = {2: "shipped form (three workgroups per CU)", 3: "two workgroups per CU", 11: "no halo requests", 12: "no weight requests", 14: "LDS fragment reads in the first K-step only",
         18: "no epilogue", 19: "no halo requests, no epilogue", 25: "MFMAs only (1+2+4+8)", 42: "halo requests all at the group's first K-step"}
for (n, hw, cin) in ((1024, 32, 64), (256, 64, 64), (256, 64, 32)):
    fl = 2.0 * n * hw * hw * cin * 64 * 25
    for rnd in range(2):
        for v in (2, 3, 11, 12, 14, 18, 19, 25, 42):
            eng.lib.pmp_debug_set_conv_variant(v)
            a, b, d, r = C.c_double(), C.c_double(), C.c_double(), C.c_double()
            eng._ck(eng.lib.pmp_debug_conv_bench(eng.h, n, hw, hw, cin, 64, 5, 10, C.byref(a), C.byref(b), C.byref(d), C.byref(r)))
            print("n%d %dx%d %d->64 5x5  %-46s %.3f ms  %.0f TF" % (n, hw, hw, cin, names[v], b.value, fl / b.value / 1e9), flush=True)
eng.lib.pmp_debug_set_conv_variant(2)
