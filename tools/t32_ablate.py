"""Timing-only ablations of the 32x16-tile 3x3 64->64 kernel (conv_f16x3_t32.hip) next to the 16x16 kernel, one layer on random data.
Needs libpmp_hip_abl.so (make -C tools/abl).  Run on the GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abl_lib  # tools/abl_lib.py: the measurement library lives in tools/abl/
from pmp_vvc_tip2023_amd import _lib, engine
_lib.load(abl_lib.ensure())
eng = engine.Engine(0, allow_synthetic_mtt=True)
eng.set_precision("f16x3")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fl = 2.0 * n * 64 * 64 * 64 * 64 * 9
names = {2: "16x16 tiles (default)", 9: "32x16 tiles", 92: "  no weight refills", 94: "  no fragment reads", 98: "  no epilogue"}
for rnd in range(2):
    for v in (2, 9, 92, 94, 98):
        assert eng.lib.pmp_debug_set_conv_variant(v) == 0
        a, b, d, r = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        eng._ck(eng.lib.pmp_debug_conv_bench(eng.h, n, 64, 64, 64, 64, 3, 10, C.byref(a), C.byref(b), C.byref(d), C.byref(r)))
        print("%-28s %.3f ms  %.0f TF   (max|diff| %.2e)" % (names[v], b.value, fl / b.value / 1e9, d.value), flush=True)
