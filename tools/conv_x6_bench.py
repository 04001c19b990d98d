"""One conv layer, fp32-MFMA kernel vs a split kernel (bf16x6, or f16x3 with `h2` as first argument), random data (run on the GPU box).
The ablate* / stamps modes need libpmp_hip_abl.so (make -C tools/abl); the product library has no timing-only kernels."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abl_lib  # tools/abl_lib.py: the measurement library lives in tools/abl/
from pmp_vvc_tip2023_amd import _lib, engine
# the timing-only (wrong-result) kernel builds live in the measurement library only: make -C tools/abl
_lib.load(abl_lib.ensure() if any(a in ("ablate", "ablate5", "ablate_h2", "stamps") for a in sys.argv[1:3]) else None)
eng = engine.Engine(0, allow_synthetic_mtt=True)
split = "bf16x6"
if len(sys.argv) > 1 and sys.argv[1] == "h2":
    split = "f16x3"; del sys.argv[1]
eng.set_precision(split)
shapes = [(256, 64, 64, 64, 64, 3), (256, 64, 64, 64, 64, 5), (256, 64, 64, 32, 64, 5), (512, 32, 32, 64, 64, 3), (512, 16, 16, 64, 32, 3), (512, 16, 16, 32, 16, 3)]
abl = [0]
if len(sys.argv) > 1 and sys.argv[1] == "ablate":
    shapes = [(256, 64, 64, 64, 64, 3)]
    if split == "f16x3": sys.argv[1] = "ablate_h2"
    abl = [0, 1, 2, 4, 8, 16, 32, 200, 128]      # bits: 1 staging, 2 weight loads, 4 x reads, 8 epilogue, 16 barrier, 32 staging from L2-resident addresses, 200 = LDS stores only, 128 = stamps
if len(sys.argv) > 1 and sys.argv[1] == "ablate5":
    shapes = [(256, 64, 64, 64, 64, 5), (256, 64, 64, 32, 64, 5)]
    abl = [0, 32, 0, 32, 1, 2, 4, 8, 9, 15]
elif len(sys.argv) > 1 and sys.argv[1] == "ab":   # same-box A/B of real (not ablated) builds: ab <bits> [shape]
    bits = int(sys.argv[2]); abl = [0, bits] * 4
    shapes = [tuple(int(v) for v in sys.argv[3].split(","))] if len(sys.argv) > 3 else [(1024, 64, 64, 64, 64, 3)]
elif len(sys.argv) > 1 and sys.argv[1] == "stamps":   # only the stamp build (with PMP_STAMP_DUMP: raw stamps for tools/stamp_overlap.py)
    shapes = [(256, 64, 64, 64, 64, 3)]; abl = [128 + (int(sys.argv[2]) if len(sys.argv) > 2 else 0)]   # + 1 no halo staging, 2 no weight refills, 4 no fragment reads (builds 0, 1, 2, 3, 7)
elif len(sys.argv) > 1 and sys.argv[1] == "ablate_h2":
    abl = [0, 32, 64, 1, 2, 4, 8, 16, 9, 15, 128]   # 128: in-kernel stamps   # 64: all blocks read and write block 0's addresses (L2-resident working set: what HBM costs); 16: epilogue without its stores (and a quarter of the conversions); 32: all halo requests with the first K-step (a real variant, not an ablation)
elif len(sys.argv) > 1 and sys.argv[1] not in ("ablate", "ablate5", "stamps", "ab"):
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for ab in abl:
  eng.lib.pmp_debug_set_conv_variant(10 + ab if ab else 2)
  for (n, h, w, ci, co, k) in shapes:
    if ab: print("ablation bits %d" % ab, end=": ")
    a, b, d, r = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    eng._ck(eng.lib.pmp_debug_conv_bench(eng.h, n, h, w, ci, co, k, 10, C.byref(a), C.byref(b), C.byref(d), C.byref(r)))
    fl = 2.0 * n * h * w * co * ci * k * k
    print(("n%d %dx%d %d->%d k%d: fp32 %.3f ms (%.0f TF)  " + split + " %.3f ms (%.0f TF)  speedup %.2fx  max|diff| %.2e (max|ref| %.1f)") % (
        n, h, w, ci, co, k, a.value, fl / a.value / 1e9, b.value, fl / b.value / 1e9, a.value / b.value, d.value, r.value), flush=True)
