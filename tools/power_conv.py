"""Socket power while ONE convolution layer runs in a loop (run on the GPU box): direct 3x3 64->64, its Winograd-x form, direct 5x5.
Usage: python tools/power_conv.py            (parent: samples rocm-smi)      python tools/power_conv.py child <k> <winograd>"""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import abl_lib  # tools/abl_lib.py: the measurement library lives in tools/abl/
    from pmp_vvc_tip2023_amd import _lib, engine
    _lib.load(abl_lib.ensure())          # the Winograd-x form lives in the measurement library
    k, wino = int(sys.argv[2]), int(sys.argv[3])
    eng = engine.Engine(0, allow_synthetic_mtt=True)
    eng.set_precision("f16x3")
    eng.lib.pmp_debug_set_winograd(eng.h, wino)
    a, b, d, r = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    for rnd in range(3):
        eng._ck(eng.lib.pmp_debug_conv_bench(eng.h, 1024, 64, 64, 64, 64, k, 1500 if k == 3 else 600, C.byref(a), C.byref(b), C.byref(d), C.byref(r)))
        print("k%d winograd %d: %.3f ms per launch" % (k, wino, b.value), flush=True)
    sys.exit(0)


def sample():
    try:
        d = json.loads(subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout)
        d = d.get("card0", d)
        return float(d["Current Socket Graphics Package Power (W)"]), int(d["sclk clock speed:"].strip("()Mhz"))
    except Exception:
        return None


for k, wino in ((3, 0), (3, 1), (5, 0)):
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(k), str(wino)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    rows = []
    while child.poll() is None:
        s = sample()
        if s:
            rows.append(s)
        time.sleep(0.1)
    busy = sorted(r for r in rows if r[0] > 600)
    print(child.stdout.read().strip())
    if busy:
        p = sorted(r[0] for r in busy); c = sorted(r[1] for r in busy)
        print("   under load (%d samples): power median %.0f W (p10 %.0f, p90 %.0f) of 1400; sclk median %d MHz (p10 %d, p90 %d)"
              % (len(busy), p[len(p) // 2], p[len(p) // 10], p[len(p) * 9 // 10], c[len(c) // 2], c[len(c) // 10], c[len(c) * 9 // 10]), flush=True)
