#!/usr/bin/env python3
"""Weight-stationary persistent 3x3 64->64 convolution (tools/abl/conv_ws.hip, measurement library) against the product's launch on the same
random tensors: bit comparison, interleaved timing, the prototype's timing-only builds.
    make -C tools/abl && python tools/ws_probe.py [blocks] [size] [once]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (loads the HIP runtime first)
import abl_lib  # tools/abl_lib.py: the measurement library lives in tools/abl/
from pmp_vvc_tip2023_amd import _lib


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    once = len(sys.argv) > 3 and sys.argv[3] == "once"       # under rocprofv3: the exact build only
    lib = _lib.open_library(abl_lib.ensure())
    ctx = C.c_void_p()
    assert lib.pmp_create(0, C.byref(ctx)) == 0
    f = lib.pmp_abl_ws_bench
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] + [C.c_int] * 9 + [C.POINTER(C.c_double)] * 2 + [C.POINTER(C.c_int64)] + [C.POINTER(C.c_double)] * 2
    flop = 2.0 * n * size * size * 64 * 64 * 9
    names = {0: "exact build", 1: "no halo DMA after the first tile", 2: "no MFMAs", 4: "no epilogue arithmetic", 8: "no residual loads",
             6: "no MFMAs, no epilogue arithmetic", 7: "DMA-less, MFMA-less, epilogue-less", 12: "no epilogue arithmetic, no residual", 13: "K-loops only (no DMA, no epilogue arithmetic, no residual)"}
    nb = int(os.environ.get("WS_NBUF", "3"))
    cases = [(1, 0, 0, 2), (1, 0, 0, 3), (1, 0, 0, 2), (1, 0, 0, 3), (0, 0, 0, 2), (0, 0, 0, 3)] if not once else [(1, 0, 0, nb)]
    if not once and os.environ.get("WS_ONLY_EXACT", "0") != "1":
        cases += [(1, a, 0, nb) for a in (1, 2, 4, 8, 12, 13, 6, 7)]
    for res, abl, grid, nbuf in cases:
        tp, tf, md, mr, bad = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        rc = f(ctx, n, size, size, res, 10, 3, abl, nbuf, grid, C.byref(tp), C.byref(tf), C.byref(bad), C.byref(md), C.byref(mr))
        assert rc == 0, lib.pmp_last_error(ctx)
        print("%4d blocks %dx%d res %d bufs %d abl %2d %-62s product %.3f ms (%.0f TF)   weight-stationary %.3f ms (%.0f TF)  %+.1f %%   exact build: %d mismatching elements, max |diff| %.3g of %.3g"
              % (n, size, size, res, nbuf, abl, names[abl], tp.value, flop / tp.value / 1e9, tf.value, flop / tf.value / 1e9, (tf.value / tp.value - 1) * 100, bad.value, md.value, mr.value), flush=True)
    lib.pmp_destroy(ctx)


if __name__ == "__main__":
    main()
