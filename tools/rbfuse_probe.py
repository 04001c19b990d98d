#!/usr/bin/env python3
"""Fused ResidualBlock(64, 64, 3) prototype (tools/abl/rbfuse_proto.hip, measurement library) against the product's launch pair on the
same random tensors: bit comparison, then interleaved timing, then the prototype's timing-only builds.
    make -C tools/abl && python tools/rbfuse_probe.py [blocks] [size]"""
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
    lib = _lib.open_library(abl_lib.ensure())
    ctx = C.c_void_p()
    assert lib.pmp_create(0, C.byref(ctx)) == 0
    f = lib.pmp_abl_rbfuse_bench
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 2 + [C.POINTER(C.c_int64)] + [C.POINTER(C.c_double)] * 2
    flop = 2 * 2.0 * n * size * size * 64 * 64 * 9
    names = {0: "exact build", 1: "no input staging after group 0", 2: "no phase-1 MFMAs", 4: "no phase-2 MFMAs", 6: "no MFMAs at all", 8: "no final epilogue",
             9: "no staging, no final epilogue", 16: "no intermediate write", 25: "no staging, no epilogue, no intermediate write (MFMAs + LDS reads + weights)",
             32: "every tile reads block 0 (L2-resident input)"}
    once = len(sys.argv) > 3 and sys.argv[3] == "once"       # under rocprofv3: the exact build only
    for abl in ((0,) if once else (0, 0, 1, 2, 4, 6, 8, 9, 16, 25, 32)):
        tp, tf, md, mr, bad = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        rc = f(ctx, n, size, size, 10, 3, abl, C.byref(tp), C.byref(tf), C.byref(bad), C.byref(md), C.byref(mr))
        assert rc == 0, lib.pmp_last_error(ctx)
        print("%4d blocks %dx%d  abl %2d %-78s launch pair %.3f ms (%.0f TF)   fused %.3f ms (%.0f TF algorithmic)  %+.1f %%   exact build: %d mismatching elements, max |diff| %.3g of %.3g"
              % (n, size, size, abl, names[abl], tp.value, flop / tp.value / 1e9, tf.value, flop / tf.value / 1e9, (tf.value / tp.value - 1) * 100, bad.value, md.value, mr.value), flush=True)
    lib.pmp_destroy(ctx)


if __name__ == "__main__":
    main()
