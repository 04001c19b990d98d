#!/usr/bin/env python3
"""Layer-pipelined persistent trunk kernel (tools/abl/conv_f16x3.hip: trunk_pipe_kernel, measurement library) against one launch per layer:
bit comparison of every layer's output, interleaved timing.
    make -C tools/abl && python tools/trunk_probe.py [blocks] [size] [layers] [delay,delay,...] [grid]"""
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
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    delays = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [6]
    grid = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    iters = int(os.environ.get("TRUNK_ITERS", "3"))
    lib = _lib.open_library(abl_lib.ensure())
    ctx = C.c_void_p()
    assert lib.pmp_create(0, C.byref(ctx)) == 0
    f = lib.pmp_abl_trunk_bench
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] + [C.c_int] * 9 + [C.POINTER(C.c_double)] * 2 + [C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    flop = 2.0 * n * size * size * 64 * 64 * 9 * L
    syncs = [int(v) for v in os.environ.get("TRUNK_SYNC", "0").split(",")]
    for d, sync in [(d, sy) for d in delays for sy in syncs]:
        tl, tp, bad, hit = C.c_double(), C.c_double(), C.c_int64(), C.c_int()
        rc = f(ctx, n, size, size, L, d, iters, 3, grid, sync, C.byref(tl), C.byref(tp), C.byref(bad), C.byref(hit))
        assert rc == 0, lib.pmp_last_error(ctx)
        print("%4d blocks %dx%d, %d layers, delay %2d blocks, sync switches %d, grid %s: %d launches %.3f ms (%.0f TF)   pipelined kernel %.3f ms (%.0f TF)  %+.1f %%   mismatching: %d   spin limit hit: %d"
              % (n, size, size, L, d, sync, grid or "3 per CU", L, tl.value, flop / tl.value / 1e9, tp.value, flop / tp.value / 1e9, (tp.value / tl.value - 1) * 100, bad.value, hit.value), flush=True)
    lib.pmp_destroy(ctx)


if __name__ == "__main__":
    main()
