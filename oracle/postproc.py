"""ORACLE (test infrastructure): ctypes front-end of oracle/pmp_oracle.c — the CPU restatement of
eli_structual_error (Metrics.py:630-637), Map_to_Partition (Map2Partition.py:98-373), the block cutter
(Inference_QBD.py:104-149) and the PartitionMat writer (Map2Partition.py:375-417).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, "libpmp_oracle.so")
        if not os.path.isfile(p):
            build()
        _LIB = C.CDLL(p)
        _LIB.pmp_oracle_write_partition_file.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def eli_structural_error(qt):
    """qt float32 [N,1,8,8] or [N,8,8] -> same shape, post-fix depths (ints as float)."""
    q = np.ascontiguousarray(qt, np.float32)
    out = np.empty_like(q)
    lib().pmp_oracle_eli_structural_error(_p(q), C.c_long(q.size // 64), _p(out))
    return out


def map_to_partition(qt, bt, dire, chroma_factor):
    """qt [N,8,8] post-fix, bt/dire [N,3,16,16] raw logits -> hor,ver u8 [N,16,16], dire i8 [N,3,16,16], leaves i64 [N]."""
    qt = np.ascontiguousarray(qt, np.float32).reshape(-1, 8, 8)
    n = qt.shape[0]
    bt = np.ascontiguousarray(bt, np.float32).reshape(n, 3, 16, 16)
    dire = np.ascontiguousarray(dire, np.float32).reshape(n, 3, 16, 16)
    hor = np.zeros((n, 16, 16), np.uint8); ver = np.zeros((n, 16, 16), np.uint8)
    dout = np.zeros((n, 3, 16, 16), np.int8); leaves = np.zeros(n, np.int64)
    lib().pmp_oracle_map_to_partition_batch(_p(qt), _p(bt), _p(dire), C.c_long(n), C.c_int(chroma_factor),
                                            _p(hor), _p(ver), _p(dout), _p(leaves))
    return hor, ver, dout, leaves


def cut_blocks(y, u, v, bitdepth=8):
    """y [F,H,W], u,v [F,H/2,W/2] (u8, or u16 for 10-bit) -> block_y [N,68,68], block_u/v [N,34,34] u8."""
    dt = np.uint8 if bitdepth == 8 else np.uint16
    y = np.ascontiguousarray(y, dt); u = np.ascontiguousarray(u, dt); v = np.ascontiguousarray(v, dt)
    F, H, W = y.shape
    n = F * (H // 64) * (W // 64)
    by = np.zeros((n, 68, 68), np.uint8); bu = np.zeros((n, 34, 34), np.uint8); bv = np.zeros((n, 34, 34), np.uint8)
    lib().pmp_oracle_cut_blocks(_p(y), _p(u), _p(v), F, H, W, bitdepth, _p(by), _p(bu), _p(bv))
    return by, bu, bv


def write_partition_file(path, frames, H, W, hor, ver, qt, dire):
    hor = np.ascontiguousarray(hor, np.uint8); ver = np.ascontiguousarray(ver, np.uint8)
    qt = np.ascontiguousarray(qt, np.float32); dire = np.ascontiguousarray(dire, np.int8)
    rc = lib().pmp_oracle_write_partition_file(path.encode(), frames, H, W, _p(hor), _p(ver), _p(qt), _p(dire))
    if rc != 0:
        raise OSError("pmp_oracle_write_partition_file failed: %d" % rc)


def seq_post_process(qt, bt, dire, comp, frames, W, H, path):
    """seq_post_process (Metrics.py:764-774): eli fix -> per-block search -> file."""
    q = eli_structural_error(qt).reshape(-1, 8, 8)
    hor, ver, dout, _ = map_to_partition(q, bt, dire, 1 if comp == "Luma" else 2)
    if path is not None:
        write_partition_file(path, frames, H, W, hor, ver, q, dout)
    return hor, ver, q, dout
