"""Run by tests/test_hostasan_cpu.py in a child process with the ASan runtime preloaded: drives every host-only entry point
of the C ABI (text / binary emission, frame tiling, f16x3 weight packing) in the sanitizer build libpmp_hostasan.so
(make -C pmp_vvc_tip2023_amd/csrc hostasan).  Any AddressSanitizer / UBSan report aborts the process (non-zero exit)."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pmp_vvc_tip2023_amd import _lib  # noqa: E402

_lib._lib = _lib.open_library(_lib.HOSTASAN_LIB_PATH, subset=True)   # before engine's helpers call _lib.load()
from pmp_vvc_tip2023_amd import engine  # noqa: E402

lib = _lib.load()
assert b"hostasan" in lib.pmp_version()
GOLD = os.path.join(ROOT, "tests", "golden")


def blocks_from_text(path, F, H, W):
    ph, pv, pq, pd = engine.read_partition_file(path, F, H, W)
    bh, bw = H // 64, W // 64

    def tb(m, s):
        return m.reshape(F, bh, s, bw, s).transpose(0, 1, 3, 2, 4).reshape(F * bh * bw, s, s)
    d8 = np.stack([tb(pd[:, k], 16) for k in range(3)], 1).astype(np.int8)
    return tb(ph, 16).astype(np.uint8), tb(pv, 16).astype(np.uint8), tb(pq, 8).astype(np.uint8), d8


def main():
    tmp = tempfile.mkdtemp()
    # 1. golden text files (reference output) -> per-block arrays -> product writer: same bytes, via both entry points
    for comp in ("Luma", "Chroma"):
        g = np.load(os.path.join(GOLD, "g5_seq_%s.npz" % comp))
        F, W, H = int(g["F"]), int(g["W"]), int(g["H"])
        src = os.path.join(GOLD, "g5_partitionmat_%s.txt" % comp)
        hor, ver, q8, d8 = blocks_from_text(src, F, H, W)
        ref = open(src, "rb").read()
        assert engine.format_partition_text(F, H, W, hor, ver, q8, d8) == ref
        p = os.path.join(tmp, comp + ".txt")
        engine.write_partition_file(p, F, H, W, hor, ver, q8, d8)
        assert open(p, "rb").read() == ref
        pb = os.path.join(tmp, comp + ".pmpb")
        engine.write_partition_binary(pb, F, H, W, hor, ver, q8, d8)
        f, h, w, bh, bv, bq, bd = engine.read_partition_binary(pb)
        th, tv, tq, td = engine.read_partition_file(src, F, H, W)
        assert np.array_equal(bh, th) and np.array_equal(bv, tv) and np.array_equal(bq, tq) and np.array_equal(bd, td)
        mh, mv, mq, md = engine.tile_partition_maps(F, H, W, hor, ver, q8, d8)
        assert np.array_equal(mh, th) and np.array_equal(md, td) and np.array_equal(mq, tq)
    # 2. the reference's real demo frame (non-multiple-of-64 geometry: 416x240)
    src = os.path.join(GOLD, "g7_racehorses_luma_qp22_frame0.txt")
    hor, ver, q8, d8 = blocks_from_text(src, 1, 240, 416)
    assert engine.format_partition_text(1, 240, 416, hor, ver, q8, d8) == open(src, "rb").read()
    # 3. arbitrary caller data: multi-digit and negative values take the slow formatter paths; exact-size contract
    rng = np.random.default_rng(3)
    for F, H, W in ((1, 64, 64), (2, 136, 200), (3, 64, 320), (0, 64, 64), (1, 63, 640)):
        n = F * (H // 64) * (W // 64)
        hor = rng.integers(0, 256, (n, 16, 16)).astype(np.uint8); ver = rng.integers(0, 2, (n, 16, 16)).astype(np.uint8)
        q8 = rng.integers(0, 256, (n, 8, 8)).astype(np.uint8); d8 = rng.integers(-128, 128, (n, 3, 16, 16)).astype(np.int8)
        text = engine.format_partition_text(F, H, W, hor, ver, q8, d8)
        vals = np.array(text.split(), dtype=np.int64)
        R, Cc = (H >> 6) * 16, (W >> 6) * 16
        assert vals.size == F * (5 * R * Cc + R * Cc // 4)
        if n:
            mh, mv, mq, md = engine.tile_partition_maps(F, H, W, hor, ver, q8, d8)
            per = 5 * R * Cc + R * Cc // 4
            v = vals.reshape(F, per)
            assert np.array_equal(v[:, :R * Cc].reshape(F, R, Cc), mh)
            assert np.array_equal(v[:, 2 * R * Cc + R * Cc // 4:].reshape(F, 3, R, Cc), md)
        # an exactly-sized buffer is filled exactly; a smaller one is refused, never overrun
        need = lib.pmp_format_partition_text(F, H, W, hor.ctypes.data, ver.ctypes.data, q8.ctypes.data, d8.ctypes.data, None, 0)
        assert need == len(text)
        buf = C.create_string_buffer(max(int(need), 1))
        got = lib.pmp_format_partition_text(F, H, W, hor.ctypes.data, ver.ctypes.data, q8.ctypes.data, d8.ctypes.data, buf, need)
        assert got == need and buf.raw[:need] == text
        small = C.create_string_buffer(max(int(need) // 2, 1))
        if n:
            assert lib.pmp_format_partition_text(F, H, W, hor.ctypes.data, ver.ctypes.data, q8.ctypes.data, d8.ctypes.data, small,
                                                 max(int(need) // 2, 1)) == -1
    # 3b. block-row formatters of the sharded emission (pmp_format_partition_rows[_records], pmp_tile_partition_rows_records): any run
    #     of rows is the frame formatter on that slice; exact-size buffers; sizes-only calls
    from pmp_vvc_tip2023_amd import parallel
    for W_, nbr, wild in ((200, 5, True), (64, 1, False), (136, 3, True), (640, 2, False)):
        bw = W_ // 64
        n = nbr * bw
        if wild:
            hor = rng.integers(0, 256, (n, 16, 16)).astype(np.uint8); q8 = rng.integers(0, 256, (n, 8, 8)).astype(np.uint8)
            d8 = rng.integers(-128, 128, (n, 3, 16, 16)).astype(np.int8)
        else:
            hor = rng.integers(0, 2, (n, 16, 16)).astype(np.uint8); q8 = rng.integers(0, 4, (n, 8, 8)).astype(np.uint8)
            d8 = rng.integers(-1, 2, (n, 3, 16, 16)).astype(np.int8)
        ver = rng.integers(0, 2, (n, 16, 16)).astype(np.uint8)
        rec = parallel.pack_records(hor, ver, q8, d8)
        for a, b in ((0, nbr), (nbr - 1, nbr), (0, 1)):
            sl = slice(a * bw, b * bw)
            buf, sizes = engine.format_partition_rows_records(W_, b - a, rec[sl])
            want = engine.format_partition_text(1, 64 * (b - a), W_, hor[sl], ver[sl], q8[sl], d8[sl])
            assert buf.raw[:int(sizes.sum())] == want
            sz2 = np.zeros((b - a, 6), np.int64)
            need = lib.pmp_format_partition_rows(W_, b - a, hor[sl].ctypes.data, ver[sl].ctypes.data, q8[sl].ctypes.data, d8[sl].ctypes.data, None, 0,
                                                 sz2.ctypes.data)
            assert need == len(want) and np.array_equal(sz2, sizes)
            out = C.create_string_buffer(max(int(need), 1))
            assert lib.pmp_format_partition_rows(W_, b - a, hor[sl].ctypes.data, ver[sl].ctypes.data, q8[sl].ctypes.data, d8[sl].ctypes.data, out, need,
                                                 None) == need and out.raw[:need] == want
            if need > 8:
                small = C.create_string_buffer(int(need) - 7)
                assert lib.pmp_format_partition_rows_records(W_, b - a, np.ascontiguousarray(rec[sl]).ctypes.data, small, need - 7, None) == -1
            oh, ov, oq, od = engine.tile_partition_rows_records(W_, b - a, rec[sl])
            th, tv, tq, td = engine.tile_partition_maps(1, 64 * (b - a), W_, hor[sl], ver[sl], q8[sl], d8[sl])
            assert np.array_equal(oh, th[0]) and np.array_equal(ov, tv[0]) and np.array_equal(oq, tq[0]) and np.array_equal(od, td[0])
    assert lib.pmp_format_partition_rows_records(64, 1, None, None, 0, None) == -1
    assert lib.pmp_tile_partition_rows_records(64, 1, None, None, None, None, None) == -1
    # 4. error paths
    assert lib.pmp_tile_partition_maps(1, 64, 64, None, None, None, None, None, None, None, None) == -1
    assert lib.pmp_write_partition_file(None, 1, 64, 64, None, None, None, None) == -1
    z = np.zeros((1, 16, 16), np.uint8)
    assert lib.pmp_write_partition_binary(os.path.join(tmp, "nodir", "x").encode(), 1, 64, 64, z.ctypes.data, z.ctypes.data,
                                          np.zeros(64, np.uint8).ctypes.data, np.zeros(768, np.int8).ctypes.data) == -4
    assert b"cannot open" in lib.pmp_last_error(None)
    # 5. weight packing for every conv shape of the four nets (and odd shapes), both buffer modes
    for cout, cin, k in ((64, 64, 3), (64, 32, 5), (64, 64, 5), (32, 64, 3), (32, 128, 3), (32, 32, 3), (16, 32, 3), (8, 16, 3),
                         (32, 3, 3), (64, 32, 1), (32, 17, 5), (8, 32, 1), (1, 1, 1)):
        w = (rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
        kexp = C.c_int(-1)
        wp = w.ctypes.data_as(C.POINTER(C.c_float))
        n = lib.pmp_debug_pack_f16x3(wp, cout, cin, k, None, 0, C.byref(kexp))
        assert n > 0 and -100 <= kexp.value <= 24
        out = np.zeros(n, np.uint16)
        assert lib.pmp_debug_pack_f16x3(wp, cout, cin, k, out.ctypes.data_as(C.POINTER(C.c_uint16)), n, C.byref(kexp)) == n
        short = np.zeros(max(n - 1, 1), np.uint16)   # too small: nothing may be written
        assert lib.pmp_debug_pack_f16x3(wp, cout, cin, k, short.ctypes.data_as(C.POINTER(C.c_uint16)), n - 1, C.byref(kexp)) == n
        assert not short.any()
    assert lib.pmp_debug_pack_f16x3(None, 1, 1, 1, None, 0, None) == -1
    # 6. the .pmpw weight container: the shipped QT-net files, and malformed files (truncated, bad magic, offsets beyond the payload)
    from pmp_vvc_tip2023_amd import weights as W
    wdir = W.default_weight_dir()
    for fn in sorted(os.listdir(wdir)):
        if not fn.endswith(".pmpw"):
            continue
        man, tens = W.load_pmpw(os.path.join(wdir, fn))
        nid, qp, nt, nfl, cs = C.c_int(), C.c_int(), C.c_int(), C.c_int64(), C.c_double()
        assert lib.pmp_debug_read_weights_file(os.path.join(wdir, fn).encode(), C.byref(nid), C.byref(qp), C.byref(nt), C.byref(nfl), C.byref(cs)) == 0
        assert (nid.value, qp.value, nt.value) == (_lib.NET_IDS[man["net"]], man["qp"], len(tens))
        ref = sum(float(np.sum(t.astype(np.float64))) for t in tens.values())
        assert abs(cs.value - ref) <= 1e-6 * max(1.0, abs(ref)) and nfl.value == sum(t.size for t in tens.values())
    # the tensor fingerprint (round 6: what ties a manifest's act_exp to its nets) under the sanitizers, against the numpy statement of it
    man, tens = W.load_pmpw(os.path.join(wdir, "Chroma_Q_27.pmpw"))
    names = list(tens)[::-1]
    descs = (_lib.TensorDesc * len(names))()
    chunks, off = [], 0
    for i, k in enumerate(names):
        a = np.ascontiguousarray(tens[k], np.float32)
        descs[i].name = k.encode(); descs[i].ndim = a.ndim
        for j, dsz in enumerate(a.shape):
            descs[i].shape[j] = dsz
        descs[i].offset = off
        chunks.append(a.reshape(-1)); off += a.size
    blob = np.concatenate(chunks)
    fp = C.c_uint64()
    assert lib.pmp_fingerprint_tensors(blob.ctypes.data_as(C.c_void_p), descs, len(names), C.byref(fp)) == 0 and fp.value == W.fingerprint(tens)
    assert lib.pmp_fingerprint_tensors(None, descs, len(names), C.byref(fp)) == -1 and lib.pmp_fingerprint_tensors(blob.ctypes.data_as(C.c_void_p), descs, 0, C.byref(fp)) == -1
    good = open(os.path.join(wdir, "Luma_Q_22.pmpw"), "rb").read()
    jl = int.from_bytes(good[6:10], "little")
    cases = {"empty": b"", "magic": b"PMPW2\n" + good[6:], "cut_manifest": good[:10 + jl // 2], "cut_payload": good[:10 + jl + 100],
             "huge_len": good[:6] + (2 ** 31).to_bytes(4, "little") + good[10:], "garbage_json": good[:10] + b"{" * jl + good[10 + jl:],
             "no_tensors": good[:6] + (2).to_bytes(4, "little") + b"{}" + good[10 + jl:]}

    def crafted(manifest, payload=b""):
        return b"PMPW1\n" + len(manifest).to_bytes(4, "little") + manifest + payload
    big = 1 << 24
    cases.update({
        # four dimensions of 2^24: the element count would overflow 64 bits if it were not bounded while multiplying
        "shape_overflow": crafted(b'{"net":"Luma_Q","qp":22,"tensors":[{"name":"a","shape":[%d,%d,%d,%d],"offset":0}]}' % (big, big, big, big), b"\0" * 64),
        # offset + count wraps around
        "offset_wrap": crafted(b'{"net":"Luma_Q","qp":22,"tensors":[{"name":"a","shape":[4],"offset":9223372036854775800}]}', b"\0" * 64),
        "offset_19_digits": crafted(b'{"net":"Luma_Q","qp":22,"tensors":[{"name":"a","shape":[4],"offset":1000000000000000000000}]}', b"\0" * 64),
        # the manifest ends in digits and nothing follows: an unbounded strtoll would read past the buffer
        "digits_at_end": crafted(b'{"net":"Luma_Q","qp":2222222222'),
        "digits_at_end2": crafted(b'{"tensors":[{"name":"a","shape":[1],"offset":12345'),
        "last_element_ok_but_one_past": crafted(b'{"net":"Luma_Q","qp":22,"tensors":[{"name":"a","shape":[4],"offset":13}]}', b"\0" * 64),
    })
    cases.update({   # the optional activation-scale exponents of an MTT net (round 5): exactly five integers 0..60
        "act_exp_four": crafted(b'{"net":"Luma_MSBD","qp":22,"act_exp":[0,0,4,0],"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64),
        "act_exp_six": crafted(b'{"net":"Luma_MSBD","qp":22,"act_exp":[0,0,4,0,8,1],"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64),
        "act_exp_range": crafted(b'{"net":"Luma_MSBD","qp":22,"act_exp":[0,0,61,0,8],"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64),
        "act_exp_attention_over_6": crafted(b'{"net":"Luma_MSBD","qp":22,"act_exp":[0,7,4,0,8],"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64),
        "act_exp_trunk_over_30": crafted(b'{"net":"Luma_MSBD","qp":22,"act_exp":[0,0,4,0,31],"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64),
        "act_fp_short": crafted(b'{"net":"Luma_MSBD","qp":22,"act_exp":[0,0,4,0,8],"act_fp":["0123","0123456789abcdef"],"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64),
        "act_fp_not_hex": crafted(b'{"net":"Luma_MSBD","qp":22,"act_fp":["0123456789abcdeg","0123456789abcdef"],"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64),
        "act_fp_one": crafted(b'{"net":"Luma_MSBD","qp":22,"act_fp":["0123456789abcdef"],"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64),
        "act_exp_negative": crafted(b'{"net":"Luma_MSBD","qp":22,"act_exp":[0,0,-1,0,8],"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64),
        "act_exp_not_array": crafted(b'{"net":"Luma_MSBD","qp":22,"act_exp":4,"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64),
        "act_exp_unterminated": crafted(b'{"net":"Luma_MSBD","qp":22,"act_exp":[0,0,4,0,8'),
    })
    ok2 = os.path.join(tmp, "act_exp_ok.pmpw")
    open(ok2, "wb").write(crafted(b'{"net":"Luma_MSBD","qp":22,"act_exp":[30, 6, 4, 0, 8],"act_fp":["0123456789abcdef", "fedcba9876543210"],"tensors":[{"name":"a","shape":[4],"offset":0}]}', b"\0" * 64))
    assert lib.pmp_debug_read_weights_file(ok2.encode(), None, None, None, None, None) == 0
    ok = os.path.join(tmp, "edge_ok.pmpw")       # the last four floats of the payload: accepted
    open(ok, "wb").write(crafted(b'{"net":"Luma_Q","qp":22,"tensors":[{"name":"a","shape":[4],"offset":12}]}', b"\0" * 64))
    assert lib.pmp_debug_read_weights_file(ok.encode(), None, None, None, None, None) == 0
    for name, blob in cases.items():
        pth = os.path.join(tmp, name + ".pmpw")
        open(pth, "wb").write(blob)
        rc = lib.pmp_debug_read_weights_file(pth.encode(), None, None, None, None, None)
        assert rc in (-1, -4), (name, rc)
    assert lib.pmp_debug_read_weights_file(os.path.join(tmp, "missing.pmpw").encode(), None, None, None, None, None) == -4
    print("hostasan checks passed")


if __name__ == "__main__":
    main()
