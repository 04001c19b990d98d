"""SURVEY 8(f) row N4 under the driver's GPU tier: the in-process VTM hook's call sequence, C ABI only.

tools/vtm_build/pmp_hook.cpp:163-191 stands in for EncAppCfg::parsePartitionMatrix
(codec/.../App/EncoderApp/EncAppCfg.cpp:4234-4404, call site encmain.cpp:184-188, tables Lib/CommonLib/Rom.h:240-248):

    pmp_create -> pmp_cut_blocks (host planes) -> per component: pmp_load_weights_file x2 -> pmp_infer_postprocess
               -> pmp_tile_partition_maps -> pmp_destroy

This test drives exactly that through ctypes with NO Python weight handling on the product side: the .pmpw containers are
files on disk (real QT weights from weights/, the documented synthetic MTT weights written by the test), read, checked and
uploaded by the library's own C++ reader.  Checked against the oracle: blocks bit-exact, logits <= 1e-3, split flags ==
oracle post-processing of the device logits, tiled frame matrices == what a parser with parsePartitionMatrix's geometry
reads from the ORACLE's text file.
"""
import ctypes as C
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3   # north_star: map logits within 1e-3


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    """A CTU_Models-style directory of .pmpw files, as the hook's PMP_MODEL_DIR: the shipped QT nets + synthetic MTT nets."""
    from pmp_vvc_tip2023_amd import synth, weights as W
    d = tmp_path_factory.mktemp("CTU_Models")
    for comp in ("Luma", "Chroma"):
        for qp in (22, 37):
            src = os.path.join(W.default_weight_dir(), "%s_Q_%d.pmpw" % (comp, qp))
            shutil.copy(src, d / os.path.basename(src))
            W.save_pmpw(str(d / ("%s_BD_%d.pmpw" % (comp, qp))), comp + "_MSBD", qp, synth.synth_msbd_weights(comp, qp),
                        source="synthetic(seed=%d)" % qp)
    return d


@pytest.mark.parametrize("bitdepth,width,height,frames,qp", [(8, 200, 136, 2, 22), (10, 136, 72, 3, 37)])
def test_hook_sequence_through_the_c_abi_only(model_dir, oracle_lib, tmp_path, bitdepth, width, height, frames, qp):
    from oracle import nets_torch as O
    from pmp_vvc_tip2023_amd import _lib, engine as E, synth, weights as W
    lib = _lib.load()
    y, u, v = synth.recipe_r_frames(frames, height, width, 900 + qp, bitdepth=bitdepth)   # both sizes drop a right/bottom remainder
    y, u, v = (np.ascontiguousarray(a) for a in (y, u, v))
    n = frames * (height // 64) * (width // 64)
    assert n > 0 and (width % 64 or height % 64)

    ctx = C.c_void_p()
    assert lib.pmp_create(0, C.byref(ctx)) == 0, lib.pmp_last_error(None)
    try:
        # ---- pmp_cut_blocks on host pointers (pmp_hook.cpp:169-170)
        by = np.empty((n, 68, 68), np.uint8); bu = np.empty((n, 34, 34), np.uint8); bv = np.empty((n, 34, 34), np.uint8)
        rc = lib.pmp_cut_blocks(ctx, _vp(y), _vp(u), _vp(v), frames, height, width, bitdepth, _vp(by), _vp(bu), _vp(bv))
        assert rc == 0, lib.pmp_last_error(ctx)
        oy, ou, ov = oracle_lib.cut_blocks(y, u, v, bitdepth)
        assert np.array_equal(by, oy) and np.array_equal(bu, ou) and np.array_equal(bv, ov)

        R, Cc = 16 * (height >> 6), 16 * (width >> 6)
        for k, comp in enumerate(("Luma", "Chroma")):
            luma = comp == "Luma"
            # ---- pmp_load_weights_file x2 (pmp_hook.cpp:176-179): the C++ .pmpw reader uploads, nothing comes from Python
            net_q, net_b = (0, 1) if luma else (2, 3)
            assert not lib.pmp_has_weights(ctx, net_q, qp) and not lib.pmp_has_weights(ctx, net_b, qp)
            pq = str(model_dir / ("%s_Q_%d.pmpw" % (comp, qp))).encode()
            pb = str(model_dir / ("%s_BD_%d.pmpw" % (comp, qp))).encode()
            # a container that holds another net or another QP is refused, the right one is accepted
            assert lib.pmp_load_weights_file(ctx, net_q, qp, pb) == _lib_err("PMP_E_INVALID")
            assert lib.pmp_load_weights_file(ctx, net_q, 27, pq) == _lib_err("PMP_E_INVALID")
            assert lib.pmp_load_weights_file(ctx, net_q, qp, pq) == 0, lib.pmp_last_error(ctx)
            assert lib.pmp_load_weights_file(ctx, net_b, qp, pb) == 0, lib.pmp_last_error(ctx)
            assert lib.pmp_has_weights(ctx, net_q, qp) and lib.pmp_has_weights(ctx, net_b, qp)

            # ---- pmp_infer_postprocess (pmp_hook.cpp:180-183); the logits are fetched too, for the oracle comparison
            hor = np.empty((n, 16, 16), np.uint8); ver = np.empty((n, 16, 16), np.uint8)
            q8 = np.empty((n, 8, 8), np.uint8); d8 = np.empty((n, 3, 16, 16), np.int8)
            qt = np.empty((n, 1, 8, 8), np.float32); bt = np.empty((n, 3, 16, 16), np.float32); dire = np.empty((n, 3, 16, 16), np.float32)
            rc = lib.pmp_infer_postprocess(ctx, k, qp, _vp(by), _vp(bu), _vp(bv), n, _vp(hor), _vp(ver), _vp(q8), _vp(d8),
                                           _vp(qt), _vp(bt), _vp(dire))
            assert rc == 0, lib.pmp_last_error(ctx)
            wq = W.load_pmpw(pq.decode())[1]; wb = W.load_pmpw(pb.decode())[1]      # the CHECKER's copy of the same files
            x = O.luma_input(oy) if luma else O.chroma_input(oy, ou, ov)
            oq, obt, odire = O.infer_qbd(wq, wb, x, luma)
            err = max(np.abs(qt - oq).max(), np.abs(bt - obt).max(), np.abs(dire - odire).max())
            assert err < TOL, "%s logits off by %g" % (comp, err)
            # the hook's own call passes NULL for the logits: same flags
            hor2 = np.empty_like(hor); ver2 = np.empty_like(ver); q82 = np.empty_like(q8); d82 = np.empty_like(d8)
            rc = lib.pmp_infer_postprocess(ctx, k, qp, _vp(by), _vp(bu), _vp(bv), n, _vp(hor2), _vp(ver2), _vp(q82), _vp(d82), None, None, None)
            assert rc == 0, lib.pmp_last_error(ctx)
            assert np.array_equal(hor, hor2) and np.array_equal(ver, ver2) and np.array_equal(q8, q82) and np.array_equal(d8, d82)
            # flags == oracle post-processing of the DEVICE logits, which also writes the oracle's text file
            txt = str(tmp_path / ("%s.txt" % comp))
            oh, ovv, oq8, od8 = oracle_lib.seq_post_process(qt, bt, dire, comp, frames, width, height, txt)
            assert np.array_equal(hor, oh) and np.array_equal(ver, ovv) and np.array_equal(q8, oq8.astype(np.uint8)) and np.array_equal(d8, od8)

            # ---- pmp_tile_partition_maps (pmp_hook.cpp:184-187) == the matrices parsePartitionMatrix would build from that file
            th = np.zeros((frames, R, Cc), np.uint8); tv = np.zeros_like(th)
            tq = np.zeros((frames, R // 2, Cc // 2), np.uint8); td = np.zeros((frames, 3, R, Cc), np.int8)
            rc = lib.pmp_tile_partition_maps(frames, height, width, _vp(hor), _vp(ver), _vp(q8), _vp(d8), _vp(th), _vp(tv), _vp(tq), _vp(td))
            assert rc == 0, lib.pmp_last_error(None)
            ph, pv, pqt, pd = E.read_partition_file(txt, frames, height, width)
            assert np.array_equal(th, ph) and np.array_equal(tv, pv) and np.array_equal(tq, pqt) and np.array_equal(td, pd)
        assert lib.pmp_get_saturation(ctx) == 0     # pmp_hook.cpp:189: the range guard stayed quiet on 8-bit content
    finally:
        lib.pmp_destroy(ctx)


def _lib_err(name):
    from pmp_vvc_tip2023_amd import _lib
    return {v: k for k, v in _lib.ERRORS.items()}[name]


def test_weight_file_errors_leave_the_context_usable(model_dir, tmp_path):
    """A missing or truncated container is an error code with a message (the hook dies on it, pmp_hook.cpp `ck`); the context
    keeps working afterwards."""
    from pmp_vvc_tip2023_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.pmp_create(0, C.byref(ctx)) == 0
    try:
        assert lib.pmp_load_weights_file(ctx, 0, 22, str(tmp_path / "nope.pmpw").encode()) < 0
        assert b"nope.pmpw" in lib.pmp_last_error(ctx)
        good = open(model_dir / "Luma_Q_22.pmpw", "rb").read()
        cut = tmp_path / "Luma_Q_22.pmpw"
        cut.write_bytes(good[:len(good) // 2])
        assert lib.pmp_load_weights_file(ctx, 0, 22, str(cut).encode()) < 0
        assert not lib.pmp_has_weights(ctx, 0, 22)
        assert lib.pmp_load_weights_file(ctx, 0, 22, str(model_dir / "Luma_Q_22.pmpw").encode()) == 0
        assert lib.pmp_has_weights(ctx, 0, 22)
    finally:
        lib.pmp_destroy(ctx)
