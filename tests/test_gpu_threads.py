"""The threading contract of the boundary (include/pmp.h: "A context is not thread-safe; different contexts are independent";
SURVEY.md 8(b): "one handle (+ one host thread or process) per device").  An in-process consumer - the N4 hook inside one encoder
process on an 8-GPU node - is one process with several contexts, each driven by its own host thread.

Two host threads, two contexts, CONCURRENTLY (ctypes drops the GIL for the duration of every call): different components, QPs and
datapaths, one context in overlap mode (two streams, two workspaces), the other provoking a range-guard re-run, both cycling
pmp_create / pmp_destroy (the process-global parked-workspace pool) and loading weights (the calibration pass on its own stream and
workspace).  Every result must be bit-identical to the same calls made serially from one thread."""
import ctypes as C
import os
import shutil
import threading
import time

import numpy as np
import pytest

import conftest  # noqa: F401 - puts tools/ on sys.path
import trained_like  # tools/trained_like.py: test-weight data (round 6: out of the product package)

pytestmark = pytest.mark.gpu


def _inputs():
    from pmp_vvc_tip2023_amd import synth
    from test_gpu_parity import _range_stress_weights
    y, u, v = synth.recipe_r_blocks(1100, 4711)
    qt, bt, dire = synth.random_partition_batch(600, 99, 1, 0.2)
    return dict(y=y, u=u, v=v, tl=trained_like.msbd_weights("Luma", 22), stress=_range_stress_weights(),
                benign=synth.synth_msbd_weights("Luma", 22), m2p=(qt, bt, dire))


def _script_overlap(inp, out, cycles=3):
    """Context A: overlap mode, default datapath, trained-like MTT weights (a calibration pass per load), a 1100-block call (two chunks
    on two streams), then an opt-in datapath on the other component, a device-pointer call left in flight across a host-pointer call."""
    import torch
    from pmp_vvc_tip2023_amd import engine
    y, u, v = inp["y"], inp["u"], inp["v"]
    dev = torch.device("cuda:0")
    for _ in range(cycles):
        e = engine.Engine(0, allow_synthetic_mtt=True)
        try:
            e.set_overlap(True)
            e.load("Luma", 22, msbd_weights=inp["tl"])
            out.append(e.infer_postprocess("Luma", 22, y, want_logits=True))
            out.append(tuple(e.activation_report("Luma", 22)["exps"]))
            d_y = torch.from_numpy(y[:300]).to(dev)
            rec = torch.empty((300, 1344), dtype=torch.uint8, device=dev)
            e.infer_postprocess_records_device("Luma", 22, d_y.data_ptr(), None, None, 300, rec.data_ptr())     # in flight ...
            e.set_precision("bf16x6")                                                                            # ... settled by the switch
            e.load("Chroma", 27)
            out.append(e.infer_postprocess("Chroma", 27, y[:64], u[:64], v[:64], want_logits=True))
            e.synchronize()
            out.append(rec.cpu().numpy())
            out.append((e.saturation_reruns(), e.saturated()))
        finally:
            e.close()


def _script_guard(inp, out, cycles=3):
    """Context B: range-stress weights with the activation scales off (every f16x3 call fires the guard and is re-run on fp32), weights
    replaced between calls, the fp32 datapath, Map2Partition on its own through the host-pointer seam."""
    from pmp_vvc_tip2023_amd import engine
    y, u, v = inp["y"], inp["u"], inp["v"]
    qt, bt, dire = inp["m2p"]
    for _ in range(cycles):
        e = engine.Engine(0, allow_synthetic_mtt=True)
        try:
            e.load("Luma", 22)
            e.set_activation_scales(False)
            e.load_pretrain_model("Luma_MSBD", 22, inp["stress"])
            out.append(e.infer_postprocess("Luma", 22, y[:48], want_logits=True))
            out.append((e.saturation_reruns(), e.saturated()))
            e.set_precision("fp32")
            e.load("Chroma", 37)
            out.append(e.inference_pre_QBD("Chroma", 37, y[100:140], u[100:140], v[100:140]))
            e.set_precision("f16x3")
            e.set_activation_scales(True)
            e.load_pretrain_model("Luma_MSBD", 22, inp["benign"])            # replaces the net: settles, re-packs, calibrates
            out.append(e.inference_pre_QBD("Luma", 22, y[200:264]))
            out.append(e.post_process(qt, bt, dire, "Luma"))
            out.append((e.saturation_reruns(), e.saturated()))
        finally:
            e.close()


def _flatten(res):
    flat = []
    for r in res:
        flat.extend(r if isinstance(r, (tuple, list)) else [r])
    return flat


def _same(a, b):
    fa, fb = _flatten(a), _flatten(b)
    assert len(fa) == len(fb)
    for i, (x, y) in enumerate(zip(fa, fb)):
        if isinstance(x, np.ndarray):
            assert np.array_equal(x, y), "result %d differs between the serial and the concurrent run" % i
        else:
            assert x == y, (i, x, y)


def _run_threads(targets):
    errs = []

    def wrap(fn, args):
        try:
            fn(*args)
        except BaseException as ex:     # noqa: B902 - reported in the main thread
            errs.append(ex)
    th = [threading.Thread(target=wrap, args=(fn, args)) for fn, args in targets]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
        assert not t.is_alive(), "a context's thread did not finish: contexts are blocking each other"
    if errs:
        raise errs[0]


def test_two_contexts_on_two_threads_equal_the_serial_run():
    from pmp_vvc_tip2023_amd import _lib
    inp = _inputs()
    serial_a, serial_b = [], []
    _script_overlap(inp, serial_a, cycles=1)
    _script_guard(inp, serial_b, cycles=1)
    assert serial_b[1] == (1, True) and serial_a[-1] == (0, False)          # the guard re-ran B's first call, never A's
    assert serial_a[1][2] > 0 or serial_a[1][4] > 0                          # A's trained-like net got activation scales from ITS calibration
    for stagger in (0.0, 0.05, 0.3):
        conc_a, conc_b = [], []

        def late_guard(inp_, out_):
            time.sleep(stagger)
            _script_guard(inp_, out_)
        _run_threads([(_script_overlap, (inp, conc_a)), (late_guard, (inp, conc_b))])
        per_a, per_b = len(serial_a), len(serial_b)
        assert len(conc_a) == 3 * per_a and len(conc_b) == 3 * per_b
        for k in range(3):      # every create/destroy cycle of both threads
            _same(serial_a, conc_a[k * per_a:(k + 1) * per_a])
            _same(serial_b, conc_b[k * per_b:(k + 1) * per_b])
    assert _lib.load().pmp_trim() == 0


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _hook_sequence(lib, model_dir, planes, frames, height, width, bitdepth, qp, out):
    """tools/vtm_build/pmp_hook.cpp:163-191 - pmp_create -> pmp_cut_blocks -> per component: pmp_load_weights_file x2 ->
    pmp_infer_postprocess (logits not fetched) -> pmp_tile_partition_maps -> pmp_destroy - through raw ctypes."""
    y, u, v = planes
    n = frames * (height // 64) * (width // 64)
    ctx = C.c_void_p()
    assert lib.pmp_create(0, C.byref(ctx)) == 0, lib.pmp_last_error(None)
    try:
        by = np.empty((n, 68, 68), np.uint8); bu = np.empty((n, 34, 34), np.uint8); bv = np.empty((n, 34, 34), np.uint8)
        assert lib.pmp_cut_blocks(ctx, _vp(y), _vp(u), _vp(v), frames, height, width, bitdepth, _vp(by), _vp(bu), _vp(bv)) == 0, lib.pmp_last_error(ctx)
        R, Cc = 16 * (height >> 6), 16 * (width >> 6)
        for k, comp in enumerate(("Luma", "Chroma")):
            net_q, net_b = (0, 1) if k == 0 else (2, 3)
            for net, kind in ((net_q, "Q"), (net_b, "BD")):
                p = str(model_dir / ("%s_%s_%d.pmpw" % (comp, kind, qp))).encode()
                assert lib.pmp_load_weights_file(ctx, net, qp, p) == 0, lib.pmp_last_error(ctx)
            hor = np.empty((n, 16, 16), np.uint8); ver = np.empty_like(hor)
            q8 = np.empty((n, 8, 8), np.uint8); d8 = np.empty((n, 3, 16, 16), np.int8)
            rc = lib.pmp_infer_postprocess(ctx, k, qp, _vp(by), _vp(bu), _vp(bv), n, _vp(hor), _vp(ver), _vp(q8), _vp(d8), None, None, None)
            assert rc == 0, lib.pmp_last_error(ctx)
            th = np.zeros((frames, R, Cc), np.uint8); tv = np.zeros_like(th)
            tq = np.zeros((frames, R // 2, Cc // 2), np.uint8); td = np.zeros((frames, 3, R, Cc), np.int8)
            assert lib.pmp_tile_partition_maps(frames, height, width, _vp(hor), _vp(ver), _vp(q8), _vp(d8), _vp(th), _vp(tv), _vp(tq), _vp(td)) == 0
            out.extend([th, tv, tq, td])
        assert lib.pmp_get_saturation(ctx) == 0
    finally:
        lib.pmp_destroy(ctx)


def test_in_process_hook_sequence_from_two_threads(tmp_path):
    """Row N4's call sequence as two encoder threads of one process would make it: an 8-bit 1080p-ish picture at QP22 beside a 10-bit
    one at QP37, each on its own context, four rounds - identical to the serial run."""
    from pmp_vvc_tip2023_amd import _lib, synth, weights as W
    lib = _lib.load()
    for comp in ("Luma", "Chroma"):
        for qp in (22, 37):
            src = os.path.join(W.default_weight_dir(), "%s_Q_%d.pmpw" % (comp, qp))
            shutil.copy(src, tmp_path / os.path.basename(src))
            W.save_pmpw(str(tmp_path / ("%s_BD_%d.pmpw" % (comp, qp))), comp + "_MSBD", qp, trained_like.msbd_weights(comp, qp),
                        source="trained-like (synth.py)")
    jobs = []
    for bitdepth, width, height, frames, qp in ((8, 1920 // 2, 1080 // 2, 3, 22), (10, 416, 240, 4, 37)):
        planes = tuple(np.ascontiguousarray(a) for a in synth.recipe_r_frames(frames, height, width, 77 + qp, bitdepth=bitdepth))
        jobs.append((planes, frames, height, width, bitdepth, qp))
    serial = []
    for j in jobs:
        o = []
        _hook_sequence(lib, tmp_path, *j, o)
        serial.append(o)
    conc = [[], []]

    def rounds(j, o):
        for _ in range(4):
            _hook_sequence(lib, tmp_path, *j, o)
    _run_threads([(rounds, (jobs[0], conc[0])), (rounds, (jobs[1], conc[1]))])
    for s, c in zip(serial, conc):
        assert len(c) == 4 * len(s)
        for k in range(4):
            _same(s, c[k * len(s):(k + 1) * len(s)])
