"""GPU parity on TRAINED-LIKE MTT weights (trained_like.msbd_weights: tensors bootstrapped from the real QT-net tensors, trunk
activations at the QT nets' 1e3 range, gated products to 1e4; the reference's *_BD_*.pkl are absent from the mount).
Goldens: tests/golden/g2b_msbd_trained_like.npz, written by tools/gen_golden.py from the REFERENCE's MTT modules
(Model_QBD.py:100-155, :198-253) holding exactly these tensors.  Tolerance: north_star's 1e-3 on the logits."""
import os

import numpy as np
import pytest

import conftest  # noqa: F401 - puts tools/ on sys.path
import trained_like  # tools/trained_like.py: test-weight data (round 6: out of the product package)
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module", params=["f16x3", "bf16x6", "fp32"])
def eng(request):
    from pmp_vvc_tip2023_amd import engine
    e = engine.Engine(0, allow_synthetic_mtt=True)
    e.set_precision(request.param)
    yield e
    e.close()


def _load_tl(e, comp, qp, **gains):
    from pmp_vvc_tip2023_amd import synth
    w = trained_like.msbd_weights(comp, qp, **gains)
    e.load(comp, qp, msbd_weights=w)                              # the real QT net from weights/, the MTT net replaced
    return w


def _golden_err(bt, dire, g2b, comp, qp):
    errs = []
    for k in range(3):
        ref = g2b["out%d_%s_%d" % (k, comp, qp)]                  # [16,2,16,16]: ch0 depth, ch1 direction
        errs += [np.abs(bt[:, k] - ref[:, 0]).max(), np.abs(dire[:, k] - ref[:, 1]).max()]
    return float(max(errs))


@pytest.mark.parametrize("comp", ["Luma", "Chroma"])
@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_trained_like_logits_vs_reference_golden(eng, comp, qp):
    """All eight nets, all three datapaths, against the reference modules' outputs; the default datapath must get there WITHOUT the
    range guard's fp32 re-run (trunks at 1e3, gate products to 1e4: inside the fp16 range)."""
    g1, g2b = golden("g1_qt.npz"), golden("g2b_msbd_trained_like.npz")
    _load_tl(eng, comp, qp)
    eng.clear_saturation()
    qt, bt, dire = eng.inference_pre_QBD(comp, qp, g1["block_y"], g1["block_u"], g1["block_v"])
    assert np.abs(qt - g1["qt_%s_%d" % (comp, qp)]).max() < TOL
    err = _golden_err(bt, dire, g2b, comp, qp)
    assert err < TOL, "%s QP%d trained-like MTT logits off by %g" % (comp, qp, err)
    assert eng.saturation_reruns() == 0 and not eng.saturated()


_ORACLE = {}


def _oracle(comp, qp, n):
    from oracle import nets_torch as O
    from pmp_vvc_tip2023_amd import synth, weights as W
    key = (comp, qp, n)
    if key not in _ORACLE:
        import os
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        y, u, v = synth.recipe_r_blocks(n, 7000 + qp + (3 if comp == "Chroma" else 0))
        y[0] = 0; u[0] = 0; v[0] = 0
        y[1] = 255; u[1] = 255; v[1] = 255
        rng = np.random.default_rng(qp)
        y[2] = rng.integers(0, 256, y[2].shape); u[2] = rng.integers(0, 256, u[2].shape); v[2] = rng.integers(0, 256, v[2].shape)   # white noise
        y[3] = np.where((np.arange(68)[:, None] // 2 + np.arange(68)[None, :] // 2) % 2, 255, 0)                                   # 2-px checkerboard
        luma = comp == "Luma"
        wq, _ = W.load_net_weights(comp + "_Q", qp)
        wbd = trained_like.msbd_weights(comp, qp)
        x = O.luma_input(y) if luma else O.chroma_input(y, u, v)
        _ORACLE[key] = (y, u, v) + tuple(O.infer_qbd(wq, wbd, x, luma, batch=64))
    return _ORACLE[key]


# Blocks of _oracle()'s set that an OPT-IN datapath puts outside the plain 1e-3 on the QT logits (Luma_Q's conditioning at low QP leaves
# bf16x6 no margin): excluded BY INDEX from the QT assertion of that datapath - every other block, and every block on
# the default datapath and on fp32, gets the plain tolerance.
_QT_KNOWN = {("bf16x6", "Luma", 22): (), ("fp32", "Luma", 22): ()}


@pytest.mark.parametrize("comp,qp", [("Luma", 22), ("Luma", 37), ("Chroma", 27), ("Chroma", 37)])
def test_trained_like_fresh_blocks_vs_oracle(eng, oracle_lib, comp, qp):
    """320 fresh blocks (flat, saturated, white-noise and checkerboard blocks included) against the torch oracle holding the same
    tensors: north_star's ABSOLUTE 1e-3 on every natural block and on every block whose logits are inside Map2Partition's operating range
    (|logit| <= 8) - only the synthetic extremes beyond it get a relative tolerance; the record of what the guard did rides along.  Then the DEVICE logits of all 320
    blocks - the checkerboard's +-300 and the noise block's +-75 among them - through pmp_postprocess against the oracle's
    post-processing of the same numbers (the kernel's +-100 saturation of the rounded depth sees real net output)."""
    y, u, v, oq, obt, odire = _oracle(comp, qp, 320)
    _load_tl(eng, comp, qp)
    eng.clear_saturation()
    hor, ver, q8, d8, qt, bt, dire = eng.infer_postprocess(comp, qp, y, u, v, want_logits=True)

    def per_block(pairs):
        e = np.max([np.abs(a - b).reshape(len(a), -1).max(axis=1) for a, b in pairs], axis=0)
        m = np.max([np.abs(b).reshape(len(b), -1).max(axis=1) for _, b in pairs], axis=0)
        return e, m
    # the QT nets (real weights; not this test's subject, but they run first): plain 1e-3 on every block
    e_q, _ = per_block([(qt, oq)])
    known = _QT_KNOWN.get((eng.get_precision(), comp, qp), ())
    over = [int(i) for i in np.flatnonzero(e_q >= TOL) if int(i) not in known]
    assert not over, "QT logits: blocks %s off by %s" % (over, e_q[over])
    assert all(e_q[i] < 1.25 * TOL for i in known)
    # (a) the MTT net ALONE on identical inputs: the oracle fed with the QT logits the HIP path produced
    from oracle import nets_torch as O
    from pmp_vvc_tip2023_amd import synth
    luma = comp == "Luma"
    x = O.luma_input(y) if luma else O.chroma_input(y, u, v)
    with torch.no_grad():
        o = O.msbd_forward(trained_like.msbd_weights(comp, qp), x, torch.from_numpy(qt), luma)
    abt = np.stack([t[:, 0].numpy() for t in o], 1); adire = np.stack([t[:, 1].numpy() for t in o], 1)
    e_a, m_a = per_block([(bt, abt), (dire, adire)])
    # (b) end to end, QT net included: the oracle's own q feeds the oracle's MTT net (the trained-like nets pass an error of q on
    # 0.5..1.7x, synth.py: _TL_Q_STEM)
    e_b, m_b = per_block([(bt, obt), (dire, odire)])
    inside = np.maximum(m_a, m_b) <= 8.0
    natural = np.arange(len(y)) >= 4
    n_out = int((natural & ~inside).sum())
    # north_star's ABSOLUTE 1e-3: on every block whose logits are inside Map2Partition's operating range (|logit| <= 8) AND on every natural
    # (recipe-R) block whatever its logits - a few per cent of them reach |logit| 10..50 on these nets (Luma QP22: 2.8 %), none needs slack
    absolute = inside | natural
    worst = int(np.argmax(np.where(absolute, e_b, 0)))
    print("trained-like %s QP%d %s: %d of %d natural blocks beyond |logit| 8 (max |logit| %.1f); absolute tolerance on %d blocks: MTT alone max %.2e, "
          "end to end max %.2e (block %d, |logit| %.1f); synthetic extremes (relative to |logit|/8): max %.2e; reruns %d"
          % (comp, qp, eng.get_precision(), n_out, int(natural.sum()), float(m_b[natural].max()), int(absolute.sum()), e_a[absolute].max(), e_b[worst], worst,
             m_b[worst], (e_b / np.maximum(1.0, m_b / 8.0))[~absolute].max() if (~absolute).any() else 0.0, eng.saturation_reruns()))
    assert (e_a[absolute] < TOL).all(), "%s QP%d MTT net on identical inputs: block %d off by %g" % (comp, qp, int(np.argmax(np.where(absolute, e_a, 0))), e_a[absolute].max())
    assert (e_b[absolute] < TOL).all(), "%s QP%d end to end: block %d off by %g (|logit| %g)" % (comp, qp, worst, e_b[worst], m_b[worst])
    assert n_out <= 0.05 * natural.sum(), "%d of %d natural blocks leave the operating range: these are no longer trained-LIKE nets" % (n_out, int(natural.sum()))
    # only the synthetic extremes beyond the operating range (2-px checkerboard: +-300, white noise: +-75; the torch oracle itself is 6.6e-4
    # from an fp64 evaluation on the checkerboard block) get the same tolerance RELATIVE to |logit| / 8
    assert (~absolute).sum() <= 4
    assert (e_a[~absolute] < TOL * np.maximum(1.0, m_a[~absolute] / 8.0)).all() and (e_b[~absolute] < TOL * np.maximum(1.0, m_b[~absolute] / 8.0)).all()
    assert max(m_b[:4].max(), m_a[:4].max()) > 50 or not luma    # the extremes ARE extreme (luma): the post-processing below sees them
    assert eng.saturation_reruns() == 0
    # the split flags of the fused call = the reference's post-processing of ITS device logits, the +-300 blocks included
    with np.errstate(invalid="ignore"):
        oh, ov, of, od = oracle_lib.seq_post_process(qt, bt, dire, comp, 1, 64 * len(y), 64, None)
    assert np.array_equal(hor, oh) and np.array_equal(ver, ov) and np.array_equal(d8, od) and np.array_equal(q8, of.astype(np.uint8))
    h2, v2, q82, d82 = eng.post_process(qt, bt, dire, comp)       # and through the host-pointer seam (pmp_postprocess)
    assert np.array_equal(hor, h2) and np.array_equal(ver, v2) and np.array_equal(q8, q82) and np.array_equal(d8, d82)


@pytest.mark.parametrize("comp,qp", [("Luma", 22), ("Chroma", 27)])
@pytest.mark.parametrize("gains", [(64.0, 1.0), (1.0, 16.0), (64.0, 16.0), (1.0, 1024.0), (4096.0, 64.0), (65536.0, 256.0), (1.0, 1.0, 1024.0), (256.0, 4.0, 2.0 ** 14)])
def test_trained_like_stress_gains_stay_on_the_default_datapath(eng, comp, qp, gains):
    """trunk_gain K / gate_gain G are exact powers of two that the heads undo (synth.py): the reference's logits do not change (pinned
    while the goldens were generated), but the trunks now run at K x 1e3 and the gated products at K x G x 1e4 - far outside fp16.
    The default datapath must still deliver the golden logits, and must do so WITHOUT falling back to the 2.8x slower fp32 re-run:
    the per-segment power-of-two activation scales chosen when the net is first used absorb the range (include/pmp.h)."""
    g1, g2b = golden("g1_qt.npz"), golden("g2b_msbd_trained_like.npz")
    K, G = gains[:2]
    A = gains[2] if len(gains) > 2 else 1.0      # att_gain: the attention trunks themselves (segments 1 and 3) at A x their range
    _load_tl(eng, comp, qp, trunk_gain=K, gate_gain=G, att_gain=A)
    eng.clear_saturation()
    qt, bt, dire = eng.inference_pre_QBD(comp, qp, g1["block_y"], g1["block_u"], g1["block_v"])
    err = _golden_err(bt, dire, g2b, comp, qp)
    print("trained-like %s QP%d K=%g G=%g A=%g %s: err %.2e reruns %d" % (comp, qp, K, G, A, eng.get_precision(), err, eng.saturation_reruns()))
    assert err < TOL, "%s QP%d K=%g G=%g A=%g off by %g" % (comp, qp, K, G, A, err)
    assert eng.saturation_reruns() == 0, "the range guard fell back to fp32 (%d re-runs)" % eng.saturation_reruns()


def test_activation_scales_report():
    """pmp_debug_activation_report: what the calibration pass saw and what it chose (include/pmp.h, "Activation scales").
      * the benign uniform MTT weights (bench.py's configs[1] workload, the G2 goldens) get exponents of zero: their arithmetic is untouched;
      * trained-like weights: 49 recorded tensors in launch order, trunks at the 1e3..1e4 level on the calibration blocks, exponents small;
      * the stress gains move the recorded maxima by exactly K (trunk) and K*G (behind the gates) - powers of two are exact on the fp32
        calibration datapath - and the exponents follow, which is what keeps those nets off the range guard's fp32 re-run."""
    from pmp_vvc_tip2023_amd import engine, synth
    e = engine.Engine(0, allow_synthetic_mtt=True)
    try:
        for comp in ("Luma", "Chroma"):
            r0 = e.activation_report(comp, 22)                    # synthetic uniform MTT weights
            assert r0["exps"] == [0, 0, 0, 0, 0], r0
            assert max(r0["seg_amax"]) < 4096
        for comp, qp in (("Luma", 22), ("Chroma", 27)):
            e.load(comp, qp, msbd_weights=trained_like.msbd_weights(comp, qp))
            base = e.activation_report(comp, qp)
            names = [t[0] for t in base["tensors"]]
            assert len(names) == 49 and names[0] == "stem" and names[1] == "trunk_M1.0.t" and names[-1] == "trunk_B3.2"
            assert [t[1] for t in base["tensors"] if t[0] in ("trunk_Att1.1", "trunk_Att2.1")] == [2, 4]   # the gated outputs open segments 2 and 4
            assert 1e3 < base["seg_amax"][0] < 1e5 and base["exps"][1] == 0 and base["exps"][3] == 0      # gates of O(10..200): no scale needed
            assert all(0 <= x <= 8 for x in base["exps"])
            print("activation scales %s QP%d: exps %s, segment maxima %s" % (comp, qp, base["exps"], ["%.3g" % m for m in base["seg_amax"]]))
            e.load(comp, qp, msbd_weights=trained_like.msbd_weights(comp, qp, trunk_gain=64.0, gate_gain=16.0))
            st = e.activation_report(comp, qp)
            assert np.isclose(st["seg_amax"][0], 64.0 * base["seg_amax"][0], rtol=1e-6)
            assert np.isclose(st["seg_amax"][1], 16.0 * base["seg_amax"][1], rtol=1e-6) or st["seg_amax"][1] >= base["seg_amax"][1]
            for sg in (2, 4):
                assert np.isclose(st["seg_amax"][sg], 1024.0 * base["seg_amax"][sg], rtol=1e-6)
                assert st["seg_amax"][sg] * 2.0 ** -st["exps"][sg] <= 4096.0
            for sg in range(5):                               # the smallest exponent >= 0 that brings the segment's maximum to 2^12 or below
                want = max(0, int(np.ceil(np.log2(st["seg_amax"][sg] / 4096.0))))
                assert st["exps"][sg] == (min(want, 6) if sg in (1, 3) else want), (sg, st)    # (attention segments: capped at 6, include/pmp.h)
            assert st["exps"][0] == base["exps"][0] + 6 or base["seg_amax"][0] <= 4096
    finally:
        e.close()


def test_attention_trunk_beyond_its_scale_cap_falls_back_to_the_guard():
    """The attention segments' exponent is capped at 6 (their input, O(1) logits, must stay out of fp16's subnormals).  A 2^18 gain inside an
    attention trunk is therefore NOT absorbed: the trunk leaves the fp16 range, the flag fires, the call is re-run on the fp32 MFMA datapath -
    and the logits are still the golden ones."""
    from pmp_vvc_tip2023_amd import engine, synth
    g1, g2b = golden("g1_qt.npz"), golden("g2b_msbd_trained_like.npz")
    e = engine.Engine(0, allow_synthetic_mtt=True)
    try:
        e.load("Luma", 22, msbd_weights=trained_like.msbd_weights("Luma", 22, att_gain=2.0 ** 18))
        rep = e.activation_report("Luma", 22)
        assert rep["exps"][1] == 6 and rep["exps"][3] == 6 and rep["seg_amax"][1] > 65504 * 64
        e.clear_saturation()
        qt, bt, dire = e.inference_pre_QBD("Luma", 22, g1["block_y"])
        assert _golden_err(bt, dire, g2b, "Luma", 22) < TOL
        assert e.saturation_reruns() == 1 and e.saturated()
    finally:
        e.close()


def test_manifest_exponents_skip_the_calibration(tmp_path):
    """A .pmpw MTT file that carries "act_exp" (tools/calibrate_pmpw.py writes it once per model directory) is loaded WITHOUT a calibration
    pass - the activation report has the file's exponents and no recorded tensors - and gives the bits the calibrated load gives."""
    import shutil
    from pmp_vvc_tip2023_amd import engine, synth, weights as W
    g1 = golden("g1_qt.npz")
    w = trained_like.msbd_weights("Luma", 22, trunk_gain=64.0)
    e1 = engine.Engine(0)
    try:
        e1.load("Luma", 22, msbd_weights=w)
        rep = e1.activation_report("Luma", 22)
        assert len(rep["tensors"]) == 49 and rep["exps"][0] >= 6
        a = e1.inference_pre_QBD("Luma", 22, g1["block_y"])
    finally:
        e1.close()
    shutil.copy(os.path.join(W.default_weight_dir(), "Luma_Q_22.pmpw"), tmp_path)
    W.save_pmpw(str(tmp_path / "Luma_BD_22.pmpw"), "Luma_MSBD", 22, w, source="test", act_exp=rep["exps"])
    e2 = engine.Engine(0, weight_dir=str(tmp_path))
    try:
        e2.load("Luma", 22)
        assert e2.provenance[("Luma_MSBD", 22)].endswith("Luma_BD_22.pmpw")
        rep2 = e2.activation_report("Luma", 22)
        assert rep2["exps"] == rep["exps"] and rep2["tensors"] == []
        b = e2.inference_pre_QBD("Luma", 22, g1["block_y"])
        assert e2.saturation_reruns() == 0
    finally:
        e2.close()
    for p, q in zip(a, b):
        assert np.array_equal(p, q)


def test_stale_manifest_exponents_are_ignored_and_a_new_qt_partner_recalibrates(tmp_path):
    """nothing tied a manifest's "act_exp" to the tensors it was calibrated on.  Now "act_fp" does (fingerprints of the MTT
    tensors and of the QT partner): a manifest written for ANOTHER QT net, or whose tensors were edited afterwards, is ignored in favour
    of a calibration pass - in both loading orders - while the matching one is still taken without a pass; and replacing only the QT net
    of a calibrated pair drops the exponents that were derived with the old one (overflow would merely cost fp32 re-runs; underflow
    would be silent)."""
    import ctypes as C
    import shutil
    from pmp_vvc_tip2023_amd import _lib, engine, synth, weights as W
    w = trained_like.msbd_weights("Luma", 22, trunk_gain=64.0)
    wq22, _ = W.load_net_weights("Luma_Q", 22)
    wq37, _ = W.load_net_weights("Luma_Q", 37)
    e = engine.Engine(0)
    try:
        e.load("Luma", 22, msbd_weights=w)
        good = e.activation_report("Luma", 22)["exps"]
        fp = C.c_uint64()
        assert e.lib.pmp_weights_fingerprint(e.h, _lib.NET_IDS["Luma_MSBD"], 22, C.byref(fp)) == 0 and fp.value == W.fingerprint(w)
        assert e.lib.pmp_weights_fingerprint(e.h, _lib.NET_IDS["Luma_Q"], 22, C.byref(fp)) == 0 and fp.value == W.fingerprint(wq22)
        assert e.lib.pmp_weights_fingerprint(e.h, _lib.NET_IDS["Luma_Q"], 27, C.byref(fp)) == -3
        # the pair with QP37's QT tensors standing in for QP22's: different logits into the MTT stem and attention inputs
        e.load_pretrain_model("Luma_Q", 22, wq37)
        rep37 = e.activation_report("Luma", 22)
        assert len(rep37["tensors"]) == 49, "replacing the QT net did not re-calibrate its MTT partner"
        e.load_pretrain_model("Luma_Q", 22, wq37)                    # the same tensors again: nothing to redo
        assert e.activation_report("Luma", 22)["exps"] == rep37["exps"]
    finally:
        e.close()
    wrong = [min(x + 3, 30) for x in good[:1]] + [0, good[2] + 2, 0, good[4] + 1]     # recognisably not what a calibration chooses
    shutil.copy(os.path.join(W.default_weight_dir(), "Luma_Q_22.pmpw"), tmp_path)
    bd = str(tmp_path / "Luma_BD_22.pmpw")
    for order in ("qt_first", "mtt_first"):
        for case, qt_partner, expect_file in (("matching", wq22, True), ("other_qt", wq37, False)):
            W.save_pmpw(bd, "Luma_MSBD", 22, w, source="test", act_exp=wrong, qt_partner=qt_partner)
            e2 = engine.Engine(0, weight_dir=str(tmp_path))
            try:
                nets = [("Luma_Q", str(tmp_path / "Luma_Q_22.pmpw")), ("Luma_MSBD", bd)]
                for net, path in (nets if order == "qt_first" else nets[::-1]):
                    assert e2.lib.pmp_load_weights_file(e2.h, _lib.NET_IDS[net], 22, path.encode()) == 0, e2.lib.pmp_last_error(e2.h)
                rep = e2.activation_report("Luma", 22)
                if expect_file:
                    assert rep["exps"] == wrong and rep["tensors"] == [], (order, case)
                else:
                    assert rep["exps"] == good and len(rep["tensors"]) == 49, (order, case, rep["exps"])
            finally:
                e2.close()
    # tensors edited after the manifest was written: its own fingerprint no longer matches
    man, tens = W.load_pmpw(bd)
    W.save_pmpw(bd, "Luma_MSBD", 22, w, source="test", act_exp=wrong, qt_partner=wq22)
    raw = bytearray(open(bd, "rb").read())
    raw[-4:] = np.float32(0.125).tobytes()                          # the last float of the payload
    open(bd, "wb").write(bytes(raw))
    e3 = engine.Engine(0, weight_dir=str(tmp_path))
    try:
        e3.load("Luma", 22)
        rep = e3.activation_report("Luma", 22)
        assert rep["exps"] != wrong and len(rep["tensors"]) == 49
    finally:
        e3.close()
    # exponents beyond the reader's bounds never reach the kernels (attention segment > 6; trunk segment > 30)
    for bad in ([0, 7, 0, 0, 0], [31, 0, 0, 0, 0]):
        with pytest.raises(ValueError):
            W.save_pmpw(bd, "Luma_MSBD", 22, w, act_exp=bad)


def test_calibrate_pmpw_tool_writes_the_exponents(tmp_path):
    """tools/calibrate_pmpw.py on a model directory (real QT files + trained-like MTT files without exponents): every MTT file comes back
    with "act_exp" in its manifest, equal to what a calibrating load reports, and with its tensors untouched."""
    import shutil
    import sys
    from pmp_vvc_tip2023_amd import engine, synth, weights as W
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import calibrate_pmpw
    for comp, qp in (("Luma", 22), ("Chroma", 37)):
        shutil.copy(os.path.join(W.default_weight_dir(), "%s_Q_%d.pmpw" % (comp, qp)), tmp_path)
        W.save_pmpw(str(tmp_path / ("%s_BD_%d.pmpw" % (comp, qp))), comp + "_MSBD", qp, trained_like.msbd_weights(comp, qp), source="test")
    done = calibrate_pmpw.calibrate_dir(str(tmp_path), 0, log=lambda *a: None)
    assert len(done) == 2
    e = engine.Engine(0)
    try:
        for comp, qp in (("Luma", 22), ("Chroma", 37)):
            man, tens = W.load_pmpw(str(tmp_path / ("%s_BD_%d.pmpw" % (comp, qp))))
            ref = trained_like.msbd_weights(comp, qp)
            assert list(tens) == list(ref) and all(np.array_equal(tens[k], ref[k]) for k in ref)
            e.load(comp, qp, msbd_weights=ref)                     # a calibrating load of the same tensors
            assert man["act_exp"] == e.activation_report(comp, qp)["exps"]
            wq, _ = W.load_net_weights(comp + "_Q", qp)
            assert man["act_fp"] == ["%016x" % W.fingerprint(ref), "%016x" % W.fingerprint(wq)]
    finally:
        e.close()


def test_acceptance_tool_end_to_end(tmp_path):
    """tools/accept_bd_weights.py, the whole flow: reference-format pickles named as the trained MTT files will be
    -> .pmpw -> calibrated manifests (exponents + fingerprints) -> fresh blocks on all three datapaths against the oracle -> JSON verdict."""
    import sys
    import torch as _t
    from pmp_vvc_tip2023_amd import weights as W
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import accept_bd_weights as A
    src = tmp_path / "CTU_Models"
    src.mkdir()
    for comp, qp in (("Luma", 22), ("Chroma", 37)):
        for net, tens in ((comp + "_Q", W.load_net_weights(comp + "_Q", qp)[0]), (comp + "_BD", trained_like.msbd_weights(comp, qp))):
            _t.save({"module." + k: _t.from_numpy(v.copy()) for k, v in tens.items()}, str(src / ("%s_%d.pkl" % (net, qp))),
                    _use_new_zipfile_serialization=False)
    v = A.accept(str(src), blocks=48, log=lambda *a: None)
    assert v["ok"] and v["on_default_datapath"], v
    assert set(v["pairs"]) == {"Luma QP22", "Chroma QP37"} and set(v["calibrate"]) == {"Luma_BD_22.pmpw", "Chroma_BD_37.pmpw"}
    p = v["pairs"]["Luma QP22"]
    assert p["manifest_act_exp"] == p["calibration"]["exps"] == p["datapaths"]["f16x3"]["activation_exps"] and len(p["manifest_act_fp"]) == 2
    assert len(p["calibration"]["tensor_amax"]) == 49 and max(p["calibration"]["segment_amax"]) > 1e3      # trained scale, recorded per tensor
    for prec in ("f16x3", "bf16x6", "fp32"):
        d = p["datapaths"][prec]
        assert d["within_tolerance"] and d["flags_bit_exact_on_device_logits"] and d["saturation_reruns"] == 0 and d["blocks_over_tolerance"] == 0
