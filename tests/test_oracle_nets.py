"""CPU: the oracle's torch restatement of the four nets against golden vectors produced by the imported
reference modules (tools/gen_golden.py).  Tolerance 1e-4 (same arithmetic library, different call graph);
the product's tolerance against these same vectors is north_star's 1e-3."""
import numpy as np
import pytest

import conftest  # noqa: F401 - puts tools/ on sys.path
import trained_like  # tools/trained_like.py: test-weight data (round 6: out of the product package)
import torch

from conftest import golden
from oracle import nets_torch as O
from pmp_vvc_tip2023_amd import synth, weights as W

TOL = 1e-4


@pytest.mark.parametrize("comp", ["Luma", "Chroma"])
@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_q_net_real_weights(comp, qp):
    g = golden("g1_qt.npz")
    luma = comp == "Luma"
    x = O.luma_input(g["block_y"]) if luma else O.chroma_input(g["block_y"], g["block_u"], g["block_v"])
    wq, src = W.load_net_weights(comp + "_Q", qp)
    assert src.endswith(".pmpw")
    with torch.no_grad():
        q = O.q_forward(wq, x, luma).numpy()
    assert q.shape == (16, 1, 8, 8)
    assert np.abs(q - g["qt_%s_%d" % (comp, qp)]).max() < TOL


@pytest.mark.parametrize("comp", ["Luma", "Chroma"])
@pytest.mark.parametrize("qp", [22, 37])
def test_msbd_net_synth_weights(comp, qp):
    g1, g2 = golden("g1_qt.npz"), golden("g2_msbd.npz")
    luma = comp == "Luma"
    x = (O.luma_input(g1["block_y"]) if luma else O.chroma_input(g1["block_y"], g1["block_u"], g1["block_v"]))[:8]
    q = torch.from_numpy(g1["qt_%s_%d" % (comp, qp)][:8])
    wbd = synth.synth_msbd_weights(comp, qp)
    taps = {}
    with torch.no_grad():
        o = O.msbd_forward(wbd, x, q, luma, taps=taps)
    for i in range(3):
        assert np.abs(o[i].numpy() - g2["out%d_%s_%d" % (i, comp, qp)]).max() < TOL
    if qp == 22:
        # in-place accumulation order (Model_QBD.py:146-147): out1.ch0 = raw + out0.ch0, ch1 untouched
        raw = taps["out1_raw"].numpy()
        assert np.abs(raw[:2] - g2["out1_raw_%s" % comp]).max() < TOL
        assert np.allclose(o[1].numpy()[:, 0], raw[:, 0] + o[0].numpy()[:, 0], atol=1e-6)
        assert np.array_equal(o[1].numpy()[:, 1], raw[:, 1])


@pytest.mark.parametrize("comp", ["Luma", "Chroma"])
def test_infer_qbd_regrouping(comp):
    """Metrics.py:399-402: bt = ch0 of the three heads, dire = ch1."""
    g1, g2 = golden("g1_qt.npz"), golden("g2_msbd.npz")
    luma = comp == "Luma"
    x = (O.luma_input(g1["block_y"]) if luma else O.chroma_input(g1["block_y"], g1["block_u"], g1["block_v"]))[:8]
    wq, _ = W.load_net_weights(comp + "_Q", 22)
    wbd = synth.synth_msbd_weights(comp, 22)
    qt, bt, dire = O.infer_qbd(wq, wbd, x, luma, batch=3)   # ragged batches: 3+3+2
    assert np.abs(qt - g2["pre_qt_%s" % comp]).max() < TOL
    assert np.abs(bt - g2["pre_bt_%s" % comp]).max() < TOL
    assert np.abs(dire - g2["pre_dire_%s" % comp]).max() < TOL


def test_synth_weights_are_deterministic_and_complete():
    for comp in ("Luma", "Chroma"):
        a = synth.synth_msbd_weights(comp, 22)
        b = synth.synth_msbd_weights(comp, 22)
        assert len(a) == 72
        assert all(np.array_equal(a[k], b[k]) for k in a)
        n = sum(v.size for v in a.values())
        assert n == (1075670 if comp == "Luma" else 1074198)   # SURVEY.md A.2 parameter counts
    # PRNG known answers (SplitMix64 reference values for seed 0)
    z = synth.splitmix64(0, 3)
    assert [int(v) for v in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


@pytest.mark.parametrize("comp", ["Luma", "Chroma"])
@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_msbd_net_trained_like_weights(comp, qp):
    """G2b: the reference's MTT modules holding trained_like.msbd_weights (bootstrapped from the real QT tensors; trunks at 1e3).
    The oracle on the same tensors, the activation maxima that ride along, and the power-of-two stress variants, which must not
    change a single bit of the logits (the heads undo the gains exactly)."""
    g1, g2b = golden("g1_qt.npz"), golden("g2b_msbd_trained_like.npz")
    luma = comp == "Luma"
    x = O.luma_input(g1["block_y"]) if luma else O.chroma_input(g1["block_y"], g1["block_u"], g1["block_v"])
    q = torch.from_numpy(g1["qt_%s_%d" % (comp, qp)])
    wbd = trained_like.msbd_weights(comp, qp)
    assert len(wbd) == 72 and all(v.dtype == np.float32 for v in wbd.values())
    taps = {}
    with torch.no_grad():
        o = O.msbd_forward(wbd, x, q, luma, taps=taps)
        o2 = O.msbd_forward(trained_like.msbd_weights(comp, qp, trunk_gain=64.0, gate_gain=16.0, att_gain=1024.0), x, q, luma)
    for i in range(3):
        assert np.abs(o[i].numpy() - g2b["out%d_%s_%d" % (i, comp, qp)]).max() < TOL
        assert torch.equal(o[i], o2[i])
    amax = g2b["amax_%s_%d" % (comp, qp)]                     # x3 x4 x5 att0 att1 xb1 xb3
    got = [taps[k].abs().max().item() for k in ("x3", "x4", "x5", "x_att0", "x_att1")]
    assert np.allclose(got, amax[:5], rtol=1e-3)
    assert 300 < amax[1] < 3000 and 300 < amax[2] < 3000      # the trunks run where the trained QT nets run (SURVEY: 3e3 on 8-bit content)
    with pytest.raises(ValueError):
        trained_like.msbd_weights(comp, qp, gate_gain=3.0)


def test_trained_like_scale_table_matches_its_generator():
    """tools/trained_like_scales.json is the output of tools/calibrate_trained_like.py on trained_like.raw(): a change to
    the bootstrap without a regenerated table (or the other way round) must not go unnoticed.  One net re-calibrated here (torch CPU
    convolutions: equal to the committed scalars up to the arithmetic of this host's convolution library)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import calibrate_trained_like as C
    c, outs = C.calibrate("Chroma", 37)
    table = json.load(open(os.path.join(root, "tools", "trained_like_scales.json")))["Chroma"]["37"]
    assert set(c.scale) == set(table) and len(table) == 72
    for name, want in table.items():
        got = c.scale[name]
        assert np.allclose(got, want, rtol=2e-3, atol=1e-4), (name, got, want)
    assert 1500 < c.stats["x3"][0] < 2500 and 0.5 < c.stats["trunk_Att1.gate"][1] < 2.0
