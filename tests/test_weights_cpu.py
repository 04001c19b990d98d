"""Weight import without a GPU: the reference's pickles (Inference_QBD.py:28-46, trained_models/*.pkl) against the product's containers.
The trained MTT nets (*_BD_*.pkl) are missing from the reference checkout, so the .pkl path for BD-named files is rehearsed on
files written here in the reference's own formats - legacy (non-zip) torch pickles, DataParallel `module.` prefixes, with and without the
{"state_dict": ...} wrapper its loader accepts - from the documented synthetic state_dict.  When the real files appear they drop in by name."""
import os
import shutil

import numpy as np
import pytest

from pmp_vvc_tip2023_amd import synth, weights as W

REF_MODELS = "/root/reference/trained_models"


def _write_ref_pkl(path, tensors, wrapper, legacy):
    import torch
    sd = {"module." + k: torch.from_numpy(v.copy()) for k, v in tensors.items()}
    obj = {"state_dict": sd, "epoch": 7} if wrapper else sd
    torch.save(obj, path, _use_new_zipfile_serialization=not legacy)


@pytest.mark.parametrize("comp", ["Luma", "Chroma"])
@pytest.mark.parametrize("wrapper", [False, True])
@pytest.mark.parametrize("legacy", [True, False])
def test_bd_named_pkl_resolves_and_round_trips(tmp_path, comp, wrapper, legacy):
    want = synth.synth_msbd_weights(comp, 22)
    p = tmp_path / ("%s_BD_22.pkl" % comp)
    _write_ref_pkl(str(p), want, wrapper, legacy)
    assert W.find_net_weights(comp + "_MSBD", 22, str(tmp_path)) == ("pkl", str(p))       # no allow_synthetic needed: the file exists
    got, src = W.load_net_weights(comp + "_MSBD", 22, str(tmp_path))
    assert src == str(p) and list(got) == list(want)                                      # names with `module.` stripped, order kept
    for k in want:
        assert got[k].dtype == np.float32 and got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), k
    # the other QPs still have no file: an error unless synthetic weights are asked for, as the CLI does with --allowSyntheticMTT
    with pytest.raises(FileNotFoundError):
        W.find_net_weights(comp + "_MSBD", 27, str(tmp_path))
    assert W.find_net_weights(comp + "_MSBD", 27, str(tmp_path), allow_synthetic=True) == ("synthetic", None)


def test_pmpw_wins_over_pkl_and_pkl_is_the_fallback(tmp_path):
    """Resolution order (weights.find_net_weights): <Comp>_{Q,BD}_<qp>.pmpw, then .pkl."""
    a = synth.synth_msbd_weights("Luma", 22)
    b = {k: v + np.float32(1) for k, v in a.items()}
    _write_ref_pkl(str(tmp_path / "Luma_BD_22.pkl"), b, False, True)
    W.save_pmpw(str(tmp_path / "Luma_BD_22.pmpw"), "Luma_MSBD", 22, a, source="test")
    got, src = W.load_net_weights("Luma_MSBD", 22, str(tmp_path))
    assert src.endswith(".pmpw") and all(np.array_equal(got[k], a[k]) for k in a)
    os.remove(tmp_path / "Luma_BD_22.pmpw")
    got, src = W.load_net_weights("Luma_MSBD", 22, str(tmp_path))
    assert src.endswith(".pkl") and all(np.array_equal(got[k], b[k]) for k in a)


def test_model_dir_with_bd_pickles_is_a_complete_model_dir(tmp_path):
    """What a user with the trained files does: --modelDir <dir holding Luma_Q_22.* and Luma_BD_22.pkl>.  The driver's resolver accepts the
    directory and both nets of the pass resolve to files - no synthetic weights involved."""
    from pmp_vvc_tip2023_amd import inference_qbd as D
    shutil.copy(os.path.join(W.default_weight_dir(), "Luma_Q_22.pmpw"), tmp_path / "Luma_Q_22.pmpw")
    _write_ref_pkl(str(tmp_path / "Luma_BD_22.pkl"), synth.synth_msbd_weights("Luma", 22), True, True)
    d = D.resolve_model_dir(str(tmp_path))
    assert W.find_net_weights("Luma_Q", 22, d)[0] == "pmpw" and W.find_net_weights("Luma_MSBD", 22, d)[0] == "pkl"


def test_acceptance_tool_converts_and_checks_a_bd_model_dir(tmp_path):
    """tools/accept_bd_weights.py, the part that needs no GPU (--convert-only): a directory as a user of the reference would have it -
    reference-format pickles named <Comp>_{Q,BD}_<qp>.pkl, here the real QT tensors and the trained-like MTT tensors - becomes a
    directory of .pmpw files with the same bits, its (QT, MTT) pairs are found, and a file with a missing tensor, a wrong shape or a
    NaN is named in the verdict instead of reaching the library."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import accept_bd_weights as A
    import trained_like
    src = tmp_path / "CTU_Models"
    src.mkdir()
    tl = {}
    for comp, qp in (("Luma", 22), ("Chroma", 37)):
        wq, _ = W.load_net_weights(comp + "_Q", qp)
        tl[comp] = trained_like.msbd_weights(comp, qp)
        _write_ref_pkl(str(src / ("%s_Q_%d.pkl" % (comp, qp))), wq, False, True)
        _write_ref_pkl(str(src / ("%s_BD_%d.pkl" % (comp, qp))), tl[comp], True, True)
    (src / "notes.pkl").write_bytes(b"not a model file")                      # skipped by name
    v = A.accept(str(src), blocks=8, log=lambda *a: None, convert_only=True)
    assert v["ok"] and v["convert"]["pairs"] == ["Luma QP22", "Chroma QP37"] and v["convert"]["problems"] == []
    man, got = W.load_pmpw(str(src / "pmpw" / "Chroma_BD_37.pmpw"))
    assert man["net"] == "Chroma_MSBD" and man["qp"] == 37 and all(np.array_equal(got[k], tl["Chroma"][k]) for k in tl["Chroma"])
    # broken inputs
    bad = dict(tl["Luma"]); del bad["trunk_M2.3.left.2.weight"]
    bad["conv_B2.weight"] = bad["conv_B2.weight"][:, :4]
    bad["conv_b1_1.bias"] = np.full_like(bad["conv_b1_1.bias"], np.nan)
    _write_ref_pkl(str(src / "Luma_BD_22.pkl"), bad, False, True)
    os.remove(src / "Chroma_Q_37.pkl"); os.remove(src / "pmpw" / "Chroma_Q_37.pmpw")
    v = A.accept(str(src), blocks=8, log=lambda *a: None, convert_only=True)
    assert not v["ok"]
    pr = {os.path.basename(p.get("file", p.get("pair", ""))): p for p in v["convert"]["problems"]}
    assert pr["Luma_BD_22.pmpw"]["missing"] == ["trunk_M2.3.left.2.weight"] and pr["Luma_BD_22.pmpw"]["wrong_shape"] == ["conv_B2.weight"]
    assert pr["Luma_BD_22.pmpw"]["non_finite"] == ["conv_b1_1.bias"] and "only Chroma_MSBD" in pr["Chroma QP37"]["error"]


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="the reference checkout is only present in the build container")
@pytest.mark.parametrize("comp", ["Luma", "Chroma"])
@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_shipped_pmpw_equals_the_reference_pickle(comp, qp):
    """The eight QT containers under weights/ ARE the reference's trained_models/*.pkl (tools/convert_weights.py): same names, same bits."""
    ref = W.load_pkl(os.path.join(REF_MODELS, "%s_Q_%d.pkl" % (comp, qp)))
    man, got = W.load_pmpw(os.path.join(W.default_weight_dir(), "%s_Q_%d.pmpw" % (comp, qp)))
    assert man["net"] == comp + "_Q" and man["qp"] == qp and set(got) == set(ref) and len(ref) == 20
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
