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
