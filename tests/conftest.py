import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tools") not in sys.path:      # tools/trained_like.py (test-weight data), tools/ref_harness.py
    sys.path.insert(1, os.path.join(ROOT, "tools"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_path(name):
    return os.path.join(GOLDEN, name)


def g3_sets():
    """Yield (tag, cf, qt f32[n,8,8], bt f32[n,3,16,16], dire, hor, ver, dout, leaves) from g3_m2p.npz."""
    g = golden("g3_m2p.npz")
    for cf in (1, 2):
        yield ("quant", cf, g["q_qt_cf%d" % cf].astype(np.float32), (g["q_bt64_cf%d" % cf] / 64.0).astype(np.float32),
               (g["q_dire64_cf%d" % cf] / 64.0).astype(np.float32), g["q_hor_cf%d" % cf], g["q_ver_cf%d" % cf],
               g["q_dout_cf%d" % cf], g["q_leaves_cf%d" % cf])
        for t in ("r", "a", "t"):
            yield ({"r": "raw", "a": "adversarial", "t": "large-trees"}[t], cf, g["%s_qt_cf%d" % (t, cf)].astype(np.float32),
                   g["%s_bt_cf%d" % (t, cf)], g["%s_dire_cf%d" % (t, cf)], g["%s_hor_cf%d" % (t, cf)],
                   g["%s_ver_cf%d" % (t, cf)], g["%s_dout_cf%d" % (t, cf)], g["%s_leaves_cf%d" % (t, cf)])


def g3b_sets():
    """Yield (cf, qt RAW logits f32[n,8,8], bt, dire, tags, fixed f32[n,8,8] (NaN where the reference keeps one), q8, hor, ver, dout, leaves)
    from g3b_m2p_range.npz: the value range the nets can emit - |logit| up to 3e38, +-inf, NaN (tools/gen_golden.py gen_g3b)."""
    g = golden("g3b_m2p_range.npz")
    for cf in (1, 2):
        yield (cf,) + tuple(g["%s_cf%d" % (k, cf)] for k in ("qt", "bt", "dire", "tag", "fixed", "q8", "hor", "ver", "dout", "leaves"))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import postproc
    postproc.build()
    return postproc
