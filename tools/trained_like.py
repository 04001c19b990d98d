"""TRAINED-LIKE MTT-net weights: test-weight data, NOT product code (round 6: moved out of pmp_vvc_tip2023_amd/synth.py together with
trained_like_scales.json).  The reference's *_BD_*.pkl are absent from the mount; the uniform synthetic MTT weights of
synth.synth_msbd_weights are benign; these stand in for a net with trained-scale activations in tests/, bench.py's extra.trained_like,
tools/gen_golden.py (G2b) and the campaigns.  numpy only.  Import with tools/ on sys.path: `import trained_like`."""
import json
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from pmp_vvc_tip2023_amd import weights as W                                        # noqa: E402
from pmp_vvc_tip2023_amd.synth import _name_seed, msbd_tensor_shapes, splitmix64    # noqa: E402

# ------------------------------------------------------------------------- trained-LIKE MTT-net weights
# The uniform weights above are BENIGN: activations stay O(1..4), the weight histograms are flat.  The trained QT nets are neither
# (activations to 3e3 on 8-bit content, per-tensor kurtosis 3..100, gains rms*sqrt(fan_in) of 1..4), and the MTT nets multiply two
# unbounded ReLU tensors twice (Model_QBD.py:143,150).  msbd_weights() builds MTT tensors out of the REAL
# {Luma,Chroma}_Q_<qp> tensors in the tree:
#   1. bootstrap (seeded resampling, pure integer indexing): every output channel of an MTT conv tensor picks a donor output channel of
#      a real tensor with the same kernel size and fills its input slices from that donor's row, so the k x k filters, their
#      magnitudes and the per-channel structure are the trained ones (the stems' 5x9 / 9x5 / 3x5 / 5x3 kernels are centre crops of real
#      9x9 / 5x5 stem filters; the plane that carries the QT logits gets a real filter too, see _TL_Q_STEM);
#   2. one scalar per tensor (trained_like_scales.json, written by tools/calibrate_trained_like.py in the build container from the
#      reference's own modules on recipe-R blocks) so that the trunks run at the QT nets' activation range (stem max ~ 4e2, x4 / x5
#      max ~ 1e3..2e3), both attention gates have rms 1, and the heads spread over the depth / direction ranges Map2Partition works on.
# The table keeps the weights bit-reproducible on every box (resampling is integer work, the scale one float32 multiply): the golden
# logits tests/golden/g2b_msbd_trained_like.npz were produced by the reference modules holding exactly these tensors.
# Stress knobs, exact for powers of two (the nets are bias-free behind the stems and ReLU is positively homogeneous, so the oracle's
# logits do not change): trunk_gain K multiplies the stems (trunk activations x K), gate_gain G the last block of both attention trunks
# (gates x G); the heads are divided by K (conv_B1) and K*G (conv_B2, conv_B3).
# How hard the MTT nets lean on the raw QT logits q they take as input (a stem plane, channel 0 of both attention trunks).  q arrives with
# the QT net's own error (Luma_Q: up to 8e-4 between any two fp32 evaluations), and north_star's 1e-3 on the MTT logits
# can only hold end to end if the MTT net does not amplify it: with a real stem filter x 32 on the q plane (to weigh 0..3 against 0..255)
# and unweighted attention inputs the bootstrapped nets amplified a perturbation of q 10..30x, measured with the oracle; with the
# two factors below 0.5..1.7x, as the uniform nets (0.8..2.1x).
_TL_Q_STEM = 1.0
_TL_Q_ATT = 0.125
_TL_SCALES = None


def _tl_scales():
    global _TL_SCALES
    if _TL_SCALES is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "trained_like_scales.json")) as f:
            _TL_SCALES = json.load(f)
    return _TL_SCALES


def _tl_pools(qp, weight_dir=None):
    """Real conv tensors of both QT nets at this QP, by kernel size: {(kh, kw): [ndarray OIHW, ...]} (+ the two stem bias vectors)."""
    pools, biases = {}, []
    for comp in ("Luma", "Chroma"):
        w, _ = W.load_net_weights(comp + "_Q", qp, weight_dir)
        for name in sorted(w):
            a = w[name]
            if a.ndim == 4:
                pools.setdefault((a.shape[2], a.shape[3]), []).append(a)
            elif name == "conv_q1.bias":
                biases.append(a)
    return pools, np.concatenate(biases)


def _tl_draw(seed, name, n, hi):
    """n integers in [0, hi) from SplitMix64 (no numpy Generator: its streams are not a stable contract)."""
    return (splitmix64(_name_seed(seed, name), n) % np.uint64(hi)).astype(np.int64)


def _tl_resample(pool, shape, seed, name):
    """OIHW tensor of `shape` bootstrapped from the tensors in `pool` (same kernel size): per output channel one donor row."""
    co, ci = shape[0], shape[1]
    rows = [(t, o) for t in range(len(pool)) for o in range(pool[t].shape[0])]
    pick = _tl_draw(seed, name + "/row", co, len(rows))
    out = np.empty(shape, np.float32)
    for o in range(co):
        t, r = rows[int(pick[o])]
        src = pool[t][r]                                      # [ci_donor][kh][kw]
        idx = _tl_draw(seed, "%s/in%d" % (name, o), ci, src.shape[0])
        out[o] = src[idx]
    return out


def _crop(a, kh, kw):
    """Centre crop of the last two axes to (kh, kw)."""
    h0, w0 = (a.shape[-2] - kh) // 2, (a.shape[-1] - kw) // 2
    return a[..., h0:h0 + kh, w0:w0 + kw]


def raw(comp, qp, weight_dir=None):
    """Step 1 above: the bootstrapped tensors before the per-tensor scale (tools/calibrate_trained_like.py starts from these)."""
    pools, stem_bias = _tl_pools(qp, weight_dir)
    luma = comp == "Luma"
    ks = 9 if luma else 5
    stem_pool = [t for t in pools[(ks, ks)] if t.shape[1] <= 3]          # conv_q1 of this component's QT net: pixel planes
    out = {}
    for name, shape in msbd_tensor_shapes(comp):
        key = comp + "/" + name
        if name.endswith(".bias"):
            if name.startswith("conv_b1_"):
                out[name] = stem_bias[_tl_draw(qp, key, shape[0], stem_bias.size)].astype(np.float32)
            else:
                out[name] = np.zeros(shape, np.float32)                  # head biases: set by the calibration table
            continue
        if name.startswith("conv_b1_"):
            co, ci, kh, kw = shape
            full = _tl_resample(stem_pool, (co, ci, ks, ks), qp, key)    # pixel planes: real stem filters
            logit = _tl_resample(stem_pool, (co, 1, ks, ks), qp, key + "/q")
            full[:, ci - 1] = logit[:, 0] * np.float32(_TL_Q_STEM)         # last plane = the upsampled QT logits (Model_QBD.py:131)
            out[name] = np.ascontiguousarray(_crop(full, kh, kw))
        else:
            out[name] = _tl_resample(pools[(shape[2], shape[3])], shape, qp, key)
            if name in ("trunk_Att1.0.left.0.weight", "trunk_Att1.0.shortcut.0.weight",
                        "trunk_Att2.0.left.0.weight", "trunk_Att2.0.shortcut.0.weight"):
                out[name][:, 0] *= np.float32(_TL_Q_ATT)                   # input channel 0 of the attention trunks = up(q) (:140, :147)
    return out


def msbd_weights(comp, qp, trunk_gain=1.0, gate_gain=1.0, att_gain=1.0, weight_dir=None):
    """Trained-like MTT-net weights (see the block comment above).  trunk_gain / gate_gain / att_gain must be powers of two.
    att_gain A multiplies the FIRST block of both attention trunks (left.0 and shortcut: everything inside the attention trunks x A, the
    gates with it), undone in conv_B2 / conv_B3 like gate_gain."""
    for g in (trunk_gain, gate_gain, att_gain):
        m, _ = np.frexp(float(g))
        if g <= 0 or m != 0.5:
            raise ValueError("trained_like.msbd_weights: gains must be powers of two (exact in fp32)")
    tab = _tl_scales()[comp][str(qp)]
    raw_t = raw(comp, qp, weight_dir)
    K, G, A = np.float32(trunk_gain), np.float32(gate_gain), np.float32(att_gain)
    out = {}
    for name, a in raw_t.items():
        t = tab[name]
        w = np.asarray(t, np.float32).reshape(a.shape) if isinstance(t, list) else (a * np.float32(t)).astype(np.float32)
        if name.startswith("conv_b1_"):
            w = w * K
        elif name in ("trunk_Att1.1.left.2.weight", "trunk_Att1.1.shortcut.0.weight",
                      "trunk_Att2.1.left.2.weight", "trunk_Att2.1.shortcut.0.weight"):
            w = w * G
        elif name in ("trunk_Att1.0.left.0.weight", "trunk_Att1.0.shortcut.0.weight",
                      "trunk_Att2.0.left.0.weight", "trunk_Att2.0.shortcut.0.weight"):
            w = w * A
        elif name == "conv_B1.weight":
            w = w / K
        elif name in ("conv_B2.weight", "conv_B3.weight"):
            w = w / (K * G * A)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out
