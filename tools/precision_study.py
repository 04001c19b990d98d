"""Precision study (run in the build container; imports the reference): which MFMA datapaths keep the
logits within the 1e-3 tolerance of north_star?  Emulates bf16-split products exactly on the CPU
(bf16*bf16 products are exact in fp32, accumulation is fp32 like the MFMA accumulator).
Usage: python tools/precision_study.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import ref_harness as R
from pmp_vvc_tip2023_amd import synth
from oracle import nets_torch as O


def split(x, terms):
    parts, r = [], x
    for _ in range(terms):
        h = r.to(torch.bfloat16).float()
        parts.append(h)
        r = r - h
    return parts


def make_conv(xa, wa, pairs):
    """pairs: list of (i,j) products x_i * w_j to accumulate."""
    def conv(x, w, b, pad):
        xs, ws = split(x, xa), split(w, wa)
        out = None
        for (i, j) in pairs:
            t = F.conv2d(xs[i], ws[j], None, padding=pad)
            out = t if out is None else out + t
        if b is not None:
            out = out + b.view(1, -1, 1, 1)
        return out
    return conv


def split16(x, terms):
    parts, r = [], x
    for _ in range(terms):
        h = r.to(torch.float16).float()     # subnormals kept, as the MFMA does (tools/probe/f16_denorm.hip)
        parts.append(h)
        r = r - h
    return parts


def make_conv_f16(scale, pairs):
    """Two-term fp16 split of x and of scale*w; the result is divided by scale (a power of two: exact)."""
    def conv(x, w, b, pad):
        xs, ws = split16(x, 2), split16(w * scale, 2)
        out = None
        for (i, j) in pairs:
            t = F.conv2d(xs[i], ws[j], None, padding=pad)
            out = t if out is None else out + t
        out = out / scale
        if b is not None:
            out = out + b.view(1, -1, 1, 1)
        return out
    return conv


MODES = {
    "f16x3 unscaled": make_conv_f16(1.0, [(0, 1), (1, 0), (0, 0)]),
    "f16x3 weights x256": make_conv_f16(256.0, [(0, 1), (1, 0), (0, 0)]),
    "bf16x1": make_conv(1, 1, [(0, 0)]),
    "bf16x3 (hh,hl,lh)": make_conv(2, 2, [(1, 0), (0, 1), (0, 0)]),
    "bf16x4 (+ll)": make_conv(2, 2, [(1, 1), (1, 0), (0, 1), (0, 0)]),
    "bf16x6 (3-term)": make_conv(3, 3, [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]),
    "fp64": None,
}

if __name__ == "__main__":
    torch.set_num_threads(8)
    y, u, v = synth.recipe_r_blocks(16, 1)
    for comp in ("Luma", "Chroma"):
        luma = comp == "Luma"
        x = O.luma_input(y) if luma else O.chroma_input(y, u, v)
        for qp in (22, 37):
            wq = {k: v_.numpy() for k, v_ in R.load_state_dict("/root/reference/trained_models/%s_Q_%d.pkl" % (comp, qp)).items()}
            wbd = synth.synth_msbd_weights(comp, qp)
            with torch.no_grad():
                q0 = O.q_forward(wq, x, luma)
                o0 = O.msbd_forward(wbd, x, q0, luma)
                for name, conv in MODES.items():
                    if name == "fp64":
                        def conv(xx, w, b, pad):
                            return F.conv2d(xx.double(), w.double(), None if b is None else b.double(), padding=pad).float()
                    q = O.q_forward(wq, x, luma, conv)
                    o = O.msbd_forward(wbd, x, q0, luma, conv)   # same q so the MTT error is isolated
                    eq = (q - q0).abs().max().item()
                    eo = max((a - b).abs().max().item() for a, b in zip(o, o0))
                    print("%-6s qp%d %-20s  QT max|d|=%.3e   MTT max|d|=%.3e" % (comp, qp, name, eq, eo))
