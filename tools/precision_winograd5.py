"""Would a one-dimensional Winograd form F(2, 5) along x of the 5x5 convolutions (6 multiplications per output pair and vertical tap
instead of 10: 0.6x the MFMA products of the direct form) keep the logits inside the 1e-3 tolerance on the f16x3 datapath?
(run in the build container; imports the reference for the trained QT weights.)  Emulation as tools/precision_winograd.py: V = B^T d in
fp32, split into two fp16 terms; U = G g in fp64, scaled by a power of two, two fp16 terms; three products, fp32 accumulation; output
transform in fp32.  Interpolation points: 0, +-1, +-a, infinity with a = 2 or 1/2.
Usage: python tools/precision_winograd5.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import ref_harness as R
from pmp_vvc_tip2023_amd import synth
from oracle import nets_torch as O
from precision_study import make_conv_f16, split16

direct = make_conv_f16(256.0, [(0, 1), (1, 0), (0, 0)])
STAT = {"vmax": 0.0}


def toom_cook(m, r, pts):
    """A^T (m x n), G (n x r), B^T (n x n) of F(m, r) (correlation form) for the finite points pts plus infinity."""
    n = m + r - 1
    assert len(pts) == n - 1
    At = np.zeros((m, n)); G = np.zeros((n, r))
    for j, p in enumerate(pts):
        Nj = np.prod([p - q for l, q in enumerate(pts) if l != j])
        At[:, j] = [p ** i for i in range(m)]
        G[j] = [p ** k / Nj for k in range(r)]
    At[m - 1, n - 1] = 1.0; G[n - 1, r - 1] = 1.0
    Bt = np.zeros((n, n))
    for i in range(n):                       # column i of B^T: y[o] = sum_j At[o, j] G[j, k] Bt[j, i] must be [i == o + k]
        rows, rhs = [], []
        for o in range(m):
            for k in range(r):
                rows.append(At[o] * G[:, k]); rhs.append(1.0 if i == o + k else 0.0)
        Bt[:, i] = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0]
    Bt = np.round(Bt * 64) / 64              # the entries are small dyadic rationals
    d = np.random.default_rng(0).normal(size=n); g = np.random.default_rng(1).normal(size=r)
    y = At @ ((G @ g) * (Bt @ d))
    assert np.allclose(y, [sum(g[k] * d[o + k] for k in range(r)) for o in range(m)], atol=1e-9), (y, Bt)
    return At, G, Bt


def make_wino5(a):
    At_, G_, Bt_ = toom_cook(2, 5, [0.0, 1.0, -1.0, a, -a])
    At, G, Bt = torch.tensor(At_, dtype=torch.float32), torch.tensor(G_, dtype=torch.float64), torch.tensor(Bt_, dtype=torch.float32)

    def wino(x, w):
        N, C, H, W = x.shape
        K = w.shape[0]
        U = torch.einsum("pj,kcij->kcip", G, w.double())                       # [K,C,ky,6]
        s = 2.0 ** np.floor(np.log2(4096.0 / U.abs().max().item()))
        u0, u1 = split16((U * s).float(), 2)
        xp = F.pad(x, (2, 2, 2, 2))
        d = xp.unfold(3, 6, 2)                                                  # [N,C,H+4,W/2,6]
        V = torch.zeros(d.shape[:-1] + (6,), dtype=torch.float32)
        for p in range(6):                                                      # fp32 FMAs, one term at a time
            acc = torch.zeros(d.shape[:-1], dtype=torch.float32)
            for i in range(6):
                if Bt[p, i] != 0:
                    acc = acc + Bt[p, i] * d[..., i]
            V[..., p] = acc
        STAT["vmax"] = max(STAT["vmax"], V.abs().max().item())
        v0, v1 = split16(V, 2)
        M = torch.zeros((N, K, H, W // 2, 6), dtype=torch.float32)
        for ky in range(5):
            for (a_, b_) in ((v0, u1), (v1, u0), (v0, u0)):
                M += torch.einsum("nchtp,kcp->nkhtp", a_[:, :, ky:ky + H], b_[:, :, ky])
        M = M / s
        Y = torch.einsum("op,nkhtp->nkhto", At, M)
        return Y.reshape(N, K, H, W)

    def conv(x, w, b, pad):
        if not (w.shape[2] == 5 and w.shape[3] == 5 and pad == 2 and x.shape[3] % 2 == 0):
            return direct(x, w, b, pad)
        out = wino(x, w)
        if b is not None:
            out = out + b.view(1, -1, 1, 1)
        return out
    return conv, Bt_, G_


if __name__ == "__main__":
    torch.set_num_threads(8)
    y, u, v = synth.recipe_r_blocks(8, 1)
    for a in (2.0, 0.5):
        conv, Bt_, G_ = make_wino5(a)
        print("points 0, +-1, +-%g, inf:  B^T =\n%s\n  row sums of |B^T|: %s" % (a, Bt_, np.abs(Bt_).sum(1)))
    for comp in ("Luma",):
        luma = comp == "Luma"
        x = O.luma_input(y) if luma else O.chroma_input(y, u, v)
        for qp in (22, 37):
            wq = {k: v_.numpy() for k, v_ in R.load_state_dict("/root/reference/trained_models/%s_Q_%d.pkl" % (comp, qp)).items()}
            wbd = synth.synth_msbd_weights(comp, qp)
            with torch.no_grad():
                q0 = O.q_forward(wq, x, luma)
                o0 = O.msbd_forward(wbd, x, q0, luma)
                for name, conv in (("f16x3 direct", direct), ("F(2,5) along x, a = 2", make_wino5(2.0)[0]), ("F(2,5) along x, a = 1/2", make_wino5(0.5)[0])):
                    STAT["vmax"] = 0.0
                    q = O.q_forward(wq, x, luma, conv)
                    o = O.msbd_forward(wbd, x, q0, luma, conv)
                    eq = (q - q0).abs().max().item()
                    eo = max((a_ - b_).abs().max().item() for a_, b_ in zip(o, o0))
                    print("%-6s qp%d %-28s QT max|d|=%.3e   MTT max|d|=%.3e   max |V| %.3g" % (comp, qp, name, eq, eo, STAT["vmax"]), flush=True)
