"""Would a Winograd F(2x2, 3x3) form of the 3x3 convolutions keep the logits inside the 1e-3 tolerance on the f16x3 datapath?
(run in the build container; imports the reference for the trained QT weights.)  Emulation: the transformed input tiles
V = B^T d B are formed in fp32 and split into two fp16 terms, the transformed weights U = G g G^T are formed in fp64, scaled by a power
of two and split into two fp16 terms; the 16 per-position channel contractions use the three products v0u0 + v0u1 + v1u0 with fp32
accumulation, the output transform A^T M A runs in fp32.  2.25x fewer MFMA products than the direct form.
Usage: python tools/precision_winograd.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import ref_harness as R
from pmp_vvc_tip2023_amd import synth
from oracle import nets_torch as O
from precision_study import make_conv_f16, split16

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
direct = make_conv_f16(256.0, [(0, 1), (1, 0), (0, 0)])
STAT = {"vmax": 0.0}


def wino(x, w, scale_bits=None, only64=False):
    N, C, H, W = x.shape
    K = w.shape[0]
    U = torch.einsum("ai,kcij,bj->kcab", G, w.double(), G)                      # [K,C,4,4], fp64
    s = 2.0 ** np.floor(np.log2(4096.0 / U.abs().max().item())) if scale_bits is None else 2.0 ** scale_bits
    u0, u1 = split16((U * s).float(), 2)
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                      # [N,C,Th,Tw,4,4]
    V = torch.einsum("ai,nctuij,bj->nctuab", Bt, d, Bt)                          # fp32
    STAT["vmax"] = max(STAT["vmax"], V.abs().max().item())
    v0, v1 = split16(V, 2)
    M = (torch.einsum("nctuab,kcab->nktuab", v0, u1) + torch.einsum("nctuab,kcab->nktuab", v1, u0)) + torch.einsum("nctuab,kcab->nktuab", v0, u0)
    M = M / s
    Y = torch.einsum("ai,nktuij,bj->nktuab", At, M, At)                          # [N,K,Th,Tw,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


def make(which):
    def conv(x, w, b, pad):
        use = w.shape[2] == 3 and w.shape[3] == 3 and pad == 1 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
        if use and which == "c64":
            use = w.shape[0] == 64 and w.shape[1] == 64
        if not use:
            return direct(x, w, b, pad)
        out = wino(x, w)
        if b is not None:
            out = out + b.view(1, -1, 1, 1)
        return out
    return conv


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
                for name, conv in (("f16x3 direct", direct), ("f16x3 winograd on 3x3 64->64", make("c64")), ("f16x3 winograd on every 3x3", make("all"))):
                    STAT["vmax"] = 0.0
                    q = O.q_forward(wq, x, luma, conv)
                    o = O.msbd_forward(wbd, x, q0, luma, conv)
                    eq = (q - q0).abs().max().item()
                    eo = max((a - b).abs().max().item() for a, b in zip(o, o0))
                    print("%-6s qp%d %-32s QT max|d|=%.3e   MTT max|d|=%.3e   max |V| %.3g" % (comp, qp, name, eq, eo, STAT["vmax"]), flush=True)
