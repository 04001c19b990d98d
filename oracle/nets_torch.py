"""ORACLE (test infrastructure, not product): CPU restatement of the four Down-Up-CNN forward passes.

Functional PyTorch-CPU fp32 restatement of the reference networks, written against a plain
{name: float32 ndarray} weight dict (the reference's state_dict names, `module.` prefix stripped):

  * residual_block      <- Model_QBD.py:23-44   (both convs bias-free, pad k//2, 1x1 bias-free shortcut iff cin!=cout)
  * luma_q / chroma_q   <- Model_QBD.py:78-98 / :176-196
  * luma_msbd / chroma_msbd <- Model_QBD.py:127-155 / :225-253 (in-place head accumulation order preserved)
  * infer_qbd           <- Metrics.py:387-419    (head regrouping bt=[o0.c0,o1.c0,o2.c0], dire=[o0.c1,o1.c1,o2.c1];
                                                  the RAW float QT logits feed the MTT net)
  * chroma_input        <- Inference_QBD.py:194-200 (max_pool2d(Y,2) ++ U ++ V)

Parity pin: tests/test_oracle_nets.py checks these against tests/golden/g1_*.npz / g2_*.npz, which were
produced by tools/gen_golden.py from the imported reference modules (torch 2.10.0 CPU, oneDNN) in the build
container.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The `conv` argument lets experiments swap the convolution arithmetic (e.g. tools/precision_study.py emulates
split-bf16 MFMA); the default is torch.nn.functional.conv2d in fp32.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _conv_default(x, w, b, pad):
    return F.conv2d(x, w, b, stride=1, padding=pad)


class _Net:
    def __init__(self, weights, conv=None):
        self.w = {k: _t(v) for k, v in weights.items()}
        self.conv = conv or _conv_default

    def c(self, x, name, pad=0):
        return self.conv(x, self.w[name + ".weight"], self.w.get(name + ".bias"), pad)

    def rb(self, x, name):
        """ResidualBlock.forward, Model_QBD.py:40-44."""
        w0 = self.w[name + ".left.0.weight"]
        pad = w0.shape[2] // 2
        out = F.relu(self.conv(x, w0, None, pad))
        out = self.conv(out, self.w[name + ".left.2.weight"], None, pad)
        sc = self.w.get(name + ".shortcut.0.weight")
        out = out + (self.conv(x, sc, None, 0) if sc is not None else x)
        return F.relu(out)

    def seq(self, x, name, n):
        for i in range(n):
            x = self.rb(x, "%s.%d" % (name, i))
        return x


def q_forward(weights, x, luma, conv=None, taps=None):
    """{Luma,Chroma}_Q_Net.forward (Model_QBD.py:78-98, :176-196). x: [N,1,68,68] or [N,3,34,34] float32."""
    n = _Net(weights, conv)
    x = _t(x)
    p = 4 if luma else 2
    x1 = F.pad(x, (0, p, 0, p))
    x2 = F.relu(n.c(x1, "conv_q1"))
    if luma:
        x3 = F.max_pool2d(n.rb(x2, "resblock_q1"), 2)
    else:
        x3 = n.rb(x2, "resblock_q1")
    x4 = F.max_pool2d(n.rb(x3, "resblock_q2"), 2)
    x5 = n.rb(x4, "resblock_q3")
    x5_1 = F.interpolate(F.max_pool2d(x5, 2), scale_factor=2)
    x5_2 = F.interpolate(F.max_pool2d(x5, 4), scale_factor=4)
    x5_3 = F.interpolate(F.max_pool2d(x5, 8), scale_factor=8)
    x6 = torch.cat([x5, x5_1, x5_2, x5_3], 1)
    x7 = n.rb(x6, "resblock_q4")
    x8 = F.max_pool2d(n.rb(x7, "resblock_q5"), 2)
    x9 = n.rb(x8, "resblock_q6")
    x10 = n.c(x9, "conv_q2", pad=1)
    if taps is not None:
        taps.update(x2=x2, x3=x3, x4=x4, x5=x5, x7=x7, x9=x9)
    return x10


def msbd_forward(weights, x, q, luma, conv=None, taps=None):
    """{Luma,Chroma}_MSBD_Net.forward (Model_QBD.py:127-155, :225-253). Returns (out0, out1, out2) [N,2,16,16]."""
    n = _Net(weights, conv)
    x = _t(x)
    q = _t(q)
    p = 4 if luma else 2
    s = 8 if luma else 4
    x1_1 = F.pad(F.interpolate(q, scale_factor=s), (p, 0, p, 0))
    x2 = torch.cat([x, x1_1], 1)
    x3_1 = F.relu(n.c(F.pad(x2, (0, p, 0, p)), "conv_b1_1"))
    x3_2 = F.relu(n.c(F.pad(x2, (0, p, 0, 0)), "conv_b1_2"))
    x3_3 = F.relu(n.c(F.pad(x2, (0, 0, 0, p)), "conv_b1_3"))
    x3 = torch.cat([x3_1, x3_2, x3_3], 1)
    m1 = n.seq(x3, "trunk_M1", 6)
    x4 = F.max_pool2d(m1, 2) if luma else m1
    x5 = F.max_pool2d(n.seq(x4, "trunk_M2", 4), 2)
    x6 = n.seq(x5, "trunk_B1", 3)
    out0 = n.c(x6, "conv_B1", pad=1)
    out0q = torch.cat([F.interpolate(q, scale_factor=2), out0], 1)
    x_att0 = n.seq(out0q, "trunk_Att1", 2)
    xb1 = x5 * x_att0
    xb2 = n.seq(xb1, "trunk_B2", 3)
    out1 = n.c(xb2, "conv_B2", pad=1)
    out1_raw = out1.clone()
    out1[:, 0:1] = out1[:, 0:1] + out0[:, 0:1]           # accumulated BEFORE it is upsampled below (:146-147)
    out1q = torch.cat([F.interpolate(q, scale_factor=4), F.interpolate(out1, scale_factor=2)], 1)
    x_att1 = n.seq(out1q, "trunk_Att2", 2)
    xb3 = x4 * x_att1
    xb4 = F.max_pool2d(n.seq(xb3, "trunk_B3", 3), 2)
    out2 = n.c(xb4, "conv_B3", pad=1)
    out2[:, 0:1] = out2[:, 0:1] + out1[:, 0:1]
    if taps is not None:
        taps.update(x3=x3, x4=x4, x5=x5, x_att0=x_att0, out1_raw=out1_raw, x_att1=x_att1)
    return out0, out1, out2


def chroma_input(block_y, block_u, block_v):
    """Inference_QBD.py:194-200: [N,68,68],[N,34,34],[N,34,34] u8 -> float32 [N,3,34,34]."""
    y = torch.from_numpy(np.ascontiguousarray(block_y)).float().unsqueeze(1)
    u = torch.from_numpy(np.ascontiguousarray(block_u)).float().unsqueeze(1)
    v = torch.from_numpy(np.ascontiguousarray(block_v)).float().unsqueeze(1)
    return torch.cat([F.max_pool2d(y, 2), u, v], 1)


def luma_input(block_y):
    return torch.from_numpy(np.ascontiguousarray(block_y)).float().unsqueeze(1)


@torch.no_grad()
def infer_qbd(wq, wbd, x, luma, batch=200, conv=None):
    """inference_pre_QBD (Metrics.py:387-419): returns numpy qt[N,1,8,8], bt[N,3,16,16], dire[N,3,16,16]."""
    x = _t(x)
    qs, bts, ds = [], [], []
    for i in range(0, x.shape[0], batch):
        xb = x[i:i + batch]
        q = q_forward(wq, xb, luma, conv)
        o0, o1, o2 = msbd_forward(wbd, xb, q, luma, conv)
        qs.append(q)
        bts.append(torch.cat([o0[:, 0:1], o1[:, 0:1], o2[:, 0:1]], 1))
        ds.append(torch.cat([o0[:, 1:2], o1[:, 1:2], o2[:, 1:2]], 1))
    return torch.cat(qs).numpy(), torch.cat(bts).numpy(), torch.cat(ds).numpy()
