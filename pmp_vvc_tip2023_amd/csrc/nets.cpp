// nets.cpp — the four Down-Up-CNN forward passes (Model_QBD.py:59-253) as sequences of HIP kernel launches.
//
// Every activation is a blocked channels-last tensor [n][C/16][H][W][16] carved from the context's workspace
// arena.  Fusions relative to the reference's op-by-op graph:
//   * conv + ReLU; conv + (identity | 1x1-conv shortcut) + ReLU                 (ResidualBlock.forward :40-44)
//   * ... + 2x2 max-pool in the same epilogue                                   (:81-82,:89,:136-137,:151)
//   * ... + attention multiply x5*att / x4*att in the epilogue of the Att trunk (:143,:150)
//   * ZeroPad2d / interpolate / cat of the stems are index arithmetic inside the stem kernel (:79,:130-135)
//   * head accumulation out_k[:,0] += out_{k-1}[:,0] inside the head kernel     (:146,:153)
#include "pmp_host.h"

namespace pmp {

namespace {

struct Act {
    float *p;
    int C, H, W;  // C padded to 16
};

struct Graph {
    pmp_ctx *c;
    const NetWeights &w;
    int n;
    int rc = PMP_OK;

    Act alloc(int C, int H, int W)
    {
        const int cp = (C + 15) & ~15;
        return Act{c->arena.get((size_t)c->chunk * cp * H * W), cp, H, W};
    }

    bool check(hipError_t e, const char *what)
    {
        if (e != hipSuccess && rc == PMP_OK) rc = hip_fail(c, e, what);
        return rc == PMP_OK;
    }

    int kclass(int k, int cin, int cout) const
    {
        if (cin >= 32 && cout == 64 && k == 3 && cin == 64) return K_CONV3_64;
        if (cout == 64 && k == 5) return K_CONV5_64;
        return K_CONV_OTHER;
    }

    // ResidualBlock (Model_QBD.py:23-44) with optional fused gate / pool.
    Act rb(const Act &x, const std::string &name, bool pool = false, const Act *gate = nullptr)
    {
        auto it = w.rb.find(name);
        if (it == w.rb.end()) { if (rc == PMP_OK) rc = set_err(c, PMP_E_INVALID, "graph: no weights for " + name); return x; }
        const RBWeights &r = it->second;
        const int H = x.H, W = x.W;
        Act t = alloc(r.cout, H, W);
        Act y = alloc(r.cout, pool ? H / 2 : H, pool ? W / 2 : W);
        if (c->arena.measuring || rc != PMP_OK) return y;
        const double px = (double)n * H * W;
        if (r.direct) {
            ConvDirectArgs a{};
            a.x = x.p; a.w = r.w0; a.out = t.p;
            a.N = n; a.H = H; a.W = W; a.Cin = r.cin; a.CinPad = x.C; a.Cout = r.cout; a.CoutPad = t.C;
            a.KH = a.KW = r.k; a.relu = 1;
            { KScope ks(c, K_SMALL, 2.0 * px * r.cout * r.cin * r.k * r.k); check(launch_conv_direct(c->stream, a), "conv_direct"); }
            ConvDirectArgs b{};
            b.x = t.p; b.w = r.w2; b.out = y.p;
            b.N = n; b.H = H; b.W = W; b.Cin = r.cout; b.CinPad = t.C; b.Cout = r.cout; b.CoutPad = y.C;
            b.KH = b.KW = r.k; b.relu = 1;
            if (r.wsc) { b.x_sc = x.p; b.w_sc = r.wsc; b.Csc = r.cin; b.CscPad = x.C; }
            else b.res = x.p;
            { KScope ks(c, K_SMALL, 2.0 * px * r.cout * (r.cout * r.k * r.k + (r.wsc ? r.cin : 0))); check(launch_conv_direct(c->stream, b), "conv_direct"); }
            return y;
        }
        ConvMfmaArgs a{};
        a.x = x.p; a.w = r.w0; a.out = t.p;
        a.N = n; a.H = H; a.W = W; a.Cin = x.C; a.Cout = t.C; a.KH = a.KW = r.k; a.relu = 1;
        { KScope ks(c, kclass(r.k, r.cin, r.cout), 2.0 * px * r.cout * r.cin * r.k * r.k); check(launch_conv_mfma(c->stream, a), "conv_mfma"); }
        ConvMfmaArgs b{};
        b.x = t.p; b.w = r.w2; b.out = y.p;
        b.N = n; b.H = H; b.W = W; b.Cin = t.C; b.Cout = y.C; b.KH = b.KW = r.k; b.relu = 1;
        b.pool = pool ? 1 : 0;
        b.gate = gate ? gate->p : nullptr;
        if (r.wsc) { b.x_sc = x.p; b.w_sc = r.wsc; b.Csc = x.C; }
        else b.res = x.p;
        { KScope ks(c, kclass(r.k, r.cout, r.cout), 2.0 * px * r.cout * (r.cout * r.k * r.k + (r.wsc ? r.cin : 0))); check(launch_conv_mfma(c->stream, b), "conv_mfma"); }
        return y;
    }

    Act stem(bool luma, bool msbd, const uint8_t *by, const uint8_t *bu, const uint8_t *bv, const float *q)
    {
        const int S = luma ? 64 : 32;
        Act o = alloc(32, S, S);
        if (c->arena.measuring || rc != PMP_OK) return o;
        StemArgs a{by, bu, bv, q, w.stem_w, w.stem_b, o.p, n};
        const int cin = (luma ? 1 : 3) + (msbd ? 1 : 0), k1 = luma ? 9 : 5, k2 = luma ? 5 : 3;
        const double macs = msbd ? (double)cin * (k1 * k1 * 16 + 2 * k1 * k2 * 8) : (double)cin * k1 * k1 * 32;
        KScope ks(c, K_STEM, 2.0 * n * S * S * macs);
        check(launch_stem(c->stream, luma, msbd, a), "stem");
        return o;
    }

    void head(const Act &x, int slot, int layer, float *qt, float *bt, float *dire)
    {
        if (c->arena.measuring || rc != PMP_OK) return;
        HeadArgs a{x.p, w.head_w[slot], w.head_b[slot], qt, bt, dire, n, x.H, layer};
        KScope ks(c, K_SMALL, 2.0 * n * x.H * x.W * 72.0 * (layer < 0 ? 1 : 2));
        check(launch_head(c->stream, a), "head");
    }
};

}  // namespace

// {Luma,Chroma}_Q_Net.forward (Model_QBD.py:78-98, :176-196)
int forward_q(pmp_ctx *c, bool luma, const NetWeights &w, const uint8_t *by, const uint8_t *bu, const uint8_t *bv,
              int n, float *qt)
{
    Graph g{c, w, n};
    Act x2 = g.stem(luma, false, by, bu, bv, nullptr);
    Act x3 = g.rb(x2, "resblock_q1", luma);            // luma: + max_pool2d(2); chroma: no pool (:179)
    Act x4 = g.rb(x3, "resblock_q2", true);
    Act x5 = g.rb(x4, "resblock_q3");
    Act x6 = g.alloc(128, 16, 16);
    if (!c->arena.measuring && g.rc == PMP_OK) {
        KScope ks(c, K_SMALL, 0.0);
        g.check(launch_multipool_concat(c->stream, x5.p, x6.p, n), "multipool_concat");
    }
    Act x7 = g.rb(x6, "resblock_q4");
    Act x8 = g.rb(x7, "resblock_q5", true);
    Act x9 = g.rb(x8, "resblock_q6");
    g.head(x9, 0, -1, qt, nullptr, nullptr);
    return g.rc;
}

// {Luma,Chroma}_MSBD_Net.forward (Model_QBD.py:127-155, :225-253)
int forward_msbd(pmp_ctx *c, bool luma, const NetWeights &w, const uint8_t *by, const uint8_t *bu, const uint8_t *bv,
                 const float *qt, int n, float *bt, float *dire)
{
    Graph g{c, w, n};
    Act x = g.stem(luma, true, by, bu, bv, qt);
    x = g.rb(x, "trunk_M1.0");
    for (int i = 1; i < 5; ++i) x = g.rb(x, "trunk_M1." + std::to_string(i));
    Act x4 = g.rb(x, "trunk_M1.5", luma);              // luma pools after M1 (:136), chroma does not (:234)
    x = x4;
    for (int i = 0; i < 3; ++i) x = g.rb(x, "trunk_M2." + std::to_string(i));
    Act x5 = g.rb(x, "trunk_M2.3", true);
    // branch B1 -> out0
    Act b = g.rb(g.rb(g.rb(x5, "trunk_B1.0"), "trunk_B1.1"), "trunk_B1.2");
    g.head(b, 0, 0, nullptr, bt, dire);
    // attention 1 gates x5 (:140-143), branch B2 -> out1 (accumulated in the head kernel, :146)
    Act ai = g.alloc(16, 16, 16);
    if (!c->arena.measuring && g.rc == PMP_OK) {
        KScope ks(c, K_SMALL, 0.0);
        g.check(launch_att_input(c->stream, qt, bt, dire, 0, ai.p, n, 16), "att_input");
    }
    Act xb1 = g.rb(g.rb(ai, "trunk_Att1.0"), "trunk_Att1.1", false, &x5);
    b = g.rb(g.rb(g.rb(xb1, "trunk_B2.0"), "trunk_B2.1"), "trunk_B2.2");
    g.head(b, 1, 1, nullptr, bt, dire);
    // attention 2 gates x4 at 32x32 (:147-150), branch B3 -> pool -> out2 (:151-153)
    Act aj = g.alloc(16, 32, 32);
    if (!c->arena.measuring && g.rc == PMP_OK) {
        KScope ks(c, K_SMALL, 0.0);
        g.check(launch_att_input(c->stream, qt, bt, dire, 1, aj.p, n, 32), "att_input");
    }
    Act xb3 = g.rb(g.rb(aj, "trunk_Att2.0"), "trunk_Att2.1", false, &x4);
    b = g.rb(g.rb(g.rb(xb3, "trunk_B3.0"), "trunk_B3.1"), "trunk_B3.2", true);
    g.head(b, 2, 2, nullptr, bt, dire);
    return g.rc;
}

}  // namespace pmp
