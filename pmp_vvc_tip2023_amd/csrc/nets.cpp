// nets.cpp — the four Down-Up-CNN forward passes (Model_QBD.py:59-253) as sequences of HIP kernel launches.
//
// Every activation is a blocked channels-last tensor [n][C/16][H][W][16] carved from the context's workspace arena,
// either plain fp32 or "split-3" (three bf16 planes, conv_bf16x6.hip) when the context runs the bf16x6 datapath.
// Fusions relative to the reference's op-by-op graph:
//   * conv + ReLU; conv + (identity | 1x1-conv shortcut) + ReLU                 (ResidualBlock.forward :40-44)
//   * ... + 2x2 max-pool in the same epilogue                                   (:81-82,:89,:136-137,:151)
//   * ... + attention multiply x5*att / x4*att in the epilogue of the Att trunk (:143,:150)
//   * ZeroPad2d / interpolate / cat of the stems are index arithmetic inside the stem kernel (:79,:130-135)
//   * head accumulation out_k[:,0] += out_{k-1}[:,0] inside the head kernel     (:146,:153)
// In bf16x6 mode tensors that only feed MFMA convs stay split-3; tensors read by the small fp32 kernels (heads,
// 8x8 tail, multi-scale pool, attention inputs) are written as fp32 by the producing conv, and the three fp32-born
// inputs of MFMA convs (stem output, x6, attention inputs) pass through a split kernel.
#include <cmath>

#include "pmp_host.h"

namespace pmp {

namespace {

struct Act {
    float *p;      // fp32 tensor, or the first split plane (three bf16 / two fp16 planes)
    int C, H, W;   // C padded to 16
    bool split;
    size_t stride; // elements between split planes
    size_t off, bytes;   // its piece of the workspace arena (Graph::release)
    const unsigned short *s() const { return reinterpret_cast<const unsigned short *>(p); }
    unsigned short *s() { return reinterpret_cast<unsigned short *>(p); }
};

struct Graph {
    pmp_ctx *c;
    const NetWeights &w;
    int n;
    int rc = PMP_OK;
    bool x6() const { return c->precision != 0; }      // a split datapath (bf16x6 or f16x3) is active
    bool h2() const { return c->precision == 2; }
    int fmt() const { return c->precision; }           // split3.h: 0 fp32, 1 split-3, 2 split-2
    unsigned *sat() const { return h2() ? c->d_sat : nullptr; }   // f16x3: the context's sticky saturation flag
    // f16x3 activation scales (pmp_host.h: NetWeights::act_exp): the segment the graph is in, and the exponent a segment's tensors carry
    int seg = 0;
    bool scaled() const { return h2() && c->act_scales && w.stem_b_h; }
    int E(int sg) const { return scaled() ? w.act_exp[sg] : 0; }
    // calibration pass (fp32 datapath, calibrate.cpp: calibrate_mtt): the largest |value| of a tensor just produced, per launch
    void note(const Act &a, const std::string &name, int sg)
    {
        if (!c->cal_on || !live() || a.split || (int)c->cal_log.size() >= PMP_CAL_SLOTS) return;
        check(launch_amax_f32(c->stream, a.p, (size_t)n * a.C * a.H * a.W, c->d_cal + c->cal_log.size()), "amax");
        c->cal_log.emplace_back(name, sg);
    }

    // Tensors are sized for the n blocks of this pass (not for the chunk) and give their bytes back with release().
    Act alloc(int C, int H, int W, bool split)
    {
        const int cp = (C + 15) & ~15;
        const size_t elems = (size_t)n * cp * H * W;
        // split-3: 3 planes of 2-byte elements = 6 bytes per element; split-2: 2 planes = 4; fp32: 4
        const size_t bytes = elems * (split && !h2() ? 6 : 4);
        const size_t off = c->arena.take(bytes);
        return Act{c->arena.ptr(off), cp, H, W, split, elems, off, bytes};
    }

    // The tensor's last consumer has been enqueued (one in-order stream): later allocations may reuse its bytes.
    void release(Act &a)
    {
        if (a.bytes) c->arena.give(a.off, a.bytes);
        a.bytes = 0;
    }

    bool check(hipError_t e, const char *what)
    {
        if (e != hipSuccess && rc == PMP_OK) rc = hip_fail(c, e, what);
        return rc == PMP_OK;
    }
    bool live() const { return !c->arena.measuring && rc == PMP_OK; }

    int kclass(int k, int cin, int cout) const
    {
        if (cout == 64 && cin == 64 && k == 3) return K_CONV3_64;
        if (cout == 64 && k == 5) return K_CONV5_64;
        return K_CONV_OTHER;
    }

    // fp32 tensor -> split planes of the active datapath (no-op in fp32 mode or for a tensor that is split already)
    Act to_conv_input(const Act &x)
    {
        if (!x6() || x.split) return x;
        Act y = alloc(x.C, x.H, x.W, true);
        if (live()) {
            KScope ks(c, K_SMALL, 0.0);
            if (h2()) check(launch_f32_to_split2(c->stream, x.p, y.s(), (size_t)n * x.C * x.H * x.W, y.stride, sat()), "f32_to_split2");
            else check(launch_f32_to_split3(c->stream, x.p, y.s(), (size_t)n * x.C * x.H * x.W, y.stride), "f32_to_split3");
        }
        return y;
    }

    // One convolution of a residual block on the MFMA path of the active datapath.
    // xexp: an extra power of two on the accumulator's way out (f16x3: the step between two segments' activation scales at a gate product)
    void conv(const Act &x, const RBWeights &r, bool second, const Act *sc_src, const Act *res, const Act *gate, bool pool,
              Act &out, double flops, int cls, int xexp = 0)
    {
        if (!live()) return;
        KScope ks(c, cls, flops);
        if (x6()) {
            ConvX6Args a{};
            a.x = x.s(); a.x_stride = x.stride;
            if (h2()) {
                a.w = second ? r.w2h : r.w0h; a.out_scale = std::ldexp(1.f, xexp - (second ? r.k2 : r.k0)); a.sat = sat();
                abl_conv_args(c, r, second, a);
            }
            else a.w = second ? r.w2x : r.w0x;
            if (sc_src) { a.x_sc = sc_src->s(); a.sc_stride = sc_src->stride; a.w_sc = h2() ? r.wsch : r.wscx; a.Csc = sc_src->C; }
            if (res) { a.res = res->s(); a.res_stride = res->stride; }
            if (gate) { a.gate = gate->s(); a.gate_stride = gate->stride; }
            if (out.split) { a.out = out.s(); a.out_stride = out.stride; }
            else a.out_f32 = out.p;
            a.N = n; a.H = x.H; a.W = x.W; a.Cin = x.C; a.Cout = out.C; a.KH = a.KW = r.k; a.relu = 1; a.pool = pool ? 1 : 0;
            if (h2()) check(launch_conv_h2(c->stream, a), "conv_h2");
            else check(launch_conv_x6(c->stream, a), "conv_x6");
        } else {
            ConvMfmaArgs a{};
            a.x = x.p; a.w = second ? r.w2 : r.w0; a.out = out.p;
            if (sc_src) { a.x_sc = sc_src->p; a.w_sc = r.wsc; a.Csc = sc_src->C; }
            if (res) a.res = res->p;
            if (gate) a.gate = gate->p;
            a.N = n; a.H = x.H; a.W = x.W; a.Cin = x.C; a.Cout = out.C; a.KH = a.KW = r.k; a.relu = 1; a.pool = pool ? 1 : 0;
            check(launch_conv_mfma(c->stream, a), "conv_mfma");
        }
    }

    // ResidualBlock (Model_QBD.py:23-44) with optional fused gate / pool.  out_f32: the consumer is not an MFMA conv.
    // consume: the caller has no further use for x_in - its bytes are released here, and an identity-shortcut block without
    // pooling writes its output IN PLACE over it (every thread reads the residual of exactly the elements it then stores,
    // the convolution itself reads the intermediate t), so a trunk of such blocks needs two tensors, not three.
    Act rb(Act &x_in, const std::string &name, bool pool = false, const Act *gate = nullptr, bool out_f32 = false, bool consume = true)
    {
        auto it = w.rb.find(name);
        if (it == w.rb.end()) { if (rc == PMP_OK) rc = set_err(c, PMP_E_INVALID, "graph: no weights for " + name); return x_in; }
        const RBWeights &r = it->second;
        const int H = x_in.H, W = x_in.W;
        const double px = (double)n * H * W;
        if (r.direct) {  // 8x8 maps: plain fp32 kernels in both modes
            Act t = alloc(r.cout, H, W, false), y = alloc(r.cout, pool ? H / 2 : H, pool ? W / 2 : W, false);
            if (live()) {
                const Act &x = x_in;
                ConvDirectArgs a{};
                a.x = x.p; a.w = r.w0; a.out = t.p;
                a.N = n; a.H = H; a.W = W; a.Cin = r.cin; a.CinPad = x.C; a.Cout = r.cout; a.CoutPad = t.C;
                a.KH = a.KW = r.k; a.relu = 1;
                { KScope ks(c, K_SMALL, 2.0 * px * r.cout * r.cin * r.k * r.k); check(launch_conv_direct(c->stream, a), "conv_direct"); }
                ConvDirectArgs b{};
                b.x = t.p; b.w = r.w2; b.out = y.p;
                b.N = n; b.H = H; b.W = W; b.Cin = r.cout; b.CinPad = t.C; b.Cout = r.cout; b.CoutPad = y.C;
                b.KH = b.KW = r.k; b.relu = 1;
                if (r.has_sc) { b.x_sc = x.p; b.w_sc = r.wsc; b.Csc = r.cin; b.CscPad = x.C; }
                else b.res = x.p;
                { KScope ks(c, K_SMALL, 2.0 * px * r.cout * (r.cout * r.k * r.k + (r.has_sc ? r.cin : 0))); check(launch_conv_direct(c->stream, b), "conv_direct"); }
            }
            note(t, name + ".t", seg);
            note(y, name, seg);
            release(t);
            if (consume) release(x_in);
            return y;
        }
        if (!gate && consume && fused32(x_in, r, pool) && (pool ? out_f32 : !out_f32)) return rb_fused32(x_in, name, r, pool);
        Act x = to_conv_input(x_in);
        const bool converted = x.p != x_in.p || x.off != x_in.off;
        if (converted && consume) release(x_in);
        const bool x_dead = converted || consume;           // x's bytes are ours to reuse after this block
        Act t = alloc(r.cout, H, W, x6());
        const bool y_split = x6() && !out_f32;
        const bool in_place = x_dead && !r.has_sc && !pool && x.split == y_split && x.C == ((r.cout + 15) & ~15);
        Act y = in_place ? x : alloc(r.cout, pool ? H / 2 : H, pool ? W / 2 : W, y_split);
        conv(x, r, false, nullptr, nullptr, nullptr, false, t, 2.0 * px * r.cout * r.cin * r.k * r.k, kclass(r.k, r.cin, r.cout));
        note(t, name + ".t", seg);
        // a gated block ends its attention segment: its output is (this segment) x (the gate operand, a trunk tensor of segment 0) and
        // opens the next segment - the three exponents meet in the out_scale of the convolution whose epilogue multiplies
        conv(t, r, true, r.has_sc ? &x : nullptr, r.has_sc ? nullptr : &x, gate, pool, y,
             2.0 * px * r.cout * (r.cout * r.k * r.k + (r.has_sc ? r.cin : 0)), kclass(r.k, r.cout, r.cout),
             gate ? E(seg) + E(0) - E(seg + 1) : 0);
        note(y, name, gate ? seg + 1 : seg);
        release(t);
        if (x_dead && !in_place) release(x);
        return y;
    }

    Act stem(bool luma, bool msbd, const uint8_t *by, const uint8_t *bu, const uint8_t *bv, const float *q)
    {
        const int S = luma ? 64 : 32;
        Act o = alloc(32, S, S, x6());     // bf16x6 mode: the stem writes split-3 planes directly
        if (!live()) return o;
        // f16x3 MTT stems write segment 0 at its activation scale: 2^-e0 on the output scale and on the biases (stem_b_h)
        const int e0 = msbd ? E(0) : 0;
        StemArgs a{by, bu, bv, q, w.stem_w, (msbd && scaled()) ? w.stem_b_h : w.stem_b, o.split ? nullptr : o.p, n, o.split ? o.s() : nullptr, o.stride, fmt(),
                   h2() ? w.stem_wh : nullptr, std::ldexp(1.f, -w.stem_k - e0), sat()};
        const int cin = (luma ? 1 : 3) + (msbd ? 1 : 0), k1 = luma ? 9 : 5, k2 = luma ? 5 : 3;
        const double macs = msbd ? (double)cin * (k1 * k1 * 16 + 2 * k1 * k2 * 8) : (double)cin * k1 * k1 * 32;
        KScope ks(c, K_STEM, 2.0 * n * S * S * macs);
        check(launch_stem(c->stream, luma, msbd, a), "stem");
        note(o, "stem", seg);
        return o;
    }

    // rbfuse32.hip (f16x3): a ResidualBlock with <= 32 output channels at 32x32 as ONE launch, the intermediate in LDS; bit-identical to rb().
    // pool_f32: + 2x2 max-pool, plain fp32 output (trunk_B3.2, read by the head kernel).
    // Exactly the three instantiations launch_rbfuse32 has (the predicate used to admit (16, 16) without pool and (32, 16) with
    // pool, which the launcher rejects - no block of the four nets has those shapes, but such a block would have failed the call instead
    // of taking the launch-per-layer path): (cin, cout, pool) = (32, 16, no) trunk_B3.1, (16, 16, yes) trunk_B3.2, (16, 32, no) trunk_Att2.0.
    bool fused32(const Act &x, const RBWeights &r, bool pool) const
    {
        return h2() && c->fuse32 && x.H == 32 && x.W == 32 && x.split && r.w0h && r.has_sc && !r.direct && r.k == 3 &&
               ((r.cin_pad == 32 && r.cout_pad == 16 && !pool) || (r.cin_pad == 16 && r.cout_pad == 16 && pool) || (r.cin_pad == 16 && r.cout_pad == 32 && !pool));
    }
    // trunk_Att2.0 with its input built from the logits inside the kernel (no att_input launch, no input tensor)
    Act rb_fused32_att(const std::string &name, const float *q, const float *bt, const float *dire, int layer, int S)
    {
        const RBWeights &r = w.rb.find(name)->second;
        Act y = alloc(r.cout, S, S, true);
        if (live()) {
            RbFuse32Args a{};
            a.w0 = r.w0h; a.w2 = r.w2h; a.wsc = r.wsch; a.s0 = std::ldexp(1.f, -r.k0); a.s2 = std::ldexp(1.f, -r.k2);
            a.out = y.s(); a.out_stride = y.stride; a.sat = sat(); a.N = n; a.H = S; a.W = S; a.cin_groups = 1; a.cout_groups = 2;
            a.q = q; a.bt = bt; a.dire = dire; a.att_layer = layer; a.att_scale = std::ldexp(1.f, -E(seg));
            KScope ks(c, K_CONV_OTHER, 2.0 * n * S * S * r.cout * (r.cin * 9 + r.cout * 9 + r.cin));
            check(launch_rbfuse32(c->stream, a), "rbfuse32(att)");
        }
        return y;
    }
    bool fused32_att(const std::string &name) const
    {
        auto it = w.rb.find(name);
        return h2() && c->fuse32 && it != w.rb.end() && it->second.w0h && it->second.has_sc && it->second.cin_pad == 16 && it->second.cout_pad == 32;
    }
    Act rb_fused32(Act &x, const std::string &name, const RBWeights &r, bool pool_f32)
    {
        const int H = x.H, W = x.W;
        Act y = alloc(r.cout, pool_f32 ? H / 2 : H, pool_f32 ? W / 2 : W, !pool_f32);
        if (live()) {
            RbFuse32Args a{};
            a.x = x.s(); a.x_stride = x.stride;
            a.w0 = r.w0h; a.w2 = r.w2h; a.wsc = r.wsch; a.s0 = std::ldexp(1.f, -r.k0); a.s2 = std::ldexp(1.f, -r.k2);
            if (pool_f32) a.out_f32 = y.p; else { a.out = y.s(); a.out_stride = y.stride; }
            a.sat = sat(); a.N = n; a.H = H; a.W = W; a.cin_groups = r.cin_pad / 16; a.cout_groups = r.cout_pad / 16; a.pool_f32 = pool_f32 ? 1 : 0;
            const double px = (double)n * H * W;
            KScope ks(c, K_CONV_OTHER, 2.0 * px * r.cout * (r.cin * 9 + r.cout * 9 + r.cin));
            check(launch_rbfuse32(c->stream, a), "rbfuse32");
        }
        release(x);
        return y;
    }

    // chain16.hip: f16x3 only, and only with every block it names loaded in that format
    bool fused16() const { return h2() && c->fuse16; }
    bool c16rb(const std::string &name, Chain16RB &o, double &flops, int px)
    {
        auto it = w.rb.find(name);
        if (it == w.rb.end() || !it->second.w0h) { if (rc == PMP_OK) rc = set_err(c, PMP_E_INVALID, "graph: no f16x3 weights for " + name); return false; }
        const RBWeights &r = it->second;
        o = Chain16RB{r.w0h, r.w2h, r.has_sc ? r.wsch : nullptr, std::ldexp(1.f, -r.k0), std::ldexp(1.f, -r.k2)};
        flops += 2.0 * n * px * r.cout * (r.cin * r.k * r.k + r.cout * r.k * r.k + (r.has_sc ? r.cin : 0));
        return true;
    }

    // f16x3 MTT heads read a tensor that carries its segment's activation scale: their weights hold the way back (head_w_h = head_w * 2^e)
    const float *head_weights(int slot, int layer) const { return (layer >= 0 && scaled()) ? w.head_w_h[slot] : w.head_w[slot]; }

    void head(const Act &x, int slot, int layer, float *qt, float *bt, float *dire)
    {
        if (!live()) return;
        HeadArgs a{x.p, head_weights(slot, layer), w.head_b[slot], qt, bt, dire, n, x.H, layer};
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
    if (g.fused16()) {   // q3 .. conv_q2 at 16x16 / 8x8: two launches, four waves per block and two blocks per CU, activations in LDS (chain16.hip)
        Chain16QtArgs a{};
        double flops = 0;
        auto q6 = w.rb.find("resblock_q6");
        if (q6 == w.rb.end() || !q6->second.direct) return set_err(c, PMP_E_INVALID, "graph: no weights for resblock_q6");
        if (!g.c16rb("resblock_q3", a.q3, flops, 256) || !g.c16rb("resblock_q4", a.q4, flops, 256) || !g.c16rb("resblock_q5", a.q5, flops, 256)) return g.rc;
        flops += 2.0 * n * 64 * 8 * (32 * 9 + 8 * 9 + 32) + 2.0 * n * 64 * 72.0;
        Act x5 = g.alloc(32, 16, 16, false);        // resblock_q3's fp32 output: from q3_rb64_kernel to qt_rest16_kernel through L2
        a.x4 = x4.s(); a.x4_stride = x4.stride; a.x5 = x5.p; a.qt = qt; a.N = n; a.sat = g.sat();
        a.d_w0 = q6->second.w0; a.d_w2 = q6->second.w2; a.d_wsc = q6->second.wsc; a.head_w = w.head_w[0]; a.head_b = w.head_b[0];
        if (g.live()) {
            KScope ks(c, K_CONV_OTHER, flops);
            g.check(launch_qt_tail16(c->stream, a), "qt_tail16");
        }
        g.release(x5);
        g.release(x4);
        return g.rc;
    }
    Act x5 = g.rb(x4, "resblock_q3", false, nullptr, true);   // fp32: read by the multi-scale pool kernel
    Act x6 = g.alloc(128, 16, 16, g.x6());
    if (g.live()) {
        KScope ks(c, K_SMALL, 0.0);
        g.check(launch_multipool_concat(c->stream, x5.p, x6.split ? nullptr : x6.p, n, x6.split ? x6.s() : nullptr, x6.stride, g.fmt(), g.sat()), "multipool_concat");
    }
    g.release(x5);
    Act x7 = g.rb(x6, "resblock_q4");
    Act x8 = g.rb(x7, "resblock_q5", true, nullptr, true);    // fp32: 8x8 tail runs on the direct kernel
    Act x9 = g.rb(x8, "resblock_q6");
    g.head(x9, 0, -1, qt, nullptr, nullptr);
    g.release(x9);
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
    x = g.rb(x4, "trunk_M2.0", false, nullptr, false, false);   // x4 stays: attention 2 gates it (:150)
    for (int i = 1; i < 3; ++i) x = g.rb(x, "trunk_M2." + std::to_string(i));
    Act x5 = g.rb(x, "trunk_M2.3", true);
    Act b{};
    if (g.fused16()) {   // B1 + conv_B1, attention 1, B2 + conv_B2 at 16x16: three launches (chain16.hip)
        Chain16MsbdArgs a{};
        double flops = 2.0 * 2 * n * 256 * 72.0 * 2;
        bool ok = true;
        for (int i = 0; i < 3 && ok; ++i)
            ok = g.c16rb("trunk_B1." + std::to_string(i), a.b1[i], flops, 256) && g.c16rb("trunk_B2." + std::to_string(i), a.b2[i], flops, 256);
        for (int i = 0; i < 2 && ok; ++i) ok = g.c16rb("trunk_Att1." + std::to_string(i), a.att[i], flops, 256);
        if (!ok) return g.rc;
        Act xb = g.alloc(64, 16, 16, true);         // x5 * att0: from att16_kernel to trunk_B2's branch16_kernel through L2
        a.x5 = x5.s(); a.x5_stride = x5.stride; a.xb = xb.s(); a.xb_stride = xb.stride; a.qt = qt; a.bt = bt; a.dire = dire; a.N = n; a.sat = g.sat();
        a.att[1].s2 = std::ldexp(a.att[1].s2, g.E(1) + g.E(0) - g.E(2));     // the gate product x5 * att0: from segments 1 and 0 into segment 2
        a.att_scale = std::ldexp(1.f, -g.E(1));
        for (int i = 0; i < 2; ++i) { a.head_w[i] = g.head_weights(i, i); a.head_b[i] = w.head_b[i]; }
        if (g.live()) {
            KScope ks(c, K_CONV_OTHER, flops);
            g.check(launch_msbd_branch16(c->stream, a), "msbd_branch16");
        }
        g.release(xb);
        g.release(x5);
    } else {
        // branch B1 -> out0 (x5 stays: attention 1 gates it, :143)
        b = g.rb(x5, "trunk_B1.0", false, nullptr, false, false);
        b = g.rb(b, "trunk_B1.1");
        b = g.rb(b, "trunk_B1.2", false, nullptr, true);
        g.head(b, 0, 0, nullptr, bt, dire);
        g.release(b);
        // attention 1 gates x5 (:140-143), branch B2 -> out1 (accumulated in the head kernel, :146)
        g.seg = 1;
        Act ai = g.alloc(16, 16, 16, g.x6());
        if (g.live()) {
            KScope ks(c, K_SMALL, 0.0);
            g.check(launch_att_input(c->stream, qt, bt, dire, 0, ai.split ? nullptr : ai.p, n, 16, ai.split ? ai.s() : nullptr, ai.stride, g.fmt(), g.sat(), std::ldexp(1.f, -g.E(1))), "att_input");
        }
        g.note(ai, "att_input1", 1);
        Act a1 = g.rb(ai, "trunk_Att1.0");
        Act xb1 = g.rb(a1, "trunk_Att1.1", false, &x5);
        g.release(x5);
        g.seg = 2;
        b = g.rb(xb1, "trunk_B2.0");
        b = g.rb(b, "trunk_B2.1");
        b = g.rb(b, "trunk_B2.2", false, nullptr, true);
        g.head(b, 1, 1, nullptr, bt, dire);
        g.release(b);
    }
    // attention 2 gates x4 at 32x32 (:147-150), branch B3 -> pool -> out2 (:151-153)
    g.seg = 3;
    Act a2{};
    if (g.fused32_att("trunk_Att2.0")) {
        a2 = g.rb_fused32_att("trunk_Att2.0", qt, bt, dire, 1, 32);
    } else {
        Act aj = g.alloc(16, 32, 32, g.x6());
        if (g.live()) {
            KScope ks(c, K_SMALL, 0.0);
            g.check(launch_att_input(c->stream, qt, bt, dire, 1, aj.split ? nullptr : aj.p, n, 32, aj.split ? aj.s() : nullptr, aj.stride, g.fmt(), g.sat(), std::ldexp(1.f, -g.E(3))), "att_input");
        }
        g.note(aj, "att_input2", 3);
        a2 = g.rb(aj, "trunk_Att2.0");
    }
    Act xb3 = g.rb(a2, "trunk_Att2.1", false, &x4);
    g.release(x4);
    g.seg = 4;
    b = g.rb(xb3, "trunk_B3.0");
    b = g.rb(b, "trunk_B3.1");
    b = g.rb(b, "trunk_B3.2", true, nullptr, true);
    g.head(b, 2, 2, nullptr, bt, dire);
    g.release(b);
    return g.rc;
}

}  // namespace pmp
