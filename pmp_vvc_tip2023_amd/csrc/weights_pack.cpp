// weights_pack.cpp — state_dict tensors (OIHW fp32, reference names) -> kernel layouts in device memory.
//
// Counterpart of load_pretrain_model (Inference_QBD.py:28-46): tensors are matched by name and shape; a missing or
// mis-shaped tensor is an error (the reference silently keeps random init for such tensors, which would be a
// silent accuracy bug here).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>

#include "pmp_host.h"

namespace pmp {

namespace {

struct Blob {
    const float *blob;
    const pmp_tensor_desc *descs;
    int ndesc;
    const pmp_tensor_desc *find(const std::string &name) const
    {
        for (int i = 0; i < ndesc; ++i)
            if (descs[i].name && name == descs[i].name) return &descs[i];
        return nullptr;
    }
};

int roundup16(int v) { return (v + 15) & ~15; }

// One device allocation and one host-to-device copy per build_net call: upload() only stages the bytes and remembers where the device
// pointer has to be written; finalize() allocates, copies and patches.  (250 hipMalloc + hipMemcpy pairs per MTT net were a third of
// the load time.)  The patched fields must have stable addresses until finalize(): NetWeights members and std::map values do.
struct Uploader {
    pmp_ctx *c;
    NetWeights *nw;
    std::vector<char> stage;
    std::vector<std::pair<void **, size_t>> fixups;
    int put(const void *src, size_t bytes, void **out)
    {
        const size_t off = (stage.size() + 255) & ~(size_t)255;
        stage.resize(off + bytes);
        if (bytes) std::memcpy(stage.data() + off, src, bytes);
        fixups.emplace_back(out, off);
        *out = reinterpret_cast<void *>(~(uintptr_t)0);     // "pending": non-null, so that nothing is staged twice; patched by finalize()
        return PMP_OK;
    }
    int upload(const std::vector<float> &h, float **out) { return put(h.data(), h.size() * sizeof(float), reinterpret_cast<void **>(out)); }
    int upload16(const std::vector<unsigned short> &h, unsigned short **out) { return put(h.data(), h.size() * sizeof(unsigned short), reinterpret_cast<void **>(out)); }
    void abandon() { for (auto &f : fixups) *f.first = nullptr; fixups.clear(); stage.clear(); }
    int finalize()
    {
        if (fixups.empty()) return PMP_OK;
        void *d = nullptr;
        hipError_t e = hipMalloc(&d, stage.size() ? stage.size() : 256);
        if (e != hipSuccess) { abandon(); return hip_fail(c, e, "hipMalloc(weights)"); }
        e = hipMemcpy(d, stage.data(), stage.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { hipFree(d); abandon(); return hip_fail(c, e, "hipMemcpy(weights)"); }
        for (auto &f : fixups) *f.first = static_cast<char *>(d) + f.second;
        nw->allocs.push_back(d);
        fixups.clear();
        stage.clear();
        return PMP_OK;
    }
};

}  // namespace

namespace {

int need(pmp_ctx *c, const Blob &b, const std::string &name, std::initializer_list<int> shape, const float **out)
{
    const pmp_tensor_desc *d = b.find(name);
    if (!d) return set_err(c, PMP_E_INVALID, "weights: missing tensor " + name);
    if (d->ndim != (int)shape.size()) return set_err(c, PMP_E_INVALID, "weights: wrong rank for " + name);
    int i = 0;
    for (int s : shape)
        if (d->shape[i++] != s) return set_err(c, PMP_E_INVALID, "weights: wrong shape for " + name);
    *out = b.blob + d->offset;
    return PMP_OK;
}

// mask: datapaths to pack for (bit = PMP_PRECISION_*).  A block that is already in the net keeps what it has; only the
// missing formats are added (ensure_datapath).
int load_rb(pmp_ctx *c, const Blob &b, Uploader &up, const std::string &name, int cin, int cout, int k, bool direct, unsigned mask)
{
    RBWeights &r = up.nw->rb[name];       // a map value: its address is stable, the uploader patches its pointers at the end
    r.cin = cin; r.cout = cout; r.k = k; r.direct = direct; r.has_sc = cin != cout;
    r.cin_pad = roundup16(cin); r.cout_pad = roundup16(cout);
    const float *w0, *w2, *wsc = nullptr;
    int rc;
    if ((rc = need(c, b, name + ".left.0.weight", {cout, cin, k, k}, &w0))) return rc;
    if ((rc = need(c, b, name + ".left.2.weight", {cout, cout, k, k}, &w2))) return rc;
    if (cin != cout && (rc = need(c, b, name + ".shortcut.0.weight", {cout, cin, 1, 1}, &wsc))) return rc;
    if (direct) {       // 8x8 maps: the direct fp32 kernel on every datapath
        if (!r.w0) {
            if ((rc = up.upload(pack_plain(w0, cout, cin, k, k), &r.w0))) return rc;
            if ((rc = up.upload(pack_plain(w2, cout, cout, k, k), &r.w2))) return rc;
            if (wsc && (rc = up.upload(pack_plain(wsc, cout, cin, 1, 1), &r.wsc))) return rc;
        }
    } else {
        if ((mask & (1u << PMP_PRECISION_F32)) && !r.w0) {
            if ((rc = up.upload(pack_mfma(w0, cout, cin, k, k, r.cout_pad, r.cin_pad), &r.w0))) return rc;
            if ((rc = up.upload(pack_mfma(w2, cout, cout, k, k, r.cout_pad, r.cout_pad), &r.w2))) return rc;
            if (wsc && (rc = up.upload(pack_mfma(wsc, cout, cin, 1, 1, r.cout_pad, r.cin_pad), &r.wsc))) return rc;
        }
        if ((mask & (1u << PMP_PRECISION_BF16X6)) && !r.w0x) {
            if ((rc = up.upload16(pack_x6(w0, cout, cin, k, k, r.cout_pad, r.cin_pad), &r.w0x))) return rc;
            if ((rc = up.upload16(pack_x6(w2, cout, cout, k, k, r.cout_pad, r.cout_pad), &r.w2x))) return rc;
            if (wsc && (rc = up.upload16(pack_x6(wsc, cout, cin, 1, 1, r.cout_pad, r.cin_pad), &r.wscx))) return rc;
        }
        if ((mask & (1u << PMP_PRECISION_F16X3)) && !r.w0h) {
            // fp16 split: the second conv and the 1x1 shortcut accumulate into one tile, so they share one scale
            r.k0 = h2_scale_exp(w0, (size_t)cout * cin * k * k);
            r.k2 = h2_scale_exp(w2, (size_t)cout * cout * k * k);
            if (wsc) r.k2 = std::min(r.k2, h2_scale_exp(wsc, (size_t)cout * cin));
            if ((rc = up.upload16(pack_h2(w0, cout, cin, k, k, r.cout_pad, r.cin_pad, r.k0), &r.w0h))) return rc;
            if ((rc = up.upload16(pack_h2(w2, cout, cout, k, k, r.cout_pad, r.cout_pad, r.k2), &r.w2h))) return rc;
            if (wsc && (rc = up.upload16(pack_h2(wsc, cout, cin, 1, 1, r.cout_pad, r.cin_pad, r.k2), &r.wsch))) return rc;
        }
        if ((rc = abl_pack_rb(w0, w2, k, cin, cout, mask, r, [&](const std::vector<unsigned short> &v, unsigned short **dst) { return up.upload16(v, dst); }))) return rc;
    }
    return PMP_OK;
}

int load_head(pmp_ctx *c, const Blob &b, Uploader &up, const std::string &name, int cout, int slot)
{
    const float *w, *bias;
    int rc;
    if ((rc = need(c, b, name + ".weight", {cout, 8, 3, 3}, &w))) return rc;
    if ((rc = need(c, b, name + ".bias", {cout}, &bias))) return rc;
    if (up.nw->head_w[slot]) return PMP_OK;
    if ((rc = up.upload(pack_plain(w, cout, 8, 3, 3), &up.nw->head_w[slot]))) return rc;
    return up.upload(std::vector<float>(bias, bias + cout), &up.nw->head_b[slot]);
}

}  // namespace

void free_net_weights(NetWeights &w)
{
    for (void *p : w.allocs) hipFree(p);
    w = NetWeights();
}

// Packs what `mask` asks for into nw (formats it already holds are kept).  nw is left as it is on failure: the caller decides.
static int build_net(pmp_ctx *c, int net_id, const Blob &b, NetWeights &nw, unsigned mask)
{
    const bool luma = net_id == PMP_NET_LUMA_Q || net_id == PMP_NET_LUMA_MSBD;
    const bool msbd = net_id == PMP_NET_LUMA_MSBD || net_id == PMP_NET_CHROMA_MSBD;
    Uploader up{c, &nw};
    int rc = PMP_OK;
    const bool want_h2 = mask & (1u << PMP_PRECISION_F16X3);
    auto fail = [&](int code) { up.abandon(); return code; };

    if (!msbd) {
        // Model_QBD.py:60-76 (luma) / :158-174 (chroma)
        const int cin = luma ? 1 : 3, k1 = luma ? 9 : 5, kq = luma ? 5 : 3;
        const float *w, *bias;
        if ((rc = need(c, b, "conv_q1.weight", {32, cin, k1, k1}, &w))) return fail(rc);
        if ((rc = need(c, b, "conv_q1.bias", {32}, &bias))) return fail(rc);
        if (!nw.stem_w) {
            if ((rc = up.upload(pack_plain(w, 32, cin, k1, k1), &nw.stem_w))) return fail(rc);
            if ((rc = up.upload(std::vector<float>(bias, bias + 32), &nw.stem_b))) return fail(rc);
        }
        if (want_h2 && !nw.stem_wh) {
            nw.stem_k = h2_scale_exp(w, (size_t)32 * cin * k1 * k1);
            if ((rc = up.upload16(pack_stem_h2(w, cin, k1, nw.stem_k), &nw.stem_wh))) return fail(rc);
        }
        if ((rc = load_rb(c, b, up, "resblock_q1", 32, 64, kq, false, mask))) return fail(rc);
        if ((rc = load_rb(c, b, up, "resblock_q2", 64, 64, kq, false, mask))) return fail(rc);
        if ((rc = load_rb(c, b, up, "resblock_q3", 64, 32, 3, false, mask))) return fail(rc);
        if ((rc = load_rb(c, b, up, "resblock_q4", 128, 32, 3, false, mask))) return fail(rc);
        if ((rc = load_rb(c, b, up, "resblock_q5", 32, 32, 3, false, mask))) return fail(rc);
        if ((rc = load_rb(c, b, up, "resblock_q6", 32, 8, 3, true, mask))) return fail(rc);   // 8x8 map: direct kernel
        if ((rc = load_head(c, b, up, "conv_q2", 1, 0))) return fail(rc);
    } else {
        // Model_QBD.py:101-125 (luma) / :199-223 (chroma)
        const int cin = luma ? 2 : 4, k1 = luma ? 9 : 5, k2 = luma ? 5 : 3;
        const float *w1, *w2, *w3, *b1, *b2, *b3;
        if ((rc = need(c, b, "conv_b1_1.weight", {16, cin, k1, k1}, &w1))) return fail(rc);
        if ((rc = need(c, b, "conv_b1_2.weight", {8, cin, k2, k1}, &w2))) return fail(rc);
        if ((rc = need(c, b, "conv_b1_3.weight", {8, cin, k1, k2}, &w3))) return fail(rc);
        if ((rc = need(c, b, "conv_b1_1.bias", {16}, &b1))) return fail(rc);
        if ((rc = need(c, b, "conv_b1_2.bias", {8}, &b2))) return fail(rc);
        if ((rc = need(c, b, "conv_b1_3.bias", {8}, &b3))) return fail(rc);
        if (!nw.stem_w) {
            std::vector<float> sw = pack_plain(w1, 16, cin, k1, k1);
            std::vector<float> t2 = pack_plain(w2, 8, cin, k2, k1), t3 = pack_plain(w3, 8, cin, k1, k2);
            sw.insert(sw.end(), t2.begin(), t2.end());
            sw.insert(sw.end(), t3.begin(), t3.end());
            std::vector<float> sb(b1, b1 + 16);
            sb.insert(sb.end(), b2, b2 + 8);
            sb.insert(sb.end(), b3, b3 + 8);
            if ((rc = up.upload(sw, &nw.stem_w))) return fail(rc);
            if ((rc = up.upload(sb, &nw.stem_b))) return fail(rc);
        }
        if (want_h2 && !nw.stem_wh) {   // the three stem convs as one top-left anchored k1 x k1 conv (the 5x9 / 9x5 kernels zero-padded), for the MFMA stem
            std::vector<float> w32((size_t)32 * cin * k1 * k1, 0.f);
            for (int co = 0; co < 16; ++co)
                for (int i = 0; i < cin * k1 * k1; ++i) w32[(size_t)co * cin * k1 * k1 + i] = w1[(size_t)co * cin * k1 * k1 + i];
            for (int co = 0; co < 8; ++co)
                for (int ci = 0; ci < cin; ++ci) {
                    for (int dy = 0; dy < k2; ++dy)
                        for (int dx = 0; dx < k1; ++dx)
                            w32[((((size_t)(16 + co)) * cin + ci) * k1 + dy) * k1 + dx] = w2[(((size_t)co * cin + ci) * k2 + dy) * k1 + dx];
                    for (int dy = 0; dy < k1; ++dy)
                        for (int dx = 0; dx < k2; ++dx)
                            w32[((((size_t)(24 + co)) * cin + ci) * k1 + dy) * k1 + dx] = w3[(((size_t)co * cin + ci) * k1 + dy) * k2 + dx];
                }
            nw.stem_k = h2_scale_exp(w32.data(), w32.size());
            if ((rc = up.upload16(pack_stem_h2(w32.data(), cin, k1, nw.stem_k), &nw.stem_wh))) return fail(rc);
        }
        if ((rc = load_rb(c, b, up, "trunk_M1.0", 32, 64, 5, false, mask))) return fail(rc);
        for (int i = 1; i < 6; ++i)
            if ((rc = load_rb(c, b, up, "trunk_M1." + std::to_string(i), 64, 64, 3, false, mask))) return fail(rc);
        for (int i = 0; i < 4; ++i)
            if ((rc = load_rb(c, b, up, "trunk_M2." + std::to_string(i), 64, 64, 3, false, mask))) return fail(rc);
        for (const char *t : {"trunk_B1", "trunk_B2", "trunk_B3"}) {
            if ((rc = load_rb(c, b, up, std::string(t) + ".0", 64, 32, 3, false, mask))) return fail(rc);
            if ((rc = load_rb(c, b, up, std::string(t) + ".1", 32, 16, 3, false, mask))) return fail(rc);
            if ((rc = load_rb(c, b, up, std::string(t) + ".2", 16, 8, 3, false, mask))) return fail(rc);
        }
        for (const char *t : {"trunk_Att1", "trunk_Att2"}) {
            if ((rc = load_rb(c, b, up, std::string(t) + ".0", 3, 32, 3, false, mask))) return fail(rc);
            if ((rc = load_rb(c, b, up, std::string(t) + ".1", 32, 64, 3, false, mask))) return fail(rc);
        }
        if ((rc = load_head(c, b, up, "conv_B1", 2, 0))) return fail(rc);
        if ((rc = load_head(c, b, up, "conv_B2", 2, 1))) return fail(rc);
        if ((rc = load_head(c, b, up, "conv_B3", 2, 2))) return fail(rc);
    }
    if ((rc = up.finalize()) != PMP_OK) return rc;
    nw.packed |= mask;
    return PMP_OK;
}

// The caller's tensors are validated (names, shapes) and packed for the context's CURRENT datapath only; the fp32 originals are kept
// on the host (1.8 - 4.3 MB per net) so that another datapath - a pmp_set_precision later, the bf16x6 re-run of the f16x3 range
// guard - is packed when it is first used (ensure_datapath): loading costs a third of packing all three formats up front.
int load_net_weights(pmp_ctx *c, int net_id, int qp, const float *blob, const pmp_tensor_desc *descs, int ndesc)
{
    if (net_id < 0 || net_id > 3 || !blob || !descs || ndesc <= 0) return set_err(c, PMP_E_INVALID, "pmp_load_weights: bad arguments");
    NetWeights nw;
    nw.net_id = net_id;
    // own copy of the tensors (the caller keeps ownership of its blob): names, shapes and values, re-based to one host vector
    size_t total = 0;
    for (int i = 0; i < ndesc; ++i) {
        if (!descs[i].name || descs[i].ndim < 0 || descs[i].ndim > 4 || descs[i].offset < 0) return set_err(c, PMP_E_INVALID, "pmp_load_weights: bad tensor descriptor");
        size_t cnt = 1;
        for (int j = 0; j < descs[i].ndim; ++j) { if (descs[i].shape[j] < 0) return set_err(c, PMP_E_INVALID, "pmp_load_weights: negative dimension"); cnt *= (size_t)descs[i].shape[j]; }
        total += cnt;
    }
    nw.host.reserve(total);
    nw.names.resize(ndesc);
    nw.descs.resize(ndesc);
    for (int i = 0; i < ndesc; ++i) {
        size_t cnt = 1;
        for (int j = 0; j < descs[i].ndim; ++j) cnt *= (size_t)descs[i].shape[j];
        nw.names[i] = descs[i].name;
        nw.descs[i] = descs[i];
        nw.descs[i].offset = (int64_t)nw.host.size();
        nw.host.insert(nw.host.end(), blob + descs[i].offset, blob + descs[i].offset + cnt);
    }
    for (int i = 0; i < ndesc; ++i) nw.descs[i].name = nw.names[i].c_str();
    nw.fp = fingerprint_tensors(nw.host.data(), nw.descs.data(), ndesc);
    Blob b{nw.host.data(), nw.descs.data(), ndesc};
    const int rc = build_net(c, net_id, b, nw, (1u << c->precision) | abl_pack_mask(c));
    if (rc != PMP_OK) { free_net_weights(nw); return rc; }
    nw.loaded = true;
    const int key = net_id * 100 + qp;
    auto it = c->nets.find(key);
    if (it != c->nets.end()) free_net_weights(it->second);
    NetWeights &slot = c->nets[key];
    slot = std::move(nw);
    for (size_t i = 0; i < slot.descs.size(); ++i) slot.descs[i].name = slot.names[i].c_str();   // the strings moved with the vector; re-point to be safe
    return PMP_OK;
}

// f16x3 activation scales of an MTT net (pmp_host.h: NetWeights::act_exp): the scaled copies of the tensors that carry the changes of
// scale - stem biases * 2^-e0 (the stem's output scale takes the same factor in nets.cpp), head weights * 2^e of the segment the head reads.
// Powers of two: every product and sum of the scaled net is the unscaled one times a power of two, exactly.
int set_activation_scales(pmp_ctx *c, NetWeights &nw, const int exps[5])
{
    if (nw.net_id != PMP_NET_LUMA_MSBD && nw.net_id != PMP_NET_CHROMA_MSBD) return set_err(c, PMP_E_INVALID, "activation scales: MTT nets only");
    Blob b{nw.host.data(), nw.descs.data(), (int)nw.descs.size()};
    std::vector<float> stage;
    int rc;
    const float *bias[3];
    static const char *stems[3] = {"conv_b1_1.bias", "conv_b1_2.bias", "conv_b1_3.bias"};
    static const int nb[3] = {16, 8, 8};
    for (int i = 0; i < 3; ++i) {
        if ((rc = need(c, b, stems[i], {nb[i]}, &bias[i]))) return rc;
        for (int j = 0; j < nb[i]; ++j) stage.push_back(std::ldexp(bias[i][j], -exps[0]));
    }
    static const char *heads[3] = {"conv_B1.weight", "conv_B2.weight", "conv_B3.weight"};
    for (int k = 0; k < 3; ++k) {
        const float *w;
        if ((rc = need(c, b, heads[k], {2, 8, 3, 3}, &w))) return rc;
        std::vector<float> p = pack_plain(w, 2, 8, 3, 3);
        for (float v : p) stage.push_back(std::ldexp(v, exps[2 * k]));
    }
    const size_t nhead = (stage.size() - 32) / 3;
    if (!nw.stem_b_h) {
        void *d = nullptr;
        hipError_t e = hipMalloc(&d, stage.size() * sizeof(float));
        if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(activation scales)");
        nw.allocs.push_back(d);
        nw.stem_b_h = static_cast<float *>(d);
        for (int k = 0; k < 3; ++k) nw.head_w_h[k] = nw.stem_b_h + 32 + k * nhead;
    }
    hipError_t e = hipMemcpy(nw.stem_b_h, stage.data(), stage.size() * sizeof(float), hipMemcpyHostToDevice);   // synchronous: nothing in flight reads an uncalibrated net's copies
    if (e != hipSuccess) return hip_fail(c, e, "hipMemcpy(activation scales)");
    for (int i = 0; i < 5; ++i) nw.act_exp[i] = exps[i];
    return PMP_OK;
}

int ensure_datapath(pmp_ctx *c, NetWeights &nw, int precision)
{
    const unsigned bit = 1u << precision;
    if (nw.packed & bit) return PMP_OK;
    Blob b{nw.host.data(), nw.descs.data(), (int)nw.descs.size()};
    return build_net(c, nw.net_id, b, nw, bit);
}

}  // namespace pmp
