// pmpw_file.cpp — reader of the product's weight container (.pmpw, written by pmp_vvc_tip2023_amd/weights.py):
//
//     "PMPW1\n" | u32 little-endian JSON length | JSON manifest | raw little-endian float32 payload
//     manifest = {"net": "Luma_Q", "qp": 22, "source": "...", "tensors": [{"name": "...", "shape": [..], "offset": N}, ...]
//                 [, "act_exp": [e0, e1, e2, e3, e4]        (MTT nets: f16x3 activation-scale exponents, then no calibration at load)
//                  , "act_fp": ["<16 hex: this net's tensors>", "<16 hex: the QT partner's>"]]}   (what the exponents were calibrated on)
//
// so that a host without Python (the in-process VTM hook) can feed pmp_load_weights.  Counterpart of
// load_pretrain_model (Inference_QBD.py:28-46).  Pure host code (no HIP): part of the sanitizer test library too.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>

#include "pmp_hostonly.h"

namespace pmp {

namespace {

// Minimal scanner for the manifest's own schema (objects, arrays, strings without escapes beyond \" and \\, integers).
struct Scan {
    const char *p, *end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool lit(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
    bool str(std::string &out)
    {
        ws();
        if (p >= end || *p != '"') return false;
        ++p;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') { if (++p >= end) return false; }
            out.push_back(*p++);
        }
        if (p >= end) return false;
        ++p;
        return true;
    }
    bool integer(long long &v)   // bounded by `end` (the manifest is not NUL-terminated) and by 18 digits (no overflow)
    {
        ws();
        const char *q = p;
        bool neg = false;
        if (q < end && *q == '-') { neg = true; ++q; }
        if (q >= end || *q < '0' || *q > '9') return false;
        long long acc = 0;
        int digits = 0;
        while (q < end && *q >= '0' && *q <= '9') {
            if (++digits > 18) return false;
            acc = acc * 10 + (*q++ - '0');
        }
        v = neg ? -acc : acc;
        p = q;
        return true;
    }
    bool skip_value()   // any JSON value
    {
        ws();
        if (p >= end) return false;
        if (*p == '"') { std::string s; return str(s); }
        if (*p == '{' || *p == '[') {
            const char open = *p, close = open == '{' ? '}' : ']';
            ++p;
            if (lit(close)) return true;
            for (;;) {
                if (open == '{') { std::string k; if (!str(k) || !lit(':')) return false; }
                if (!skip_value()) return false;
                if (lit(',')) continue;
                return lit(close);
            }
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n') ++p;   // number / true / false / null
        return true;
    }
};

}  // namespace

int read_pmpw(const char *path, WeightFile &wf)
{
    wf = WeightFile();
    if (!path) return set_err_global(PMP_E_INVALID, "pmp weights file: null path");
    FILE *fp = fopen(path, "rb");
    if (!fp) return set_err_global(PMP_E_IO, std::string("cannot open ") + path);
    std::vector<char> raw;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), fp)) > 0) raw.insert(raw.end(), buf, buf + got);
    fclose(fp);
    if (raw.size() < 10 || memcmp(raw.data(), "PMPW1\n", 6) != 0) return set_err_global(PMP_E_INVALID, std::string(path) + ": not a PMPW1 file");
    uint32_t jl = 0;
    for (int i = 0; i < 4; ++i) jl |= (uint32_t)(unsigned char)raw[6 + i] << (8 * i);
    if ((size_t)jl > raw.size() - 10) return set_err_global(PMP_E_INVALID, std::string(path) + ": manifest length exceeds the file");
    const size_t pay_off = 10 + (size_t)jl, nfl = (raw.size() - pay_off) / 4;
    wf.payload.resize(nfl);
    if (nfl) memcpy(wf.payload.data(), raw.data() + pay_off, nfl * 4);   // little-endian host (x86-64): raw copy

    Scan s{raw.data() + 10, raw.data() + 10 + jl};
    auto bad = [&](const char *what) { return set_err_global(PMP_E_INVALID, std::string(path) + ": manifest: " + what); };
    if (!s.lit('{')) return bad("not an object");
    bool have_tensors = false;
    if (!s.lit('}')) {
        for (;;) {
            std::string key;
            if (!s.str(key) || !s.lit(':')) return bad("bad key");
            if (key == "net") { if (!s.str(wf.net)) return bad("net is not a string"); }
            else if (key == "qp") { long long v; if (!s.integer(v)) return bad("qp is not an integer"); wf.qp = (int)v; }
            else if (key == "act_exp") {   // optional: [e0, e1, e2, e3, e4], written by the conversion tool after a calibration on the target GPU
                if (!s.lit('[')) return bad("act_exp is not an array");
                if (!s.lit(']')) {
                    for (;;) {
                        long long v;
                        // bounded like the calibration's own choices (calibrate.cpp): a trunk segment up to 2^-30, an attention segment up to 2^-6
                        // (its O(1) input would sink into fp16's subnormals beyond that and silently lose the logits; overflow has the range
                        // guard behind it, underflow has nothing)
                        const size_t sg = wf.act_exp.size();
                        if (!s.integer(v) || v < 0 || sg >= 5 || v > ((sg == 1 || sg == 3) ? PMP_ACT_EXP_ATT_MAX : PMP_ACT_EXP_MAX))
                            return bad("act_exp entry (five integers: 0..30 for segments 0, 2, 4; 0..6 for the attention segments 1, 3)");
                        wf.act_exp.push_back((int)v);
                        if (s.lit(',')) continue;
                        if (!s.lit(']')) return bad("act_exp end");
                        break;
                    }
                }
                if (wf.act_exp.size() != 5) return bad("act_exp needs five entries");
            }
            else if (key == "act_fp") {    // optional: what "act_exp" was calibrated on - [this net's fingerprint, its QT partner's], 16 hex digits each
                std::string a, b;
                if (!s.lit('[') || !s.str(a) || !s.lit(',') || !s.str(b) || !s.lit(']')) return bad("act_fp is not an array of two strings");
                auto hex = [](const std::string &h, uint64_t &out) {
                    if (h.size() != 16) return false;
                    out = 0;
                    for (char ch : h) {
                        const int d = ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : -1;
                        if (d < 0) return false;
                        out = (out << 4) | (uint64_t)d;
                    }
                    return true;
                };
                if (!hex(a, wf.act_mtt_fp) || !hex(b, wf.act_qt_fp)) return bad("act_fp entry (16 lower-case hex digits)");
                wf.have_fp = true;
            }
            else if (key == "tensors") {
                have_tensors = true;
                if (!s.lit('[')) return bad("tensors is not an array");
                if (!s.lit(']')) {
                    for (;;) {
                        WeightTensor t;
                        bool have_off = false;
                        if (!s.lit('{')) return bad("tensor entry is not an object");
                        for (;;) {
                            std::string k;
                            if (!s.str(k) || !s.lit(':')) return bad("bad tensor key");
                            if (k == "name") { if (!s.str(t.name)) return bad("tensor name"); }
                            else if (k == "offset") { long long v; if (!s.integer(v) || v < 0) return bad("tensor offset"); t.offset = v; have_off = true; }
                            else if (k == "shape") {
                                if (!s.lit('[')) return bad("tensor shape");
                                if (!s.lit(']')) {
                                    for (;;) {
                                        long long v;
                                        if (!s.integer(v) || v < 0 || v > (1 << 24) || t.ndim >= 4) return bad("tensor shape entry");
                                        t.shape[t.ndim++] = (int)v;
                                        if (s.lit(',')) continue;
                                        if (!s.lit(']')) return bad("tensor shape end");
                                        break;
                                    }
                                }
                            } else if (!s.skip_value()) return bad("tensor value");
                            if (s.lit(',')) continue;
                            if (!s.lit('}')) return bad("tensor entry end");
                            break;
                        }
                        long long cnt = 1;   // bounded while multiplying: four dimensions of up to 2^24 would overflow 64 bits
                        for (int i = 0; i < t.ndim; ++i) {
                            cnt *= t.shape[i];
                            if (cnt > (long long)nfl) return bad("tensor outside the payload");
                        }
                        if (t.name.empty() || !have_off || t.offset > (long long)nfl - cnt) return bad("tensor outside the payload");
                        wf.tensors.push_back(t);
                        if (s.lit(',')) continue;
                        if (!s.lit(']')) return bad("tensors end");
                        break;
                    }
                }
            } else if (!s.skip_value()) return bad("bad value");
            if (s.lit(',')) continue;
            if (!s.lit('}')) return bad("object end");
            break;
        }
    }
    if (!have_tensors || wf.tensors.empty()) return bad("no tensors");
    return PMP_OK;
}

// Fingerprint of a tensor set, independent of the order and layout the caller hands it over in: tensors sorted by name; per tensor
// FNV-1a over the name, the shape folded in, then a position-weighted sum of the float BIT patterns (vectorisable: weights.py computes
// the same number with numpy).  Not cryptographic - it catches the stale file, not the adversary.
uint64_t fingerprint_tensors(const float *blob, const pmp_tensor_desc *descs, int ndesc)
{
    constexpr uint64_t P = 0x100000001b3ull;
    std::vector<int> order(ndesc);
    for (int i = 0; i < ndesc; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return strcmp(descs[a].name, descs[b].name) < 0; });
    uint64_t fp = 0xcbf29ce484222325ull;
    for (int i : order) {
        uint64_t h = 0xcbf29ce484222325ull;
        for (const char *c = descs[i].name; *c; ++c) h = (h ^ (uint64_t)(unsigned char)*c) * P;
        uint64_t cnt = 1;
        for (int j = 0; j < descs[i].ndim; ++j) { h = (h ^ (uint64_t)descs[i].shape[j]) * P; cnt *= (uint64_t)descs[i].shape[j]; }
        const float *w = blob + descs[i].offset;
        uint64_t sum = 0;
        for (uint64_t k = 0; k < cnt; ++k) {
            uint32_t bits;
            memcpy(&bits, w + k, 4);
            sum += ((uint64_t)bits + 0x9E3779B97F4A7C15ull) * (2 * k + 1);
        }
        h = (h ^ sum) * P;
        fp = (fp ^ h) * P;
    }
    return fp;
}

int net_id_of(const std::string &net)
{
    if (net == "Luma_Q") return PMP_NET_LUMA_Q;
    if (net == "Luma_MSBD") return PMP_NET_LUMA_MSBD;
    if (net == "Chroma_Q") return PMP_NET_CHROMA_Q;
    if (net == "Chroma_MSBD") return PMP_NET_CHROMA_MSBD;
    return -1;
}

}  // namespace pmp

extern "C" int pmp_fingerprint_tensors(const float *blob, const pmp_tensor_desc *descs, int ndesc, uint64_t *out)
{
    if (!blob || !descs || ndesc <= 0 || !out) return pmp::set_err_global(PMP_E_INVALID, "pmp_fingerprint_tensors: bad arguments");
    for (int i = 0; i < ndesc; ++i)
        if (!descs[i].name || descs[i].ndim < 0 || descs[i].ndim > 4 || descs[i].offset < 0) return pmp::set_err_global(PMP_E_INVALID, "pmp_fingerprint_tensors: bad tensor descriptor");
    *out = pmp::fingerprint_tensors(blob, descs, ndesc);
    return PMP_OK;
}

extern "C" int pmp_debug_read_weights_file(const char *path, int *net_id, int *qp, int *ntensors, int64_t *nfloats, double *checksum)
{
    pmp::WeightFile wf;
    const int rc = pmp::read_pmpw(path, wf);
    if (rc != PMP_OK) return rc;
    if (net_id) *net_id = pmp::net_id_of(wf.net);
    if (qp) *qp = wf.qp;
    if (ntensors) *ntensors = (int)wf.tensors.size();
    if (nfloats) *nfloats = (int64_t)wf.payload.size();
    if (checksum) {   // sum over the tensors' elements (order of the manifest): what a caller of pmp_load_weights would see
        double s = 0;
        for (const auto &t : wf.tensors) {
            long long cnt = 1;
            for (int i = 0; i < t.ndim; ++i) cnt *= t.shape[i];
            for (long long i = 0; i < cnt; ++i) s += wf.payload[(size_t)(t.offset + i)];
        }
        *checksum = s;
    }
    return PMP_OK;
}
