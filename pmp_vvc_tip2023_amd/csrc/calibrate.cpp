// calibrate.cpp — the f16x3 activation scales of the MTT nets: calibration content, the calibration pass, its triggers and its two hooks of
// the C ABI (include/pmp.h, "Activation scales"; pmp_host.h: NetWeights::act_exp; the graph side is in nets.cpp, the scaled tensors in
// weights_pack.cpp: set_activation_scales).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pmp_host.h"

namespace pmp {

// ---- calibration of the f16x3 activation scales (pmp_host.h: NetWeights::act_exp; include/pmp.h) -------------------------------------
// The library's own calibration content: PMP_CAL_BLOCKS blocks that span what 8-bit pictures can do to a first layer - flat black and
// white, 1- and 2-pixel checkerboards, stripes, step edges, white noise, and smooth random content of three grain sizes (the kind
// recipe R makes).  Deterministic (a 64-bit LCG), so every context, rank and run derives the same exponents from the same weights.
constexpr int PMP_CAL_BLOCKS = 32;
constexpr int PMP_CAL_ATT_MAX_EXP = PMP_ACT_EXP_ATT_MAX;   // 6: largest exponent of an attention segment (its input, built from O(1) logits, must stay out of fp16's subnormals); pmp_hostonly.h - the .pmpw reader applies the same bounds to a manifest's exponents
constexpr int PMP_CAL_PASS = 16;            // blocks per calibration pass (its private workspace: 44 MB)
constexpr int PMP_CAL_TARGET_EXP = 12;      // a segment whose calibration maximum exceeds 2^12 is scaled down to it: 16x headroom to 65504

static void make_calibration_blocks(std::vector<uint8_t> &y, std::vector<uint8_t> &u, std::vector<uint8_t> &v)
{
    y.assign((size_t)PMP_CAL_BLOCKS * 68 * 68, 0); u.assign((size_t)PMP_CAL_BLOCKS * 34 * 34, 0); v.assign((size_t)PMP_CAL_BLOCKS * 34 * 34, 0);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(st >> 33); };
    auto fill = [&](uint8_t *p, int S, int kind, int b) {
        if (kind < 10) {
            for (int r = 0; r < S; ++r)
                for (int c = 0; c < S; ++c) {
                    int val = 0;
                    switch (kind) {
                    case 0: val = 0; break;
                    case 1: val = 255; break;
                    case 2: val = ((r + c) & 1) ? 255 : 0; break;                 // 1-px checkerboard
                    case 3: val = (((r >> 1) + (c >> 1)) & 1) ? 255 : 0; break;   // 2-px checkerboard
                    case 4: val = (c & 1) ? 255 : 0; break;                       // vertical stripes, period 2
                    case 5: val = ((r >> 1) & 1) ? 255 : 0; break;                // horizontal stripes, period 4
                    case 6: val = c < S / 2 ? 0 : 255; break;                     // vertical step edge
                    case 7: val = r < S / 2 ? 255 : 0; break;                     // horizontal step edge
                    case 8: val = (((r >> 2) + (c >> 2)) & 1) ? 235 : 16; break;  // 4-px checkerboard, video range
                    default: val = (r * 255) / (S - 1); break;                    // ramp
                    }
                    p[r * S + c] = (uint8_t)val;
                }
        } else if (kind < 14) {
            for (int i = 0; i < S * S; ++i) p[i] = (uint8_t)(rnd() & 255);      // white noise
        } else {   // smooth random: bilinear interpolation of a coarse random grid (grain 4, 8 or 16 px) + a little noise
            const int grain = 4 << (b % 3), G = S / grain + 2;
            std::vector<int> grid((size_t)G * G);
            for (auto &gv : grid) gv = (int)(rnd() & 255);
            for (int r = 0; r < S; ++r)
                for (int c = 0; c < S; ++c) {
                    const int gy = r / grain, gx = c / grain, fy = r % grain, fx = c % grain;
                    const int a00 = grid[gy * G + gx], a01 = grid[gy * G + gx + 1], a10 = grid[(gy + 1) * G + gx], a11 = grid[(gy + 1) * G + gx + 1];
                    int val = (a00 * (grain - fy) * (grain - fx) + a01 * (grain - fy) * fx + a10 * fy * (grain - fx) + a11 * fy * fx) / (grain * grain);
                    val += (int)(rnd() % 13) - 6;
                    p[r * S + c] = (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
                }
        }
    };
    for (int b = 0; b < PMP_CAL_BLOCKS; ++b) {
        fill(y.data() + (size_t)b * 68 * 68, 68, b, b);
        fill(u.data() + (size_t)b * 34 * 34, 34, b, b + 1);
        fill(v.data() + (size_t)b * 34 * 34, 34, b, b + 2);
    }
}

// Runs the (QT, MTT) pair of a component once on the calibration blocks - on the fp32 MFMA datapath, launch per layer, with the largest
// |value| of every MTT tensor recorded (nets.cpp: Graph::note) - and derives the segment exponents of the f16x3 datapath from them:
// e = the smallest exponent >= 0 that brings the segment's maximum to 2^PMP_CAL_TARGET_EXP or below (the attention trunks, segments 1 and
// 3, take theirs where their input is built from the logits).  Synchronises the stream (once per net: first use on the f16x3 datapath).
int calibrate_mtt(pmp_ctx *c, bool luma, NetWeights &wq, NetWeights &wb)
{
    int rc;
    if ((rc = ensure_datapath(c, wq, PMP_PRECISION_F32)) != PMP_OK || (rc = ensure_datapath(c, wb, PMP_PRECISION_F32)) != PMP_OK) return rc;
    static std::vector<uint8_t> hy, hu, hv;
    static std::once_flag once;
    std::call_once(once, [] { make_calibration_blocks(hy, hu, hv); });
    const int n = PMP_CAL_BLOCKS;
    const size_t o_u = ((size_t)n * 68 * 68 + 255) & ~(size_t)255, o_v = o_u + (((size_t)n * 34 * 34 + 255) & ~(size_t)255);
    const size_t o_q = o_v + (((size_t)n * 34 * 34 + 255) & ~(size_t)255), o_bt = o_q + (size_t)n * 64 * 4, o_dr = o_bt + (size_t)n * 768 * 4;
    if ((rc = ensure(c, c->d_calbuf, o_dr + (size_t)n * 768 * 4)) != PMP_OK) return rc;
    hipError_t e = hipSuccess;
    if (!c->d_cal) e = hipMalloc((void **)&c->d_cal, PMP_CAL_SLOTS * sizeof(unsigned));
    if (e == hipSuccess && !c->cal_stream) e = hipStreamCreateWithFlags(&c->cal_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return hip_fail(c, e, "calibration: stream");
    // Its OWN stream and workspace: the calibration depends on nothing the context's stream is doing (weights are uploaded synchronously,
    // its blocks and logits are its own), so it runs beside the passes in flight instead of behind them - a driver that loads the next
    // (component, QP) while the GPU works on this one (pmp_load_weights calibrates a pair as soon as it is complete) pays host time only.
    // Round 5's first form ran it on the context's stream inside the first inference call: 8 x 25 ms of exposed serialisation per 8-file job.
    hipStream_t user_stream = c->stream;
    const size_t user_need = c->ws_need;
    c->stream = c->cal_stream;
    std::swap(c->ws, c->ws_cal);
    struct Restore {
        pmp_ctx *c; hipStream_t s; size_t need;
        ~Restore() { c->stream = s; std::swap(c->ws, c->ws_cal); c->ws_need = need; c->cal_on = 0; }
    } restore{c, user_stream, user_need};
    char *base = static_cast<char *>(c->d_calbuf.p);
    if (e == hipSuccess) e = hipMemcpyAsync(base, hy.data(), hy.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(base + o_u, hu.data(), hu.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(base + o_v, hv.data(), hv.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_cal, 0, PMP_CAL_SLOTS * sizeof(unsigned), c->stream);
    if (e != hipSuccess) return hip_fail(c, e, "calibration: staging");
    const uint8_t *dy = (const uint8_t *)base, *du = (const uint8_t *)(base + o_u), *dv = (const uint8_t *)(base + o_v);
    float *dq = (float *)(base + o_q), *dbt = (float *)(base + o_bt), *ddr = (float *)(base + o_dr);
    const int saved = c->precision;
    c->precision = PMP_PRECISION_F32;
    // passes of PMP_CAL_PASS blocks in the private workspace (the context's own stays what its calls need: a 4-block call in 11 MB,
    // include/pmp.h); every pass folds into the same slots, so the log is that of the first pass
    rc = PMP_OK;
    for (int o = 0; o < n && rc == PMP_OK; o += PMP_CAL_PASS) {
        const int m = std::min(PMP_CAL_PASS, n - o);
        c->cal_log.clear();
        rc = run_graph_fn(c, [&] { return forward_q(c, luma, wq, dy + (size_t)o * 68 * 68, du + (size_t)o * 34 * 34, dv + (size_t)o * 34 * 34, m, dq + (size_t)o * 64); });
        c->cal_on = 1;
        if (rc == PMP_OK)
            rc = run_graph_fn(c, [&] { return forward_msbd(c, luma, wb, dy + (size_t)o * 68 * 68, du + (size_t)o * 34 * 34, dv + (size_t)o * 34 * 34, dq + (size_t)o * 64, m,
                                                        dbt + (size_t)o * 768, ddr + (size_t)o * 768); });
        c->cal_on = 0;
    }
    c->precision = saved;
    if (rc != PMP_OK) return rc;
    std::vector<unsigned> bits(PMP_CAL_SLOTS);
    e = hipMemcpyAsync(bits.data(), c->d_cal, PMP_CAL_SLOTS * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return hip_fail(c, e, "calibration: maxima");
    float seg_max[5] = {0, 0, 0, 0, 0};
    wb.cal_names.clear(); wb.cal_seg.clear(); wb.cal_amax.clear();
    for (size_t i = 0; i < c->cal_log.size(); ++i) {
        float m;
        std::memcpy(&m, &bits[i], 4);
        wb.cal_names.push_back(c->cal_log[i].first); wb.cal_seg.push_back(c->cal_log[i].second); wb.cal_amax.push_back(m);
        const int sg = c->cal_log[i].second;
        if (sg >= 0 && sg < 5 && (m > seg_max[sg] || m != m)) seg_max[sg] = m;
    }
    int exps[5] = {0, 0, 0, 0, 0};
    for (int sg = 0; sg < 5; ++sg) {           // a NaN / inf maximum leaves e = 0: that net needs the range guard's fp32 re-run anyway
        const float m = seg_max[sg];
        if (!(m == m) || std::isinf(m)) continue;
        int ex = 0;
        while (ex < PMP_ACT_EXP_MAX && m > std::ldexp(1.f, PMP_CAL_TARGET_EXP + ex)) ++ex;
        // An attention segment BEGINS with its smallest tensor - three channels of logits, O(1) - and one exponent serves the whole segment:
        // beyond 2^-6 that input would sink into fp16's subnormals (measured: a 2^18 gain inside an attention trunk, fully absorbed, cost
        // 1e-2 on the logits).  Capped there; a trunk that still leaves the range raises the flag and the call re-runs on fp32.
        if ((sg == 1 || sg == 3) && ex > PMP_CAL_ATT_MAX_EXP) ex = PMP_CAL_ATT_MAX_EXP;
        exps[sg] = ex;
    }
    if ((rc = set_activation_scales(c, wb, exps)) != PMP_OK) return rc;
    wb.calibrated = true;
    wb.act_from_file = false;
    wb.act_fp_known = true;          // these exponents belong to exactly this QT partner
    wb.act_qt_fp = wq.fp;
    return PMP_OK;
}

// A (QT, MTT) pair that has just become complete is calibrated at once (f16x3 activation scales, on the calibration's own stream): at
// LOAD time, where a pipelined host hides it, not inside its first inference call.  The lazy check in infer_passes stays for pairs loaded
// under another datapath.
int calibrate_if_ready(pmp_ctx *c, int net_id, int qp)
{
    if (c->precision != PMP_PRECISION_F16X3 || !c->act_scales) return PMP_OK;
    const bool luma = net_id == PMP_NET_LUMA_Q || net_id == PMP_NET_LUMA_MSBD;
    NetWeights *wq = find_net(c, luma ? PMP_NET_LUMA_Q : PMP_NET_CHROMA_Q, qp), *wb = find_net(c, luma ? PMP_NET_LUMA_MSBD : PMP_NET_CHROMA_MSBD, qp);
    if (!wq || !wb || wb->calibrated) return PMP_OK;
    return calibrate_mtt(c, luma, *wq, *wb);
}

// The exponents of an MTT net depend on its QT partner too - the raw QT logits feed the MTT stem and both attention inputs
// (Model_QBD.py:130,140,147).  Called when a QT net has been (re)loaded: exponents that were calibrated with, or whose manifest names,
// ANOTHER QT net are dropped (overflow would only cost fp32 re-runs; underflow would be silent), so that calibrate_if_ready - or the lazy
// check at the first f16x3 call - derives them again.  Exponents from a manifest without fingerprints (files written before round 6) are
// trusted at the moment of loading only: a QT net that arrives later cannot be checked against them.
void qt_partner_changed(pmp_ctx *c, int qt_net_id, int qp)
{
    const bool luma = qt_net_id == PMP_NET_LUMA_Q;
    NetWeights *wq = find_net(c, qt_net_id, qp), *wb = find_net(c, luma ? PMP_NET_LUMA_MSBD : PMP_NET_CHROMA_MSBD, qp);
    if (!wq || !wb || !wb->calibrated) return;
    if (wb->act_fp_known && wb->act_qt_fp == wq->fp) return;
    wb->calibrated = false;
    wb->act_from_file = false;
    wb->act_fp_known = false;
}

}  // namespace pmp

using namespace pmp;

#define CHECK_CTX(c) do { if (!(c)) return set_err(nullptr, PMP_E_INVALID, "null context"); hipSetDevice((c)->device); } while (0)

extern "C" {

int pmp_debug_set_activation_scales(pmp_ctx *c, int on)
{
    CHECK_CTX(c);
    const int rc = settle(c);
    if (rc != PMP_OK) return rc;
    c->act_scales = on ? 1 : 0;
    return PMP_OK;
}

int pmp_debug_activation_report(pmp_ctx *c, int comp, int qp, int exps[5], float seg_amax[5], char *buf, int64_t cap)
{
    CHECK_CTX(c);
    if (comp != PMP_LUMA && comp != PMP_CHROMA) return set_err(c, PMP_E_INVALID, "pmp_debug_activation_report: bad comp");
    const bool luma = comp == PMP_LUMA;
    NetWeights *wq = find_net(c, luma ? PMP_NET_LUMA_Q : PMP_NET_CHROMA_Q, qp), *wb = find_net(c, luma ? PMP_NET_LUMA_MSBD : PMP_NET_CHROMA_MSBD, qp);
    if (!wq || !wb) return set_err(c, PMP_E_NOWEIGHTS, "pmp_debug_activation_report: weights for this (comp, qp) are not loaded");
    int rc = settle(c);
    if (rc != PMP_OK) return rc;
    if (!wb->calibrated && (rc = calibrate_mtt(c, luma, *wq, *wb)) != PMP_OK) return rc;
    std::string out;
    float sm[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < wb->cal_names.size(); ++i) {
        char line[160];
        std::snprintf(line, sizeof line, "%s %d %.9g\n", wb->cal_names[i].c_str(), wb->cal_seg[i], (double)wb->cal_amax[i]);
        out += line;
        const int sg = wb->cal_seg[i];
        if (sg >= 0 && sg < 5 && (wb->cal_amax[i] > sm[sg] || wb->cal_amax[i] != wb->cal_amax[i])) sm[sg] = wb->cal_amax[i];
    }
    for (int i = 0; i < 5; ++i) { if (exps) exps[i] = wb->act_exp[i]; if (seg_amax) seg_amax[i] = sm[i]; }
    if (buf && cap > 0) {
        const size_t k = std::min((size_t)cap - 1, out.size());
        std::memcpy(buf, out.data(), k);
        buf[k] = 0;
    }
    return (int)wb->cal_names.size();
}

}  // extern "C"
