// abl/abl_host.cpp — MEASUREMENT library: the host side of what the product compiles as no-op hooks (hooks/abl_hooks.h): the process-wide
// kernel-form selector and its environment knob, the Winograd-x weight streams, the in-kernel stamp reports of pmp_debug_conv_bench.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pmp_host.h"

namespace pmp {

// Winograd F(2, 3) along x for a 3x3 64 -> 64 convolution (conv_f16x3_wx.hip).  U_p[ky] = sum_kx G[p][kx] w[ky][kx] in fp64, scaled by
// S = 2^k (max |S U| in [4096, 8192)), two fp16 terms.  Stream [pair P 2][step s 3][position p 4][split 2][cout group 4][64 lanes][8]:
// a K-step is 16 channels x two vertical taps -  s = 0: (ky0, ky1) of group 2P;  s = 1: ky2 of group 2P and ky2 of group 2P + 1;
// s = 2: (ky0, ky1) of group 2P + 1.  Lane l of cout group ct: cout 16 ct + (l & 15), channels 8 ((l >> 4) & 1) + j of the tap/group
// selected by l >> 5.
std::vector<unsigned short> pack_h2_wx(const float *w, int *scale_exp)
{
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> U((size_t)4 * 3 * 64 * 64);          // [p][ky][co][ci]
    for (int p = 0; p < 4; ++p)
        for (int ky = 0; ky < 3; ++ky)
            for (int co = 0; co < 64; ++co)
                for (int ci = 0; ci < 64; ++ci) {
                    double u = 0;
                    for (int kx = 0; kx < 3; ++kx) u += G[p][kx] * (double)w[(((size_t)co * 64 + ci) * 3 + ky) * 3 + kx];
                    U[(((size_t)p * 3 + ky) * 64 + co) * 64 + ci] = (float)u;     // |u| <= 1.5 max|w|: a float holds it to 2^-24
                }
    const int kexp = h2_scale_exp(U.data(), U.size());
    if (scale_exp) *scale_exp = kexp;
    const float S = std::ldexp(1.f, kexp);
    std::vector<unsigned short> out((size_t)2 * 3 * 4 * 2 * 4 * 64 * 8, 0);
    for (int P = 0; P < 2; ++P)
        for (int st = 0; st < 3; ++st) {
            const int cbA = st == 2 ? 2 * P + 1 : 2 * P, kyA = st == 1 ? 2 : 0;
            const int cbB = st == 0 ? 2 * P : 2 * P + 1, kyB = st == 1 ? 2 : 1;
            for (int p = 0; p < 4; ++p)
                for (int ct = 0; ct < 4; ++ct)
                    for (int l = 0; l < 64; ++l) {
                        const int g = l >> 4, cb = (g >> 1) ? cbB : cbA, ky = (g >> 1) ? kyB : kyA;
                        const int co = ct * 16 + (l & 15), ci0 = cb * 16 + 8 * (g & 1);
                        float v[8];
                        for (int j = 0; j < 8; ++j) v[j] = U[(((size_t)p * 3 + ky) * 64 + co) * 64 + ci0 + j] * S;
                        const size_t base = ((((size_t)(P * 3 + st) * 4 + p) * 2) * 4 + ct) * 64 + l;
                        h2_split8(v, out.data() + base * 8, out.data() + (base + 4 * 64) * 8);
                    }
        }
    return out;
}

const char *abl_version() { return "pmp-hip 0.5-abl (gfx950; f16x3 / bf16x6 split MFMA + fp32 MFMA; MEASUREMENT BUILD with timing-only kernels)"; }

void abl_on_create()
{
    if (const char *v = getenv("PMP_CONV_VARIANT")) g_conv_variant = atoi(v);
}

bool abl_set_conv_variant(int variant, int *rc)
{
    if (variant < 0 || variant > 4095) *rc = set_err(nullptr, PMP_E_INVALID, "pmp_debug_set_conv_variant: 0..9, or 10 + bits for the timing-only builds");
    else { g_conv_variant = variant; *rc = PMP_OK; }
    return true;
}

bool abl_set_winograd(pmp_ctx *c, int on, int *rc)
{
    c->abl.winograd = on ? 1 : 0;
    *rc = PMP_OK;
    return true;
}

unsigned abl_pack_mask(const pmp_ctx *c) { return c->abl.winograd ? 1u << 3 : 0u; }     // pseudo-datapath 3: the Winograd-x streams

int abl_prepare_pass(pmp_ctx *c, NetWeights &wq, NetWeights &wb)
{
    if (!c->abl.winograd || c->precision != PMP_PRECISION_F16X3) return PMP_OK;
    int rc = ensure_datapath(c, wq, 3);
    return rc != PMP_OK ? rc : ensure_datapath(c, wb, 3);
}

void abl_conv_args(const pmp_ctx *c, const RBWeights &r, bool second, ConvX6Args &a)
{
    a.abl.zeros = c->d_sat + 16;
    if (c->abl.winograd && r.abl.w0w) { a.abl.w_wx = second ? r.abl.w2w : r.abl.w0w; a.abl.wx_out_scale = std::ldexp(1.f, -(second ? r.abl.k2w : r.abl.k0w)); }
}

int abl_pack_rb(const float *w0, const float *w2, int k, int cin, int cout, unsigned mask, RBWeights &r,
                const std::function<int(const std::vector<unsigned short> &, unsigned short **)> &upload16)
{
    if (!(mask & (1u << 3)) || k != 3 || cin != 64 || cout != 64 || r.abl.w0w) return PMP_OK;
    int rc = upload16(pack_h2_wx(w0, &r.abl.k0w), &r.abl.w0w);
    return rc != PMP_OK ? rc : upload16(pack_h2_wx(w2, &r.abl.k2w), &r.abl.w2w);
}

void abl_bench_prepare(pmp_ctx *c, AblBench &ab, const float *w, int k, int cin, int cout, bool h2, ConvX6Args &b)
{
    b.abl.zeros = c->d_sat + 16;
    ab.wino = h2 && c->abl.winograd && k == 3 && cin == 64 && cout == 64;     // the Winograd-x form of this layer (conv_f16x3_wx.hip)
    if (!ab.wino) return;
    const std::vector<unsigned short> ww = pack_h2_wx(w, &ab.kexp_w);
    if (hipMalloc((void **)&ab.dww, ww.size() * 2) != hipSuccess) { ab.wino = false; return; }
    hipMemcpy(ab.dww, ww.data(), ww.size() * 2, hipMemcpyHostToDevice);
    b.abl.w_wx = ab.dww;
    b.abl.wx_out_scale = std::ldexp(1.f, -ab.kexp_w);
}

void abl_bench_free(AblBench &ab)
{
    if (ab.dww) hipFree(ab.dww);
    ab.dww = nullptr;
}

// in-kernel stamp reports of the diagnostic builds (conv variants 10 + 128 ...)
void abl_bench_report(pmp_ctx *c, AblBench &, bool h2, int n, int h, int w, int k, int cout, ConvX6Args &b, const std::function<hipError_t()> &launch_split)
{
        if (!h2 && g_conv_variant == 10 + 128 && k == 3 && cout == 64) {   // in-kernel stamp report (diagnostic build)
            const int wgs = n * (h / 16) * (w / 16);
            unsigned long long *ddbg = nullptr;
            if (hipMalloc((void **)&ddbg, (size_t)wgs * 8 * 8) == hipSuccess) {
                hipMemset(ddbg, 0, (size_t)wgs * 8 * 8);
                b.abl.dbg = ddbg;
                launch_conv_x6(c->stream, b);
                hipStreamSynchronize(c->stream);
                std::vector<unsigned long long> hd((size_t)wgs * 8);
                hipMemcpy(hd.data(), ddbg, hd.size() * 8, hipMemcpyDeviceToHost);
                double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                unsigned long long tmin = ~0ull, tmax = 0;
                for (int i = 0; i < wgs; ++i) {
                    for (int j = 0; j < 8; ++j) if (j != 5) s[j] += (double)hd[(size_t)i * 8 + j];
                    if (hd[(size_t)i * 8 + 5] < tmin) tmin = hd[(size_t)i * 8 + 5];
                    if (hd[(size_t)i * 8 + 5] + hd[(size_t)i * 8 + 3] + hd[(size_t)i * 8 + 4] > tmax) tmax = hd[(size_t)i * 8 + 5] + hd[(size_t)i * 8 + 3] + hd[(size_t)i * 8 + 4];
                }
                fprintf(stderr, "stamps (mean ticks per workgroup, wave 0): prologue %.0f | K-steps %.0f | wait for staged loads %.0f | "
                                "LDS store %.0f | barrier %.0f | accumulate total %.0f | epilogue incl. store ack %.0f\n",
                        s[0] / wgs, s[1] / wgs, s[6] / wgs, s[2] / wgs, s[7] / wgs, s[3] / wgs, s[4] / wgs);
                b.abl.dbg = nullptr;
                hipFree(ddbg);
            }
        }
        if (h2 && ((g_conv_variant >= 10 + 128 && g_conv_variant < 10 + 144) || g_conv_variant == 10 + 1152) && k == 3 && cout == 64) {   // 128 + ablation bits 1/2/4; 1152: the three-workgroup form   // in-kernel stamp report (diagnostic build)
            const int wgs = n * (h / 16) * (w / 16);
            unsigned long long *ddbg = nullptr;
            if (hipMalloc((void **)&ddbg, (size_t)wgs * 16 * 8) == hipSuccess) {
                hipMemset(ddbg, 0, (size_t)wgs * 16 * 8);
                b.abl.dbg = ddbg;
                launch_split();
                hipStreamSynchronize(c->stream);
                std::vector<unsigned long long> hd((size_t)wgs * 16);
                hipMemcpy(hd.data(), ddbg, hd.size() * 8, hipMemcpyDeviceToHost);
                double s[6] = {0, 0, 0, 0, 0, 0};
                unsigned long long tmin = ~0ull, tmax = 0;
                for (int i = 0; i < wgs; ++i) {
                    for (int j = 0; j < 6; ++j) s[j] += (double)hd[(size_t)i * 16 + j];
                    tmin = std::min(tmin, hd[(size_t)i * 16 + 6]);
                    tmax = std::max(tmax, hd[(size_t)i * 16 + 7]);
                }
                fprintf(stderr, "f16x3 stamps (mean ticks per workgroup, wave 0): prologue %.0f | K-steps %.0f | halo LDS store incl. its wait %.0f | "
                                "group barrier %.0f | accumulate total %.0f | epilogue incl. store ack %.0f | kernel span %.0f ticks, %d workgroups\n",
                        s[0] / wgs, s[1] / wgs, s[2] / wgs, s[3] / wgs, s[4] / wgs, s[5] / wgs, (double)(tmax - tmin), wgs);
                if (const char *dump = getenv("PMP_STAMP_DUMP")) {   // raw stamps, 16 x u64 per workgroup (tools/stamp_overlap.py)
                    if (FILE *f = fopen(dump, "wb")) { fwrite(hd.data(), 8, hd.size(), f); fclose(f); }
                }
                b.abl.dbg = nullptr;
                hipFree(ddbg);
            }
        }
}

}  // namespace pmp
