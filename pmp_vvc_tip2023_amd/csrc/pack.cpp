// pack.cpp — OIHW fp32 conv weights -> the kernels' fragment-order streams (pure host code, no HIP: also part of the
// sanitizer test library, make hostasan).  Layout contracts: conv_mfma.hip, conv_bf16x6.hip, conv_f16x3.hip, conv_misc.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>

#include "pmp_hostonly.h"

namespace pmp {

// OIHW conv weight -> MFMA A-operand fragments [Cin_pad/16][KH*KW][Cout_pad/16][64 lanes][4]:
// lane l of cout-tile nt holds W[cout = 16nt + (l&15)][channel = 16cb + 4(l>>4) + j], j = 0..3 (conv_mfma.hip).
std::vector<float> pack_mfma(const float *w, int cout, int cin, int kh, int kw, int cout_pad, int cin_pad)
{
    const int taps = kh * kw, CB = cin_pad / 16, NT = cout_pad / 16;
    std::vector<float> out((size_t)CB * taps * NT * 64 * 4, 0.f);
    for (int cb = 0; cb < CB; ++cb)
        for (int t = 0; t < taps; ++t)
            for (int nt = 0; nt < NT; ++nt)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 4; ++j) {
                        const int co = nt * 16 + (l & 15), ci = cb * 16 + 4 * (l >> 4) + j;
                        float v = 0.f;
                        if (co < cout && ci < cin) v = w[((size_t)co * cin + ci) * taps + t];
                        out[((((size_t)cb * taps + t) * NT + nt) * 64 + l) * 4 + j] = v;
                    }
    return out;
}
// fp32 -> bf16, round to nearest even (weights are finite)
static inline unsigned short bf16_rne(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf16_f32(unsigned short h)
{
    unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// ---- two fp16 terms of 8 consecutive values (v = h0 + h1, round to nearest even, subnormals kept): the inner step of every f16x3
// packer.  Through compiler-rt's soft-float conversions this is what weight loading spends its time on (1.1 ms for one 3x3 64->64
// tensor, 15 ms per MTT net); with the F16C conversions of the host CPU - selected at run time - the same bits come 10x faster.
static void split8_generic(const float *v, unsigned short *h0, unsigned short *h1)
{
    for (int j = 0; j < 8; ++j) {
        const _Float16 a = (_Float16)v[j];
        const _Float16 b = (_Float16)(v[j] - (float)a);
        std::memcpy(h0 + j, &a, 2);
        std::memcpy(h1 + j, &b, 2);
    }
}
#if defined(__x86_64__)
}  // namespace pmp
#include <immintrin.h>
namespace pmp {
__attribute__((target("avx,f16c"))) static void split8_f16c(const float *v, unsigned short *h0, unsigned short *h1)
{
    const __m256 x = _mm256_loadu_ps(v);
    const __m128i a = _mm256_cvtps_ph(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
    const __m256 r = _mm256_sub_ps(x, _mm256_cvtph_ps(a));          // exact in fp32
    const __m128i b = _mm256_cvtps_ph(r, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
    _mm_storeu_si128(reinterpret_cast<__m128i *>(h0), a);
    _mm_storeu_si128(reinterpret_cast<__m128i *>(h1), b);
}
static bool have_f16c() { static const bool f = __builtin_cpu_supports("avx") && __builtin_cpu_supports("f16c"); return f; }
static inline void split8(const float *v, unsigned short *h0, unsigned short *h1) { if (have_f16c()) split8_f16c(v, h0, h1); else split8_generic(v, h0, h1); }
#else
static inline void split8(const float *v, unsigned short *h0, unsigned short *h1) { split8_generic(v, h0, h1); }
#endif

void h2_split8(const float *v, unsigned short *h0, unsigned short *h1) { split8(v, h0, h1); }


// OIHW conv weight -> split-3 bf16 MFMA A-operand fragments (conv_bf16x6.hip), one K-step = 16 channels x 2 taps:
// [K-step][3 splits][Cout_pad/16][64 lanes][8]; lane l of cout-tile nt holds W[cout = 16nt + (l&15)][channel = 16cb +
// 8((l>>4)&1) + j][tap], where (cb, tap) of the lane's half (l>>5) follows the kernel's K-step order:
//   plain  (odd number of channel groups, or even tap count): per group ceil(T/2) K-steps (2ks, 2ks+1), zero beyond T;
//   paired (even number of groups, odd T): group 2p: (T-1)/2 K-steps (2ks, 2ks+1); group 2p+1: the same, then one K-step
//          pairing tap T-1 of group 2p (lanes l<32) with tap T-1 of group 2p+1 (lanes l>=32).
std::vector<unsigned short> pack_x6(const float *w, int cout, int cin, int kh, int kw, int cout_pad, int cin_pad)
{
    const int taps = kh * kw, CB = cin_pad / 16, NT = cout_pad / 16;
    const bool paired = (CB % 2 == 0) && (taps % 2 == 1);
    struct Half { int cb, tap; };
    std::vector<std::pair<Half, Half>> steps;
    if (paired) {
        for (int cb = 0; cb < CB; ++cb) {   // the cross-group pair is the FIRST K-step of the odd group
            if (cb & 1) steps.push_back({{cb - 1, taps - 1}, {cb, taps - 1}});
            for (int ks = 0; ks < (taps - 1) / 2; ++ks) steps.push_back({{cb, 2 * ks}, {cb, 2 * ks + 1}});
        }
    } else {
        for (int cb = 0; cb < CB; ++cb)
            for (int ks = 0; ks < (taps + 1) / 2; ++ks) steps.push_back({{cb, 2 * ks}, {cb, 2 * ks + 1}});   // tap >= T -> zeros
    }
    std::vector<unsigned short> out(steps.size() * 3 * NT * 64 * 8, 0);
    for (size_t st = 0; st < steps.size(); ++st)
        for (int nt = 0; nt < NT; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int g = l >> 4;
                    const Half h = (g >> 1) ? steps[st].second : steps[st].first;
                    const int co = nt * 16 + (l & 15), ci = h.cb * 16 + 8 * (g & 1) + j, t = h.tap;
                    float v = 0.f;
                    if (co < cout && ci < cin && t < taps) v = w[((size_t)co * cin + ci) * taps + t];
                    const unsigned short h0 = bf16_rne(v);
                    const float r1 = v - bf16_f32(h0);
                    const unsigned short h1 = bf16_rne(r1);
                    const float r2 = r1 - bf16_f32(h1);
                    const unsigned short h2 = bf16_rne(r2);
                    const unsigned short hs[3] = {h0, h1, h2};
                    for (int sp = 0; sp < 3; ++sp) out[(((st * 3 + sp) * NT + nt) * 64 + l) * 8 + j] = hs[sp];
                }
    return out;
}

// Power-of-two scale for the fp16 split (conv_f16x3.hip): S = 2^k with max|S*w| in [4096, 8192); k in [-100, 24] (negative for a tensor
// that holds weights of 8192 and more: unscaled, those would leave the fp16 range when they are split - round 5, found by the stress gains
// of tests/test_gpu_trained_like.py; k used to be clamped at 0).
int h2_scale_exp(const float *w, size_t n)
{
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) m = std::fmax(m, std::fabs(w[i]));
    if (!(m > 0.f) || !std::isfinite(m)) return 0;
    int e = 0;
    std::frexp(m, &e);            // m = f * 2^e, f in [0.5, 1)
    int k = 13 - e;               // S*m = f * 2^13 in [4096, 8192)
    return k < -100 ? -100 : (k > 24 ? 24 : k);
}

// Same K-step stream as pack_x6, two fp16 terms of S*w per element: [step][2 splits][cout_pad/16][64 lanes][8].
std::vector<unsigned short> pack_h2(const float *w, int cout, int cin, int kh, int kw, int cout_pad, int cin_pad, int scale_exp)
{
    const int taps = kh * kw, CB = cin_pad / 16, NT = cout_pad / 16;
    const bool paired = (CB % 2 == 0) && (taps % 2 == 1);
    const float S = std::ldexp(1.f, scale_exp);
    struct Half { int cb, tap; };
    std::vector<std::pair<Half, Half>> steps;
    if (paired) {
        for (int cb = 0; cb < CB; ++cb) {
            if (cb & 1) steps.push_back({{cb - 1, taps - 1}, {cb, taps - 1}});
            for (int ks = 0; ks < (taps - 1) / 2; ++ks) steps.push_back({{cb, 2 * ks}, {cb, 2 * ks + 1}});
        }
    } else {
        for (int cb = 0; cb < CB; ++cb)
            for (int ks = 0; ks < (taps + 1) / 2; ++ks) steps.push_back({{cb, 2 * ks}, {cb, 2 * ks + 1}});
    }
    std::vector<unsigned short> out(steps.size() * 2 * NT * 64 * 8, 0);
    for (size_t st = 0; st < steps.size(); ++st)
        for (int nt = 0; nt < NT; ++nt) {
            unsigned short *o0 = out.data() + ((st * 2 + 0) * NT + nt) * 512, *o1 = out.data() + ((st * 2 + 1) * NT + nt) * 512;
            for (int l = 0; l < 64; ++l) {
                const int g = l >> 4;
                const Half h = (g >> 1) ? steps[st].second : steps[st].first;
                const int co = nt * 16 + (l & 15), ci0 = h.cb * 16 + 8 * (g & 1), t = h.tap;
                float v[8];
                for (int j = 0; j < 8; ++j) v[j] = (co < cout && ci0 + j < cin && t < taps) ? w[((size_t)co * cin + ci0 + j) * taps + t] * S : 0.f;
                split8(v, o0 + l * 8, o1 + l * 8);
            }
        }
    return out;
}

// Stem weights for the MFMA stem (conv_misc.hip: stem_mfma_kernel): ONE k1 x k1 convolution with 32 outputs, top-left
// anchored (the smaller kernels of the MTT stems are zero-padded into it).  w32: [32][cin][k1][k1].  K is ordered plane by
// plane; a K-step of 32 slots holds RPS whole kernel rows of one plane (k1 = 9: 3 rows = 27 slots; k1 = 5: 5 rows = 25 slots), slot
// k = (dy - r0) * k1 + dx, the rest zero.
// Stream: [K-step][3 splits][2 cout groups][64 lanes][8] fp16 terms of 2^scale_exp * w.  THREE terms (33 significand bits: the fp32 weight
// to 2^-37 of the tensor's maximum - exactly, unless it lies more than 2^13 below that maximum), not the two of the other layers: the pixels are exact in fp16, so with exact weights every product of the first layer is exact
// - and this is the layer whose weight error the Luma_Q net amplifies most (raw 0..255 inputs, outputs in the thousands, logits of order
// 1).  A weight's 2^-23 representation error is the same for every pixel it meets, so it adds up coherently where rounding noise
// averages out: on the worst block of the 15 840-block campaign two-term stem weights alone cost 5e-4 of the 1e-3 tolerance, three
// cost 1e-4 (profiles/r04_campaign_config4.txt).
std::vector<unsigned short> pack_stem_h2(const float *w32, int cin, int k1, int scale_exp)
{
    const int RPS = k1 > 8 ? 3 : 5, KPP = (k1 + RPS - 1) / RPS, KS = cin * KPP;   // kernel rows per K-step, K-steps per plane (stem_mfma_kernel)
    const float S = std::ldexp(1.f, scale_exp);
    std::vector<unsigned short> out((size_t)KS * 3 * 2 * 64 * 8, 0);
    for (int ks = 0; ks < KS; ++ks)
        for (int nt = 0; nt < 2; ++nt)
            for (int l = 0; l < 64; ++l) {
                const int g = l >> 4, co = nt * 16 + (l & 15), ci = ks / KPP, r0 = (ks % KPP) * RPS;
                float v[8];
                for (int j = 0; j < 8; ++j) {
                    const int kk = 8 * g + j, dyl = kk / k1, dx = kk % k1, dy = r0 + dyl;
                    v[j] = (dyl < RPS && dy < k1) ? w32[(((size_t)co * cin + ci) * k1 + dy) * k1 + dx] * S : 0.f;
                }
                unsigned short *o0 = out.data() + ((((size_t)ks * 3 + 0) * 2 + nt) * 64 + l) * 8, *o1 = o0 + 2 * 64 * 8, *o2 = o1 + 2 * 64 * 8;
                split8(v, o0, o1);
                float r[8];      // what the two terms leave over: exact in fp32 (both subtractions are), at most 11 significant bits
                unsigned short dummy[8];
                for (int j = 0; j < 8; ++j) {
                    _Float16 h0, h1;
                    std::memcpy(&h0, o0 + j, 2); std::memcpy(&h1, o1 + j, 2);
                    r[j] = (v[j] - (float)h0) - (float)h1;
                }
                split8(r, o2, dummy);
            }
    return out;
}

// OIHW -> [tap][cin][cout] (direct kernels, stems, heads)
std::vector<float> pack_plain(const float *w, int cout, int cin, int kh, int kw)
{
    const int taps = kh * kw;
    std::vector<float> out((size_t)taps * cin * cout);
    for (int t = 0; t < taps; ++t)
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co) out[((size_t)t * cin + ci) * cout + co] = w[((size_t)co * cin + ci) * taps + t];
    return out;
}

}  // namespace pmp

using namespace pmp;
#define set_err(ctx_unused, code, msg) set_err_global((code), (msg))

extern "C" {

int64_t pmp_debug_pack_f16x3(const float *w, int cout, int cin, int k, uint16_t *out, int64_t cap, int *scale_exp)
{
    if (!w || cout <= 0 || cin <= 0 || (k != 1 && k != 3 && k != 5) || !scale_exp)
        return set_err(nullptr, PMP_E_INVALID, "pmp_debug_pack_f16x3: bad arguments");
    const int kexp = h2_scale_exp(w, (size_t)cout * cin * k * k);
    *scale_exp = kexp;
    const std::vector<unsigned short> v = pack_h2(w, cout, cin, k, k, (cout + 15) & ~15, (cin + 15) & ~15, kexp);
    if (out && (int64_t)v.size() <= cap) memcpy(out, v.data(), v.size() * sizeof(unsigned short));
    return (int64_t)v.size();
}

}  // extern "C"
