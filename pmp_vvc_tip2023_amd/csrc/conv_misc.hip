// conv_misc.hip — the non-MFMA kernels of the four nets: stems straight from the u8 blocks, the tiny tail convs,
// the heads, and the gather-style glue (multi-scale pool concat, attention inputs).  Together < 3 % of the FLOPs.
#include "pmp_kernels.h"
#include "split3.h"

namespace pmp {

__device__ __forceinline__ size_t act_idx(int n, int c, int y, int x, int CB, int H, int W)
{
    return ((((size_t)n * CB + (c >> 4)) * H + y) * W + x) * 16 + (c & 15);
}

// =============================================================================================== stems
// One workgroup = one block.  The zero-padded input planes and all stem weights live in LDS; every thread owns one
// output column and walks it in groups of 4 rows x 8 output channels (32 accumulators), sliding a register window
// down the column so each LDS pixel read feeds up to KH taps.
//
//   luma   Q    : 1 plane 72x72,  conv 9x9 -> 32                      (Model_QBD.py:79-80)
//   luma   MSBD : 2 planes 72x72, conv 9x9 -> 16, 5x9 -> 8, 9x5 -> 8  (Model_QBD.py:130-135)
//   chroma Q    : 3 planes 36x36, conv 5x5 -> 32                      (Model_QBD.py:177-178)
//   chroma MSBD : 4 planes 36x36, conv 5x5 -> 16, 3x5 -> 8, 5x3 -> 8  (Model_QBD.py:228-233)
// Right/bottom zero padding of 4 (2) is common to all convs: the (5,9) conv pads right only but never reads below
// row y+4 <= 67, the (9,5) conv pads bottom only but never reads right of x+4 <= 67.
template <int KH, int KW, int CIN, int PS, int ROWS, int FMT>
__device__ __forceinline__ void stem_conv(const float *__restrict__ planes, const float *__restrict__ w,
                                          const float *__restrict__ bias, int cout, int x, int y0, int OUT,
                                          const ActOut &out, size_t out_n, int ch_off)
{
    // planes: [CIN][PS][PS];  w: [KH*KW][CIN][cout];  computes rows y0..y0+ROWS-1 at column x for all cout.
#pragma unroll 1
    for (int c0 = 0; c0 < cout; c0 += 8) {
        float acc[ROWS][8];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[r][k] = bias[c0 + k];
        for (int ci = 0; ci < CIN; ++ci) {
            const float *pl = planes + ci * PS * PS;
#pragma unroll
            for (int dx = 0; dx < KW; ++dx) {
                float col[ROWS + KH - 1];
#pragma unroll
                for (int r = 0; r < ROWS + KH - 1; ++r) col[r] = pl[(y0 + r) * PS + x + dx];
#pragma unroll
                for (int dy = 0; dy < KH; ++dy) {
                    const float *wp = w + ((dy * KW + dx) * CIN + ci) * cout + c0;
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wp);
                    const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wp + 4);
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) {
                        const float v = col[r + dy];
                        acc[r][0] = fmaf(v, w0.x, acc[r][0]); acc[r][1] = fmaf(v, w0.y, acc[r][1]);
                        acc[r][2] = fmaf(v, w0.z, acc[r][2]); acc[r][3] = fmaf(v, w0.w, acc[r][3]);
                        acc[r][4] = fmaf(v, w1.x, acc[r][4]); acc[r][5] = fmaf(v, w1.y, acc[r][5]);
                        acc[r][6] = fmaf(v, w1.z, acc[r][6]); acc[r][7] = fmaf(v, w1.w, acc[r][7]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int c = ch_off + c0;  // 8 consecutive channels inside one 16-channel group
            const size_t o = out_n + (((size_t)(c >> 4) * OUT + (y0 + r)) * OUT + x) * 16 + (c & 15);
            f32x4 v0 = {fmaxf(acc[r][0], 0.f), fmaxf(acc[r][1], 0.f), fmaxf(acc[r][2], 0.f), fmaxf(acc[r][3], 0.f)};
            f32x4 v1 = {fmaxf(acc[r][4], 0.f), fmaxf(acc[r][5], 0.f), fmaxf(acc[r][6], 0.f), fmaxf(acc[r][7], 0.f)};
            if (FMT != FMT_F32) { out.store4(o, v0); out.store4(o + 4, v1); }
            else { *reinterpret_cast<f32x4 *>(out.f32 + o) = v0; *reinterpret_cast<f32x4 *>(out.f32 + o + 4) = v1; }
        }
    }
}

template <bool LUMA, bool MSBD, int FMT>
__global__ __launch_bounds__(256) void stem_kernel(StemArgs a)
{
    constexpr int S = LUMA ? 68 : 34, P = LUMA ? 4 : 2, PS = S + P, OUT = S - P;  // 72/36 planes, 64/32 outputs
    constexpr int CIN = (LUMA ? 1 : 3) + (MSBD ? 1 : 0);
    constexpr int K1 = LUMA ? 9 : 5, K2 = LUMA ? 5 : 3;
    constexpr int NW = MSBD ? (K1 * K1 * CIN * 16 + 2 * K2 * K1 * CIN * 8) : (K1 * K1 * CIN * 32);
    extern __shared__ float smem[];
    float *planes = smem;                 // [CIN][PS][PS]
    float *wl = smem + CIN * PS * PS;     // NW floats, then 32 biases
    float *bl = wl + NW;
    const int n = blockIdx.x, tid = threadIdx.x;

    for (int i = tid; i < NW; i += 256) wl[i] = a.w[i];
    if (tid < 32) bl[tid] = a.bias[tid];
    const uint8_t *by = a.by + (size_t)n * 68 * 68;
    for (int i = tid; i < PS * PS; i += 256) {
        const int r = i / PS, c = i - r * PS;
        const bool in = r < S && c < S;
        if (LUMA) {
            planes[i] = in ? (float)by[r * 68 + c] : 0.f;
        } else {
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (in) {
                const uint8_t *p = by + (2 * r) * 68 + 2 * c;  // F.max_pool2d(Y, 2), Inference_QBD.py:197
                const int m = max(max((int)p[0], (int)p[1]), max((int)p[68], (int)p[69]));
                v0 = (float)m;
                v1 = (float)a.bu[(size_t)n * 34 * 34 + r * 34 + c];
                v2 = (float)a.bv[(size_t)n * 34 * 34 + r * 34 + c];
            }
            planes[i] = v0; planes[PS * PS + i] = v1; planes[2 * PS * PS + i] = v2;
        }
        if (MSBD) {  // padding_lu(interpolate(q, 8 | 4)), Model_QBD.py:130 / :228
            float qv = 0.f;
            if (in && r >= P && c >= P) {
                constexpr int SC = LUMA ? 8 : 4;
                qv = a.q[(size_t)n * 64 + ((r - P) / SC) * 8 + (c - P) / SC];
            }
            planes[(CIN - 1) * PS * PS + i] = qv;
        }
    }
    __syncthreads();

    const size_t out_n = (size_t)n * 2 * OUT * OUT * 16;
    const ActOut out{a.out, a.out_s3, a.s3_stride, FMT, a.sat};
    constexpr int COLS_PER_WG = 256 / OUT;       // luma: 4 row-bands of 16 rows; chroma: 8 bands of 4 rows
    constexpr int BAND = OUT / COLS_PER_WG;
    const int x = tid % OUT, band = tid / OUT;
    for (int y0 = band * BAND; y0 < (band + 1) * BAND; y0 += 4) {
        if (MSBD) {
            stem_conv<K1, K1, CIN, PS, 4, FMT>(planes, wl, bl, 16, x, y0, OUT, out, out_n, 0);
            stem_conv<K2, K1, CIN, PS, 4, FMT>(planes, wl + K1 * K1 * CIN * 16, bl + 16, 8, x, y0, OUT, out, out_n, 16);
            stem_conv<K1, K2, CIN, PS, 4, FMT>(planes, wl + K1 * K1 * CIN * 16 + K2 * K1 * CIN * 8, bl + 24, 8, x, y0, OUT,
                                          out, out_n, 24);
        } else {
            stem_conv<K1, K1, CIN, PS, 4, FMT>(planes, wl, bl, 32, x, y0, OUT, out, out_n, 0);
        }
    }
}

// ---- MFMA stem (f16x3 datapath).  The same first layers as stem_kernel, as ONE top-left anchored K1 x K1 convolution with 32
// outputs on the fp16 matrix cores: pixels (0..255) are exact in fp16, so they need one term; the plane built from the QT
// logits gets the usual two (x0, x1); weights are THREE scaled fp16 terms for the pixel planes (pack_stem_h2: the fp32 weight exactly,
// so every pixel product is exact - this layer's weight error is the one the Luma_Q net amplifies most) and the usual two for the
// logit plane.  Products: x0*w2 + x0*w1 + x0*w0 (pixels), x0*w1 + x0*w0 + x1*w0 (logits).
// K order: plane by plane, a K-step = RPS whole kernel rows of one plane (9x9: 3 rows = 27 of the 32 slots, 3 K-steps per plane;
// 5x5: all 5 rows = 25 slots, one K-step per plane).  There is no channel dimension to make a lane's 8 K-values contiguous, so the B
// fragments are gathered with 2-byte LDS reads (lane (g, j): slot k = 8g + j -> tap (k / K1, k % K1) of the step's rows); one gathered
// fragment feeds 6 MFMAs (two cout groups x the products).  The fragments of a batch (output rows m = 0..7) form a sliding window over
// input rows: the next K-step of a plane needs RPS new ones.  Workgroup = one block; a wave walks batches of 8 rows x 16 columns (64
// accumulator registers) and streams the weight fragments from L2 once per batch and K-step, one K-step ahead.
// (Until round 4 a K-step was 2 rows x 16 column slots of ALL planes: 5 K-steps for 81 taps, 9 with the logit plane - 40-50 % more MFMAs.)
template <bool LUMA, bool MSBD>
__global__ __launch_bounds__(256, 2) void stem_mfma_kernel(StemArgs a)
{
    constexpr int S = LUMA ? 68 : 34, P = LUMA ? 4 : 2, PS = S + P, OUT = S - P;   // 72/36 planes, 64/32 outputs
    constexpr int CIN = (LUMA ? 1 : 3) + (MSBD ? 1 : 0);
    constexpr int K1 = LUMA ? 9 : 5;
    constexpr int RPS = LUMA ? 3 : 5, KPP = (K1 + RPS - 1) / RPS, KS = CIN * KPP;     // kernel rows per K-step, K-steps per plane
    static_assert(RPS * K1 <= 32 && KPP * RPS == K1, "a K-step holds whole kernel rows, a plane whole K-steps");
    constexpr int RS = PS + 16;                          // row stride of the planes in LDS
    constexpr int SEGS = OUT / 16, RB = 8, NBATCH = SEGS * (OUT / RB);
    constexpr int NPIX = CIN - (MSBD ? 1 : 0);           // pixel planes (one fp16 term: exact)
    __shared__ _Float16 h0[NPIX * PS * RS + 16];
    __shared__ unsigned hq[MSBD ? PS * RS + 16 : 4];     // the logit plane: both fp16 terms of a value in one word (low | high << 16), so that
                                                         // ONE 4-byte LDS read per K slot fetches what two 2-byte gathers did
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, xl = lane & 15, g = lane >> 4;

    const uint8_t *by = a.by + (size_t)n * 68 * 68;
    for (int i = tid; i < PS * RS; i += 256) {
        const int r = i / RS, c = i - r * RS;
        const bool in = r < S && c < S;
        if (LUMA) {
            h0[i] = in ? (_Float16)(float)by[r * 68 + c] : (_Float16)0.f;
        } else {
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (in) {
                const uint8_t *p = by + (2 * r) * 68 + 2 * c;  // F.max_pool2d(Y, 2), Inference_QBD.py:197
                v0 = (float)max(max((int)p[0], (int)p[1]), max((int)p[68], (int)p[69]));
                v1 = (float)a.bu[(size_t)n * 34 * 34 + r * 34 + c];
                v2 = (float)a.bv[(size_t)n * 34 * 34 + r * 34 + c];
            }
            h0[i] = (_Float16)v0; h0[PS * RS + i] = (_Float16)v1; h0[2 * PS * RS + i] = (_Float16)v2;
        }
        if (MSBD) {  // padding_lu(interpolate(q, 8 | 4)), Model_QBD.py:130 / :228
            float qv = 0.f;
            if (in && r >= P && c >= P) {
                constexpr int SC = LUMA ? 8 : 4;
                qv = a.q[(size_t)n * 64 + ((r - P) / SC) * 8 + (c - P) / SC];
            }
            _Float16 q0, q1;
            sat_report(a.sat, fabsf(qv));   // checked where it is stored, no register carried through the kernel
            split2(qv, q0, q1);
            unsigned short b0, b1;
            __builtin_memcpy(&b0, &q0, 2); __builtin_memcpy(&b1, &q1, 2);
            hq[i] = (unsigned)b0 | ((unsigned)b1 << 16);
        }
    }
    __syncthreads();

    const f16x8 *wl = reinterpret_cast<const f16x8 *>(a.wh) + lane;
    const float inv_scale = a.out_scale;
    const size_t out_n = (size_t)n * 2 * OUT * OUT * 16;
    int eo[8];      // element offsets of this lane's 8 K slots inside a fragment: tap (dy, dx) of the K-step's rows (slots past RPS * K1: zero weights, any valid address)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kk = 8 * g + j;
        eo[j] = min(kk / K1, RPS - 1) * RS + kk % K1;
    }
    auto gather = [&](const _Float16 *p) __attribute__((always_inline)) {
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[eo[j]];
        return v;
    };
    auto gather2 = [&](const unsigned *p, f16x8 &x0v, f16x8 &x1v) __attribute__((always_inline)) {
        unsigned v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[eo[j]];
        u32x4 lo, hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = __builtin_amdgcn_perm(v[2 * k + 1], v[2 * k], 0x05040100u);     // the low halves of two words
            hi[k] = __builtin_amdgcn_perm(v[2 * k + 1], v[2 * k], 0x07060302u);     // the high halves
        }
        __builtin_memcpy(&x0v, &lo, 16);
        __builtin_memcpy(&x1v, &hi, 16);
    };
    for (int b = wave; b < NBATCH; b += 4) {
        const int seg = b % SEGS, y0 = (b / SEGS) * RB, x0 = seg * 16;
        f32x4 acc[RB][2];
        f16x8 win[RB], win1[MSBD ? RB : 1];
#pragma unroll
        for (int m = 0; m < RB; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        // weight fragments [w0 | w1 | w2][nt] of a K-step, requested one K-step ahead; the fences keep hipcc from hoisting the requests of
        // all KS steps to the top of the batch (KS x 24 registers)
        f16x8 wq[2][6];
        auto wload = [&](int ks, bool third) __attribute__((always_inline)) {
            const f16x8 *wk = wl + (size_t)ks * (3 * 2 * 64);
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (i < 4 || third) wq[ks & 1][i] = wk[i * 64];
        };
        wload(0, true);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            constexpr int dummy = 0; (void)dummy;
            const int plane = ks / KPP, s = ks % KPP, r0 = s * RPS;
            const bool isq = MSBD && plane == CIN - 1;
            const _Float16 *base = h0 + ((isq ? 0 : plane) * PS + y0) * RS + x0 + xl;
            const unsigned *baseq = hq + y0 * RS + x0 + xl;
            if (s == 0) {      // a new plane: the whole window
#pragma unroll
                for (int m = 0; m < RB; ++m) {
                    if (isq) gather2(baseq + m * RS, win[m], win1[m]);
                    else win[m] = gather(base + m * RS);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KS) wload(ks + 1, !(MSBD && (ks + 1) / KPP == CIN - 1));
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 w00 = wq[ks & 1][0], w01 = wq[ks & 1][1], w10 = wq[ks & 1][2], w11 = wq[ks & 1][3];
#pragma unroll
            for (int m = 0; m < RB; ++m) {
                const f16x8 bx = win[(m + r0) & (RB - 1)];
                if (!isq) {
                    const f16x8 w20 = wq[ks & 1][4], w21 = wq[ks & 1][5];
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w20, bx, acc[m][0], 0, 0, 0);     // smallest terms first
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w21, bx, acc[m][1], 0, 0, 0);
                }
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w10, bx, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w11, bx, acc[m][1], 0, 0, 0);
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w00, bx, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01, bx, acc[m][1], 0, 0, 0);
                if (isq) {
                    const f16x8 b1 = win1[(m + r0) & (RB - 1)];
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w00, b1, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w01, b1, acc[m][1], 0, 0, 0);
                }
                if (s + 1 < KPP && m < RPS) {   // slide the window: input row RB + r0 + m takes the slot this row just left
                    const int p = RB + r0 + m;
                    if (isq) gather2(baseq + p * RS, win[p & (RB - 1)], win1[p & (RB - 1)]);
                    else win[p & (RB - 1)] = gather(base + p * RS);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(a.bias + nt * 16 + g * 4);
            // 16-byte stores, as the convolution epilogues (conv_f16x3.hip): one v_permlane16_swap per register turns {rows m, m+1} x {couts 4g..}
            // into the 8 consecutive channels 8(g>>1).. of row m + (g&1)
#pragma unroll
            for (int m = 0; m < RB; m += 2) {
                f32x4 v0 = acc[m][nt] * inv_scale + bias, v1 = acc[m + 1][nt] * inv_scale + bias;
                v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                sat_report(a.sat, sat_amax4(sat_amax4(0.f, v0), v1));
                u32x4 p, q;
                split2_rows(v0, v1, p, q);
                rows16_swap(p);
                rows16_swap(q);
                const size_t o = out_n + (((size_t)nt * OUT + (y0 + m + (g & 1))) * OUT + x0 + xl) * 16 + 8 * (g >> 1);
                // non-temporal stores: stems -6.5 % (0.384 -> 0.359 ms per 1024 blocks), same-box A/B
                __builtin_nontemporal_store(p, reinterpret_cast<u32x4 *>(a.out_s3 + o));
                __builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(a.out_s3 + o + a.s3_stride));
            }
        }
    }
}

template <bool LUMA, bool MSBD>
static hipError_t launch_stem_t(hipStream_t s, const StemArgs &a)
{
    if (a.out_s3 && a.fmt == FMT_H2 && a.wh) {
        hipLaunchKernelGGL((stem_mfma_kernel<LUMA, MSBD>), dim3(a.N), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    constexpr int S = LUMA ? 68 : 34, P = LUMA ? 4 : 2, PS = S + P;
    constexpr int CIN = (LUMA ? 1 : 3) + (MSBD ? 1 : 0);
    constexpr int K1 = LUMA ? 9 : 5, K2 = LUMA ? 5 : 3;
    constexpr int NW = MSBD ? (K1 * K1 * CIN * 16 + 2 * K2 * K1 * CIN * 8) : (K1 * K1 * CIN * 32);
    const size_t smem = (size_t)(CIN * PS * PS + NW + 32) * sizeof(float);
    if (a.out_s3 && a.fmt == FMT_H2) hipLaunchKernelGGL((stem_kernel<LUMA, MSBD, FMT_H2>), dim3(a.N), dim3(256), smem, s, a);
    else if (a.out_s3) hipLaunchKernelGGL((stem_kernel<LUMA, MSBD, FMT_B3>), dim3(a.N), dim3(256), smem, s, a);
    else hipLaunchKernelGGL((stem_kernel<LUMA, MSBD, FMT_F32>), dim3(a.N), dim3(256), smem, s, a);
    return hipGetLastError();
}

hipError_t launch_stem(hipStream_t s, bool luma, bool msbd, const StemArgs &a)
{
    if (luma) return msbd ? launch_stem_t<true, true>(s, a) : launch_stem_t<true, false>(s, a);
    return msbd ? launch_stem_t<false, true>(s, a) : launch_stem_t<false, false>(s, a);
}

// =============================================================================================== small direct conv
// One thread per (n, y, x, cout).  Only used where the whole layer is a few hundred MAC per output on 8x8 maps.
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvDirectArgs a)
{
    const size_t total = (size_t)a.N * a.H * a.W * a.CoutPad;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % a.CoutPad);
    size_t r = i / a.CoutPad;
    const int x = (int)(r % a.W); r /= a.W;
    const int y = (int)(r % a.H);
    const int n = (int)(r / a.H);
    const int CBo = a.CoutPad >> 4;
    float acc = 0.f;
    if (co < a.Cout) {
        if (a.bias) acc = a.bias[co];
        const int py = a.KH / 2, px = a.KW / 2, CBi = a.CinPad >> 4;
        for (int dy = 0; dy < a.KH; ++dy) {
            const int yy = y + dy - py;
            if (yy < 0 || yy >= a.H) continue;
            for (int dx = 0; dx < a.KW; ++dx) {
                const int xx = x + dx - px;
                if (xx < 0 || xx >= a.W) continue;
                const float *wp = a.w + (size_t)((dy * a.KW + dx) * a.Cin) * a.Cout + co;
                for (int ci = 0; ci < a.Cin; ++ci)
                    acc = fmaf(a.x[act_idx(n, ci, yy, xx, CBi, a.H, a.W)], wp[(size_t)ci * a.Cout], acc);
            }
        }
        if (a.x_sc) {
            const int CBs = a.CscPad >> 4;
            for (int ci = 0; ci < a.Csc; ++ci)
                acc = fmaf(a.x_sc[act_idx(n, ci, y, x, CBs, a.H, a.W)], a.w_sc[(size_t)ci * a.Cout + co], acc);
        }
        if (a.res) acc += a.res[act_idx(n, co, y, x, CBo, a.H, a.W)];
        if (a.relu) acc = fmaxf(acc, 0.f);
    }
    a.out[act_idx(n, co, y, x, CBo, a.H, a.W)] = acc;  // padded channels are written as zeros
}

// Fast path of the same operation for Cout <= 8 (the 8x8 tail of the QT nets): one thread per pixel computes all couts,
// weights sit in LDS as [tap][cin][8], activations come as 16-channel float4 quads.  Same accumulation order as the
// generic kernel (bias, taps row-major, channels ascending, shortcut, residual), so the result is bit-identical.
__global__ __launch_bounds__(256) void conv_direct8_kernel(ConvDirectArgs a)
{
    extern __shared__ float wl[];   // [KH*KW*Cin][8], then [Csc][8]
    const int taps = a.KH * a.KW, nw = taps * a.Cin;
    for (int i = threadIdx.x; i < (nw + (a.x_sc ? a.Csc : 0)) * 8; i += 256) {
        const int row = i >> 3, co = i & 7;
        float v = 0.f;
        if (co < a.Cout) v = row < nw ? a.w[(size_t)row * a.Cout + co] : a.w_sc[(size_t)(row - nw) * a.Cout + co];
        wl[i] = v;
    }
    __syncthreads();
    const size_t total = (size_t)a.N * a.H * a.W;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % a.W), y = (int)((i / a.W) % a.H), n = (int)(i / ((size_t)a.W * a.H));
    const int CBi = a.CinPad >> 4, CBo = a.CoutPad >> 4, py = a.KH / 2, px = a.KW / 2;
    float acc[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) acc[co] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    auto mac16 = [&](const float *xp, const float *wp, int nch) {   // up to 16 consecutive channels of one pixel
        const f32x4 q0 = *reinterpret_cast<const f32x4 *>(xp), q1 = *reinterpret_cast<const f32x4 *>(xp + 4),
                    q2 = *reinterpret_cast<const f32x4 *>(xp + 8), q3 = *reinterpret_cast<const f32x4 *>(xp + 12);
        const float v[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < nch) {
                const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wp + c * 8), w1 = *reinterpret_cast<const f32x4 *>(wp + c * 8 + 4);
                acc[0] = fmaf(v[c], w0.x, acc[0]); acc[1] = fmaf(v[c], w0.y, acc[1]); acc[2] = fmaf(v[c], w0.z, acc[2]); acc[3] = fmaf(v[c], w0.w, acc[3]);
                acc[4] = fmaf(v[c], w1.x, acc[4]); acc[5] = fmaf(v[c], w1.y, acc[5]); acc[6] = fmaf(v[c], w1.z, acc[6]); acc[7] = fmaf(v[c], w1.w, acc[7]);
            }
        }
    };
    for (int dy = 0; dy < a.KH; ++dy) {
        const int yy = y + dy - py;
        if (yy < 0 || yy >= a.H) continue;
        for (int dx = 0; dx < a.KW; ++dx) {
            const int xx = x + dx - px;
            if (xx < 0 || xx >= a.W) continue;
            for (int c0 = 0; c0 < a.Cin; c0 += 16)
                mac16(a.x + act_idx(n, c0, yy, xx, CBi, a.H, a.W), wl + ((dy * a.KW + dx) * a.Cin + c0) * 8, min(16, a.Cin - c0));
        }
    }
    if (a.x_sc) {
        const int CBs = a.CscPad >> 4;
        for (int c0 = 0; c0 < a.Csc; c0 += 16)
            mac16(a.x_sc + act_idx(n, c0, y, x, CBs, a.H, a.W), wl + (nw + c0) * 8, min(16, a.Csc - c0));
    }
    float o[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) o[co] = 0.f;
#pragma unroll
    for (int co = 0; co < 8; ++co) {
        float v = acc[co];
        if (co < a.Cout) {
            if (a.res) v += a.res[act_idx(n, co, y, x, CBo, a.H, a.W)];
            if (a.relu) v = fmaxf(v, 0.f);
            o[co] = v;
        }
    }
    float *op = a.out + act_idx(n, 0, y, x, CBo, a.H, a.W);
    *reinterpret_cast<f32x4 *>(op) = (f32x4){o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4 *>(op + 4) = (f32x4){o[4], o[5], o[6], o[7]};
    *reinterpret_cast<f32x4 *>(op + 8) = (f32x4){0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4 *>(op + 12) = (f32x4){0.f, 0.f, 0.f, 0.f};
}

hipError_t launch_conv_direct(hipStream_t s, const ConvDirectArgs &a)
{
    if (a.Cout <= 8 && a.CoutPad == 16 && (a.CinPad & 15) == 0 && (!a.x_sc || (a.CscPad & 15) == 0)) {
        const size_t px = (size_t)a.N * a.H * a.W;
        const size_t smem = (size_t)(a.KH * a.KW * a.Cin + (a.x_sc ? a.Csc : 0)) * 8 * sizeof(float);
        if (smem <= 48 * 1024) {
            hipLaunchKernelGGL(conv_direct8_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), smem, s, a);
            return hipGetLastError();
        }
    }
    const size_t total = (size_t)a.N * a.H * a.W * a.CoutPad;
    hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =============================================================================================== heads
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a)
{
    const int S = a.S, cout = a.layer < 0 ? 1 : 2;
    const size_t total = (size_t)a.N * S * S;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % S), y = (int)((i / S) % S), n = (int)(i / ((size_t)S * S));
    float acc0 = a.bias[0], acc1 = cout > 1 ? a.bias[1] : 0.f;
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        if (yy < 0 || yy >= S) continue;
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = x + dx - 1;
            if (xx < 0 || xx >= S) continue;
            const float *xp = a.x + (((size_t)n * S + yy) * S + xx) * 16;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(xp), v1 = *reinterpret_cast<const f32x4 *>(xp + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const float *wp = a.w + (dy * 3 + dx) * 8 * cout;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                acc0 = fmaf(v[ci], wp[ci * cout], acc0);
                if (cout > 1) acc1 = fmaf(v[ci], wp[ci * cout + 1], acc1);
            }
        }
    }
    if (a.layer < 0) {
        a.qt[(size_t)n * 64 + y * 8 + x] = acc0;
    } else {
        const size_t o = ((size_t)n * 3 + a.layer) * 256 + y * 16 + x;
        if (a.layer > 0) acc0 += a.bt[o - 256];  // out_k[:,0] += out_{k-1}[:,0]  (Model_QBD.py:146, :153)
        a.bt[o] = acc0;
        a.dire[o] = acc1;
    }
}

hipError_t launch_head(hipStream_t s, const HeadArgs &a)
{
    const size_t total = (size_t)a.N * a.S * a.S;
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =============================================================================================== glue
// cat[x5, up2(mp2(x5)), up4(mp4(x5)), up8(mp8(x5))]: one workgroup per (block, 16-channel group of x5).
__global__ __launch_bounds__(256) void multipool_concat_kernel(const float *__restrict__ x5, ActOut x6)
{
    __shared__ __attribute__((aligned(16))) float t[16 * 16 * 16];
    __shared__ __attribute__((aligned(16))) float p2[8 * 8 * 16], p4[4 * 4 * 16], p8[2 * 2 * 16];
    const int n = blockIdx.x >> 1, cb = blockIdx.x & 1, tid = threadIdx.x;
    const float *src = x5 + ((size_t)n * 2 + cb) * 4096;
    for (int i = tid; i < 4096; i += 256) t[i] = src[i];
    __syncthreads();
    for (int i = tid; i < 8 * 8 * 16; i += 256) {
        const int c = i & 15, x = (i >> 4) & 7, y = i >> 7;
        const float *q = t + ((2 * y) * 16 + 2 * x) * 16 + c;
        p2[i] = fmaxf(fmaxf(q[0], q[16]), fmaxf(q[256], q[272]));
    }
    __syncthreads();
    if (tid < 4 * 4 * 16) {
        const int c = tid & 15, x = (tid >> 4) & 3, y = tid >> 6;
        const float *q = p2 + ((2 * y) * 8 + 2 * x) * 16 + c;
        p4[tid] = fmaxf(fmaxf(q[0], q[16]), fmaxf(q[128], q[144]));
    }
    __syncthreads();
    if (tid < 2 * 2 * 16) {
        const int c = tid & 15, x = (tid >> 4) & 1, y = tid >> 5;
        const float *q = p4 + ((2 * y) * 4 + 2 * x) * 16 + c;
        p8[tid] = fmaxf(fmaxf(q[0], q[16]), fmaxf(q[64], q[80]));
    }
    __syncthreads();
    const size_t dst = (size_t)n * 8 * 4096;  // channel groups: x5 -> 0,1; mp2 -> 2,3; mp4 -> 4,5; mp8 -> 6,7
    for (int i = tid * 4; i < 4096; i += 1024) {   // 4 consecutive channels per thread and step
        const int c = i & 15, x = (i >> 4) & 15, y = i >> 8;
        x6.store4(dst + (size_t)(0 + cb) * 4096 + i, *reinterpret_cast<const f32x4 *>(t + i));
        x6.store4(dst + (size_t)(2 + cb) * 4096 + i, *reinterpret_cast<const f32x4 *>(p2 + ((y >> 1) * 8 + (x >> 1)) * 16 + c));
        x6.store4(dst + (size_t)(4 + cb) * 4096 + i, *reinterpret_cast<const f32x4 *>(p4 + ((y >> 2) * 4 + (x >> 2)) * 16 + c));
        x6.store4(dst + (size_t)(6 + cb) * 4096 + i, *reinterpret_cast<const f32x4 *>(p8 + ((y >> 3) * 2 + (x >> 3)) * 16 + c));
    }
}

// ---- calibration of the f16x3 activation scales (calibrate.cpp: calibrate_mtt): largest |value| of an fp32 tensor, folded into *slot as
// the magnitude's bit pattern (ordered like the floats; a NaN ranks above everything, split3.h).  Runs on a few dozen blocks once per
// net, never on the inference path.
__global__ __launch_bounds__(256) void amax_f32_kernel(const float *__restrict__ x, size_t n4, unsigned *slot)
{
    __shared__ unsigned part[4];
    float amax = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        amax = sat_amax4(amax, *reinterpret_cast<const f32x4 *>(x + i * 4));
    unsigned b = sat_bits(amax);
    for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(slot, max(max(part[0], part[1]), max(part[2], part[3])));
}

hipError_t launch_amax_f32(hipStream_t s, const float *x, size_t n, unsigned *slot)
{
    const size_t n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 > 1024 ? 1024 : (n4 + 255) / 256);
    if (n4) hipLaunchKernelGGL(amax_f32_kernel, dim3(grid), dim3(256), 0, s, x, n4, slot);
    return hipGetLastError();
}

hipError_t launch_multipool_concat(hipStream_t s, const float *x5, float *x6, int N, unsigned short *x6_s3, size_t s3_stride, int fmt, unsigned *sat)
{
    hipLaunchKernelGGL(multipool_concat_kernel, dim3(N * 2), dim3(256), 0, s, x5, ActOut{x6, x6_s3, s3_stride, fmt, sat});
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void att_input_kernel(const float *__restrict__ q, const float *__restrict__ bt,
                                                        const float *__restrict__ dire, int layer, ActOut out, int N,
                                                        int S, float scale)
{
    const size_t total = (size_t)N * S * S;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % S), y = (int)((i / S) % S), n = (int)(i / ((size_t)S * S));
    const int sq = S / 8, sh = S / 16;
    const size_t o = ((size_t)n * 3 + layer) * 256 + (y / sh) * 16 + (x / sh);
    f32x4 v = {q[(size_t)n * 64 + (y / sq) * 8 + (x / sq)], bt[o], dire[o], 0.f};
    v *= scale;        // f16x3 activation scale of the attention segment: a power of two (exact), 1 elsewhere
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    out.store4(i * 16, v); out.store4(i * 16 + 4, z); out.store4(i * 16 + 8, z); out.store4(i * 16 + 12, z);
}

hipError_t launch_att_input(hipStream_t s, const float *q, const float *bt, const float *dire, int layer, float *out,
                            int N, int S, unsigned short *out_s3, size_t s3_stride, int fmt, unsigned *sat, float scale)
{
    const size_t total = (size_t)N * S * S;
    hipLaunchKernelGGL(att_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, q, bt, dire, layer,
                       ActOut{out, out_s3, s3_stride, fmt, sat}, N, S, scale);
    return hipGetLastError();
}

}  // namespace pmp
