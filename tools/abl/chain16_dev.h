// chain16_dev.h — ROUND 4 form of the LDS-resident layer chains (six halo images, eight waves, one workgroup per CU; the product uses csrc/tail16_dev.h since round 6), kept for the fused-ResidualBlock prototype rbfuse_proto.hip: halo-image geometry, the K-step list of
// pack_h2, the unrolled convolution pass, the epilogue, a ResidualBlock on LDS images.  Header so that the measurement library's
// fused-ResidualBlock prototype (abl/rbfuse_proto.hip) builds on the same code.
#pragma once
#include "pmp_kernels.h"
#include "split3.h"

namespace pmp {

namespace {

constexpr int C16_PLN = 18 * 18 * 32;        // bytes per fp16 plane of one 16-channel group
constexpr int C16_SLOT = 2 * C16_PLN;        // one group, both planes
constexpr int C16_ROW = 18 * 32;
constexpr int C16_CENTER = C16_ROW + 32;     // pixel (0, 0) of the image inside its halo
constexpr int C16_NSLOT = 6;
#define C16_GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ constexpr int c16_tapoff(int t) { return ((t / 3) * 18 + t % 3) * 32; }

// K-step list of a pass = pack_h2's (pack.cpp): with an even group count and an odd tap count the last tap of an even group is paired
// with the last tap of the odd group that follows; otherwise the last pair of a group is zero-padded on the weight side (the pixels of
// its first tap are read twice, as conv_f16x3.hip does).  Byte offset (group + tap) of K-half `half` of step `st`:
template <int T, int CB>
__device__ __forceinline__ constexpr int c16_step_off(int st, int half)
{
    int cb = 0, tap = 0;
    if (!(CB & 1) && (T & 1)) {
        const int h = (T - 1) / 2, pr = st / T, j = st % T;
        if (j < h) { cb = 2 * pr; tap = 2 * j + half; }
        else if (j == h) { cb = 2 * pr + half; tap = T - 1; }
        else { cb = 2 * pr + 1; tap = 2 * (j - h - 1) + half; }
    } else {
        const int per = (T + 1) / 2, ks = st % per;
        cb = st / per;
        tap = 2 * ks + half < T ? 2 * ks + half : 2 * ks;
    }
    return cb * C16_SLOT + (T == 1 ? C16_CENTER : c16_tapoff(tap));
}

// Wave tile of a layer with NT output channel groups: RW image rows x ONE group.  Every wave then requests 2 weight fragments per K-step
// for 3 RW MFMAs (a wave tile of 2 rows x all groups, the first form of this file, requested 2 NT for 6 NT: the L1 -> register path, 64 B
// per clock and CU, needed 512 cycles per K-step for 384 cycles of MFMAs).  NT = 4: 8 waves = 2 row halves x 4 groups; NT = 2: 4 row quarters
// x 2 groups; NT = 1: waves 0..3 take four rows each, waves 4..7 only keep the barriers company (one wave per SIMD fills its matrix pipe
// for these 48-MFMA K-steps as well as two would).
template <int NT>
struct C16Tile {
    static constexpr int RW = NT == 4 ? 8 : 4;
    static constexpr int NWAVE = (16 / RW) * NT;
    __device__ __forceinline__ static bool active() { return (int)(threadIdx.x >> 6) < NWAVE; }
    __device__ __forceinline__ static int ct() { return (int)(threadIdx.x >> 6) % NT; }
    __device__ __forceinline__ static int row0() { return ((int)(threadIdx.x >> 6) / NT) * RW; }
};

// The weight ring of one convolution pass: the fragments of the next D K-steps.  c16_wstart requests the first D - the caller does that
// as early as the registers allow, in front of the epilogue and the barrier that precede the pass (weights do not depend on activations):
// a pass that starts with its own requests waits a full L2 round trip before its first MFMA, 24 times per block.
template <int T, int NT, int CB>
struct C16Pass {
    static constexpr bool paired = !(CB & 1) && (T & 1);
    static constexpr int NS = paired ? (CB / 2) * T : CB * ((T + 1) / 2);
    static constexpr int D = NS < 6 ? NS : 6;           // K-steps of lead (8 registers each)
    f16x8 wq[D][2];
};

template <int T, int NT, int CB>
__device__ __forceinline__ void c16_wload(C16Pass<T, NT, CB> &p, const unsigned short *wpk, int st)
{
    // (explicitly global: a pointer that reached this point through a struct is generic to hipcc, and a flat load counts on both wait counters)
    const C16_GLOBAL f16x8 *wl = (const C16_GLOBAL f16x8 *)wpk + (threadIdx.x & 63) + C16Tile<NT>::ct() * 64;
    p.wq[st % C16Pass<T, NT, CB>::D][0] = wl[(size_t)st * (2 * NT * 64)];
    p.wq[st % C16Pass<T, NT, CB>::D][1] = wl[(size_t)st * (2 * NT * 64) + NT * 64];
}

template <int T, int NT, int CB>
__device__ __forceinline__ void c16_wstart(C16Pass<T, NT, CB> &p, const unsigned short *wpk)
{
    if (!C16Tile<NT>::active()) return;
#pragma unroll
    for (int st = 0; st < C16Pass<T, NT, CB>::D; ++st) c16_wload(p, wpk, st);
}

// One convolution pass (T = 9: 3x3, T = 1: 1x1 on the same halo images) over the CB source groups at `src` into the wave's accumulators;
// its first weight fragments are on their way (c16_wstart).  Fully unrolled.  A K-step runs in sub-steps of four rows; the weight
// fragments (L2 / L1 hits) are requested D K-steps ahead into the ring, the pixel fragments of the next sub-step are read from LDS during
// this one's MFMAs.  Per accumulator the order is the launch path's: x0*w1, x0*w0, x1*w0, K-step after K-step.
template <int T, int NT, int CB>
__device__ __forceinline__ void c16_accumulate(const char *src, const unsigned short *wpk, f32x4 (&acc)[C16Tile<NT>::RW], C16Pass<T, NT, CB> &p)
{
    typedef C16Tile<NT> WT;
    constexpr int NS = C16Pass<T, NT, CB>::NS, D = C16Pass<T, NT, CB>::D;
    constexpr int SUB = WT::RW / 4, NTK = NS * SUB;
    if (!WT::active()) return;
    const int lane = threadIdx.x & 63, xl = lane & 15, g = lane >> 4;
    const char *pbase = src + (WT::row0() * 18 + xl) * 32 + (g & 1) * 16;
    const bool hi = (g >> 1) != 0;
    f16x8 (&wq)[D][2] = p.wq;
    f16x8 xq[2][2][4];
    auto wload = [&](int st) __attribute__((always_inline)) { c16_wload(p, wpk, st); };
    auto xload = [&](int tk) __attribute__((always_inline)) {
        const int st = tk / SUB, h = tk % SUB;
        const char *p = pbase + (hi ? c16_step_off<T, CB>(st, 1) : c16_step_off<T, CB>(st, 0)) + h * 4 * C16_ROW;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            xq[tk & 1][0][m] = *reinterpret_cast<const f16x8 *>(p + m * C16_ROW);
            xq[tk & 1][1][m] = *reinterpret_cast<const f16x8 *>(p + C16_PLN + m * C16_ROW);
        }
    };
    xload(0);
#pragma unroll
    for (int tk = 0; tk < NTK; ++tk) {
        const int st = tk / SUB, h = tk % SUB;
        // fences: hipcc's scheduler otherwise sinks every request to just before its first use (fewer live registers, no lead at all)
        __builtin_amdgcn_sched_barrier(0);
        if (tk + 1 < NTK) xload(tk + 1);
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 w0 = wq[st % D][0], w1 = wq[st % D][1];
        f16x8 (&x0)[4] = xq[tk & 1][0], (&x1)[4] = xq[tk & 1][1];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[h * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x0[m], acc[h * 4 + m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[h * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x0[m], acc[h * 4 + m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[h * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x1[m], acc[h * 4 + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (h == SUB - 1 && st + D < NS) wload(st + D);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NT>
__device__ __forceinline__ void c16_zero(f32x4 (&acc)[C16Tile<NT>::RW])
{
#pragma unroll
    for (int m = 0; m < C16Tile<NT>::RW; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// The epilogue of conv_f16x3.hip's h2_epilogue on a wave's rows: x 1/S (+ identity residual), ReLU, x gate, then ONE of: two-term
// split into a halo image, plain fp32 [group][256 px][16], or 2x2 max-pool to fp32 [group][64 px][16].  Returns the running |max| of
// what the split clamps (the range flag).
enum { C16_IMG = 0, C16_F32 = 1, C16_POOL = 2 };
struct C16Epi {
    float inv_scale;
    const char *res;                    // RES: LDS halo image of the identity residual (its group 0)
    const unsigned short *gate;         // GATE: global split-2 tensor of this block [NT][256 px][16]
    size_t gate_stride;
    char *dst_img;                      // C16_IMG: LDS halo image (group 0 of the output)
    float *dst_f32;                     // C16_F32 / C16_POOL: LDS fp32 output
};

template <int NT, bool RES, bool GATE, int OUT>
__device__ __forceinline__ float c16_epilogue(f32x4 (&acc)[C16Tile<NT>::RW], const C16Epi &e, float amax)
{
    typedef C16Tile<NT> WT;
    if (!WT::active()) return amax;
    const int lane = threadIdx.x & 63, xl = lane & 15, g = lane >> 4, nt = WT::ct(), row0 = WT::row0();
#pragma unroll
    for (int m = 0; m < WT::RW; ++m) {
        const int row = row0 + m;
        f32x4 v = acc[m];
        if (RES) {
            const char *rp = e.res + nt * C16_SLOT + ((row + 1) * 18 + xl + 1) * 32 + g * 8;
            const u32x2_t a = *reinterpret_cast<const u32x2_t *>(rp), b = *reinterpret_cast<const u32x2_t *>(rp + C16_PLN);
            v = v * e.inv_scale + (f32x4){h2_sum_lo(a.x, b.x), h2_sum_hi(a.x, b.x), h2_sum_lo(a.y, b.y), h2_sum_hi(a.y, b.y)};
        } else {
            v = v * e.inv_scale;
        }
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (GATE) {
            const C16_GLOBAL unsigned short *gp = (const C16_GLOBAL unsigned short *)e.gate + (size_t)((nt * 256 + row * 16 + xl) * 16 + g * 4);
            const u32x2_t ga = *reinterpret_cast<const C16_GLOBAL u32x2_t *>(gp), gb = *reinterpret_cast<const C16_GLOBAL u32x2_t *>(gp + e.gate_stride);
            v *= (f32x4){h2_sum_lo(ga.x, gb.x), h2_sum_hi(ga.x, gb.x), h2_sum_lo(ga.y, gb.y), h2_sum_hi(ga.y, gb.y)};     // load_split2_4
        }
        amax = sat_amax4(amax, v);
        acc[m] = v;
    }
#pragma unroll
    for (int m = 0; m < WT::RW; m += 2) {
        if (OUT == C16_POOL) {
            f32x4 v = acc[m], u = acc[m + 1];
            v.x = fmaxf(v.x, u.x); v.y = fmaxf(v.y, u.y); v.z = fmaxf(v.z, u.z); v.w = fmaxf(v.w, u.w);
            f32x4 o;
            o.x = __shfl_xor(v.x, 1); o.y = __shfl_xor(v.y, 1); o.z = __shfl_xor(v.z, 1); o.w = __shfl_xor(v.w, 1);
            v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
            if ((xl & 1) == 0) *reinterpret_cast<f32x4 *>(e.dst_f32 + ((nt * 64 + ((row0 + m) >> 1) * 8 + (xl >> 1)) * 16 + g * 4)) = v;
        } else if (OUT == C16_F32) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                *reinterpret_cast<f32x4 *>(e.dst_f32 + ((nt * 256 + (row0 + m + k) * 16 + xl) * 16 + g * 4)) = acc[m + k];
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned p0, q0, p1, q1;
                h2_split_pair(acc[m + k].x, acc[m + k].y, p0, q0);
                h2_split_pair(acc[m + k].z, acc[m + k].w, p1, q1);
                char *dp = e.dst_img + nt * C16_SLOT + ((row0 + m + k + 1) * 18 + xl + 1) * 32 + g * 8;
                *reinterpret_cast<u32x2_t *>(dp) = (u32x2_t){p0, p1};
                *reinterpret_cast<u32x2_t *>(dp + C16_PLN) = (u32x2_t){q0, q1};
            }
        }
    }
    return amax;
}

struct C16RB {                    // a ResidualBlock's f16x3 streams (RBWeights): first conv, second conv, 1x1 shortcut, and 1/S of each pass
    const unsigned short *w0, *w2, *wsc;
    float s0, s2;
};

// ResidualBlock on LDS images: t = relu(conv3x3(in)) -> `mid`; out = relu(conv3x3(t) + (shortcut1x1(in) | in)) [* gate] -> `out` in the
// form OUT.  NT = output channel groups, CB_IN = input groups, SC: 1x1 shortcut convolution (cin != cout) or identity residual.
// `out` may alias `in` or `mid` (registers -> barrier -> LDS).  `p1` = the first pass's weight ring, started by the caller;
// `start_next()` starts whatever pass follows this block (it runs in front of the block's last epilogue).
template <int NT, int CB_IN, bool SC, bool GATE, int OUT, class Next>
__device__ __forceinline__ float c16_rb(const C16RB w, const char *in, char *mid, C16Epi e, float amax, C16Pass<9, NT, CB_IN> &p1, Next start_next)
{
    f32x4 acc[C16Tile<NT>::RW];
    c16_zero<NT>(acc);
    c16_accumulate<9, NT, CB_IN>(in, w.w0, acc, p1);
    C16Pass<9, NT, NT> p2;
    C16Pass<1, NT, CB_IN> p3;
    c16_wstart(p2, w.w2);
    if (SC) c16_wstart(p3, w.wsc);
    amax = c16_epilogue<NT, false, false, C16_IMG>(acc, C16Epi{w.s0, nullptr, nullptr, 0, mid, nullptr}, amax);   // `mid` is nobody's source: no barrier before
    __syncthreads();
    c16_zero<NT>(acc);
    c16_accumulate<9, NT, NT>(mid, w.w2, acc, p2);
    if (SC) c16_accumulate<1, NT, CB_IN>(in, w.wsc, acc, p3);      // ResidualBlock, Model_QBD.py:33-38
    start_next();
    e.res = in;
    e.inv_scale = w.s2;
    __syncthreads();                  // every wave is done reading `in` and `mid`: the output may land on either
    amax = c16_epilogue<NT, !SC, GATE, OUT>(acc, e, amax);
    __syncthreads();
    return amax;
}

// global split-2 tensor of one block [G][256 px][16 ch] (two planes) -> G halo images
__device__ __forceinline__ void c16_load_image(char *dst, const unsigned short *x, size_t plane_stride, int G)
{
    for (int i = threadIdx.x; i < G * 2 * 512; i += blockDim.x) {
        const int j = i & 511, sp = (i >> 9) & 1, cb = i >> 10, px = j >> 1, half = j & 1;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(x + sp * plane_stride + (size_t)(cb * 256 + px) * 16 + half * 8);
        *reinterpret_cast<u32x4 *>(dst + cb * C16_SLOT + sp * C16_PLN + (((px >> 4) + 1) * 18 + (px & 15) + 1) * 32 + half * 16) = v;
    }
}

// 4 consecutive channels of one pixel -> both planes of a halo image (ActOut::store4 with FMT_H2)
__device__ __forceinline__ float c16_store_split(char *img, int cb, int px, int c4, f32x4 v, float amax)
{
    unsigned p0, q0, p1, q1;
    h2_split_pair(v.x, v.y, p0, q0);
    h2_split_pair(v.z, v.w, p1, q1);
    char *dp = img + cb * C16_SLOT + (((px >> 4) + 1) * 18 + (px & 15) + 1) * 32 + c4 * 2;
    *reinterpret_cast<u32x2_t *>(dp) = (u32x2_t){p0, p1};
    *reinterpret_cast<u32x2_t *>(dp + C16_PLN) = (u32x2_t){q0, q1};
    return sat_amax4(amax, v);
}

// conv_misc.hip's head_kernel on an LDS-resident fp32 map [S*S px][16] (channels 0..7): 3x3, 8 -> cout, bias, no activation
template <int S>
__device__ __forceinline__ void c16_head(const float *f, const float *w, const float *bias, int cout, int t, float &acc0, float &acc1)
{
    const int x = t % S, y = t / S;
    acc0 = bias[0];
    acc1 = cout > 1 ? bias[1] : 0.f;
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        if (yy < 0 || yy >= S) continue;
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = x + dx - 1;
            if (xx < 0 || xx >= S) continue;
            const float *xp = f + (yy * S + xx) * 16;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(xp), v1 = *reinterpret_cast<const f32x4 *>(xp + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const float *wp = w + (dy * 3 + dx) * 8 * cout;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                acc0 = fmaf(v[ci], wp[ci * cout], acc0);
                if (cout > 1) acc1 = fmaf(v[ci], wp[ci * cout + 1], acc1);
            }
        }
    }
}

__device__ __forceinline__ void c16_clear_slots(char *slots)
{
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < C16_NSLOT * C16_SLOT / 16; i += blockDim.x) reinterpret_cast<u32x4 *>(slots)[i] = z;
}

}  // namespace

}  // namespace pmp
