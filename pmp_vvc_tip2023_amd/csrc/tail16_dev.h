// tail16_dev.h — device building blocks of the LDS-resident 16x16 tails (chain16.hip), round 6 form: TWO workgroups per CU.
//
// Round 4's form kept six halo images (143 KB) and 168-230 VGPRs per lane: one workgroup of eight waves per CU, whose barriers, epilogues
// and heads nothing could overlap (matrix pipes 0.32-0.46 busy).  This form fits a block into <= 80 KB and four waves:
//   * FOUR slots of one 16-channel group each.  A 64-channel tensor is never resident as a whole: it is streamed from global memory (L2)
//     through a two-group window, pair by pair, in the order pack_h2's K-step list consumes it (K-steps 0..8 touch groups 0 and 1 only,
//     9..17 groups 2 and 3), and a 64-channel INTERMEDIATE is produced and consumed pair by pair the same way.
//   * halo images with a row pitch of 17 pixels: the right halo cell of row r IS the left halo cell of row r + 1 (both zero), 307 cells
//     instead of 324, and the two 8-channel halves of a pixel in separate arrays [half][cell][16 B] - a lane's MFMA B fragment (8 channels
//     of one pixel) is one 16-byte read and 16 lanes read 256 contiguous bytes; the epilogue's 8-byte writes of lanes g and g ^ 1 are
//     adjacent: no bank conflicts on either side (the [cell][32 B] images of round 4: 22 % / 11 % of the LDS cycles).
//   * workgroups of 256 threads: one wave per SIMD and block, two blocks per CU.  A wave's tile is 4 NT rows x one output group (NT =
//     output groups of the layer), so every wave works in every layer; per accumulator the order of products is the launch path's.
// BIT-IDENTICAL to the launch-per-layer path, as before: same pack_h2 streams, same K-step list, x0*w1, x0*w0, x1*w0 per K-step, main pass
// before the shortcut pass, the same epilogue arithmetic.
#pragma once
#include "pmp_kernels.h"
#include "split3.h"

namespace pmp {

namespace {

constexpr int T16_PITCH = 17;
constexpr int T16_CELLS = 18 * T16_PITCH + 1;   // cell(r, x) = (r + 1) * 17 + x + 1 for r, x in -1..16: 0 .. 306
// 8 channels x all cells of one fp16 plane, padded to a multiple of the 256-byte bank row: ds_read_b128 serves lanes {0-3, 12-15} of one
// 16-lane quarter together with lanes {4-11} of the NEXT quarter (MI355X_MICROARCH.md, LDS) - the other 8-channel half of the same
// pixels - so the two halves must sit a whole number of bank rows apart to be each other's complement (307 cells = 4912 B apart: 3 of 16
// slots collide, measured 49 % of the LDS cycles as bank conflicts, the LDS array 79 % busy).  4 slots = 81 920 B = exactly half a CU's LDS.
constexpr int T16_HALF = ((T16_CELLS * 16 + 255) / 256) * 256;
constexpr int T16_PLN = 2 * T16_HALF;           // one fp16 plane of a 16-channel group
constexpr int T16_SLOT = 2 * T16_PLN;           // 20 480 B: one group, both planes
constexpr int T16_ROW = T16_PITCH * 16;         // bytes between image rows inside a half
constexpr int T16_CENTER = (T16_PITCH + 1) * 16;
constexpr int T16_NSLOT = 4;
constexpr int T16_THREADS = 256;
#define T16_GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ constexpr int t16_tapoff(int t) { return ((t / 3) * T16_PITCH + t % 3) * 16; }

// K-step list of a pass = pack_h2's (pack.cpp): with an even group count and an odd tap count the last tap of an even group is paired
// with the last tap of the odd group that follows; otherwise the last pair of a group is zero-padded on the weight side (the pixels of
// its first tap are read twice, as conv_f16x3.hip does).  Byte offset (group + tap) of K-half `half` of step `st`:
template <int T, int CB>
__device__ __forceinline__ constexpr int t16_step_off(int st, int half)
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
    return cb * T16_SLOT + (T == 1 ? T16_CENTER : t16_tapoff(tap));
}

// Where a wave works in a layer with NT output groups: 4 NT rows x one group.  `ntw` / `ct0`: the weight stream may hold more groups than
// this pass produces (a 64-channel intermediate made pair by pair: NT = 2 of ntw = 4, first group ct0).
template <int NT>
struct T16Tile {
    static constexpr int RW = 4 * NT;
    __device__ __forceinline__ static int ct() { return (int)(threadIdx.x >> 6) % NT; }
    __device__ __forceinline__ static int row0() { return ((int)(threadIdx.x >> 6) / NT) * RW; }
};

// The weight ring of one convolution pass over CB source groups: the fragments of the next D K-steps.  t16_wstart requests the first D -
// the caller does that as early as the registers allow, in front of the epilogue and the barrier that precede the pass.
template <int T, int CB>
struct T16Pass {
    static constexpr bool paired = !(CB & 1) && (T & 1);
    static constexpr int NS = paired ? (CB / 2) * T : CB * ((T + 1) / 2);
    static constexpr int D = NS < 4 ? NS : 4;           // K-steps of lead (8 registers each)
    f16x8 wq[D][2];
    const T16_GLOBAL f16x8 *wl;                         // this lane's fragment of K-step 0, split 0
    int kstride, split1;                                // f16x8 units: between K-steps, between the two splits
};

// wpk: the stream at the pass's first K-step; ntw: groups in the stream; ct: the group this wave produces
template <int T, int CB>
__device__ __forceinline__ void t16_wstart(T16Pass<T, CB> &p, const unsigned short *wpk, int ntw, int ct)
{
    // (explicitly global: a pointer that reached this point through a struct is generic to hipcc, and a flat load counts on both wait counters)
    p.wl = (const T16_GLOBAL f16x8 *)wpk + (threadIdx.x & 63) + ct * 64;
    p.kstride = 2 * ntw * 64;
    p.split1 = ntw * 64;
#pragma unroll
    for (int st = 0; st < T16Pass<T, CB>::D; ++st) {
        p.wq[st][0] = p.wl[(size_t)st * p.kstride];
        p.wq[st][1] = p.wl[(size_t)st * p.kstride + p.split1];
    }
}

// One convolution pass (T = 9: 3x3, T = 1: 1x1 on the same halo images) over the CB source groups at `src` into the wave's RW rows; its
// first weight fragments are on their way (t16_wstart).  Fully unrolled.  A K-step runs in sub-steps of four rows; the pixel fragments of
// the next sub-step are read from LDS during this one's MFMAs.  Per accumulator the order is the launch path's: x0*w1, x0*w0, x1*w0.
template <int T, int CB, int RW>
__device__ __forceinline__ void t16_accumulate(const char *src, int row0, f32x4 (&acc)[RW], T16Pass<T, CB> &p)
{
    constexpr int NS = T16Pass<T, CB>::NS, D = T16Pass<T, CB>::D;
    constexpr int SUB = RW / 4, NTK = NS * SUB;
    const int lane = threadIdx.x & 63, xl = lane & 15, g = lane >> 4;
    const char *pbase = src + (row0 * T16_PITCH + xl) * 16 + (g & 1) * T16_HALF;
    const bool hi = (g >> 1) != 0;
    f16x8 xq[2][2][4];
    auto xload = [&](int tk) __attribute__((always_inline)) {
        const int st = tk / SUB, h = tk % SUB;
        const char *q = pbase + (hi ? t16_step_off<T, CB>(st, 1) : t16_step_off<T, CB>(st, 0)) + h * 4 * T16_ROW;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            xq[tk & 1][0][m] = *reinterpret_cast<const f16x8 *>(q + m * T16_ROW);
            xq[tk & 1][1][m] = *reinterpret_cast<const f16x8 *>(q + T16_PLN + m * T16_ROW);
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
        const f16x8 w0 = p.wq[st % D][0], w1 = p.wq[st % D][1];
        f16x8 (&x0)[4] = xq[tk & 1][0], (&x1)[4] = xq[tk & 1][1];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[h * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x0[m], acc[h * 4 + m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[h * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x0[m], acc[h * 4 + m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[h * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x1[m], acc[h * 4 + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (h == SUB - 1 && st + D < NS) {
            p.wq[st % D][0] = p.wl[(size_t)(st + D) * p.kstride];
            p.wq[st % D][1] = p.wl[(size_t)(st + D) * p.kstride + p.split1];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// A 1x1 pass over ONE pair of source groups whose values are fp32 and not resident as an image: the B fragments - 8 channels of a pixel, both
// fp16 terms - are split in registers from `src(group, row, x)` (16 floats of that pixel), exactly as t16_store_split would have written them
// into an image, and fed to the same three products.  No image, no barrier: resblock_q4's shortcut over the four pairs of x6.
template <int RW, class Src>
__device__ __forceinline__ void t16_accumulate_1x1_f32(Src src, int row0, f32x4 (&acc)[RW], T16Pass<1, 2> &p)
{
    const int lane = threadIdx.x & 63, xl = lane & 15, g = lane >> 4;
    const f16x8 w0 = p.wq[0][0], w1 = p.wq[0][1];
    f32x4 va[RW], vb[RW];
#pragma unroll
    for (int m = 0; m < RW; ++m) {
        const float *s = src(g >> 1, row0 + m, xl) + (g & 1) * 8;      // K-half = group of the pair, then the 8-channel half
        va[m] = *reinterpret_cast<const f32x4 *>(s);
        vb[m] = *reinterpret_cast<const f32x4 *>(s + 4);
    }
#pragma unroll
    for (int m = 0; m < RW; ++m) {
        unsigned p0, q0, p1, q1, p2, q2, p3, q3;
        h2_split_pair(va[m].x, va[m].y, p0, q0);
        h2_split_pair(va[m].z, va[m].w, p1, q1);
        h2_split_pair(vb[m].x, vb[m].y, p2, q2);
        h2_split_pair(vb[m].z, vb[m].w, p3, q3);
        const f16x8 h0 = __builtin_bit_cast(f16x8, (u32x4){p0, p1, p2, p3}), h1 = __builtin_bit_cast(f16x8, (u32x4){q0, q1, q2, q3});
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, h0, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, h0, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, h1, acc[m], 0, 0, 0);
    }
}

template <int RW>
__device__ __forceinline__ void t16_zero(f32x4 (&acc)[RW])
{
#pragma unroll
    for (int m = 0; m < RW; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// byte offset of a lane's 4 consecutive channels (g * 4 ..) of pixel (row, xl) inside a group's plane
__device__ __forceinline__ int t16_px4(int row, int xl, int g) { return (g >> 1) * T16_HALF + ((row + 1) * T16_PITCH + xl + 1) * 16 + (g & 1) * 8; }

// The epilogue of conv_f16x3.hip's h2_epilogue on a wave's rows: x 1/S (+ identity residual), ReLU, x gate, then ONE of: two-term split
// into a halo image (T16_IMG), the same to a global split-2 tensor [group][256 px][16] (T16_GLB), plain fp32 [group][256 px][16] in LDS or
// global memory (T16_F32), 2x2 max-pool to fp32 [group][64 px][16] (T16_POOL).  Returns the running |max| of what the split clamps.
enum { T16_IMG = 0, T16_F32 = 1, T16_POOL = 2, T16_GLB = 3 };
struct T16Epi {
    float inv_scale;
    const char *res;                    // RES: LDS halo image of the identity residual (the slot of THIS wave's group)
    const unsigned short *gate;         // GATE: global split-2 tensor of this block [groups][256 px][16], at this wave's group
    size_t gate_stride;
    char *dst_img;                      // T16_IMG: the slot of this wave's output group
    float *dst_f32;                     // T16_F32 / T16_POOL: output of this wave's group (LDS or global)
    unsigned short *dst_glb;            // T16_GLB: this wave's group of a global split-2 tensor
    size_t glb_stride;
    int f32_pitch;                      // T16_F32: floats per pixel (16; 20 where a head reads the map: conflict-free 16-byte reads of consecutive pixels)
};

template <int RW, bool RES, bool GATE, int OUT>
__device__ __forceinline__ float t16_epilogue(f32x4 (&acc)[RW], int row0, const T16Epi &e, float amax)
{
    const int lane = threadIdx.x & 63, xl = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 0; m < RW; ++m) {
        const int row = row0 + m;
        f32x4 v = acc[m];
        if (RES) {
            const char *rp = e.res + t16_px4(row, xl, g);
            const u32x2_t a = *reinterpret_cast<const u32x2_t *>(rp), b = *reinterpret_cast<const u32x2_t *>(rp + T16_PLN);
            v = v * e.inv_scale + (f32x4){h2_sum_lo(a.x, b.x), h2_sum_hi(a.x, b.x), h2_sum_lo(a.y, b.y), h2_sum_hi(a.y, b.y)};
        } else {
            v = v * e.inv_scale;
        }
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (GATE) {
            const T16_GLOBAL unsigned short *gp = (const T16_GLOBAL unsigned short *)e.gate + (size_t)((row * 16 + xl) * 16 + g * 4);
            const u32x2_t ga = *reinterpret_cast<const T16_GLOBAL u32x2_t *>(gp), gb = *reinterpret_cast<const T16_GLOBAL u32x2_t *>(gp + e.gate_stride);
            v *= (f32x4){h2_sum_lo(ga.x, gb.x), h2_sum_hi(ga.x, gb.x), h2_sum_lo(ga.y, gb.y), h2_sum_hi(ga.y, gb.y)};     // load_split2_4
        }
        if (OUT != T16_F32 && OUT != T16_POOL) amax = sat_amax4(amax, v);     // only what is split can clamp
        acc[m] = v;
    }
#pragma unroll
    for (int m = 0; m < RW; m += 2) {
        if (OUT == T16_POOL) {
            f32x4 v = acc[m], u = acc[m + 1];
            v.x = fmaxf(v.x, u.x); v.y = fmaxf(v.y, u.y); v.z = fmaxf(v.z, u.z); v.w = fmaxf(v.w, u.w);
            f32x4 o;
            o.x = __shfl_xor(v.x, 1); o.y = __shfl_xor(v.y, 1); o.z = __shfl_xor(v.z, 1); o.w = __shfl_xor(v.w, 1);
            v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
            if ((xl & 1) == 0) *reinterpret_cast<f32x4 *>(e.dst_f32 + ((((row0 + m) >> 1) * 8 + (xl >> 1)) * 16 + g * 4)) = v;
        } else if (OUT == T16_F32) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                *reinterpret_cast<f32x4 *>(e.dst_f32 + (((row0 + m + k) * 16 + xl) * e.f32_pitch + g * 4)) = acc[m + k];
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned p0, q0, p1, q1;
                h2_split_pair(acc[m + k].x, acc[m + k].y, p0, q0);
                h2_split_pair(acc[m + k].z, acc[m + k].w, p1, q1);
                if (OUT == T16_IMG) {
                    char *dp = e.dst_img + t16_px4(row0 + m + k, xl, g);
                    *reinterpret_cast<u32x2_t *>(dp) = (u32x2_t){p0, p1};
                    *reinterpret_cast<u32x2_t *>(dp + T16_PLN) = (u32x2_t){q0, q1};
                } else {
                    T16_GLOBAL unsigned short *dp = (T16_GLOBAL unsigned short *)e.dst_glb + (size_t)(((row0 + m + k) * 16 + xl) * 16 + g * 4);
                    *reinterpret_cast<T16_GLOBAL u32x2_t *>(dp) = (u32x2_t){p0, p1};
                    *reinterpret_cast<T16_GLOBAL u32x2_t *>(dp + e.glb_stride) = (u32x2_t){q0, q1};
                }
            }
        }
    }
    return amax;
}

struct T16RB {                    // a ResidualBlock's f16x3 streams (RBWeights): first conv, second conv, 1x1 shortcut, and 1/S of each pass
    const unsigned short *w0, *w2, *wsc;
    float s0, s2;
};
constexpr size_t T16_KSTEP = 2 * 64 * 8;          // halves per K-step and output group of a pack_h2 stream

// global split-2 tensor of one block [G][256 px][16 ch] (two planes), groups g0 .. g0 + 1 -> two slots at dst.  In two halves so that the
// requests can be issued early (t16_fetch: 4 x 16 B per thread and plane pair) and parked behind a barrier (t16_park).
struct T16Fetch { u32x4 v[8]; };
__device__ __forceinline__ void t16_fetch(T16Fetch &f, const unsigned short *x, size_t plane_stride, int g0)
{
    const T16_GLOBAL unsigned short *xg = (const T16_GLOBAL unsigned short *)x + (size_t)g0 * 4096;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = threadIdx.x + k * T16_THREADS;            // 0 .. 2047: (group, plane, 8 pixels, half, pixel of the 8)
        const int j = i & 511, sp = (i >> 9) & 1, cb = i >> 10, half = (j >> 3) & 1, px = ((j >> 4) << 3) | (j & 7);   // 8 lanes = 128 contiguous LDS bytes
        f.v[k] = *reinterpret_cast<const T16_GLOBAL u32x4 *>(xg + sp * plane_stride + (size_t)(cb * 256 + px) * 16 + half * 8);
    }
}
__device__ __forceinline__ void t16_park(const T16Fetch &f, char *dst)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = threadIdx.x + k * T16_THREADS;
        const int j = i & 511, sp = (i >> 9) & 1, cb = i >> 10, half = (j >> 3) & 1, px = ((j >> 4) << 3) | (j & 7);
        *reinterpret_cast<u32x4 *>(dst + cb * T16_SLOT + sp * T16_PLN + half * T16_HALF + (((px >> 4) + 1) * T16_PITCH + (px & 15) + 1) * 16) = f.v[k];
    }
}

// 4 consecutive channels c4 .. c4 + 3 of pixel px -> both planes of group slot `img` (ActOut::store4 with FMT_H2)
__device__ __forceinline__ float t16_store_split(char *img, int px, int c4, f32x4 v, float amax)
{
    unsigned p0, q0, p1, q1;
    h2_split_pair(v.x, v.y, p0, q0);
    h2_split_pair(v.z, v.w, p1, q1);
    char *dp = img + (c4 >> 3) * T16_HALF + (((px >> 4) + 1) * T16_PITCH + (px & 15) + 1) * 16 + (c4 & 7) * 2;
    *reinterpret_cast<u32x2_t *>(dp) = (u32x2_t){p0, p1};
    *reinterpret_cast<u32x2_t *>(dp + T16_PLN) = (u32x2_t){q0, q1};
    return sat_amax4(amax, v);
}

// conv_misc.hip's head_kernel on an LDS-resident fp32 map [S*S px][PITCH] (channels 0..7): 3x3, 8 -> cout, bias, no activation.  The
// weights [9][8][cout] and the bias are in LDS too (`w`, `bias`: the caller staged them; every lane reads the same address - a broadcast).
#define T16_LDS __attribute__((address_space(3)))
// (Branch-free: a tap outside the map is computed on a clamped pixel and dropped by a select - the reference skips it, the chain of the
// taps that count is the same fmaf sequence; with `continue` in the loops hipcc cannot hoist the LDS reads and every tap pays their latency.)
template <int S, int PITCH>
__device__ __forceinline__ void t16_head(const float *f_, const float *w_, const float *bias_, int cout, int t, float &acc0, float &acc1)
{
    const T16_LDS float *f = (const T16_LDS float *)f_, *w = (const T16_LDS float *)w_, *bias = (const T16_LDS float *)bias_;
    const int x = t % S, y = t / S;
    acc0 = bias[0];
    acc1 = cout > 1 ? bias[1] : 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        const bool oky = yy >= 0 && yy < S;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = x + dx - 1;
            const bool ok = oky && xx >= 0 && xx < S;
            const T16_LDS float *xp = f + ((oky ? yy : y) * S + (xx >= 0 && xx < S ? xx : x)) * PITCH;
            const f32x4 v0 = *reinterpret_cast<const T16_LDS f32x4 *>(xp), v1 = *reinterpret_cast<const T16_LDS f32x4 *>(xp + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const T16_LDS float *wp = w + (dy * 3 + dx) * 8 * cout;
            float t0 = acc0, t1 = acc1;
            if (cout > 1) {
                float wv[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w4 = *reinterpret_cast<const T16_LDS f32x4 *>(wp + q * 4);
                    wv[q * 4] = w4.x; wv[q * 4 + 1] = w4.y; wv[q * 4 + 2] = w4.z; wv[q * 4 + 3] = w4.w;
                }
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) {
                    t0 = fmaf(v[ci], wv[ci * 2], t0);
                    t1 = fmaf(v[ci], wv[ci * 2 + 1], t1);
                }
            } else {
                const f32x4 wa = *reinterpret_cast<const T16_LDS f32x4 *>(wp), wb = *reinterpret_cast<const T16_LDS f32x4 *>(wp + 4);
                const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) t0 = fmaf(v[ci], wv[ci], t0);
            }
            acc0 = ok ? t0 : acc0;
            acc1 = ok ? t1 : acc1;
        }
    }
}

// Zero the halo cells of `nslots` group slots (both planes, both halves): all an image needs before its interior is written - 51 of 307
// cells: row -1 (cells 0..17, which include (0, -1)), the shared right / left halo cell of rows 0..15, and row 16 (289..306).
__device__ __forceinline__ void t16_clear_borders(char *p, int nslots)
{
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < nslots * 4 * 51; i += T16_THREADS) {
        const int arr = i / 51, k = i % 51;
        const int cell = k < 18 ? k : k < 33 ? 34 + (k - 18) * T16_PITCH : 289 + (k - 33);
        *reinterpret_cast<u32x4 *>(p + arr * T16_HALF + cell * 16) = z;
    }
}

__device__ __forceinline__ void t16_clear(char *p, int bytes)
{
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < bytes / 16; i += T16_THREADS) reinterpret_cast<u32x4 *>(p)[i] = z;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ResidualBlock(64 -> 32, 3x3, 1x1 shortcut) whose 64-channel input X lives in global memory (split-2, [4][256 px][16]): trunk_B1.0 /
// trunk_B2.0 of the MTT nets, resblock_q3 of the QT nets.  All four slots A..D:
//   conv 1:   X groups 0, 1 -> A, B and 2, 3 -> C, D (both requested by the caller at kernel start) | 18 K-steps, uninterrupted
//   t = relu(.) -> A, B (groups 0, 1 are dead);  conv 2: 9 K-steps from A, B
//   shortcut: X groups 0, 1 again -> A, B (requested during conv 2) | K-step 0 from A, B | K-step 1 from C, D (groups 2, 3 are still there)
// The accumulators come back UNFINISHED (no epilogue): the caller applies the block's second epilogue in the form it needs, after the
// closing barrier (every slot is free then).  `p1` = the first pass's ring, started by the caller.
template <class Next>
__device__ __forceinline__ float t16_rb64(const T16RB w, const unsigned short *x, size_t x_stride, char *slots, T16Fetch &f01, const T16Fetch &f23,
                                          f32x4 (&acc)[8], float amax, T16Pass<9, 2> &p1, Next start_next)
{
    typedef T16Tile<2> WT;
    char *A = slots, *C = slots + 2 * T16_SLOT;
    const int ct = WT::ct(), row0 = WT::row0();
    t16_park(f01, A);
    t16_park(f23, C);
    T16Pass<9, 2> p1b;
    t16_wstart(p1b, w.w0 + 9 * 2 * T16_KSTEP, 2, ct);
    __syncthreads();
    t16_zero<8>(acc);
    t16_accumulate<9, 2, 8>(A, row0, acc, p1);
    t16_accumulate<9, 2, 8>(C, row0, acc, p1b);
    T16Pass<9, 2> p2;
    t16_wstart(p2, w.w2, 2, ct);
    t16_fetch(f01, x, x_stride, 0);                    // for the shortcut pass, behind conv 2
    __syncthreads();                                   // every wave is done with groups 0, 1: the intermediate takes their slots
    amax = t16_epilogue<8, false, false, T16_IMG>(acc, row0, T16Epi{w.s0, nullptr, nullptr, 0, A + ct * T16_SLOT, nullptr, nullptr, 0, 16}, amax);
    __syncthreads();
    t16_zero<8>(acc);
    t16_accumulate<9, 2, 8>(A, row0, acc, p2);
    T16Pass<1, 2> p3a, p3b;
    t16_wstart(p3a, w.wsc, 2, ct);
    t16_wstart(p3b, w.wsc + 1 * 2 * T16_KSTEP, 2, ct);
    __syncthreads();                                   // every wave is done with the intermediate
    t16_park(f01, A);
    __syncthreads();
    t16_accumulate<1, 2, 8>(A, row0, acc, p3a);        // ResidualBlock, Model_QBD.py:33-38: the shortcut pass follows the main pass
    t16_accumulate<1, 2, 8>(C, row0, acc, p3b);
    start_next();
    __syncthreads();                                   // all four slots are free
    return amax;
}

// ResidualBlock on LDS images with at most 32 channels on either side: t = relu(conv3x3(in)) -> `mid`; out = relu(conv3x3(t) +
// (shortcut1x1(in) | in)) -> left UNFINISHED in acc for the caller's epilogue (after a barrier: every slot it names may be overwritten).
template <int NT, int CB_IN, bool SC, class Next>
__device__ __forceinline__ float t16_rb(const T16RB w, const char *in, char *mid, f32x4 (&acc)[4 * NT], float amax, T16Pass<9, CB_IN> &p1, Next start_next)
{
    typedef T16Tile<NT> WT;
    const int ct = WT::ct(), row0 = WT::row0();
    t16_zero<4 * NT>(acc);
    t16_accumulate<9, CB_IN, 4 * NT>(in, row0, acc, p1);
    T16Pass<9, NT> p2;
    T16Pass<1, CB_IN> p3;
    t16_wstart(p2, w.w2, NT, ct);
    if (SC) t16_wstart(p3, w.wsc, NT, ct);
    amax = t16_epilogue<4 * NT, false, false, T16_IMG>(acc, row0, T16Epi{w.s0, nullptr, nullptr, 0, mid + ct * T16_SLOT, nullptr, nullptr, 0, 16}, amax);   // `mid` is nobody's source: no barrier before
    __syncthreads();
    t16_zero<4 * NT>(acc);
    t16_accumulate<9, NT, 4 * NT>(mid, row0, acc, p2);
    if (SC) t16_accumulate<1, CB_IN, 4 * NT>(in, row0, acc, p3);
    start_next();
    __syncthreads();                                   // every wave is done reading `in` and `mid`
    return amax;
}

}  // namespace

}  // namespace pmp
