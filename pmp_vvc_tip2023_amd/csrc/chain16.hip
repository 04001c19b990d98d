// chain16.hip — the 16x16-resolution tails of the four nets as ONE workgroup per block with every activation resident in LDS
// (f16x3 datapath).  At 16x16 a block IS a tile: a 64-channel tensor is 65 KB, so the chains below - until round 3 about thirty launches per
// pass that existed only to move <= 65 KB per block through HBM between tiny layers - need no halo recompute and no HBM round trip:
//
//   MTT nets (Model_QBD.py:116-124, :138-146 / :214-223, :236-244), msbd_branch16_kernel:
//       x5 -> trunk_B1 (RB 64->32, 32->16, 16->8) -> conv_B1 -> out0
//          -> cat[up2(q), out0] -> trunk_Att1 (RB 3->32, 32->64) * x5 -> trunk_B2 -> conv_B2 -> out1, out1[:,0] += out0[:,0]
//   QT nets (Model_QBD.py:71-76, :83-91 / :169-174, :181-189), qt_tail16_kernel:
//       x4 -> resblock_q3 -> cat[x5, up2(mp2), up4(mp4), up8(mp8)] -> resblock_q4 -> resblock_q5, max_pool2d(2) -> resblock_q6 -> conv_q2
//
// BIT-IDENTICAL to the launch-per-layer path (nets.cpp with fusion off; tests/test_gpu_parity.py): the convolutions consume the same
// packed weight streams (pack_h2, pack.cpp) in the same K-step order with the same three products per K-step (x0*w1, x0*w0, x1*w0) into
// the same fp32 accumulators, the epilogues apply the same operations in the same order (1/S, residual or 1x1 shortcut pass, ReLU, gate,
// 2x2 pool, two-term fp16 split with the range clamp), tensors that the launch path hands to its fp32 kernels stay fp32 here too, and the
// fp32 kernels (heads, 8x8 direct convolutions, multi-scale pool) are restated with their accumulation order.  What changes is where
// the tensors live.
//
// LDS: six "slots" of one 16-channel group each in the halo-image form the MFMA kernels stage ([plane][18 x 18 px][16 ch] fp16, zero
// border: 20 736 B), i.e. a 64-channel input next to its 32-channel intermediate, plus an fp32 area; 143-147 KB: one workgroup of eight
// waves per CU.  A wave owns four or eight image rows x one 16-channel group of a layer's output (C16Tile); every layer ends "registers -> barrier -> LDS" so that an
// output may overwrite the slots of a tensor that was still being read.
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

// One convolution pass (T = 9: 3x3, T = 1: 1x1 on the same halo images) over the CB source groups at `src` into the wave's accumulators.
// Fully unrolled.  A K-step runs in sub-steps of four rows; the weight fragments (L2 / L1 hits) are requested D K-steps ahead into a
// register ring, the pixel fragments of the next sub-step are read from LDS during this one's MFMAs.  Per accumulator the order is the
// launch path's: x0*w1, x0*w0, x1*w0, K-step after K-step.
template <int T, int NT, int CB>
__device__ __forceinline__ void c16_accumulate(const char *src, const unsigned short *wpk, f32x4 (&acc)[C16Tile<NT>::RW])
{
    typedef C16Tile<NT> WT;
    constexpr bool paired = !(CB & 1) && (T & 1);
    constexpr int NS = paired ? (CB / 2) * T : CB * ((T + 1) / 2);
    constexpr int D = NS < 6 ? NS : 6;                  // K-steps of lead (8 registers each)
    constexpr int SUB = WT::RW / 4, NTK = NS * SUB;
    if (!WT::active()) return;
    const int lane = threadIdx.x & 63, xl = lane & 15, g = lane >> 4;
    const char *pbase = src + (WT::row0() * 18 + xl) * 32 + (g & 1) * 16;
    const bool hi = (g >> 1) != 0;
    // (explicitly global: a pointer that reached this point through a struct is generic to hipcc, and a flat load counts on both wait counters)
    const C16_GLOBAL f16x8 *wl = (const C16_GLOBAL f16x8 *)wpk + lane + WT::ct() * 64;
    f16x8 wq[D][2], xq[2][2][4];
    auto wload = [&](int st) __attribute__((always_inline)) {
        wq[st % D][0] = wl[(size_t)st * (2 * NT * 64)];
        wq[st % D][1] = wl[(size_t)st * (2 * NT * 64) + NT * 64];
    };
    auto xload = [&](int tk) __attribute__((always_inline)) {
        const int st = tk / SUB, h = tk % SUB;
        const char *p = pbase + (hi ? c16_step_off<T, CB>(st, 1) : c16_step_off<T, CB>(st, 0)) + h * 4 * C16_ROW;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            xq[tk & 1][0][m] = *reinterpret_cast<const f16x8 *>(p + m * C16_ROW);
            xq[tk & 1][1][m] = *reinterpret_cast<const f16x8 *>(p + C16_PLN + m * C16_ROW);
        }
    };
#pragma unroll
    for (int st = 0; st < D; ++st) wload(st);
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
// `out` may alias `in` or `mid` (registers -> barrier -> LDS).
template <int NT, int CB_IN, bool SC, bool GATE, int OUT>
__device__ __forceinline__ float c16_rb(const C16RB w, const char *in, char *mid, C16Epi e, float amax)
{
    f32x4 acc[C16Tile<NT>::RW];
    c16_zero<NT>(acc);
    c16_accumulate<9, NT, CB_IN>(in, w.w0, acc);
    amax = c16_epilogue<NT, false, false, C16_IMG>(acc, C16Epi{w.s0, nullptr, nullptr, 0, mid, nullptr}, amax);   // `mid` is nobody's source: no barrier before
    __syncthreads();
    c16_zero<NT>(acc);
    c16_accumulate<9, NT, NT>(mid, w.w2, acc);
    if (SC) c16_accumulate<1, NT, CB_IN>(in, w.wsc, acc);      // ResidualBlock, Model_QBD.py:33-38
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

// ===================================================================================================== MTT nets: B1, Att1, B2
struct MsbdBranch16Args {
    const unsigned short *x5; size_t x5_stride;     // [N][4][16][16][16] split-2: trunk_M2's pooled output (and attention 1's gate operand)
    const float *qt;                                // raw QT logits [N][64]
    float *bt, *dire;                               // [N][3][256]: layers 0 and 1 are written here
    C16RB b1[3], att[2], b2[3];
    const float *head_w[2], *head_b[2];
    unsigned *sat;
};

__global__ __launch_bounds__(512, 2) void msbd_branch16_kernel(MsbdBranch16Args a)
{
    __shared__ __attribute__((aligned(16))) char slots[C16_NSLOT * C16_SLOT];
    __shared__ __attribute__((aligned(16))) float f0[256 * 16];
    __shared__ float s_bt[256], s_dire[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const unsigned short *x5 = a.x5 + (size_t)n * 4 * 4096;
    char *S0 = slots, *S2 = slots + 2 * C16_SLOT, *S4 = slots + 4 * C16_SLOT;
    float amax = 0.f;
    c16_clear_slots(slots);
    __syncthreads();
    c16_load_image(S0, x5, a.x5_stride, 4);
    __syncthreads();

    auto branch = [&](const C16RB r0, const C16RB r1, const C16RB r2, const float *hw, const float *hb, int layer) __attribute__((always_inline)) {
        // trunk_B: 64 ch in S0..3 -> 32 ch (S0..1) -> 16 ch (S2) -> 8 ch fp32 (f0) -> head
        amax = c16_rb<2, 4, true, false, C16_IMG>(r0, S0, S4, C16Epi{0.f, nullptr, nullptr, 0, S0, nullptr}, amax);
        amax = c16_rb<1, 2, true, false, C16_IMG>(r1, S0, S4, C16Epi{0.f, nullptr, nullptr, 0, S2, nullptr}, amax);
        amax = c16_rb<1, 1, true, false, C16_F32>(r2, S2, S4, C16Epi{0.f, nullptr, nullptr, 0, nullptr, f0}, amax);
        if (tid < 256) {
            float acc0, acc1;
            c16_head<16>(f0, hw, hb, 2, tid, acc0, acc1);
            if (layer > 0) acc0 += s_bt[tid];          // out1[:, 0] += out0[:, 0]  (Model_QBD.py:146)
            const size_t o = ((size_t)n * 3 + layer) * 256 + tid;
            a.bt[o] = acc0;
            a.dire[o] = acc1;
            s_bt[tid] = acc0;
            s_dire[tid] = acc1;
        }
        __syncthreads();
    };
    branch(a.b1[0], a.b1[1], a.b1[2], a.head_w[0], a.head_b[0], 0);

    // attention input cat[up2(q), out0] (conv_misc.hip: att_input_kernel, S = 16) -> S0, channels 3..15 zero
    if (tid < 256) {
        const int x = tid & 15, y = tid >> 4;
        const f32x4 v = {a.qt[(size_t)n * 64 + (y >> 1) * 8 + (x >> 1)], s_bt[tid], s_dire[tid], 0.f};
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        amax = c16_store_split(S0, 0, tid, 0, v, amax);
        c16_store_split(S0, 0, tid, 4, z, 0.f);
        c16_store_split(S0, 0, tid, 8, z, 0.f);
        c16_store_split(S0, 0, tid, 12, z, 0.f);
    }
    __syncthreads();
    // trunk_Att1: 3 -> 32 (S4..5), 32 -> 64 gated by x5 (S0..3)
    amax = c16_rb<2, 1, true, false, C16_IMG>(a.att[0], S0, S4, C16Epi{0.f, nullptr, nullptr, 0, S4, nullptr}, amax);
    amax = c16_rb<4, 2, true, true, C16_IMG>(a.att[1], S4, S0, C16Epi{0.f, nullptr, x5, a.x5_stride, S0, nullptr}, amax);
    branch(a.b2[0], a.b2[1], a.b2[2], a.head_w[1], a.head_b[1], 1);
    sat_report(a.sat, amax);
}

// ===================================================================================================== QT nets: q3 .. conv_q2
struct QtTail16Args {
    const unsigned short *x4; size_t x4_stride;     // [N][4][16][16][16] split-2: resblock_q2's pooled output
    float *qt;                                      // [N][64]
    C16RB q3, q4, q5;
    const float *d_w0, *d_w2, *d_wsc;               // resblock_q6 on the 8x8 map: plain fp32 [tap][cin][8], [tap][8][8], [32][8] (conv_direct8_kernel)
    const float *head_w, *head_b;
    unsigned *sat;
};

__global__ __launch_bounds__(512, 2) void qt_tail16_kernel(QtTail16Args a)
{
    __shared__ __attribute__((aligned(16))) char lds[C16_NSLOT * C16_SLOT + 8192];
    const int n = blockIdx.x, tid = threadIdx.x;
    char *S0 = lds, *S2 = lds + 2 * C16_SLOT, *S4 = lds + 4 * C16_SLOT;
    float *F = reinterpret_cast<float *>(S4);       // fp32 area = slots 4, 5 and the 8 KB behind them (no halo image lives there after q3)
    float *x5 = F;                                  // [2][256][16]
    float *p2 = F + 8192, *p4 = p2 + 2 * 1024, *p8 = p4 + 2 * 256;     // [2][64][16], [2][16][16], [2][4][16]
    float amax = 0.f;
    c16_clear_slots(lds);
    __syncthreads();
    c16_load_image(S0, a.x4 + (size_t)n * 4 * 4096, a.x4_stride, 4);
    __syncthreads();
    // resblock_q3: 64 -> 32, output fp32 (the multi-scale pool reads it).  Its intermediate uses S4..5, which then becomes the fp32 area.
    amax = c16_rb<2, 4, true, false, C16_F32>(a.q3, S0, S4, C16Epi{0.f, nullptr, nullptr, 0, nullptr, x5}, amax);
    // multi-scale pool (conv_misc.hip: multipool_concat_kernel), both groups
    for (int i = tid; i < 2 * 1024; i += 512) {
        const int cb = i >> 10, j = i & 1023, c = j & 15, x = (j >> 4) & 7, y = j >> 7;
        const float *q = x5 + cb * 4096 + ((2 * y) * 16 + 2 * x) * 16 + c;
        p2[i] = fmaxf(fmaxf(q[0], q[16]), fmaxf(q[256], q[272]));
    }
    __syncthreads();
    {
        const int cb = tid >> 8, j = tid & 255, c = j & 15, x = (j >> 4) & 3, y = j >> 6;
        const float *q = p2 + cb * 1024 + ((2 * y) * 8 + 2 * x) * 16 + c;
        p4[tid] = fmaxf(fmaxf(q[0], q[16]), fmaxf(q[128], q[144]));
    }
    __syncthreads();
    if (tid < 128) {
        const int cb = tid >> 6, j = tid & 63, c = j & 15, x = (j >> 4) & 1, y = j >> 5;
        const float *q = p4 + cb * 256 + ((2 * y) * 4 + 2 * x) * 16 + c;
        p8[tid] = fmaxf(fmaxf(q[0], q[16]), fmaxf(q[64], q[80]));
    }
    __syncthreads();
    // x6 = cat[x5, up2(mp2), up4(mp4), up8(mp8)], 128 channels: never materialised as a whole - channel-group pair p (= source p) is
    // written to S0..1 when the convolution needs it
    auto x6_pair = [&](int p) __attribute__((always_inline)) {
        for (int i = tid; i < 2048; i += 512) {        // (group, pixel, quad of channels)
            const int c4 = (i & 3) * 4, px = (i >> 2) & 255, cb = i >> 10, x = px & 15, y = px >> 4;
            const float *s = p == 0 ? x5 + cb * 4096 + px * 16
                           : p == 1 ? p2 + cb * 1024 + ((y >> 1) * 8 + (x >> 1)) * 16
                           : p == 2 ? p4 + cb * 256 + ((y >> 2) * 4 + (x >> 2)) * 16
                                    : p8 + cb * 64 + ((y >> 3) * 2 + (x >> 3)) * 16;
            amax = c16_store_split(S0, cb, px, c4, *reinterpret_cast<const f32x4 *>(s + c4), amax);
        }
    };
    f32x4 acc[C16Tile<2>::RW];
    // resblock_q4: 128 -> 32.  First conv: 8 groups = 4 pairs of 9 K-steps; second conv: 32 -> 32 from S2..3, then the 1x1 shortcut over x6
    c16_zero<2>(acc);
    for (int p = 0; p < 4; ++p) {
        x6_pair(p);
        __syncthreads();
        c16_accumulate<9, 2, 2>(S0, a.q4.w0 + (size_t)p * 9 * (2 * 2 * 64 * 8), acc);
        __syncthreads();
    }
    amax = c16_epilogue<2, false, false, C16_IMG>(acc, C16Epi{a.q4.s0, nullptr, nullptr, 0, S2, nullptr}, amax);
    __syncthreads();
    c16_zero<2>(acc);
    c16_accumulate<9, 2, 2>(S2, a.q4.w2, acc);
    for (int p = 0; p < 4; ++p) {
        x6_pair(p);
        __syncthreads();
        c16_accumulate<1, 2, 2>(S0, a.q4.wsc + (size_t)p * (2 * 2 * 64 * 8), acc);
        __syncthreads();
    }
    amax = c16_epilogue<2, false, false, C16_IMG>(acc, C16Epi{a.q4.s2, nullptr, nullptr, 0, S0, nullptr}, amax);   // x7 -> S0..1 (every wave is past the last barrier)
    __syncthreads();
    // resblock_q5: 32 -> 32, identity shortcut, max_pool2d(2) -> fp32 [2][64][16] over the dead x5
    float *x8 = F;
    amax = c16_rb<2, 2, false, false, C16_POOL>(a.q5, S0, S2, C16Epi{0.f, nullptr, nullptr, 0, nullptr, x8}, amax);
    // resblock_q6 on the 8x8 map (conv_misc.hip: conv_direct8_kernel - fp32 FMA chains: taps row-major, channels ascending, shortcut last);
    // one (pixel, cout) chain per thread
    float *t8 = F + 2048, *y8 = t8 + 1024;          // [64][16] each, channels 8..15 zero
    float *wd0 = F + 4096, *wd2 = wd0 + 9 * 32 * 8, *wds = wd2 + 9 * 8 * 8;     // the block's three weight tensors: 12.5 KB
    for (int i = tid; i < 9 * 32 * 8; i += 512) wd0[i] = a.d_w0[i];
    for (int i = tid; i < 9 * 8 * 8; i += 512) wd2[i] = a.d_w2[i];
    if (tid < 32 * 8) wds[tid] = a.d_wsc[tid];
    __syncthreads();
    {
        const int co = tid & 7, px = tid >> 3, x = px & 7, y = px >> 3;
        float v = 0.f;
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            if (yy < 0 || yy >= 8) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x + dx - 1;
                if (xx < 0 || xx >= 8) continue;
                const float *wp = wd0 + ((dy * 3 + dx) * 32) * 8 + co;
                const float *xp = x8 + (yy * 8 + xx) * 16;
#pragma unroll
                for (int ci = 0; ci < 32; ++ci) v = fmaf(xp[(ci >> 4) * 1024 + (ci & 15)], wp[ci * 8], v);
            }
        }
        t8[px * 16 + co] = fmaxf(v, 0.f);
        t8[px * 16 + 8 + co] = 0.f;
        __syncthreads();
        v = 0.f;
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            if (yy < 0 || yy >= 8) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x + dx - 1;
                if (xx < 0 || xx >= 8) continue;
                const float *wp = wd2 + ((dy * 3 + dx) * 8) * 8 + co;
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) v = fmaf(t8[(yy * 8 + xx) * 16 + ci], wp[ci * 8], v);
            }
        }
#pragma unroll
        for (int ci = 0; ci < 32; ++ci) v = fmaf(x8[(ci >> 4) * 1024 + px * 16 + (ci & 15)], wds[ci * 8 + co], v);
        y8[px * 16 + co] = fmaxf(v, 0.f);
        __syncthreads();
    }
    if (tid < 64) {
        float acc0, acc1;
        c16_head<8>(y8, a.head_w, a.head_b, 1, tid, acc0, acc1);
        a.qt[(size_t)n * 64 + tid] = acc0;
    }
    sat_report(a.sat, amax);
}

// ===================================================================================================== host side
hipError_t launch_msbd_branch16(hipStream_t s, const Chain16MsbdArgs &h)
{
    if (h.N <= 0) return hipErrorInvalidValue;
    MsbdBranch16Args a{};
    a.x5 = h.x5; a.x5_stride = h.x5_stride; a.qt = h.qt; a.bt = h.bt; a.dire = h.dire; a.sat = h.sat;
    for (int i = 0; i < 3; ++i) {
        a.b1[i] = C16RB{h.b1[i].w0, h.b1[i].w2, h.b1[i].wsc, h.b1[i].s0, h.b1[i].s2};
        a.b2[i] = C16RB{h.b2[i].w0, h.b2[i].w2, h.b2[i].wsc, h.b2[i].s0, h.b2[i].s2};
    }
    for (int i = 0; i < 2; ++i) {
        a.att[i] = C16RB{h.att[i].w0, h.att[i].w2, h.att[i].wsc, h.att[i].s0, h.att[i].s2};
        a.head_w[i] = h.head_w[i]; a.head_b[i] = h.head_b[i];
    }
    hipLaunchKernelGGL(msbd_branch16_kernel, dim3(h.N), dim3(512), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_qt_tail16(hipStream_t s, const Chain16QtArgs &h)
{
    if (h.N <= 0) return hipErrorInvalidValue;
    QtTail16Args a{};
    a.x4 = h.x4; a.x4_stride = h.x4_stride; a.qt = h.qt; a.sat = h.sat;
    a.q3 = C16RB{h.q3.w0, h.q3.w2, h.q3.wsc, h.q3.s0, h.q3.s2};
    a.q4 = C16RB{h.q4.w0, h.q4.w2, h.q4.wsc, h.q4.s0, h.q4.s2};
    a.q5 = C16RB{h.q5.w0, h.q5.w2, h.q5.wsc, h.q5.s0, h.q5.s2};
    a.d_w0 = h.d_w0; a.d_w2 = h.d_w2; a.d_wsc = h.d_wsc; a.head_w = h.head_w; a.head_b = h.head_b;
    hipLaunchKernelGGL(qt_tail16_kernel, dim3(h.N), dim3(512), 0, s, a);
    return hipGetLastError();
}

}  // namespace pmp
