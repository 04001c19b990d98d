// chain16.hip — the 16x16-resolution tails of the four nets with every activation of a ResidualBlock resident in LDS (f16x3 datapath),
// TWO blocks per CU (round 6; building blocks and the memory plan: tail16_dev.h).  At 16x16 a block IS a tile, so the chains below - until
// round 3 about thirty launches per pass that existed only to move <= 65 KB per block through HBM between tiny layers - need no halo
// recompute, and only the two 64-channel tensors that do not fit beside their consumers make a round trip through L2:
//
//   MTT nets (Model_QBD.py:116-124, :138-146 / :214-223, :236-244):
//       branch16_kernel   x5 -> trunk_B1 (RB 64->32, 32->16, 16->8) -> conv_B1 -> out0
//       att16_kernel      cat[up2(q), out0] -> trunk_Att1 (RB 3->32, 32->64) * x5 -> xb1 (global, 64 KB per block)
//       branch16_kernel   xb1 -> trunk_B2 -> conv_B2 -> out1, out1[:,0] += out0[:,0]
//   QT nets (Model_QBD.py:71-76, :83-91 / :169-174, :181-189):
//       q3_rb64_kernel    x4 -> resblock_q3 -> x5 (global, fp32, 32 KB per block: the multi-scale pool reads fp32)
//       qt_rest16_kernel  cat[x5, up2(mp2), up4(mp4), up8(mp8)] -> resblock_q4 -> resblock_q5, max_pool2d(2) -> resblock_q6 -> conv_q2
//
// Round 4's form did each net's tail in ONE kernel (msbd_branch16_kernel, qt_tail16_kernel: 143 KB of LDS, 168-230 VGPRs, eight waves: one
// workgroup per CU, matrix pipes 0.46 / 0.32 busy behind 24 dependent passes).  Here a workgroup is four waves with <= 80 KB of LDS, two
// per CU: one block's barriers, epilogues and heads run beside the other's MFMAs.
//
// BIT-IDENTICAL to the launch-per-layer path (nets.cpp with fusion off; tests/test_gpu_parity.py): the convolutions consume the same
// packed weight streams (pack_h2, pack.cpp) in the same K-step order with the same three products per K-step (x0*w1, x0*w0, x1*w0) into
// the same fp32 accumulators, the epilogues apply the same operations in the same order (1/S, residual or 1x1 shortcut pass, ReLU, gate,
// 2x2 pool, two-term fp16 split with the range clamp), tensors that the launch path hands to its fp32 kernels stay fp32 here too, and the
// fp32 kernels (heads, 8x8 direct convolutions, multi-scale pool) are restated with their accumulation order.
#include "tail16_dev.h"

namespace pmp {

// ===================================================================================================== RB(64 -> 32) -> fp32 (resblock_q3)
struct Q3Args {
    const unsigned short *x4; size_t x4_stride;     // [N][4][16][16][16] split-2: resblock_q2's pooled output
    float *x5;                                      // [N][2][256][16] fp32
    T16RB q3;
    unsigned *sat;
};

__global__ __launch_bounds__(T16_THREADS, 2) void q3_rb64_kernel(Q3Args a)
{
    __shared__ __attribute__((aligned(16))) char slots[T16_NSLOT * T16_SLOT];
    typedef T16Tile<2> WT;
    const int n = blockIdx.x;
    const unsigned short *x = a.x4 + (size_t)n * 4 * 4096;
    T16Pass<9, 2> p1;
    t16_wstart(p1, a.q3.w0, 2, WT::ct());
    T16Fetch f01, f23;
    t16_fetch(f01, x, a.x4_stride, 0);
    t16_fetch(f23, x, a.x4_stride, 2);
    t16_clear_borders(slots, T16_NSLOT);
    __syncthreads();
    f32x4 acc[8];
    const float amax = t16_rb64(a.q3, x, a.x4_stride, slots, f01, f23, acc, 0.f, p1, []() {});
    t16_epilogue<8, false, false, T16_F32>(acc, WT::row0(), T16Epi{a.q3.s2, nullptr, nullptr, 0, nullptr, a.x5 + ((size_t)n * 2 + WT::ct()) * 4096, nullptr, 0, 16}, 0.f);
    sat_report(a.sat, amax);                        // the block's intermediate is split; its fp32 output is not clamped
}

// ===================================================================================================== QT nets: x6 .. conv_q2
struct QtRest16Args {
    const float *x5;                                // [N][2][256][16] fp32: resblock_q3's output
    float *qt;                                      // [N][64]
    T16RB q4, q5;
    const float *d_w0, *d_w2, *d_wsc;               // resblock_q6 on the 8x8 map: plain fp32 [tap][cin][8], [tap][8][8], [32][8] (conv_direct8_kernel)
    const float *head_w, *head_b;
    unsigned *sat;
};

__global__ __launch_bounds__(T16_THREADS, 2) void qt_rest16_kernel(QtRest16Args a)
{
    __shared__ __attribute__((aligned(16))) char slots[T16_NSLOT * T16_SLOT];
    typedef T16Tile<2> WT;
    const int n = blockIdx.x, tid = threadIdx.x, ct = WT::ct(), row0 = WT::row0();
    char *A = slots, *C = slots + 2 * T16_SLOT;
    const float *x5 = a.x5 + (size_t)n * 2 * 4096;
    float *p2 = reinterpret_cast<float *>(C), *p4 = p2 + 2 * 1024, *p8 = p4 + 2 * 256;     // [2][64][16], [2][16][16], [2][4][16]: 10.5 KB of slot C
    float amax = 0.f;
    T16Pass<9, 2> pp0;
    t16_wstart(pp0, a.q4.w0, 2, ct);
    f32x4 v0[8];                                    // pair 0 of x6 = x5 itself: its values for the window, requested before the pooling reads the same lines
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + k * T16_THREADS, c4 = (i & 3) * 4, px = (i >> 2) & 255, cb = i >> 10;
        v0[k] = *reinterpret_cast<const f32x4 *>(x5 + cb * 4096 + px * 16 + c4);
    }
    t16_clear_borders(slots, 2);                    // the window A, B; C, D become images only as resblock_q5's intermediate
    // multi-scale pool (conv_misc.hip: multipool_concat_kernel), both groups
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + k * T16_THREADS;
        const int cb = i >> 10, j = i & 1023, c = j & 15, x = (j >> 4) & 7, y = j >> 7;
        const float *q = x5 + cb * 4096 + ((2 * y) * 16 + 2 * x) * 16 + c;
        p2[i] = fmaxf(fmaxf(q[0], q[16]), fmaxf(q[256], q[272]));
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + k * T16_THREADS;
        const int cb = i >> 8, j = i & 255, c = j & 15, x = (j >> 4) & 3, y = j >> 6;
        const float *q = p2 + cb * 1024 + ((2 * y) * 8 + 2 * x) * 16 + c;
        p4[i] = fmaxf(fmaxf(q[0], q[16]), fmaxf(q[128], q[144]));
    }
    __syncthreads();
    if (tid < 128) {
        const int cb = tid >> 6, j = tid & 63, c = j & 15, x = (j >> 4) & 1, y = j >> 5;
        const float *q = p4 + cb * 256 + ((2 * y) * 4 + 2 * x) * 16 + c;
        p8[tid] = fmaxf(fmaxf(q[0], q[16]), fmaxf(q[64], q[80]));
    }
    __syncthreads();
    // x6 = cat[x5, up2(mp2), up4(mp4), up8(mp8)], 128 channels: never materialised as a whole.  16 floats of pixel (y, x) of group cb of
    // channel-group pair p (= source p):
    auto x6_src = [&](int p, int cb, int y, int x) __attribute__((always_inline)) -> const float * {
        return p == 0 ? x5 + cb * 4096 + (y * 16 + x) * 16
             : p == 1 ? p2 + cb * 1024 + ((y >> 1) * 8 + (x >> 1)) * 16
             : p == 2 ? p4 + cb * 256 + ((y >> 2) * 4 + (x >> 2)) * 16
                      : p8 + cb * 64 + ((y >> 3) * 2 + (x >> 3)) * 16;
    };
    // ... pair p as halo images in the window A, B, for the 3x3 convolution
    auto x6_pair = [&](int p) __attribute__((always_inline)) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {               // (group, pixel, quad of channels)
            const int i = tid + k * T16_THREADS, c4 = (i & 3) * 4, px = (i >> 2) & 255, cb = i >> 10;
            v[k] = p == 0 ? v0[k] : *reinterpret_cast<const f32x4 *>(x6_src(p, cb, px >> 4, px & 15) + c4);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = tid + k * T16_THREADS, c4 = (i & 3) * 4, px = (i >> 2) & 255, cb = i >> 10;
            amax = t16_store_split(A + cb * T16_SLOT, px, c4, v[k], amax);
        }
    };
    f32x4 acc[8];
    // resblock_q4: 128 -> 32.  First conv: 8 groups = 4 pairs of 9 K-steps; second conv: 32 -> 32, then the 1x1 shortcut over x6 again
    t16_zero<8>(acc);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        T16Pass<9, 2> pp;
        if (p > 0) t16_wstart(pp, a.q4.w0 + (size_t)p * 9 * 2 * T16_KSTEP, 2, ct);
        x6_pair(p);
        __syncthreads();
        t16_accumulate<9, 2, 8>(A, row0, acc, p == 0 ? pp0 : pp);
        __syncthreads();
    }
    T16Pass<9, 2> pq4;
    t16_wstart(pq4, a.q4.w2, 2, ct);
    amax = t16_epilogue<8, false, false, T16_IMG>(acc, row0, T16Epi{a.q4.s0, nullptr, nullptr, 0, A + ct * T16_SLOT, nullptr, nullptr, 0, 16}, amax);   // the window is free: the intermediate takes it
    __syncthreads();
    t16_zero<8>(acc);
    t16_accumulate<9, 2, 8>(A, row0, acc, pq4);
    // the shortcut's B fragments are split in registers from the fp32 sources (no image, no barrier): K-step p = pair p
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        T16Pass<1, 2> pp;
        t16_wstart(pp, a.q4.wsc + (size_t)p * 2 * T16_KSTEP, 2, ct);
        t16_accumulate_1x1_f32<8>([&](int cb, int y, int x) __attribute__((always_inline)) { return x6_src(p, cb, y, x); }, row0, acc, pp);
    }
    T16Pass<9, 2> pq5;
    t16_wstart(pq5, a.q5.w0, 2, ct);
    __syncthreads();                                // every wave is done with the intermediate (A, B) and the pools (C)
    t16_clear_borders(C, 2);                        // C, D become resblock_q5's intermediate
    amax = t16_epilogue<8, false, false, T16_IMG>(acc, row0, T16Epi{a.q4.s2, nullptr, nullptr, 0, A + ct * T16_SLOT, nullptr, nullptr, 0, 16}, amax);   // x7 -> A, B
    __syncthreads();
    // resblock_q6's and the head's weights (12.8 KB of fp32, the same for every block): requested here, parked in LDS behind resblock_q5
    float wr0[9], wr2[3], wrs, wrh;
#pragma unroll
    for (int k = 0; k < 9; ++k) wr0[k] = a.d_w0[tid + k * T16_THREADS];
#pragma unroll
    for (int k = 0; k < 3; ++k) wr2[k] = a.d_w2[min(tid + k * T16_THREADS, 9 * 8 * 8 - 1)];
    wrs = a.d_wsc[tid];
    wrh = tid < 72 ? a.head_w[tid] : a.head_b[0];
    // resblock_q5: 32 -> 32, identity shortcut, max_pool2d(2) -> fp32 [2][64][16]
    amax = t16_rb<2, 2, false>(a.q5, A, C, acc, amax, pq5, []() {});
    float *x8 = reinterpret_cast<float *>(C);       // over the dead intermediate
    t16_epilogue<8, true, false, T16_POOL>(acc, row0, T16Epi{a.q5.s2, A + ct * T16_SLOT, nullptr, 0, nullptr, x8 + ct * 1024, nullptr, 0, 16}, 0.f);
    // resblock_q6 on the 8x8 map (conv_misc.hip: conv_direct8_kernel - fp32 FMA chains: taps row-major, channels ascending, shortcut last).
    // Its weights go to LDS TRANSPOSED - [tap][cout][cin] with a pitch that spreads the eight couts over the banks - so that a chain's 32
    // (8) input channels of a tap are 16-byte reads; a thread runs the chains of pixels px and px + 32 of one cout side by side (same weights).
    float *D_ = reinterpret_cast<float *>(slots + 3 * T16_SLOT);       // D is dead (resblock_q5's intermediate); A, B are still the residual
    float *wd0 = D_, *wd2 = wd0 + 9 * 8 * 36, *wds = wd2 + 9 * 8 * 12, *hw = wds + 8 * 36;     // 2592 + 864 + 288 + 73 floats = 15.3 KB
#pragma unroll
    for (int k = 0; k < 9; ++k) { const int i = tid + k * T16_THREADS; wd0[((i >> 8) * 8 + (i & 7)) * 36 + ((i >> 3) & 31)] = wr0[k]; }      // [tap][ci][co] -> [tap][co][36]
#pragma unroll
    for (int k = 0; k < 3; ++k) { const int i = tid + k * T16_THREADS; if (i < 9 * 8 * 8) wd2[((i >> 6) * 8 + (i & 7)) * 12 + ((i >> 3) & 7)] = wr2[k]; }   // [tap][ci][co] -> [tap][co][12]
    wds[(tid & 7) * 36 + (tid >> 3)] = wrs;                                                                                    // [ci][co] -> [co][36]
    if (tid < 73) hw[tid] = wrh;
    __syncthreads();
    float *t8 = reinterpret_cast<float *>(A), *y8 = t8 + 1024;     // [64][16] each, channels 8..15 zero (A: the residual is dead now)
    {
        // (branch-free: a tap outside the map runs on a clamped pixel and is dropped by a select, so that the LDS reads of a tap can be
        // issued together - with `continue` in the loops every 16-byte read waited for its own latency: 23 k of a block's 91 k cycles)
        const int co = tid & 7, px0 = tid >> 3, x = px0 & 7, y0 = px0 >> 3;      // chains: pixel px0 (rows 0..3) and px0 + 32 (rows 4..7)
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ya = y0 + dy - 1, yb = ya + 4;                             // -1 .. 4 and 3 .. 8
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x + dx - 1;
                const bool okx = xx >= 0 && xx < 8, oka = okx && ya >= 0, okb = okx && yb < 8;
                const float *wp = wd0 + ((dy * 3 + dx) * 8 + co) * 36;
                const float *xa = x8 + ((ya >= 0 ? ya : 0) * 8 + (okx ? xx : x)) * 16, *xb = x8 + ((yb < 8 ? yb : 7) * 8 + (okx ? xx : x)) * 16;
                float t0 = v0, t1 = v1;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wp + q * 4);
                    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(xa + (q >> 2) * 1024 + (q & 3) * 4);
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(xb + (q >> 2) * 1024 + (q & 3) * 4);
                    t0 = fmaf(a4.x, w4.x, t0); t0 = fmaf(a4.y, w4.y, t0); t0 = fmaf(a4.z, w4.z, t0); t0 = fmaf(a4.w, w4.w, t0);
                    t1 = fmaf(b4.x, w4.x, t1); t1 = fmaf(b4.y, w4.y, t1); t1 = fmaf(b4.z, w4.z, t1); t1 = fmaf(b4.w, w4.w, t1);
                }
                v0 = oka ? t0 : v0;
                v1 = okb ? t1 : v1;
            }
        }
        t8[px0 * 16 + co] = fmaxf(v0, 0.f);
        t8[px0 * 16 + 8 + co] = 0.f;
        t8[(px0 + 32) * 16 + co] = fmaxf(v1, 0.f);
        t8[(px0 + 32) * 16 + 8 + co] = 0.f;
        __syncthreads();
        v0 = 0.f; v1 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ya = y0 + dy - 1, yb = ya + 4;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = x + dx - 1;
                const bool okx = xx >= 0 && xx < 8, oka = okx && ya >= 0, okb = okx && yb < 8;
                const float *wp = wd2 + ((dy * 3 + dx) * 8 + co) * 12;
                const float *xa = t8 + ((ya >= 0 ? ya : 0) * 8 + (okx ? xx : x)) * 16, *xb = t8 + ((yb < 8 ? yb : 7) * 8 + (okx ? xx : x)) * 16;
                float t0 = v0, t1 = v1;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wp + q * 4);
                    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(xa + q * 4), b4 = *reinterpret_cast<const f32x4 *>(xb + q * 4);
                    t0 = fmaf(a4.x, w4.x, t0); t0 = fmaf(a4.y, w4.y, t0); t0 = fmaf(a4.z, w4.z, t0); t0 = fmaf(a4.w, w4.w, t0);
                    t1 = fmaf(b4.x, w4.x, t1); t1 = fmaf(b4.y, w4.y, t1); t1 = fmaf(b4.z, w4.z, t1); t1 = fmaf(b4.w, w4.w, t1);
                }
                v0 = oka ? t0 : v0;
                v1 = okb ? t1 : v1;
            }
        }
        const float *wp = wds + co * 36;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wp + q * 4);
            const f32x4 xa = *reinterpret_cast<const f32x4 *>(x8 + (q >> 2) * 1024 + px0 * 16 + (q & 3) * 4);
            const f32x4 xb = *reinterpret_cast<const f32x4 *>(x8 + (q >> 2) * 1024 + (px0 + 32) * 16 + (q & 3) * 4);
            v0 = fmaf(xa.x, w4.x, v0); v0 = fmaf(xa.y, w4.y, v0); v0 = fmaf(xa.z, w4.z, v0); v0 = fmaf(xa.w, w4.w, v0);
            v1 = fmaf(xb.x, w4.x, v1); v1 = fmaf(xb.y, w4.y, v1); v1 = fmaf(xb.z, w4.z, v1); v1 = fmaf(xb.w, w4.w, v1);
        }
        y8[px0 * 16 + co] = fmaxf(v0, 0.f);
        y8[(px0 + 32) * 16 + co] = fmaxf(v1, 0.f);
    }
    __syncthreads();
    if (tid < 64) {
        float acc0, acc1;
        t16_head<8, 16>(y8, hw, hw + 72, 1, tid, acc0, acc1);
        a.qt[(size_t)n * 64 + tid] = acc0;
    }
    sat_report(a.sat, amax);
}

// ===================================================================================================== MTT nets: trunk_B + head
struct Branch16Args {
    const unsigned short *x; size_t x_stride;       // [N][4][16][16][16] split-2: x5 (layer 0) or x5 * att0 (layer 1)
    float *bt, *dire;                               // [N][3][256]: plane `layer` is written; layer 1 adds plane 0 of bt (Model_QBD.py:146)
    T16RB r[3];
    const float *head_w, *head_b;
    unsigned *sat;
    int layer;
};

__global__ __launch_bounds__(T16_THREADS, 2) void branch16_kernel(Branch16Args a)
{
    __shared__ __attribute__((aligned(16))) char slots[T16_NSLOT * T16_SLOT];
    typedef T16Tile<2> WT;
    const int n = blockIdx.x, tid = threadIdx.x;
    char *A = slots, *B = slots + T16_SLOT, *C = slots + 2 * T16_SLOT, *D = slots + 3 * T16_SLOT;
    const unsigned short *x = a.x + (size_t)n * 4 * 4096;
    float amax = 0.f;
    T16Pass<9, 2> pb0;
    t16_wstart(pb0, a.r[0].w0, 2, WT::ct());
    T16Fetch f01, f23;
    t16_fetch(f01, x, a.x_stride, 0);
    t16_fetch(f23, x, a.x_stride, 2);
    const float hreg = tid < 144 ? a.head_w[tid] : tid < 146 ? a.head_b[tid - 144] : 0.f;     // the head's weights, on their way since the start
    const float prev = a.layer > 0 ? a.bt[((size_t)n * 3 + a.layer - 1) * 256 + tid] : 0.f;   // out1[:, 0] += out0[:, 0]  (Model_QBD.py:146)
    t16_clear_borders(slots, T16_NSLOT);
    __syncthreads();
    // trunk_B.0: 64 ch (global, through the slots) -> 32 ch (A, B)
    T16Pass<9, 2> pb1;
    {
        f32x4 acc[8];
        amax = t16_rb64(a.r[0], x, a.x_stride, slots, f01, f23, acc, amax, pb0, [&]() __attribute__((always_inline)) { t16_wstart(pb1, a.r[1].w0, 1, 0); });
        amax = t16_epilogue<8, false, false, T16_IMG>(acc, WT::row0(), T16Epi{a.r[0].s2, nullptr, nullptr, 0, A + WT::ct() * T16_SLOT, nullptr, nullptr, 0, 16}, amax);
        __syncthreads();
    }
    // trunk_B.1: 32 -> 16 (D), trunk_B.2: 16 -> 8 as fp32 over A (pixel pitch 20 floats: exactly the slot), then the head
    const int row1 = T16Tile<1>::row0();
    float *f0 = reinterpret_cast<float *>(A), *hw = reinterpret_cast<float *>(B);
    T16Pass<9, 1> pb2;
    f32x4 acc[4];
    amax = t16_rb<1, 2, true>(a.r[1], A, C, acc, amax, pb1, [&]() __attribute__((always_inline)) { t16_wstart(pb2, a.r[2].w0, 1, 0); });
    amax = t16_epilogue<4, false, false, T16_IMG>(acc, row1, T16Epi{a.r[1].s2, nullptr, nullptr, 0, D, nullptr, nullptr, 0, 16}, amax);
    __syncthreads();
    if (tid < 146) hw[tid] = hreg;                  // B is dead since trunk_B.1's last barrier
    amax = t16_rb<1, 1, true>(a.r[2], D, C, acc, amax, pb2, []() {});
    t16_epilogue<4, false, false, T16_F32>(acc, row1, T16Epi{a.r[2].s2, nullptr, nullptr, 0, nullptr, f0, nullptr, 0, 20}, 0.f);
    __syncthreads();
    {
        float acc0, acc1;
        t16_head<16, 20>(f0, hw, hw + 144, 2, tid, acc0, acc1);
        const size_t o = ((size_t)n * 3 + a.layer) * 256 + tid;
        if (a.layer > 0) acc0 += prev;
        a.bt[o] = acc0;
        a.dire[o] = acc1;
    }
    sat_report(a.sat, amax);
}

// ===================================================================================================== MTT nets: attention 1
struct Att16Args {
    const float *qt, *bt, *dire;                    // raw QT logits [N][64]; out0 = plane 0 of [N][3][256]
    const unsigned short *x5; size_t x5_stride;     // the gate operand [N][4][16][16][16] split-2
    unsigned short *xb; size_t xb_stride;           // out: x5 * att0, same layout
    T16RB att[2];
    unsigned *sat;
    float att_scale;
};

__global__ __launch_bounds__(T16_THREADS, 2) void att16_kernel(Att16Args a)
{
    __shared__ __attribute__((aligned(16))) char slots[T16_NSLOT * T16_SLOT];
    typedef T16Tile<2> W2;
    const int n = blockIdx.x, tid = threadIdx.x, wv = tid >> 6;
    char *A = slots, *B = slots + T16_SLOT, *C = slots + 2 * T16_SLOT;
    float amax = 0.f;
    T16Pass<9, 1> pa0;
    t16_wstart(pa0, a.att[0].w0, 2, W2::ct());
    // attention input cat[up2(q), out0] (conv_misc.hip: att_input_kernel, S = 16) -> B, channels 3..15 zero
    f32x4 v;
    {
        const int x = tid & 15, y = tid >> 4;
        v = (f32x4){a.qt[(size_t)n * 64 + (y >> 1) * 8 + (x >> 1)], a.bt[(size_t)n * 768 + tid], a.dire[(size_t)n * 768 + tid], 0.f};
    }
    t16_clear(B, T16_SLOT);                         // the whole slot: 13 of its 16 channels are never written
    t16_clear_borders(A, 1);
    t16_clear_borders(C, 2);
    __syncthreads();
    v *= a.att_scale;          // the attention segment's activation scale (a power of two: exact)
    amax = t16_store_split(B, tid, 0, v, amax);
    __syncthreads();
    // trunk_Att1.0: 3 -> 32, intermediate in C, D, output over A, B
    T16Pass<9, 2> pm;
    {
        f32x4 acc[8];
        amax = t16_rb<2, 1, true>(a.att[0], B, C, acc, amax, pa0, [&]() __attribute__((always_inline)) { t16_wstart(pm, a.att[1].w0, 4, W2::ct()); });
        amax = t16_epilogue<8, false, false, T16_IMG>(acc, W2::row0(), T16Epi{a.att[0].s2, nullptr, nullptr, 0, A + W2::ct() * T16_SLOT, nullptr, nullptr, 0, 16}, amax);
        __syncthreads();
    }
    // trunk_Att1.1: 32 (A, B) -> 64, gated by x5.  Its 64-channel intermediate is produced and consumed pair by pair through C, D: the second
    // convolution's K-steps 0..8 read groups 0 and 1 only, 9..17 groups 2 and 3; every wave owns one output group x all 16 rows.
    f32x4 acc2[16];
    t16_zero<16>(acc2);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 accm[8];
        t16_zero<8>(accm);
        t16_accumulate<9, 2, 8>(A, W2::row0(), accm, pm);
        T16Pass<9, 2> p2;
        t16_wstart(p2, a.att[1].w2 + (size_t)h * 9 * 4 * T16_KSTEP, 4, wv);
        amax = t16_epilogue<8, false, false, T16_IMG>(accm, W2::row0(), T16Epi{a.att[1].s0, nullptr, nullptr, 0, C + W2::ct() * T16_SLOT, nullptr, nullptr, 0, 16}, amax);
        __syncthreads();
        t16_accumulate<9, 2, 16>(C, 0, acc2, p2);
        if (h == 0) {
            t16_wstart(pm, a.att[1].w0, 4, 2 + W2::ct());
            __syncthreads();                           // before the next pair lands in C, D
        }
    }
    {
        T16Pass<1, 2> p3;
        t16_wstart(p3, a.att[1].wsc, 4, wv);
        t16_accumulate<1, 2, 16>(A, 0, acc2, p3);
    }
    const size_t go = ((size_t)n * 4 + wv) * 4096;
    amax = t16_epilogue<16, false, true, T16_GLB>(acc2, 0, T16Epi{a.att[1].s2, nullptr, a.x5 + go, a.x5_stride, nullptr, nullptr, a.xb + go, a.xb_stride, 16}, amax);
    sat_report(a.sat, amax);
}

// ===================================================================================================== host side
static T16RB rbw(const Chain16RB &r) { return T16RB{r.w0, r.w2, r.wsc, r.s0, r.s2}; }

hipError_t launch_msbd_branch16(hipStream_t s, const Chain16MsbdArgs &h)
{
    if (h.N <= 0 || !h.xb) return hipErrorInvalidValue;
    Branch16Args b{};
    b.x = h.x5; b.x_stride = h.x5_stride; b.bt = h.bt; b.dire = h.dire; b.sat = h.sat; b.layer = 0;
    for (int i = 0; i < 3; ++i) b.r[i] = rbw(h.b1[i]);
    b.head_w = h.head_w[0]; b.head_b = h.head_b[0];
    hipLaunchKernelGGL(branch16_kernel, dim3(h.N), dim3(T16_THREADS), 0, s, b);
    Att16Args a{};
    a.qt = h.qt; a.bt = h.bt; a.dire = h.dire; a.x5 = h.x5; a.x5_stride = h.x5_stride; a.xb = h.xb; a.xb_stride = h.xb_stride;
    a.att[0] = rbw(h.att[0]); a.att[1] = rbw(h.att[1]); a.sat = h.sat; a.att_scale = h.att_scale;
    hipLaunchKernelGGL(att16_kernel, dim3(h.N), dim3(T16_THREADS), 0, s, a);
    b.x = h.xb; b.x_stride = h.xb_stride; b.layer = 1;
    for (int i = 0; i < 3; ++i) b.r[i] = rbw(h.b2[i]);
    b.head_w = h.head_w[1]; b.head_b = h.head_b[1];
    hipLaunchKernelGGL(branch16_kernel, dim3(h.N), dim3(T16_THREADS), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_qt_tail16(hipStream_t s, const Chain16QtArgs &h)
{
    if (h.N <= 0 || !h.x5) return hipErrorInvalidValue;
    Q3Args q{h.x4, h.x4_stride, h.x5, rbw(h.q3), h.sat};
    hipLaunchKernelGGL(q3_rb64_kernel, dim3(h.N), dim3(T16_THREADS), 0, s, q);
    QtRest16Args a{};
    a.x5 = h.x5; a.qt = h.qt; a.sat = h.sat;
    a.q4 = rbw(h.q4); a.q5 = rbw(h.q5);
    a.d_w0 = h.d_w0; a.d_w2 = h.d_w2; a.d_wsc = h.d_wsc; a.head_w = h.head_w; a.head_b = h.head_b;
    hipLaunchKernelGGL(qt_rest16_kernel, dim3(h.N), dim3(T16_THREADS), 0, s, a);
    return hipGetLastError();
}

}  // namespace pmp
