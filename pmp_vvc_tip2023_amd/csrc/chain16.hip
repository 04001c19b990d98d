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
#include "chain16_dev.h"

namespace pmp {


// ===================================================================================================== MTT nets: B1, Att1, B2
struct MsbdBranch16Args {
    const unsigned short *x5; size_t x5_stride;     // [N][4][16][16][16] split-2: trunk_M2's pooled output (and attention 1's gate operand)
    const float *qt;                                // raw QT logits [N][64]
    float *bt, *dire;                               // [N][3][256]: layers 0 and 1 are written here
    C16RB b1[3], att[2], b2[3];
    const float *head_w[2], *head_b[2];
    unsigned *sat;
    float att_scale;
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
    C16Pass<9, 2, 4> pb0;            // first pass of a branch trunk: started in front of whatever precedes it
    c16_wstart(pb0, a.b1[0].w0);
    c16_clear_slots(slots);
    __syncthreads();
    c16_load_image(S0, x5, a.x5_stride, 4);
    __syncthreads();

    auto branch = [&](const C16RB r0, const C16RB r1, const C16RB r2, const float *hw, const float *hb, int layer) __attribute__((always_inline)) {
        // trunk_B: 64 ch in S0..3 -> 32 ch (S0..1) -> 16 ch (S2) -> 8 ch fp32 (f0) -> head
        C16Pass<9, 1, 2> pb1;
        C16Pass<9, 1, 1> pb2;
        amax = c16_rb<2, 4, true, false, C16_IMG>(r0, S0, S4, C16Epi{0.f, nullptr, nullptr, 0, S0, nullptr}, amax, pb0, [&]() __attribute__((always_inline)) { c16_wstart(pb1, r1.w0); });
        amax = c16_rb<1, 2, true, false, C16_IMG>(r1, S0, S4, C16Epi{0.f, nullptr, nullptr, 0, S2, nullptr}, amax, pb1, [&]() __attribute__((always_inline)) { c16_wstart(pb2, r2.w0); });
        amax = c16_rb<1, 1, true, false, C16_F32>(r2, S2, S4, C16Epi{0.f, nullptr, nullptr, 0, nullptr, f0}, amax, pb2, []() {});
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
        f32x4 v = {a.qt[(size_t)n * 64 + (y >> 1) * 8 + (x >> 1)], s_bt[tid], s_dire[tid], 0.f};
        v *= a.att_scale;      // the attention segment's activation scale (a power of two: exact)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        amax = c16_store_split(S0, 0, tid, 0, v, amax);
        c16_store_split(S0, 0, tid, 4, z, 0.f);
        c16_store_split(S0, 0, tid, 8, z, 0.f);
        c16_store_split(S0, 0, tid, 12, z, 0.f);
    }
    __syncthreads();
    // trunk_Att1: 3 -> 32 (S4..5), 32 -> 64 gated by x5 (S0..3)
    C16Pass<9, 2, 1> pa0;
    C16Pass<9, 4, 2> pa1;
    c16_wstart(pa0, a.att[0].w0);
    amax = c16_rb<2, 1, true, false, C16_IMG>(a.att[0], S0, S4, C16Epi{0.f, nullptr, nullptr, 0, S4, nullptr}, amax, pa0, [&]() __attribute__((always_inline)) { c16_wstart(pa1, a.att[1].w0); });
    amax = c16_rb<4, 2, true, true, C16_IMG>(a.att[1], S4, S0, C16Epi{0.f, nullptr, x5, a.x5_stride, S0, nullptr}, amax, pa1, [&]() __attribute__((always_inline)) { c16_wstart(pb0, a.b2[0].w0); });
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
    C16Pass<9, 2, 4> pq3;
    c16_wstart(pq3, a.q3.w0);
    c16_clear_slots(lds);
    __syncthreads();
    c16_load_image(S0, a.x4 + (size_t)n * 4 * 4096, a.x4_stride, 4);
    __syncthreads();
    // resblock_q3: 64 -> 32, output fp32 (the multi-scale pool reads it).  Its intermediate uses S4..5, which then becomes the fp32 area.
    amax = c16_rb<2, 4, true, false, C16_F32>(a.q3, S0, S4, C16Epi{0.f, nullptr, nullptr, 0, nullptr, x5}, amax, pq3, []() {});
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
        C16Pass<9, 2, 2> pp;
        c16_wstart(pp, a.q4.w0 + (size_t)p * 9 * (2 * 2 * 64 * 8));
        x6_pair(p);
        __syncthreads();
        c16_accumulate<9, 2, 2>(S0, a.q4.w0 + (size_t)p * 9 * (2 * 2 * 64 * 8), acc, pp);
        __syncthreads();
    }
    C16Pass<9, 2, 2> pq4;
    c16_wstart(pq4, a.q4.w2);
    amax = c16_epilogue<2, false, false, C16_IMG>(acc, C16Epi{a.q4.s0, nullptr, nullptr, 0, S2, nullptr}, amax);
    __syncthreads();
    c16_zero<2>(acc);
    c16_accumulate<9, 2, 2>(S2, a.q4.w2, acc, pq4);
    for (int p = 0; p < 4; ++p) {
        C16Pass<1, 2, 2> pp;
        c16_wstart(pp, a.q4.wsc + (size_t)p * (2 * 2 * 64 * 8));
        x6_pair(p);
        __syncthreads();
        c16_accumulate<1, 2, 2>(S0, a.q4.wsc + (size_t)p * (2 * 2 * 64 * 8), acc, pp);
        __syncthreads();
    }
    C16Pass<9, 2, 2> pq5;
    c16_wstart(pq5, a.q5.w0);
    amax = c16_epilogue<2, false, false, C16_IMG>(acc, C16Epi{a.q4.s2, nullptr, nullptr, 0, S0, nullptr}, amax);   // x7 -> S0..1 (every wave is past the last barrier)
    __syncthreads();
    // resblock_q5: 32 -> 32, identity shortcut, max_pool2d(2) -> fp32 [2][64][16] over the dead x5
    float *x8 = F;
    amax = c16_rb<2, 2, false, false, C16_POOL>(a.q5, S0, S2, C16Epi{0.f, nullptr, nullptr, 0, nullptr, x8}, amax, pq5, []() {});
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
    a.x5 = h.x5; a.x5_stride = h.x5_stride; a.qt = h.qt; a.bt = h.bt; a.dire = h.dire; a.sat = h.sat; a.att_scale = h.att_scale;
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
