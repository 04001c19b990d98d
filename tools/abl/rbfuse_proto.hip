// abl/rbfuse_proto.hip — MEASUREMENT LIBRARY ONLY (make abl): the fused ResidualBlock(64, 64, 3) that VERDICT r3 asked to have measured
// instead of estimated.  One launch computes  out = relu(conv3x3(relu(conv3x3(x, w1)), w2) + x)  for a 16x16 output tile per
// workgroup with the 64-channel intermediate resident in LDS: 2 tensor passes through HBM (x in, out) instead of the launch pair's 5
// (x, t out, t in, x as residual, out).  EXACT - same weight streams, same K-step order, same epilogue arithmetic as the two launches of
// conv_f16x3.hip, so the result is bit-identical (tools/rbfuse_probe.py checks it) - and built with the timing-only switches every other
// kernel experiment of this repository got (ABL bits below).
//
// Structure (512 threads = 8 waves, ONE workgroup per CU: 134 KB of LDS):
//   phase 1  conv1 on the 18x18 halo of the tile = 324 pixels = 21 MFMA pixel tiles (16 would do without the halo: +31 % MFMAs here, +15.6 %
//            on the block) x 4 output groups.  Input channel groups (20x20 halo images, 25.6 KB) stream through two LDS buffers, staged
//            through registers one group ahead; wave = (cout half, quarter of the pixel tiles): 6 or 5 tiles x 2 groups = 48 accumulators.
//            Epilogue: 1/S, ReLU, zero outside the image (conv2's padding), two-term split -> the 18x18 intermediate image in LDS (83 KB).
//   phase 2  conv2 from that image (no staging, no barrier), chain16_dev.h's pass: wave = (cout group, row half).  Epilogue as h2_epilogue:
//            1/S + identity residual (16-byte loads from global - the tile's input is long gone from LDS), ReLU, split, 16-byte stores.
#include "chain16_dev.h"

namespace pmp {

struct RbFuseArgs {
    const unsigned short *x; size_t x_stride;      // [N][4][H][W][16] split-2: input and identity residual
    const unsigned short *w1, *w2; float s1, s2;   // pack_h2 streams of the two convolutions and their 1/S
    unsigned short *out; size_t out_stride;
    int N, H, W;
    unsigned *sat;
};

namespace {

constexpr int RBF_IPLN = 20 * 20 * 32, RBF_IBUF = 2 * RBF_IPLN;      // input halo image of one channel group: two planes of 400 pixels x 32 B
constexpr int RBF_NLD = 4;                                           // 16-byte pieces per thread and group (1600 pieces, 512 threads)

__device__ __forceinline__ constexpr int rbf_tap20(int t) { return ((t / 3) * 20 + t % 3) * 32; }

// timing-only switches (wrong results): 1 no input staging after group 0, 2 no phase-1 MFMAs, 4 no phase-2 MFMAs, 8 no final epilogue,
// 16 no intermediate write, 32 every tile reads block 0 (L2-resident input)
template <int ABL>
__global__ __launch_bounds__(512, 2) void rb64_fused_kernel(RbFuseArgs a)
{
    __shared__ __attribute__((aligned(16))) char mid[4 * C16_SLOT];
    __shared__ __attribute__((aligned(16))) char inb[2 * RBF_IBUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, xl = lane & 15, g = lane >> 4;
    const int H = a.H, W = a.W, tiles_x = W >> 4, tiles = tiles_x * (H >> 4);
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);      // XCD-aware tile order, as conv_h2_kernel
    const int n0 = bid / tiles, t = bid - n0 * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
    const int n = (ABL & 32) ? 0 : n0;
    const size_t grp_sz = (size_t)H * W * 16;
    const unsigned short *xg = a.x + (size_t)n * 4 * grp_sz;
    float amax = 0.f;

    // ---- staging plan: piece i = [plane][pixel of the 20x20 halo][half]
    unsigned soff[RBF_NLD], sdst[RBF_NLD], svalid = 0;
#pragma unroll
    for (int k = 0; k < RBF_NLD; ++k) {
        const int i = min(tid + k * 512, 1599), sp = i / 800, j = i - sp * 800, pix = j >> 1, half = j & 1;
        const int row = pix / 20, col = pix - row * 20, gy = ty * 16 + row - 2, gx = tx * 16 + col - 2;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W && tid + k * 512 < 1600) svalid |= 1u << k;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
        soff[k] = (unsigned)(sp * a.x_stride + ((size_t)cy * W + cx) * 16 + half * 8);
        sdst[k] = (unsigned)(tid + k * 512 < 1600 ? sp * RBF_IPLN + pix * 32 + half * 16 : 2 * RBF_IBUF);      // (spare: never stored)
    }
    u32x4 sr[RBF_NLD];
    auto stage_load = [&](int cb) __attribute__((always_inline)) {
        const unsigned short *grp = xg + (size_t)cb * grp_sz;
#pragma unroll
        for (int k = 0; k < RBF_NLD; ++k) sr[k] = *reinterpret_cast<const C16_GLOBAL u32x4 *>((const C16_GLOBAL unsigned short *)grp + soff[k]);
    };
    auto stage_store = [&](int buf) __attribute__((always_inline)) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < RBF_NLD; ++k)
            if (k < 3 || tid < 64) *reinterpret_cast<u32x4 *>(inb + buf * RBF_IBUF + sdst[k]) = ((svalid >> k) & 1u) ? sr[k] : z;
    };

    // ---- phase 1
    const int ch = wave & 1, q = wave >> 1, t0 = q == 0 ? 0 : 1 + 5 * q, cnt = q == 0 ? 6 : 5;
    int pb[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int p = min((t0 + i) * 16 + xl, 323);
        pb[i] = ((p / 18) * 20 + p % 18) * 32 + (g & 1) * 16;
    }
    const bool hi = (g >> 1) != 0;
    f32x4 acc1[6][2];
#pragma unroll
    for (int i = 0; i < 6; ++i) { acc1[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[i][1] = acc1[i][0]; }
    {
        constexpr int D = 3;
        const C16_GLOBAL f16x8 *wl = (const C16_GLOBAL f16x8 *)a.w1 + lane + (2 * ch) * 64;
        f16x8 wq[D][2][2];
        auto wload = [&](int st) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) wq[st % D][s][j] = wl[(size_t)st * (2 * 4 * 64) + (s * 4 + j) * 64];
        };
#pragma unroll
        for (int st = 0; st < D; ++st) wload(st);
        stage_load(0);
        stage_store(0);
        __syncthreads();
        int st = 0;
        // one K-step: the K halves of lanes g < 2 / g >= 2 come from (buffer, tap) A / B; sub-steps of three pixel tiles
        auto kstep = [&](const int STc, int offA, int offB) __attribute__((always_inline)) {
            const char *base = inb + (hi ? offB : offA);
#pragma unroll
            for (int h = 0; h < 2; ++h) {            // sub-steps of three pixel tiles (the partner wave of the SIMD covers the LDS round trip)
                f16x8 x0[3], x1[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int i = h * 3 + k;
                    if (i < 5 || cnt == 6) {
                        x0[k] = *reinterpret_cast<const f16x8 *>(base + pb[i]);
                        x1[k] = *reinterpret_cast<const f16x8 *>(base + pb[i] + RBF_IPLN);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int i = h * 3 + k;
                    if (i < 5 || cnt == 6) {
                        if (ABL & 2) { asm volatile("" ::"v"(x0[k]), "v"(x1[k])); continue; }      // timing-only: the reads stay, the MFMAs go
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[STc % D][1][j], x0[k], acc1[i][j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[STc % D][0][j], x0[k], acc1[i][j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[STc % D][0][j], x1[k], acc1[i][j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ABL & 2) asm volatile("" ::"v"(wq[STc % D][0][0]), "v"(wq[STc % D][0][1]), "v"(wq[STc % D][1][0]), "v"(wq[STc % D][1][1]));
            __builtin_amdgcn_sched_barrier(0);
            if (STc + D < 18) wload(STc + D);
            __builtin_amdgcn_sched_barrier(0);
        };
        (void)st;
#define RBF_GROUP(CBc)                                                                                                   \
        {                                                                                                                \
            constexpr int cb = CBc, S0c = (cb / 2) * 9 + (cb & 1) * 4;      /* first K-step of the group in the stream */ \
            const int bo = (cb & 1) * RBF_IBUF, po = ((cb & 1) ^ 1) * RBF_IBUF;                                           \
            if (cb < 3 && !(ABL & 1)) stage_load(cb + 1);                                                                \
            if (cb & 1) {                                                                                                \
                kstep(S0c, po + rbf_tap20(8), bo + rbf_tap20(8));                                                        \
                h2_lds_barrier();            /* every wave has read the partner buffer's last tap */                      \
                kstep(S0c + 1, bo + rbf_tap20(0), bo + rbf_tap20(1));                                                    \
                kstep(S0c + 2, bo + rbf_tap20(2), bo + rbf_tap20(3));                                                    \
                kstep(S0c + 3, bo + rbf_tap20(4), bo + rbf_tap20(5));                                                    \
                kstep(S0c + 4, bo + rbf_tap20(6), bo + rbf_tap20(7));                                                    \
            } else {                                                                                                     \
                kstep(S0c, bo + rbf_tap20(0), bo + rbf_tap20(1));                                                        \
                kstep(S0c + 1, bo + rbf_tap20(2), bo + rbf_tap20(3));                                                    \
                kstep(S0c + 2, bo + rbf_tap20(4), bo + rbf_tap20(5));                                                    \
                kstep(S0c + 3, bo + rbf_tap20(6), bo + rbf_tap20(7));                                                    \
            }                                                                                                            \
            if (cb < 3 && !(ABL & 1)) stage_store((cb + 1) & 1);                                                         \
            __syncthreads();                                                                                             \
        }
        RBF_GROUP(0) RBF_GROUP(1) RBF_GROUP(2) RBF_GROUP(3)
#undef RBF_GROUP
    }
    C16Pass<9, 4, 4> pw2;              // phase 2's first weight fragments travel while the intermediate is written
    if (!(ABL & 4)) c16_wstart(pw2, a.w2);
    // intermediate: 1/S, ReLU, zero outside the image, split -> mid[group][18x18 px][16 ch]
    if (!(ABL & 16)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < 5 || cnt == 6) {
                const int p = (t0 + i) * 16 + xl, my = p / 18, mx = p - my * 18, gy = ty * 16 - 1 + my, gx = tx * 16 - 1 + mx;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 v = acc1[i][j] * a.s1;
                    v.x = in ? fmaxf(v.x, 0.f) : 0.f; v.y = in ? fmaxf(v.y, 0.f) : 0.f; v.z = in ? fmaxf(v.z, 0.f) : 0.f; v.w = in ? fmaxf(v.w, 0.f) : 0.f;
                    amax = sat_amax4(amax, v);
                    unsigned p0, q0, p1, q1;
                    h2_split_pair(v.x, v.y, p0, q0);
                    h2_split_pair(v.z, v.w, p1, q1);
                    if (p < 324) {
                        char *dp = mid + (2 * ch + j) * C16_SLOT + p * 32 + g * 8;
                        *reinterpret_cast<u32x2_t *>(dp) = (u32x2_t){p0, p1};
                        *reinterpret_cast<u32x2_t *>(dp + C16_PLN) = (u32x2_t){q0, q1};
                    }
                }
            }
        }
    } else {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) s += acc1[i][0].x + acc1[i][1].y;
        if (s == 123.456f) mid[tid] = 1;
    }
    __syncthreads();

    // ---- phase 2
    f32x4 acc[8];
    c16_zero<4>(acc);
    if (!(ABL & 4)) c16_accumulate<9, 4, 4>(mid, a.w2, acc, pw2);
    const int ct = C16Tile<4>::ct(), row0 = C16Tile<4>::row0();
    if (ABL & 8) {
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) s += acc[m].x + acc[m].y + acc[m].z + acc[m].w;
        if (s == 123.456f) a.out[0] = 1;
        sat_report(a.sat, amax);
        return;
    }
    // h2_epilogue's 16-byte form: lane (xl, g) moves the 8 channels 8(g>>1).. of row m + (g&1)
    const unsigned row_el = (unsigned)W * 16;
    const unsigned off0w = (unsigned)(((size_t)n0 * 4 + ct) * grp_sz + ((size_t)(ty * 16 + row0) * W + tx * 16 + xl) * 16 + 8 * (g >> 1)) + (unsigned)(g & 1) * row_el;
    const unsigned roff0w = (ABL & 32) ? off0w - (unsigned)((size_t)n0 * 4 * grp_sz) : off0w;
    u32x4 ra[4], rb[4];
#pragma unroll
    for (int m = 0; m < 8; m += 2) {
        ra[m >> 1] = *reinterpret_cast<const C16_GLOBAL u32x4 *>((const C16_GLOBAL unsigned short *)a.x + roff0w + (unsigned)m * row_el);
        rb[m >> 1] = *reinterpret_cast<const C16_GLOBAL u32x4 *>((const C16_GLOBAL unsigned short *)a.x + roff0w + (unsigned)m * row_el + a.x_stride);
    }
#pragma unroll
    for (int m = 0; m < 8; m += 2) {
        u32x4 p = ra[m >> 1], qv = rb[m >> 1];
        rows16_swap(p);
        rows16_swap(qv);
        acc[m] = acc[m] * a.s2 + h2_sum4_lo(p, qv);
        acc[m + 1] = acc[m + 1] * a.s2 + h2_sum4_hi(p, qv);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x4 v = acc[m];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        amax = sat_amax4(amax, v);
        acc[m] = v;
    }
#pragma unroll
    for (int m = 0; m < 8; m += 2) {
        u32x4 p, qv;
        split2_rows(acc[m], acc[m + 1], p, qv);
        rows16_swap(p);
        rows16_swap(qv);
        *reinterpret_cast<C16_GLOBAL u32x4 *>((C16_GLOBAL unsigned short *)a.out + off0w + (unsigned)m * row_el) = p;
        *reinterpret_cast<C16_GLOBAL u32x4 *>((C16_GLOBAL unsigned short *)a.out + off0w + (unsigned)m * row_el + a.out_stride) = qv;
    }
    sat_report(a.sat, amax);
}

}  // namespace

hipError_t launch_rb64_fused(hipStream_t s, const RbFuseArgs &a, int abl)
{
    if ((a.H & 15) || (a.W & 15) || a.N <= 0) return hipErrorInvalidValue;
    const int grid = a.N * (a.H >> 4) * (a.W >> 4);
    switch (abl) {
    case 0: hipLaunchKernelGGL(rb64_fused_kernel<0>, dim3(grid), dim3(512), 0, s, a); break;
    case 1: hipLaunchKernelGGL(rb64_fused_kernel<1>, dim3(grid), dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL(rb64_fused_kernel<2>, dim3(grid), dim3(512), 0, s, a); break;
    case 4: hipLaunchKernelGGL(rb64_fused_kernel<4>, dim3(grid), dim3(512), 0, s, a); break;
    case 6: hipLaunchKernelGGL(rb64_fused_kernel<6>, dim3(grid), dim3(512), 0, s, a); break;
    case 8: hipLaunchKernelGGL(rb64_fused_kernel<8>, dim3(grid), dim3(512), 0, s, a); break;
    case 9: hipLaunchKernelGGL(rb64_fused_kernel<9>, dim3(grid), dim3(512), 0, s, a); break;
    case 16: hipLaunchKernelGGL(rb64_fused_kernel<16>, dim3(grid), dim3(512), 0, s, a); break;
    case 25: hipLaunchKernelGGL(rb64_fused_kernel<25>, dim3(grid), dim3(512), 0, s, a); break;
    case 32: hipLaunchKernelGGL(rb64_fused_kernel<32>, dim3(grid), dim3(512), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace pmp
