// conv_bf16x6.hip — fp32-accurate convolution on the bf16 matrix cores by 3-term operand splitting.
//
// Every fp32 value v is carried as three bf16 values v = v0 + v1 + v2 (exact: 3 x 8 significand bits = 24).
// A product x*w is evaluated as the six bf16 MFMA products with weight >= 2^-16:
//      x0w0 + (x0w1 + x1w0) + (x0w2 + x1w1 + x2w0)            (dropped terms are <= 2^-24 relative)
// accumulated in the MFMA's fp32 accumulator.  tools/precision_study.py: on the real Luma_Q_22 net this is as
// close to the reference as fp32 arithmetic itself (9.2e-5 vs 8.2e-5 for fp64-vs-fp32), while the 3-product split
// (4.6e-3) and single bf16 (0.79) break the 1e-3 tolerance.  Six v_mfma_f32_16x16x32_bf16 (16 cycles, K=32) replace
// sixteen v_mfma_f32_16x16x4_f32 (32 cycles, K=4 each) per 32 channels: 2.67x the fp32-MFMA peak.
//
// Activation format ("split-3"): three bf16 planes, each blocked channels-last [n][C/16][H][W][16] (32 B per pixel and
// group), plane stride = N*C*H*W elements.  Same bytes-per-lane coalescing as the fp32 layout: 16 pixels x 32 B = 512 B
// contiguous per plane per tile row.
//
// GEMM mapping: D[cout][pixel], A = weights, B = pixels (as conv_mfma.hip).  One MFMA K-step (K = 32) covers
// 16 channels x a PAIR of taps: lane group g = l>>4 reads channels 8(g&1)..+7 of tap 2p+(g>>1).  An odd tap count pads
// the last pair with zero weights (3x3: 10 % padding, 5x5: 4 %).
#include <type_traits>

#include "abl_kernels.h"
#include "split3.h"

namespace pmp {

// In-kernel stamps (diagnostic builds only, ABL bit 128): shader-clock ticks of wave 0 at phase boundaries, written to a
// debug buffer nothing else reads (cdna guide, 'In-kernel stamps').
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// Per-lane staging plan, computed ONCE per workgroup: element offset (inside one 16-channel group, split plane
// included) of each 16-B piece this thread copies, and a validity mask for the zero padding.  The per-group work is then
// one add per piece; recomputing the div/mod chain per group cost ~80 VALU issue slots per K-step, which matters when a
// 16-cycle bf16 MFMA leaves only 8 free issue cycles.
template <int KH, int KW>
struct StagePlan {
    unsigned off[GeoX<KH, KW>::NLD];
    unsigned valid;
};

template <int KH, int KW>
__device__ __forceinline__ void x6_plan(StagePlan<KH, KW> &p, size_t plane_stride, int H, int W, int ty, int tx)
{
    typedef GeoX<KH, KW> G;
    constexpr int PY = KH / 2, PX = KW / 2;
    p.valid = 0;
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
        const int i = min((int)threadIdx.x + k * 256, G::PIECES - 1);
        const int sp = i / G::PLANE, j = i - sp * G::PLANE, pix = j >> 1, half = j & 1;
        const int row = pix / G::TW, col = pix - row * G::TW;
        const int gy = ty * 16 + row - PY, gx = tx * 16 + col - PX;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W && (int)threadIdx.x + k * 256 < G::PIECES;
        if (in) p.valid |= 1u << k;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);   // clamped: loads stay unconditional
        p.off[k] = (unsigned)(sp * plane_stride + ((size_t)cy * W + cx) * 16 + half * 8);
    }
}

template <int KH, int KW>
__device__ __forceinline__ void x6_stage_load(const StagePlan<KH, KW> &p, const unsigned short *__restrict__ grp,
                                              u32x4 (&r)[GeoX<KH, KW>::NLD], int k0 = 0, int k1 = 1 << 20)
{
#pragma unroll
    for (int k = 0; k < GeoX<KH, KW>::NLD; ++k) {
        if (k < k0 || k >= k1) continue;   // folds away: callers pass constants into unrolled code
        r[k] = *reinterpret_cast<const u32x4 *>(grp + p.off[k]);
    }
}

template <int KH, int KW>
__device__ __forceinline__ void x6_stage_store(const StagePlan<KH, KW> &p, u32x4 *lds, const u32x4 (&r)[GeoX<KH, KW>::NLD])
{
    typedef GeoX<KH, KW> G;
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
        const int i = threadIdx.x + k * 256;
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (i < G::PIECES) lds[i] = ((p.valid >> k) & 1u) ? r[k] : z;   // LDS image: [split][pixel][2 halves], linear
    }
}

template <int KH, int KW, int NT, int ABL = 0>
__device__ __forceinline__ void x6_accumulate(const unsigned short *__restrict__ x, size_t plane_stride,
                                              const unsigned short *__restrict__ wpk, int C, int H, int W, int n, int ty,
                                              int tx, u32x4 *lds, f32x4 (&acc)[4][NT], unsigned long long *dbg = nullptr)
{
    typedef GeoX<KH, KW> G;
    unsigned long long t_pro = 0, t_k = 0, t_s = 0, t_w = 0, t_b = 0, tmark = 0;   // diagnostic accumulators (ABL & 128)
    if (ABL & 128) tmark = stamp_now();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, xl = lane & 15, g = lane >> 4;
    const int CB = C >> 4;
    const size_t grp_sz = (size_t)H * W * 16;
    const unsigned short *grp0 = x + (size_t)n * CB * grp_sz;
    u32x4 r[G::NLD];
    StagePlan<KH, KW> plan;
    x6_plan<KH, KW>(plan, plane_stride, H, W, ty, tx);
    __syncthreads();
    x6_stage_load<KH, KW>(plan, grp0, r);
    x6_stage_store<KH, KW>(plan, lds, r);
    __syncthreads();
    if (ABL & 128) { const unsigned long long t = stamp_now(); t_pro = t - tmark; tmark = t; }
    // ---- K-step schedule -------------------------------------------------------------------------------------
    // Weight fragments live in ONE register set that is refilled in place, split by split, as soon as its last MFMA
    // of the K-step has been issued: w2 is used once (with x0), w1 twice, w0 three times, so the products are ordered
    //   phase A: x0*w2            -> request next K-step's w2
    //   phase B: x0*w1, x1*w1     -> request next w1 and the next K-step's x0 fragments (second register set)
    //   phase C: x0*w0, x1*w0, x2*w0 -> request next w0
    // Every request has at least one full phase (256..768 MFMA cycles) before its first use.  All loads are
    // unconditional so hipcc's s_waitcnt vmcnt(N) are exact counts; scheduling fences pin the phase order.
    // Tap pairing.  A K-step covers 16 channels x 2 taps.  With an odd tap count the last tap has no partner inside its
    // channel group; when the number of groups is even ("paired" packing, see pack_x6) the last tap of an EVEN group is
    // deferred and paired with the last tap of the following ODD group, as that group's first K-step - both halo tiles
    // are resident then, the even one in the other LDS buffer (it is overwritten only at the end of the odd group) - so
    // no MFMA work is spent on zero padding (3x3: 9 K-steps per two groups instead of 10).
    //   mode 0 (plain): NKS = ceil(T/2) per group, last pair padded with zero weights
    //   mode 1 (even group of a pair): (T-1)/2 K-steps;  mode 2 (odd group): (T-1)/2 + 1 K-steps
    const bool paired = (CB & 1) == 0 && (G::TAPS & 1);
    const bf16x8 *wl = reinterpret_cast<const bf16x8 *>(wpk) + lane;
    const int last = paired ? (CB / 2) * G::TAPS - 1 : CB * G::NKS - 1;   // last K-step of the weight stream
    bf16x8 w0[NT], w1[NT], w2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { w2[nt] = wl[(2 * NT + nt) * 64]; w1[nt] = wl[(1 * NT + nt) * 64]; w0[nt] = wl[(0 * NT + nt) * 64]; }
    const int pb = ((wave * 4 * G::TW + xl) * 2 + (g & 1)) * 16;   // bytes inside a split plane, tap (0,0)
    int stream = 0;                                                  // K-step index inside the weight stream
    int tapsel = g >> 1;

    auto group = [&](auto mode_tag, int cb) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int NK = MODE == 0 ? G::NKS : (MODE == 1 ? (G::TAPS - 1) / 2 : (G::TAPS - 1) / 2 + 1);
        constexpr int PER = (G::NLD + (NK > 0 ? NK : 1) - 1) / (NK > 0 ? NK : 1);   // staging loads issued per K-step
        const bool more = cb + 1 < CB;
        const unsigned short *nxt_grp = ((ABL & 32) ? x : grp0) + (size_t)min(cb + 1, CB - 1) * grp_sz;   // clamped: loads stay unconditional
        const char *buf = reinterpret_cast<const char *>(lds + (cb & 1) * G::PIECES);
        const char *prv = reinterpret_cast<const char *>(lds + ((cb + 1) & 1) * G::PIECES);
        // per-lane address of this lane's tap of K-step ks (lanes g < 2: first tap of the pair, g >= 2: second)
        auto xaddr = [&](int ks) -> const char * {
            int tA = 2 * ks, tB = 2 * ks + 1;
            bool prevA = false;
            if (MODE == 0 && tB >= G::TAPS) tB = tA;
            if (MODE == 2) {   // first K-step of an odd group: the deferred tap of the even group + this group's last tap
                if (ks == 0) { tA = tB = G::TAPS - 1; prevA = true; }
                else { tA = 2 * (ks - 1); tB = tA + 1; }
            }
            const int oA = ((tA / KW) * G::TW + tA % KW) * 32, oB = ((tB / KW) * G::TW + tB % KW) * 32;
            return (tapsel ? buf + oB : (prevA ? prv : buf) + oA) + pb;
        };
        bf16x8 xa[4], xb[4], x1[4], x2[4];   // x0 fragments alternate between xa (even K-steps) and xb (odd)
        if (NK == 0) {   // 1x1 source, even group: nothing to compute yet, only fetch the partner group
            if (!(ABL & 1)) x6_stage_load<KH, KW>(plan, nxt_grp, r);
        } else {
            const char *p0 = xaddr(0);
#pragma unroll
            for (int m = 0; m < 4; ++m) xa[m] = *reinterpret_cast<const bf16x8 *>(p0 + m * G::TW * 32);
        }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            asm volatile("" : "+v"(tapsel));   // keeps hipcc from hoisting every K-step's tap offset out of the group loop
            const char *px = xaddr(ks);
            ++stream;
            const bf16x8 *wk = wl + (size_t)min(stream, last) * (3 * NT * 64);
            bf16x8 (&x0)[4] = (ks & 1) ? xb : xa;
            bf16x8 (&x0n)[4] = (ks & 1) ? xa : xb;
            if (!(ABL & 4) || ks == 0) {
#pragma unroll
                for (int m = 0; m < 4; ++m) x1[m] = *reinterpret_cast<const bf16x8 *>(px + G::PLANE * 16 + m * G::TW * 32);
#pragma unroll
                for (int m = 0; m < 4; ++m) x2[m] = *reinterpret_cast<const bf16x8 *>(px + 2 * G::PLANE * 16 + m * G::TW * 32);
            }
            if (!(ABL & 1)) x6_stage_load<KH, KW>(plan, nxt_grp, r, ks * PER, (ks + 1) * PER);
            __builtin_amdgcn_sched_barrier(0);
            // phase A
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[nt], x0[m], acc[m][nt], 0, 0, 0);
            if (!(ABL & 2)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) w2[nt] = wk[(2 * NT + nt) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            // phase B
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[nt], x0[m], acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[nt], x1[m], acc[m][nt], 0, 0, 0);
                }
            if (!(ABL & 2)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) w1[nt] = wk[(1 * NT + nt) * 64];
            }
            if (ks + 1 < NK && !(ABL & 4)) {
                const char *pn = xaddr(ks + 1);
#pragma unroll
                for (int m = 0; m < 4; ++m) x0n[m] = *reinterpret_cast<const bf16x8 *>(pn + m * G::TW * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
            // phase C
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[nt], x0[m], acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[nt], x1[m], acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[nt], x2[m], acc[m][nt], 0, 0, 0);
                }
            if (!(ABL & 2)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) w0[nt] = wk[(0 * NT + nt) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            // The cross-group K-step is the only one that reads the OTHER halo buffer; every wave must be past it before any
            // wave overwrites that buffer at the end of this group (the waves of a workgroup are only loosely in step).
            if (MODE == 2 && ks == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (ABL & 128) { const unsigned long long t = stamp_now(); t_k += t - tmark; tmark = t; }
        // the partner buffer is only overwritten here, after the last K-step that may read the previous group from it
        if ((ABL & 128) && more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long t = stamp_now(); t_w += t - tmark; tmark = t; }
        if (more && !(ABL & 1)) {
            if (ABL & 256) {   // timing-only: keep the loads alive, skip the LDS stores
#pragma unroll
                for (int k = 0; k < G::NLD; ++k) asm volatile("" ::"v"(r[k]));
            } else
                x6_stage_store<KH, KW>(plan, lds + ((cb + 1) & 1) * G::PIECES, r);
        }
        if (ABL & 128) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t = stamp_now(); t_s += t - tmark; tmark = t; }
        if (!(ABL & 16)) __syncthreads();
        if (ABL & 128) { const unsigned long long t = stamp_now(); t_b += t - tmark; tmark = t; }
    };

    if (paired) {
        for (int cb = 0; cb < CB; cb += 2) {
            group(std::integral_constant<int, 1>{}, cb);
            group(std::integral_constant<int, 2>{}, cb + 1);
        }
    } else {
        for (int cb = 0; cb < CB; ++cb) group(std::integral_constant<int, 0>{}, cb);
    }
    if ((ABL & 128) && dbg && threadIdx.x == 0) { dbg[0] = t_pro; dbg[1] = t_k; dbg[2] = t_s; dbg[6] = t_w; dbg[7] = t_b; }
}

template <int KH, int KW, int NT, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_x6_kernel(ConvX6Args a)
{
    typedef GeoX<KH, KW> G;
    __shared__ u32x4 lds[2 * G::PIECES];
    const int tiles_x = a.W >> 4, tiles = tiles_x * (a.H >> 4);
    const int n = blockIdx.x / tiles, t = blockIdx.x - n * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, xl = lane & 15, g = lane >> 4;

    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned long long t_begin = (ABL & 128) ? stamp_now() : 0;
    x6_accumulate<KH, KW, NT, ABL>(a.x, a.x_stride, a.w, a.Cin, a.H, a.W, n, ty, tx, lds, acc, a.abl.dbg ? a.abl.dbg + (size_t)blockIdx.x * 8 : nullptr);
    const unsigned long long t_acc = (ABL & 128) ? stamp_now() : 0;
    if (a.x_sc) x6_accumulate<1, 1, NT>(a.x_sc, a.sc_stride, a.w_sc, a.Csc, a.H, a.W, n, ty, tx, lds, acc);

    const int H = a.H, W = a.W;
    const size_t grp = (size_t)H * W * 16;
    if (ABL & 8) {  // timing-only build: skip the epilogue but keep the accumulators live
        float sacc = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int m = 0; m < 4; ++m) sacc += acc[m][nt].x + acc[m][nt].y + acc[m][nt].z + acc[m][nt].w;
        if (sacc == 123.456f) a.out[0] = 1;
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int y = ty * 16 + wave * 4 + m, x = tx * 16 + xl;
            const size_t off = ((size_t)n * NT + nt) * grp + ((size_t)y * W + x) * 16 + g * 4;
            f32x4 v = acc[m][nt];
            if (a.res) v += load_split4(a.res + off, a.res_stride);
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (a.gate) v *= load_split4(a.gate + off, a.gate_stride);
            acc[m][nt] = v;
        }
        if (!a.pool) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int y = ty * 16 + wave * 4 + m, x = tx * 16 + xl;
                const size_t off = ((size_t)n * NT + nt) * grp + ((size_t)y * W + x) * 16 + g * 4;
                if (a.out_f32) *reinterpret_cast<f32x4 *>(a.out_f32 + off) = acc[m][nt];
                else store_split4(a.out + off, a.out_stride, acc[m][nt]);
            }
        } else {
            const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
            for (int m = 0; m < 4; m += 2) {
                f32x4 v = acc[m][nt], u = acc[m + 1][nt];
                v.x = fmaxf(v.x, u.x); v.y = fmaxf(v.y, u.y); v.z = fmaxf(v.z, u.z); v.w = fmaxf(v.w, u.w);
                f32x4 o;
                o.x = __shfl_xor(v.x, 1); o.y = __shfl_xor(v.y, 1); o.z = __shfl_xor(v.z, 1); o.w = __shfl_xor(v.w, 1);
                v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
                if ((xl & 1) == 0) {
                    const int yo = ty * 8 + wave * 2 + (m >> 1), xo = tx * 8 + (xl >> 1);
                    const size_t off = (((size_t)n * NT + nt) * Ho + yo) * Wo * 16 + (size_t)xo * 16 + g * 4;
                    if (a.out_f32) *reinterpret_cast<f32x4 *>(a.out_f32 + off) = v;
                    else store_split4(a.out + off, a.out_stride, v);
                }
            }
        }
    }
    if ((ABL & 128) && a.abl.dbg && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // include the store acknowledgements in the epilogue span
        const unsigned long long t_end = stamp_now();
        unsigned long long *d = a.abl.dbg + (size_t)blockIdx.x * 8;
        d[3] = t_acc - t_begin; d[4] = t_end - t_acc; d[5] = t_begin;
    }
}

template <int KH, int KW>
static hipError_t launch_x6(hipStream_t s, const ConvX6Args &a)
{
    const int grid = a.N * (a.H >> 4) * (a.W >> 4);
    switch (a.Cout >> 4) {
    case 1: hipLaunchKernelGGL((conv_x6_kernel<KH, KW, 1>), dim3(grid), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((conv_x6_kernel<KH, KW, 2>), dim3(grid), dim3(256), 0, s, a); break;
    case 4:
#ifdef PMP_ABLATION   // timing-only builds (wrong results): only in libpmp_hip_abl.so, never in the product library
        if (KH == 3 && g_conv_variant >= 10) {  // timing-only ablation builds (tools/conv_x6_bench.py)
            switch (g_conv_variant - 10) {
            case 1: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 1>), dim3(grid), dim3(256), 0, s, a); break;
            case 2: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 2>), dim3(grid), dim3(256), 0, s, a); break;
            case 4: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 4>), dim3(grid), dim3(256), 0, s, a); break;
            case 8: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 8>), dim3(grid), dim3(256), 0, s, a); break;
            case 15: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 15>), dim3(grid), dim3(256), 0, s, a); break;
            case 16: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 16>), dim3(grid), dim3(256), 0, s, a); break;
            case 31: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 31>), dim3(grid), dim3(256), 0, s, a); break;
            case 32: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 32>), dim3(grid), dim3(256), 0, s, a); break;
            case 40: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 40>), dim3(grid), dim3(256), 0, s, a); break;
            case 128: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 128>), dim3(grid), dim3(256), 0, s, a); break;
            case 200: hipLaunchKernelGGL((conv_x6_kernel<3, 3, 4, 256>), dim3(grid), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((conv_x6_kernel<KH, KW, 4>), dim3(grid), dim3(256), 0, s, a); break;
            }
            break;
        }
#endif
            hipLaunchKernelGGL((conv_x6_kernel<KH, KW, 4>), dim3(grid), dim3(256), 0, s, a);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_x6(hipStream_t s, const ConvX6Args &a)
{
    if ((a.H & 15) || (a.W & 15) || (a.Cin & 15) || (a.Cout & 15) || (a.x_sc && (a.Csc & 15)) || a.N <= 0)
        return hipErrorInvalidValue;
    if (a.pool && a.gate) return hipErrorInvalidValue;
    if (a.KH == 3 && a.KW == 3) return launch_x6<3, 3>(s, a);
    if (a.KH == 5 && a.KW == 5) return launch_x6<5, 5>(s, a);
    if (a.KH == 1 && a.KW == 1) return launch_x6<1, 1>(s, a);
    return hipErrorInvalidValue;
}

// ---- format converters ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f32_to_split3_kernel(const float *__restrict__ x, unsigned short *__restrict__ out,
                                                            size_t n4, size_t plane_stride)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        store_split4(out + i * 4, plane_stride, *reinterpret_cast<const f32x4 *>(x + i * 4));
}

__global__ __launch_bounds__(256) void split3_to_f32_kernel(const unsigned short *__restrict__ x, float *__restrict__ out,
                                                            size_t n4, size_t plane_stride)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        *reinterpret_cast<f32x4 *>(out + i * 4) = load_split4(x + i * 4, plane_stride);
}

hipError_t launch_f32_to_split3(hipStream_t s, const float *x, unsigned short *out, size_t n, size_t plane_stride)
{
    const size_t n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256);
    if (n4) hipLaunchKernelGGL(f32_to_split3_kernel, dim3(grid), dim3(256), 0, s, x, out, n4, plane_stride);
    return hipGetLastError();
}

hipError_t launch_split3_to_f32(hipStream_t s, const unsigned short *x, float *out, size_t n, size_t plane_stride)
{
    const size_t n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256);
    if (n4) hipLaunchKernelGGL(split3_to_f32_kernel, dim3(grid), dim3(256), 0, s, x, out, n4, plane_stride);
    return hipGetLastError();
}

}  // namespace pmp
