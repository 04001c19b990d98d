// EXPERIMENT, not built into libpmp_hip.so (round 1).  Parity-correct (the f16x3 GPU suite passed with every eligible launch
// routed here) but 3-6 % slower than conv_f16x3.hip:   3x3 64->64, 1024 blocks: 0.88-0.90 ms vs 0.84 ms; 5x5: 0.49 vs 0.47 ms.
// What the builds showed (in-kernel stamps, s_memtime / s_memrealtime):
//   * 6-wave workgroups (4 compute + 2 loader waves, 168 VGPRs, 16x16 tiles): only ONE workgroup per CU is ever resident
//     (6 waves do not fill the 4 SIMDs evenly), so the kernel took exactly twice its main loop - although that loop needed
//     29 % fewer cycles per tile than conv_f16x3.hip's.
//   * this build (8 waves on a 16x8 tile, <= 128 VGPRs, two workgroups resident): a compute wave's K-step is 24 MFMAs
//     (384 cycles) and takes ~900 ticks with weights double-buffered; the epilogue of a 16x8 tile costs 3.6 k ticks per
//     23 k-tick tile; the register cap leaves 2-3 spills whose reloads wait behind the weight prefetches.
// To build it: add it to the Makefile, declare conv_h2_ws_eligible / launch_conv_h2_ws in pmp_kernels.h, route launch_conv_h2.
//
// conv_f16x3_ws.hip — the f16x3 convolution (conv_f16x3.hip) with dedicated LOADER waves, persistent over tiles.
// Shapes: Cout = 64, even number of 16-channel groups, odd tap count (3x3 / 5x5), no 1x1 shortcut source.
//
// Why.  The counter that orders a wave's vector-memory operations (vmcnt) is in-order: in conv_f16x3.hip a wave that waits
// for a weight fragment (L2, ~1 us) also waits for every halo request it issued before it (HBM, several us), and the halo
// staging registers (48) push the kernel to 254 VGPRs, which rules out chaining tiles.  Here
//   waves 0-3 (compute): LDS fragment reads, weight refills, MFMAs, and their tile's epilogue - the only HBM reads on their
//                        counter are the residual's, at the very end of a tile;
//   waves 4-7 (loaders): request halo tiles HBM -> registers two channel groups ahead of the compute waves and write them
//                        to LDS one group ahead; they run straight through tile boundaries, so a tile's first halo tiles
//                        arrive during the previous tile's MFMAs and its output stores drain under the next tile's.
// Workgroup = 8 waves on a 16 x 8 pixel tile (compute wave = 4 rows x 32 couts), 2 workgroups per CU = 4 waves per SIMD
// (two compute, two loaders) at <= 128 VGPRs - a wave count that fills every SIMD evenly, so both workgroups are always
// resident (6-wave workgroups at 168 VGPRs were placed one per CU).  One LDS-only barrier per channel group; weight
// prefetches and halo requests stay in flight across it.
#include <type_traits>

#include "pmp_kernels.h"
#include "split3.h"

namespace pmp {

__device__ __forceinline__ unsigned long long ws_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <int KH, int KW, bool STAMP = false>
__global__ __launch_bounds__(512, 4) void conv_h2_ws_kernel(ConvX6Args a)
{
    constexpr int TR = 8, RW = 4, CW = 2, RSPLIT = TR / RW;   // tile rows, rows and cout groups per compute wave
    typedef GeoH<KH, KW, TR> G;
    constexpr int NT = 4, T = G::TAPS, HP = (T - 1) / 2;
    static_assert((T & 1) == 1, "tap pairing");
    __shared__ u32x4 lds[2 * G::PIECES];

    const int tid = threadIdx.x, lane = tid & 63, xl = lane & 15, g = lane >> 4;
    const int H = a.H, W = a.W, CB = a.Cin >> 4;
    const int tiles_x = W >> 4, tiles = tiles_x * (H / TR), ntiles = a.N * tiles;
    const size_t grp_sz = (size_t)H * W * 16;
    const int first = blockIdx.x, stride = gridDim.x;
    const int my_tiles = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;

    if (tid < 256) {
        // ================================================================================== compute waves
        const int wave = tid >> 6, rh = wave % RSPLIT, ch = wave / RSPLIT;   // rows RW*rh.., cout groups CW*ch..
        const f16x8 *wl = reinterpret_cast<const f16x8 *>(a.w) + lane + ch * CW * 64;
        const int last = (CB / 2) * T - 1;   // last K-step of the weight stream; it wraps - the weights are per launch
        // A compute wave's K-step is only 24 MFMAs (384 cycles), shorter than an L2 round trip: weight fragments and the
        // split-0 pixel fragments are double-buffered, requested one whole K-step ahead.  Set parity is static: the pair of
        // groups (2*HP+1 K-steps, odd) is instantiated for both parities and the pair loop is unrolled by two.
        f16x8 wa0[CW], wa1[CW], wb0[CW], wb1[CW];
#pragma unroll
        for (int nt = 0; nt < CW; ++nt) { wa0[nt] = wl[(0 * NT + nt) * 64]; wa1[nt] = wl[(1 * NT + nt) * 64]; }
        const int pb = ((rh * RW * G::TW + xl) * 2 + (g & 1)) * 16;   // bytes inside a split plane, tap (0,0)
        int stream = 0, tapsel = g >> 1, par = 0;
        constexpr int O_LAST = (((T - 1) / KW) * G::TW + (T - 1) % KW) * 32;
        f16x8 xa[RW], x1[RW];   // split-0 / split-1 pixel fragments of the current K-step
        f32x4 acc[RW][CW];

        // K-steps of one channel group (see conv_f16x3.hip): MODE 1 even group (HP tap pairs, then lanes g < 2 pick up the
        // deferred last tap), MODE 2 odd group (cross-group pair first, then HP pairs).  WP = weight-set parity at its start.
        auto group = [&](auto mode_tag, auto wp_tag, const char *buf) __attribute__((always_inline)) {
            constexpr int MODE = decltype(mode_tag)::value, WP = decltype(wp_tag)::value;
            constexpr int NK = MODE == 1 ? HP : HP + 1;
            auto xaddr = [&](int ks) -> const char * {
                const int j = MODE == 2 ? ks - 1 : ks;
                const int tA = 2 * j, tB = 2 * j + 1;
                const int oA = ((tA / KW) * G::TW + tA % KW) * 32, oB = ((tB / KW) * G::TW + tB % KW) * 32;
                return buf + (tapsel ? oB : oA) + pb;
            };
            if (MODE == 2) {
                if (g >= 2) {
                    const char *pl = buf + O_LAST + pb;
#pragma unroll
                    for (int m = 0; m < RW; ++m) {
                        xa[m] = *reinterpret_cast<const f16x8 *>(pl + m * G::TW * 32);
                        x1[m] = *reinterpret_cast<const f16x8 *>(pl + G::PLANE * 16 + m * G::TW * 32);
                    }
                }
            } else if (NK > 0) {
                const char *p0x = xaddr(0);
#pragma unroll
                for (int m = 0; m < RW; ++m) xa[m] = *reinterpret_cast<const f16x8 *>(p0x + m * G::TW * 32);
            }
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                asm volatile("" : "+v"(tapsel));   // keeps hipcc from hoisting every K-step's tap offset out of the loop
                stream = stream == last ? 0 : stream + 1;
                const f16x8 *wf = wl + (size_t)stream * (2 * NT * 64);
                const bool odd = ((WP + ks) & 1) != 0;   // compile-time after unrolling
                f16x8 (&w0)[CW] = odd ? wb0 : wa0;
                f16x8 (&w1)[CW] = odd ? wb1 : wa1;
                f16x8 (&w0n)[CW] = odd ? wa0 : wb0;
                f16x8 (&w1n)[CW] = odd ? wa1 : wb1;
                f16x8 (&x0)[RW] = xa;
                // requests for the NEXT K-step: weights (L2) and split-0 pixels (LDS); this K-step's split-1 pixels
#pragma unroll
                for (int nt = 0; nt < CW; ++nt) { w1n[nt] = wf[(1 * NT + nt) * 64]; w0n[nt] = wf[(0 * NT + nt) * 64]; }
                if (!(MODE == 2 && ks == 0)) {
                    const char *px = xaddr(ks);
#pragma unroll
                    for (int m = 0; m < RW; ++m) x1[m] = *reinterpret_cast<const f16x8 *>(px + G::PLANE * 16 + m * G::TW * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < RW; ++m)
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) {
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[nt], x0[m], acc[m][nt], 0, 0, 0);
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], x0[m], acc[m][nt], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < NK) {   // x0 is free: read the next K-step's split-0 pixels into it (128 VGPRs leave no second set)
                    const char *pn = xaddr(ks + 1);
#pragma unroll
                    for (int m = 0; m < RW; ++m) x0[m] = *reinterpret_cast<const f16x8 *>(pn + m * G::TW * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < RW; ++m)
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], x1[m], acc[m][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 1 && g < 2) {   // deferred last tap of the even group: its buffer is rewritten during the odd group
                const char *pl = buf + O_LAST + pb;
#pragma unroll
                for (int m = 0; m < RW; ++m) {
                    xa[m] = *reinterpret_cast<const f16x8 *>(pl + m * G::TW * 32);
                    x1[m] = *reinterpret_cast<const f16x8 *>(pl + G::PLANE * 16 + m * G::TW * 32);
                }
            }
        };

        unsigned long long t_k = 0, t_b = 0, t_e = 0, t0 = 0, tm = 0;   // diagnostic build only
        if (STAMP) t0 = ws_stamp();
        h2_lds_barrier();   // group 0 of the first tile is staged
        const unsigned long long t1 = STAMP ? ws_stamp() : 0;
        const unsigned long long r1 = STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
        if (STAMP) tm = t1;
        auto bar = [&]() __attribute__((always_inline)) {
            if (STAMP) { const unsigned long long t = ws_stamp(); t_k += t - tm; tm = t; }
            h2_lds_barrier(); par ^= 1;
            if (STAMP) { const unsigned long long t = ws_stamp(); t_b += t - tm; tm = t; }
        };
        int tile = first;
        const int ppt = CB >> 1, total_pairs = my_tiles * ppt;   // pairs of channel groups per tile / in this workgroup's list
        int pit = 0;                                              // pairs done inside the current tile
#pragma unroll
        for (int m = 0; m < RW; ++m)
#pragma unroll
            for (int nt = 0; nt < CW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // epilogue of the tile that just finished (as conv_f16x3.hip); the loader waves are already fetching the next tile
        auto epilogue = [&]() __attribute__((always_inline)) {
            // lane- and wave-derived address parts are recomputed here: hoisted out of the tile loop they would have to live
            // through the K-steps, and at 128 VGPRs that means scratch (whose reloads wait behind the weight prefetches)
            int lane_e = lane, wave_e = wave;
            asm volatile("" : "+v"(lane_e), "+v"(wave_e));
            const int xl = lane_e & 15, g = lane_e >> 4, rh = wave_e % RSPLIT, ch = wave_e / RSPLIT;
            const int n = tile / tiles, tt = tile - n * tiles, ty = tt / tiles_x, tx = tt - ty * tiles_x;
            const float inv_scale = a.out_scale;
            const unsigned off0 = (unsigned)(((size_t)n * NT + ch * CW) * grp_sz + ((size_t)(ty * TR + rh * RW) * W + tx * 16 + xl) * 16 + g * 4);
            const unsigned row_el = (unsigned)W * 16;
            if (a.res) {
                f16x4 ra[RW][CW], rbv[RW][CW];
#pragma unroll
                for (int nt = 0; nt < CW; ++nt)
#pragma unroll
                    for (int m = 0; m < RW; ++m) {
                        const unsigned off = off0 + (unsigned)m * row_el + (unsigned)nt * (unsigned)grp_sz;
                        ra[m][nt] = *reinterpret_cast<const f16x4 *>(a.res + off);
                        rbv[m][nt] = *reinterpret_cast<const f16x4 *>(a.res + off + a.res_stride);
                    }
#pragma unroll
                for (int nt = 0; nt < CW; ++nt)
#pragma unroll
                    for (int m = 0; m < RW; ++m) {
                        f32x4 v = acc[m][nt] * inv_scale;
                        const f16x4 p = ra[m][nt], q = rbv[m][nt];
                        v.x += (float)p.x + (float)q.x; v.y += (float)p.y + (float)q.y; v.z += (float)p.z + (float)q.z; v.w += (float)p.w + (float)q.w;
                        acc[m][nt] = v;
                    }
            } else {
#pragma unroll
                for (int nt = 0; nt < CW; ++nt)
#pragma unroll
                    for (int m = 0; m < RW; ++m) acc[m][nt] = acc[m][nt] * inv_scale;
            }
#pragma unroll
            for (int nt = 0; nt < CW; ++nt) {
                if (a.relu) {
#pragma unroll
                    for (int m = 0; m < RW; ++m) {
                        f32x4 v = acc[m][nt];
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        acc[m][nt] = v;
                    }
                }
                if (!a.pool) {
#pragma unroll
                    for (int m = 0; m < RW; ++m) {
                        const unsigned off = off0 + (unsigned)m * row_el + (unsigned)nt * (unsigned)grp_sz;
                        store_split2_4(a.out + off, a.out_stride, acc[m][nt]);
                    }
                } else {
                    const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
                    for (int m = 0; m < RW; m += 2) {
                        f32x4 v = acc[m][nt], u = acc[m + 1][nt];
                        v.x = fmaxf(v.x, u.x); v.y = fmaxf(v.y, u.y); v.z = fmaxf(v.z, u.z); v.w = fmaxf(v.w, u.w);
                        f32x4 o;
                        o.x = __shfl_xor(v.x, 1); o.y = __shfl_xor(v.y, 1); o.z = __shfl_xor(v.z, 1); o.w = __shfl_xor(v.w, 1);
                        v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
                        if ((xl & 1) == 0) {
                            const int yo = ty * (TR / 2) + rh * (RW / 2) + (m >> 1), xo = tx * 8 + (xl >> 1);
                            const size_t off = (((size_t)n * NT + ch * CW + nt) * Ho + yo) * Wo * 16 + (size_t)xo * 16 + g * 4;
                            store_split2_4(a.out + off, a.out_stride, v);
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < RW; ++m)
#pragma unroll
                for (int nt = 0; nt < CW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            tile += stride;
            if (STAMP) { const unsigned long long t = ws_stamp(); t_e += t - tm; tm = t; }
        };
        // one pair of channel groups; P = weight-set parity at its start (the next pair starts with the other one)
        auto pair = [&](auto p_tag) __attribute__((always_inline)) {
            constexpr int P = decltype(p_tag)::value;
            group(std::integral_constant<int, 1>{}, std::integral_constant<int, P>{}, reinterpret_cast<const char *>(lds + par * G::PIECES));
            bar();
            group(std::integral_constant<int, 2>{}, std::integral_constant<int, (P + HP) & 1>{}, reinterpret_cast<const char *>(lds + par * G::PIECES));
            bar();
            if (++pit == ppt) { pit = 0; epilogue(); }
        };
        int gp = 0;
        for (; gp + 1 < total_pairs; gp += 2) {
            pair(std::integral_constant<int, 0>{});
            pair(std::integral_constant<int, 1>{});
        }
        if (gp < total_pairs) pair(std::integral_constant<int, 0>{});
        if (STAMP && a.dbg && (tid == 0 || tid == 64)) {
            unsigned long long *d = a.dbg + (size_t)blockIdx.x * 16 + (tid >> 6) * 8;
            d[0] = t1 - t0; d[1] = tm - t1; d[2] = t_k; d[3] = t_b; d[4] = t_e; d[5] = __builtin_amdgcn_s_memrealtime() - r1;
        }
    } else {
        // ================================================================================== loader waves
        constexpr int PY = KH / 2, PX = KW / 2;
        constexpr int NLW = (G::PIECES + 255) / 256;   // 16-B pieces per loader thread and channel group
        const int lt = tid - 256;
        unsigned off[NLW];
        unsigned vplan = 0;
        u32x4 ra[NLW], rb[NLW];
        unsigned va = 0, vb = 0;
        const unsigned short *grp0 = a.x;
        int rq_tile = first, rq_cb = 0, rq_left = my_tiles;   // next group to request
        auto make_plan = [&](int tile) {
            const int n = tile / tiles, tt = tile - n * tiles, ty = tt / tiles_x, tx = tt - ty * tiles_x;
            vplan = 0;
#pragma unroll
            for (int k = 0; k < NLW; ++k) {
                const int i = min(lt + k * 256, G::PIECES - 1);
                const int sp = i / G::PLANE, j = i - sp * G::PLANE, pix = j >> 1, half = j & 1;
                const int row = pix / G::TW, col = pix - row * G::TW;
                const int gy = ty * TR + row - PY, gx = tx * 16 + col - PX;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) vplan |= 1u << k;
                const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);   // clamped: the loads stay unconditional
                off[k] = (unsigned)(sp * a.x_stride + ((size_t)cy * W + cx) * 16 + half * 8);
            }
            grp0 = a.x + (size_t)n * CB * grp_sz;
        };
        // request the next group of this workgroup's tile list into r (past the end: the last group again - never stored)
        auto request = [&](u32x4 (&r)[NLW], unsigned &v) {
            if (rq_cb == 0 && rq_left > 0) make_plan(rq_tile);
            const unsigned short *grp = grp0 + (size_t)rq_cb * grp_sz;
#pragma unroll
            for (int k = 0; k < NLW; ++k) r[k] = *reinterpret_cast<const u32x4 *>(grp + off[k]);
            v = vplan;
            if (rq_left > 0 && ++rq_cb == CB) { rq_cb = 0; rq_tile += stride; --rq_left; }
        };
        auto store = [&](u32x4 *dst, const u32x4 (&r)[NLW], unsigned v) {
#pragma unroll
            for (int k = 0; k < NLW; ++k) {
                const int i = lt + k * 256;
                const u32x4 z = {0u, 0u, 0u, 0u};
                if (k < NLW - 1 || i < G::PIECES) dst[i] = ((v >> k) & 1u) ? r[k] : z;   // only the last piece can fall outside
            }
        };
        const int Q = my_tiles * CB;   // channel groups this workgroup computes; CB is even, so is Q
        if (Q > 0) {
            request(ra, va);              // group 0
            request(rb, vb);              // group 1
            store(lds, ra, va);
            request(ra, va);              // group 2
        }
        h2_lds_barrier();
        // step q (the compute waves are in group q): write group q+1 - requested two steps ago - to the buffer nobody reads
        // now, then request group q+3 into the registers just freed
        // (everything unconditional, so hipcc's vmcnt waits are exact counts: the store waits for loads two steps old while
        // the newer set stays in flight; past the end the requests repeat the last group and the extra store goes to the
        // buffer nobody reads any more)
        for (int q = 0; q < Q; q += 2) {
            store(lds + G::PIECES, rb, vb);
            request(rb, vb);
            h2_lds_barrier();
            store(lds, ra, va);
            request(ra, va);
            h2_lds_barrier();
        }
    }
}

bool conv_h2_ws_eligible(const ConvX6Args &a)
{
    const int CB = a.Cin >> 4;
    return a.Cout == 64 && !a.x_sc && !a.gate && !a.out_f32 && a.out && (CB & 1) == 0 && a.KH == a.KW && (a.KH == 3 || a.KH == 5);
}

hipError_t launch_conv_h2_ws(hipStream_t s, const ConvX6Args &a, int rows_per_wave)
{
    const int ntiles = a.N * (a.H >> 3) * (a.W >> 4);   // 16 x 8 pixel tiles
    const int grid = ntiles < 512 ? ntiles : 512;   // two persistent workgroups on each of the 256 CUs
    (void)rows_per_wave;
    if (a.dbg && a.KH == 3) hipLaunchKernelGGL((conv_h2_ws_kernel<3, 3, true>), dim3(grid), dim3(512), 0, s, a);
    else if (a.KH == 3) hipLaunchKernelGGL((conv_h2_ws_kernel<3, 3>), dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv_h2_ws_kernel<5, 5>), dim3(grid), dim3(512), 0, s, a);
    return hipGetLastError();
}

}  // namespace pmp
