// EXPERIMENT, not built into libpmp_hip.so (round 1).  Correct (the GPU parity suite passed with it routed in for every
// eligible launch) but not faster than conv_bf16x6.hip, and the measurements say why:
//   4 compute + 4 loader waves (1 compute wave/SIMD, 256 VGPRs): 3x3 64->64 199 TF vs 203 TF; barriers 3 %, hand-over 2 %
//     of the compute waves' time, yet a K-step takes 2050 cycles instead of 1536 - one wave per SIMD does not keep the
//     matrix pipe full (with weight refills AND fragment reads removed it still only reaches 211 TF).
//   8 compute + 4 loader waves (2 compute waves/SIMD, 168 VGPRs, each wave 4 rows x 32 couts): 197 TF.  The older compute
//     wave of each SIMD wins arbitration, finishes its channel group early and waits 32 % of its life at the group
//     barrier while the younger runs alone; in-kernel clock 1.5-1.7 GHz (s_memtime ticks / wall time).
// Conclusion: the kernel is power/clock-limited (DVFS) rather than stall-limited - cycles saved by hiding the staging
// loads and the epilogue come back as a lower clock.  The lever that worked is fewer MFMAs per result (conv_f16x3.hip).
// To build it again: add it to the Makefile, declare conv_x6_ws_eligible/launch_conv_x6_ws in pmp_kernels.h and route
// launch_conv_x6 to it.
//
// conv_x6_ws.hip — persistent, wave-specialised build of the split-3 convolution (conv_bf16x6.hip) for the dominant
// shapes: Cout = 64, no 1x1 shortcut source, even number of 16-channel groups, odd tap count (3x3 / 5x5).
//
// Why.  In conv_bf16x6.hip every wave issues three kinds of global access on ONE in-order counter (vmcnt): the weight
// fragments it needs one phase later, the halo pieces it stages for the next channel group, and the epilogue's residual
// loads / output stores.  A wait for weights therefore also waits for every older HBM staging load (profiles/ r01e:
// staging costs 17 % of the kernel, the epilogue another 15 %).  Here the roles are split:
//   waves 0-3 (compute): LDS fragment reads, weight refills (L2-resident) and MFMAs - nothing else touches their vmcnt;
//   waves 4-7 (loaders): stage the next halo tile HBM -> registers -> LDS, and run the PREVIOUS tile's epilogue
//                        (accumulators handed over through LDS: + residual, ReLU, 2x2 pool, split-3, stores).
// One workgroup per CU (126-141 KB LDS, 12 waves x 168 VGPRs), persistent over a strided list of tiles, so neither the
// prologue (first halo tile) nor the epilogue of a tile is exposed: both overlap the MFMAs of the neighbouring tiles.
// Barriers: one per channel group plus one per tile (accumulator hand-over); they are plain s_barrier after an LDS-only
// wait, so the compute waves' weight prefetches stay in flight across them.
#include <type_traits>

#include "pmp_kernels.h"
#include "split3.h"

namespace pmp {

namespace {

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct TileCoord { int n, ty, tx; };

__device__ __forceinline__ unsigned long long ws_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

}  // namespace

template <int KH, int KW, bool STAMP, int ABL = 0>
__global__ __launch_bounds__(768, 1) void conv_x6_ws_kernel(ConvX6Args a, int ntiles)
{
    typedef GeoX<KH, KW> G;
    constexpr int NT = 4, NTW = 2, T = G::TAPS, HP = (T - 1) / 2;   // NT cout groups per tile, NTW of them per compute wave
    static_assert((T & 1) == 1 && (HP & 1) == 0, "tap pairing below assumes an even number of in-group K-steps");
    __shared__ u32x4 xbuf[2 * G::PIECES];
    __shared__ f32x4 accbuf[16 * 256];   // [cout/4][pixel]: one tile's accumulators on their way to the loader waves

    const int tid = threadIdx.x, lane = tid & 63, xl = lane & 15, g = lane >> 4;
    const int lt = tid & 255;                               // loader thread index (tid - 512)
    const int rg = (tid >> 6) & 3, ch = (tid >> 8) & 1;     // compute wave: rows 4rg..4rg+3, cout groups 2ch, 2ch+1
    const int H = a.H, W = a.W, CB = a.Cin >> 4;
    const int tiles_x = W >> 4, tiles = tiles_x * (H >> 4);
    const size_t grp_sz = (size_t)H * W * 16;
    auto coord = [&](int tile) {
        TileCoord c;
        c.n = tile / tiles;
        const int t = tile - c.n * tiles;
        c.ty = t / tiles_x;
        c.tx = t - c.ty * tiles_x;
        return c;
    };
    int par = 0;

    if (tid < 512) {
        // ================================================================================== compute waves
        f32x4 acc[4][NTW];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bf16x8 *wl = reinterpret_cast<const bf16x8 *>(a.w) + lane + ch * NTW * 64;
        const int last = (CB / 2) * T - 1;   // last K-step of the weight stream; it wraps, the weights are per launch
        bf16x8 w0[NTW], w1[NTW], w2[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) { w2[nt] = wl[(2 * NT + nt) * 64]; w1[nt] = wl[(1 * NT + nt) * 64]; w0[nt] = wl[(0 * NT + nt) * 64]; }
        const int pb = ((rg * 4 * G::TW + xl) * 2 + (g & 1)) * 16;   // bytes inside a split plane, tap (0,0)
        int stream = 0;
        int tapsel = g >> 1;
        bf16x8 xa[4], xb[4], x1[4], x2[4];   // x0 fragments alternate between xa (even K-steps) and xb (odd)
        constexpr int O_LAST = (((T - 1) / KW) * G::TW + (T - 1) % KW) * 32;

        // K-steps of one channel group.  MODE 1 = even group (HP in-group tap pairs, then the deferred last tap is picked
        // up into lanes g < 2 of the fragment registers before the barrier), MODE 2 = odd group (cross-group pair first:
        // lanes g >= 2 read this group's last tap, lanes g < 2 still hold the even group's; then HP in-group pairs).
        auto group = [&](auto mode_tag, const char *buf) {
            constexpr int MODE = decltype(mode_tag)::value;
            constexpr int NK = MODE == 1 ? HP : HP + 1;
            auto xaddr = [&](int ks) -> const char * {   // in-group pair of K-step ks
                const int j = MODE == 1 ? ks : ks - 1;
                const int tA = 2 * j, tB = 2 * j + 1;
                const int oA = ((tA / KW) * G::TW + tA % KW) * 32, oB = ((tB / KW) * G::TW + tB % KW) * 32;
                return buf + (tapsel ? oB : oA) + pb;
            };
            if (MODE == 1) {
                const char *p0 = xaddr(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) xa[m] = *reinterpret_cast<const bf16x8 *>(p0 + m * G::TW * 32);
            } else if (g >= 2) {
                const char *p0 = buf + O_LAST + pb;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    xa[m] = *reinterpret_cast<const bf16x8 *>(p0 + m * G::TW * 32);
                    x1[m] = *reinterpret_cast<const bf16x8 *>(p0 + G::PLANE * 16 + m * G::TW * 32);
                    x2[m] = *reinterpret_cast<const bf16x8 *>(p0 + 2 * G::PLANE * 16 + m * G::TW * 32);
                }
            }
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                asm volatile("" : "+v"(tapsel));   // keeps hipcc from hoisting every K-step's tap offset out of the loop
                stream = stream == last ? 0 : stream + 1;
                const bf16x8 *wk = wl + (size_t)stream * (3 * NT * 64);
                bf16x8 (&x0)[4] = (ks & 1) ? xb : xa;
                bf16x8 (&x0n)[4] = (ks & 1) ? xa : xb;
                if (!(MODE == 2 && ks == 0) && (!(ABL & 4) || (MODE == 1 && ks == 0))) {
                    const char *px = xaddr(ks);
#pragma unroll
                    for (int m = 0; m < 4; ++m) x1[m] = *reinterpret_cast<const bf16x8 *>(px + G::PLANE * 16 + m * G::TW * 32);
#pragma unroll
                    for (int m = 0; m < 4; ++m) x2[m] = *reinterpret_cast<const bf16x8 *>(px + 2 * G::PLANE * 16 + m * G::TW * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
                // phase A: x0*w2
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[nt], x0[m], acc[m][nt], 0, 0, 0);
                if (!(ABL & 2)) {
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) w2[nt] = wk[(2 * NT + nt) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                // phase B: x0*w1, x1*w1
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[nt], x0[m], acc[m][nt], 0, 0, 0);
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[nt], x1[m], acc[m][nt], 0, 0, 0);
                    }
                if (!(ABL & 2)) {
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) w1[nt] = wk[(1 * NT + nt) * 64];
                }
                if (ks + 1 < NK && !(ABL & 4)) {
                    const char *pn = xaddr(ks + 1);
#pragma unroll
                    for (int m = 0; m < 4; ++m) x0n[m] = *reinterpret_cast<const bf16x8 *>(pn + m * G::TW * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
                // phase C: x0*w0, x1*w0, x2*w0
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[nt], x0[m], acc[m][nt], 0, 0, 0);
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[nt], x1[m], acc[m][nt], 0, 0, 0);
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[nt], x2[m], acc[m][nt], 0, 0, 0);
                    }
                if (!(ABL & 2)) {
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) w0[nt] = wk[(0 * NT + nt) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 1 && g < 2) {   // deferred last tap of the even group: its buffer is overwritten during the odd group
                const char *p0 = buf + O_LAST + pb;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    xa[m] = *reinterpret_cast<const bf16x8 *>(p0 + m * G::TW * 32);
                    x1[m] = *reinterpret_cast<const bf16x8 *>(p0 + G::PLANE * 16 + m * G::TW * 32);
                    x2[m] = *reinterpret_cast<const bf16x8 *>(p0 + 2 * G::PLANE * 16 + m * G::TW * 32);
                }
            }
        };

        unsigned long long t_bar = 0, t_hand = 0, t0 = 0, tm = 0;   // diagnostic build only
        auto chunk_barrier = [&]() {
            if (STAMP) tm = ws_stamp();
            lds_barrier(); par ^= 1;
            if (STAMP) t_bar += ws_stamp() - tm;
        };
        if (STAMP) t0 = ws_stamp();
        lds_barrier();   // group 0 of the first tile is staged
        const unsigned long long t1 = STAMP ? ws_stamp() : 0;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            for (int cb = 0; cb < CB; cb += 2) {
                group(std::integral_constant<int, 1>{}, reinterpret_cast<const char *>(xbuf + par * G::PIECES));
                chunk_barrier();
                group(std::integral_constant<int, 2>{}, reinterpret_cast<const char *>(xbuf + par * G::PIECES));
                chunk_barrier();
            }
            if (STAMP) tm = ws_stamp();
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    accbuf[((ch * NTW + nt) * 4 + g) * 256 + (rg * 4 + m) * 16 + xl] = acc[m][nt];
                    acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            lds_barrier();   // hand-over: the loader waves own accbuf until the end of the next tile's last group
            if (STAMP) t_hand += ws_stamp() - tm;
        }
        if (STAMP && a.dbg && (tid == 0 || tid == 256)) {   // the two compute waves of SIMD 0
            unsigned long long *d = a.dbg + (size_t)blockIdx.x * 8 + (tid >> 8) * 4;
            const unsigned long long t2 = ws_stamp();
            d[0] = t1 - t0; d[1] = t2 - t1; d[2] = t_bar; d[3] = t_hand;
        }
    } else {
        // ================================================================================== loader waves
        constexpr int PY = KH / 2, PX = KW / 2;
        // Two register sets: the pieces of group q+2 are requested while group q+1 (requested one step earlier, so long
        // arrived) is written to LDS - an HBM round trip is hidden behind a whole group of MFMAs, not a fraction of one.
        unsigned off[G::NLD];
        u32x4 ra[G::NLD], rb[G::NLD];
        unsigned va = 0, vb = 0, vplan = 0;
        const unsigned short *grp0 = nullptr;
        auto make_plan = [&](const TileCoord &c) {
            vplan = 0;
#pragma unroll
            for (int k = 0; k < G::NLD; ++k) {
                const int i = min(lt + k * 256, G::PIECES - 1);
                const int sp = i / G::PLANE, j = i - sp * G::PLANE, pix = j >> 1, half = j & 1;
                const int row = pix / G::TW, col = pix - row * G::TW;
                const int gy = c.ty * 16 + row - PY, gx = c.tx * 16 + col - PX;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) vplan |= 1u << k;
                const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
                off[k] = (unsigned)(sp * a.x_stride + ((size_t)cy * W + cx) * 16 + half * 8);
            }
            grp0 = a.x + (size_t)c.n * CB * grp_sz;
        };
        auto stage_load = [&](int cb, u32x4 (&r)[G::NLD], unsigned &v) {
            const unsigned short *grp = grp0 + (size_t)cb * grp_sz;
#pragma unroll
            for (int k = 0; k < G::NLD; ++k) r[k] = *reinterpret_cast<const u32x4 *>(grp + off[k]);
            v = vplan;
        };
        auto stage_store = [&](u32x4 *dst, const u32x4 (&r)[G::NLD], unsigned v) {
#pragma unroll
            for (int k = 0; k < G::NLD; ++k) {
                const int i = lt + k * 256;
                const u32x4 z = {0u, 0u, 0u, 0u};
                if (i < G::PIECES) dst[i] = ((v >> k) & 1u) ? r[k] : z;
            }
        };
        // 1/CB of a finished tile's epilogue: accumulators from LDS, + residual, ReLU, (2x2 max), split-3, store
        auto ep_slice = [&](const TileCoord &c, int slice) {
            if (!a.pool) {
                const int per = 4096 / CB;
                for (int it = slice * per + lt; it < (slice + 1) * per; it += 256) {
                    const int cg = it >> 8, pix = it & 255;
                    const int y = c.ty * 16 + (pix >> 4), x = c.tx * 16 + (pix & 15);
                    const size_t off = ((size_t)c.n * NT + (cg >> 2)) * grp_sz + ((size_t)y * W + x) * 16 + (cg & 3) * 4;
                    f32x4 v = accbuf[it];
                    if (a.res) v += load_split4(a.res + off, a.res_stride);
                    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    store_split4(a.out + off, a.out_stride, v);
                }
            } else {
                const int per = 1024 / CB, Ho = H >> 1, Wo = W >> 1;
                for (int it = slice * per + lt; it < (slice + 1) * per; it += 256) {
                    const int cg = it >> 6, pp = it & 63, py = pp >> 3, qx = pp & 7;
                    f32x4 best = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int ry = 2 * py + (d >> 1), rx = 2 * qx + (d & 1);
                        f32x4 v = accbuf[cg * 256 + ry * 16 + rx];
                        if (a.res) {
                            const size_t off = ((size_t)c.n * NT + (cg >> 2)) * grp_sz + ((size_t)(c.ty * 16 + ry) * W + c.tx * 16 + rx) * 16 + (cg & 3) * 4;
                            v += load_split4(a.res + off, a.res_stride);
                        }
                        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        if (d == 0) best = v;
                        else { best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w); }
                    }
                    const size_t off = (((size_t)c.n * NT + (cg >> 2)) * Ho + c.ty * 8 + py) * Wo * 16 + (size_t)(c.tx * 8 + qx) * 16 + (cg & 3) * 4;
                    store_split4(a.out + off, a.out_stride, best);
                }
            }
        };

        // Flat sequence of channel groups q = 0 .. my_tiles*CB-1 (CB is even, so q and the group index share parity).
        const int first = blockIdx.x, stride = gridDim.x;
        const int my_tiles = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;
        const int Q = my_tiles * CB;
        int ld_tile = first, ld_cb = 0;   // next group to request
        auto request = [&](u32x4 (&r)[G::NLD], unsigned &v) {
            if (ld_tile >= ntiles) return;
            if (ld_cb == 0) make_plan(coord(ld_tile));
            stage_load(ld_cb, r, v);
            if (++ld_cb == CB) { ld_cb = 0; ld_tile += stride; }
        };
        if (Q > 0) {
            request(ra, va);
            stage_store(xbuf, ra, va);
            request(rb, vb);   // group 1
        }
        lds_barrier();
        TileCoord prev = {0, 0, 0};
        bool have_prev = false;
        int tile = first;
        // step(q): request group q+2 into `rn`, run a slice of the previous tile's epilogue, store group q+1 from `ro`
        auto step = [&](int cb, u32x4 (&rn)[G::NLD], unsigned &vn, const u32x4 (&ro)[G::NLD], unsigned vo, bool store_next) {
            // order matters: hipcc's waits are vmcnt(0) here (runtime trip counts), so everything that is waited for must be
            // a step old - store the old group first, only then issue this step's loads and stores
            if (store_next) stage_store(xbuf + (par ^ 1) * G::PIECES, ro, vo);
            request(rn, vn);
            if (have_prev) ep_slice(prev, cb);
            lds_barrier(); par ^= 1;
        };
        for (int t = 0; t < my_tiles; ++t, tile += stride) {
            for (int cb = 0; cb < CB; cb += 2) {
                const int q = t * CB + cb;
                step(cb, ra, va, rb, vb, q + 1 < Q);
                step(cb + 1, rb, vb, ra, va, q + 2 < Q);
            }
            lds_barrier();   // accbuf now holds `tile`
            prev = coord(tile); have_prev = true;
        }
        if (have_prev)
            for (int cb = 0; cb < CB; ++cb) ep_slice(prev, cb);
    }
}

bool conv_x6_ws_eligible(const ConvX6Args &a)
{
    const int CB = a.Cin >> 4;
    return a.Cout == 64 && !a.x_sc && !a.gate && !a.out_f32 && a.out && (CB & 1) == 0 && CB <= 16 && a.KH == a.KW && (a.KH == 3 || a.KH == 5);
}

hipError_t launch_conv_x6_ws(hipStream_t s, const ConvX6Args &a, int num_cus)
{
    const int ntiles = a.N * (a.H >> 4) * (a.W >> 4);
    const int grid = ntiles < num_cus ? ntiles : num_cus;
    if (a.KH == 3 && g_conv_variant >= 4 && g_conv_variant <= 6) {   // timing-only ablations: 4 no weight refills, 5 no fragment reads, 6 neither
        if (g_conv_variant == 4) hipLaunchKernelGGL((conv_x6_ws_kernel<3, 3, false, 2>), dim3(grid), dim3(768), 0, s, a, ntiles);
        else if (g_conv_variant == 5) hipLaunchKernelGGL((conv_x6_ws_kernel<3, 3, false, 4>), dim3(grid), dim3(768), 0, s, a, ntiles);
        else hipLaunchKernelGGL((conv_x6_ws_kernel<3, 3, false, 6>), dim3(grid), dim3(768), 0, s, a, ntiles);
    } else if (a.dbg) {   // diagnostic build with in-kernel stamps (tools/conv_x6_bench.py ws)
        if (a.KH == 3) hipLaunchKernelGGL((conv_x6_ws_kernel<3, 3, true>), dim3(grid), dim3(768), 0, s, a, ntiles);
        else hipLaunchKernelGGL((conv_x6_ws_kernel<5, 5, true>), dim3(grid), dim3(768), 0, s, a, ntiles);
    } else if (a.KH == 3) hipLaunchKernelGGL((conv_x6_ws_kernel<3, 3, false>), dim3(grid), dim3(768), 0, s, a, ntiles);
    else hipLaunchKernelGGL((conv_x6_ws_kernel<5, 5, false>), dim3(grid), dim3(768), 0, s, a, ntiles);
    return hipGetLastError();
}

}  // namespace pmp
