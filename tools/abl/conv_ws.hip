// abl/conv_ws.hip — MEASUREMENT LIBRARY ONLY (make abl): a WEIGHT-STATIONARY, persistent form of the 3x3 64->64 convolution (f16x3).
// The product kernel (conv_f16x3.hip) streams every wave's weight fragments through the L1 once per tile: 21 GB of L1 traffic per launch
// for 6.6 GB of activations, and a weight fragment (an L1 hit) queued behind a halo request waits for that request's HBM round trip (the
// vector-memory path returns in order).  Here a workgroup is resident for the whole launch (one per CU, 8 waves) and every wave keeps the
// weights of ITS 16 output channels for all 18 K-steps in registers (18 x 2 terms x 4 = 144 registers); the K-loop then has no
// vector-memory operation at all - pixel fragments from LDS, weights from registers - and the halo image of the NEXT tile arrives by
// LDS-DMA (global_load_lds_dwordx4, no staging registers, no LDS stores) while this one is computed.
//   tile      16 x 8 output pixels, all 64 input channels resident: 4 groups x 2 planes x 10 x 18 halo pixels x 32 B = 46 KB (48 KB reserved),
//             two or three buffers (one or two tiles of lookahead)
//   wave      cout group (wave & 3) x row half (wave >> 2): 4 rows x 16 px x 16 couts = 16 accumulator registers; the two waves of a SIMD
//             share a cout group, i.e. hold the same weights.  Every pixel fragment feeds ONE output group (3 MFMAs): 0.67 ds_read_b128
//             per MFMA, twice the product kernel's LDS rate - the price of weights that never move.
//   phases    the row halves are de-phased by where they cross the tile barrier: waves 0-3  K-loop, epilogue, barrier;  waves 4-7  K-loop,
//             barrier, epilogue - so one wave of every SIMD is in its K-loop while the other converts and stores.
// EXACT: pack_h2's K-step list, the products x0*w1, x0*w0, x1*w0 per K-step and accumulator, conv_f16x3.hip's epilogue arithmetic.
// Timing-only switches (ABL, wrong results): 1 no DMA after the first tile, 2 no MFMAs, 4 no epilogue (stores of zeros stay), 8 no residual.
#include "chain16_dev.h"

namespace pmp {

struct ConvWsArgs {
    const unsigned short *x; size_t x_stride;       // [N][4][H][W][16] split-2
    const unsigned short *w; float inv_scale;       // pack_h2 stream of the layer and its 1/S
    const unsigned short *res; size_t res_stride;   // identity residual (same geometry) or nullptr
    unsigned short *out; size_t out_stride;
    int N, H, W;
    unsigned *sat;
    const void *zeros;                              // >= 16 zero bytes in global memory (source of the padding pieces)
};

namespace {

constexpr int WS_TH = 8, WS_HR = WS_TH + 2, WS_HC = 18;
constexpr int WS_NDMA = 6;                          // DMA instructions per lane and tile: wave w fills image (group w >> 1, plane w & 1), 6 x 64 pieces of 16 B
constexpr int WS_ROW = WS_HC * 32, WS_PLN = WS_NDMA * 1024, WS_GRP = 2 * WS_PLN;      // 576 B; a plane's 5760 B padded to 6 KB (24 dummy pieces)
constexpr int WS_PIECES = WS_HR * WS_HC * 2;        // 360 valid pieces per plane image
constexpr int WS_BUF = 8 * WS_PLN;                  // 49 152 B per buffer

// pack_h2's K-step list for 9 taps over 4 groups (chain16_dev.h: c16_step_off<9, 4>) on this tile's image geometry
__device__ __forceinline__ constexpr int ws_step_off(int st, int half)
{
    const int pr = st / 9, j = st % 9;
    int cb = 0, tap = 0;
    if (j < 4) { cb = 2 * pr; tap = 2 * j + half; }
    else if (j == 4) { cb = 2 * pr + half; tap = 8; }
    else { cb = 2 * pr + 1; tap = 2 * (j - 5) + half; }
    return cb * WS_GRP + ((tap / 3) * WS_HC + tap % 3) * 32;
}

__device__ __forceinline__ void ws_dma16(const void *src, unsigned lds_byte_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_byte_base) : "memory");
}

}  // namespace

template <bool RES, int NBUF, int ABL>
__global__ __launch_bounds__(512, 1) void conv_ws_kernel(ConvWsArgs a)
{
    __shared__ __attribute__((aligned(16))) char lds[NBUF * WS_BUF];
    const int tid = threadIdx.x, lane = tid & 63, xl = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), ct = wave & 3, rh = wave >> 2;
    const int H = a.H, W = a.W, tiles_y = H / WS_TH, tpb = tiles_y * (W >> 4);
    const int T = a.N * tpb, G = (int)gridDim.x;
    int b = (int)blockIdx.x;
    if ((G & 7) == 0) b = (b & 7) * (G >> 3) + (b >> 3);       // workgroup ids go round-robin over the XCDs: contiguous tile runs per XCD
    const int t0 = (int)((long long)T * b / G), t1 = (int)((long long)T * (b + 1) / G);
    const size_t grp_sz = (size_t)H * W * 16;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    float amax = 0.f;

    // tile t -> (block, tile row, tile column); tile rows fastest: consecutive tiles of a workgroup share two halo rows in L2
    auto coords = [&](int t, int &n, int &ty, int &tx) __attribute__((always_inline)) {
        n = t / tpb;
        const int r = t - n * tpb;
        tx = r / tiles_y;
        ty = r - tx * tiles_y;
    };
    // halo image of tile t -> buffer `buf`: wave w moves plane w & 1 of group w >> 1, piece j = k * 64 + lane = [row][col][half]; what
    // depends on the lane is worked out once (offset of the piece relative to the tile's origin, which image borders it touches), what
    // depends on the tile is scalar: ~6 vector instructions per piece
    int rel[WS_NDMA];
    unsigned pflags = 0;         // 5 bits per piece: halo row 0, row 9, column 0, column 17, dummy piece
#pragma unroll
    for (int k = 0; k < WS_NDMA; ++k) {
        const int j = k * 64 + lane, jc = min(j, WS_PIECES - 1), pix = jc >> 1, half = jc & 1, row = pix / WS_HC, col = pix - row * WS_HC;
        rel[k] = (((row - 1) * W + (col - 1)) * 16 + half * 8) * 2;
        pflags |= (unsigned)((row == 0) | ((row == WS_HR - 1) << 1) | ((col == 0) << 2) | ((col == WS_HC - 1) << 3) | ((j >= WS_PIECES) << 4)) << (5 * k);
    }
    auto dma_tile = [&](int t, int buf) __attribute__((always_inline)) {
        int n, ty, tx;
        coords(t, n, ty, tx);
        const char *base = reinterpret_cast<const char *>(a.x + (size_t)(wave & 1) * a.x_stride + ((size_t)n * 4 + (wave >> 1)) * grp_sz +
                                                         ((size_t)(ty * WS_TH) * W + tx * 16) * 16);
        const unsigned border = (ty == 0 ? 1u : 0u) | (ty == tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == (W >> 4) - 1 ? 8u : 0u) | 16u;
#pragma unroll
        for (int k = 0; k < WS_NDMA; ++k) {
            const void *src = ((pflags >> (5 * k)) & border) ? a.zeros : (const void *)(base + rel[k]);
            ws_dma16(src, lds_base + (unsigned)(buf * WS_BUF + wave * WS_PLN + k * 1024));
        }
    };

    // ---- the wave's weights: both terms of its 16 output channels for all 18 K-steps
    f16x8 wq[18][2];
    {
        const C16_GLOBAL f16x8 *wl = (const C16_GLOBAL f16x8 *)a.w + lane + ct * 64;
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            wq[st][0] = wl[(size_t)st * (2 * 4 * 64)];
            wq[st][1] = wl[(size_t)st * (2 * 4 * 64) + 4 * 64];
        }
    }
    constexpr int LA = NBUF - 1;      // tiles of lookahead
#pragma unroll
    for (int i = 0; i < LA; ++i)
        if (t0 + i < t1) dma_tile(t0 + i, i);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    const int lane_off = ((rh * 4) * WS_HC + xl) * 32 + (g & 1) * 16;
    const bool hi = (g >> 1) != 0;

    int buf = 0, nbuf = LA % NBUF;        // buffer of tile t, buffer of tile t + LA
    for (int t = t0; t < t1; ++t) {
        int n, ty, tx;
        coords(t, n, ty, tx);
        if (t + LA < t1 && (!(ABL & 1))) dma_tile(t + LA, nbuf);
        // this wave's output rows: element offset of (row 0, cout 4g) in a [n][4][H][W][16] tensor
        const size_t o0 = (((size_t)n * 4 + ct) * H + (ty * WS_TH + rh * 4)) * W * 16 + (size_t)(tx * 16 + xl) * 16 + g * 4;
        u32x2_t ra[4], rb[4];
        if (RES && !(ABL & 8)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const C16_GLOBAL unsigned short *rp = (const C16_GLOBAL unsigned short *)a.res + o0 + (size_t)m * W * 16;
                ra[m] = *reinterpret_cast<const C16_GLOBAL u32x2_t *>(rp);
                rb[m] = *reinterpret_cast<const C16_GLOBAL u32x2_t *>(rp + a.res_stride);
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- K-loop: 18 K-steps x 2 half-steps of two rows; the fragments of the next half-step are read during this one's MFMAs
        const char *pbase = lds + buf * WS_BUF + lane_off;
        f16x8 xq[2][2][2];
        auto xload = [&](int tk) __attribute__((always_inline)) {
            const int st = tk >> 1, h = tk & 1;
            const char *p = pbase + (hi ? ws_step_off(st, 1) : ws_step_off(st, 0)) + h * 2 * WS_ROW;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                xq[tk & 1][0][m] = *reinterpret_cast<const f16x8 *>(p + m * WS_ROW);
                xq[tk & 1][1][m] = *reinterpret_cast<const f16x8 *>(p + WS_PLN + m * WS_ROW);
            }
        };
        xload(0);
#pragma unroll
        for (int tk = 0; tk < 36; ++tk) {
            const int st = tk >> 1, h = tk & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (tk + 1 < 36) xload(tk + 1);
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 w0 = wq[st][0], w1 = wq[st][1];
            f16x8 (&x0)[2] = xq[tk & 1][0], (&x1)[2] = xq[tk & 1][1];
            if (ABL & 2) { asm volatile("" ::"v"(x0[0]), "v"(x0[1]), "v"(x1[0]), "v"(x1[1]), "v"(w0), "v"(w1)); continue; }
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[h * 2 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x0[m], acc[h * 2 + m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[h * 2 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x0[m], acc[h * 2 + m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[h * 2 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x1[m], acc[h * 2 + m], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next tile's image has landed: everything older than this iteration's requests (the image of tile t + LA, the residual) is
        // complete - loads return in order.  The stores below are not waited for.
        if (LA > 1) asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(WS_NDMA + (RES ? 8 : 0)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        auto epilogue = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 v = acc[m];
                if (ABL & 4) { asm volatile("" ::"v"(v)); v = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                else if (RES && !(ABL & 8))
                    v = v * a.inv_scale + (f32x4){h2_sum_lo(ra[m].x, rb[m].x), h2_sum_hi(ra[m].x, rb[m].x), h2_sum_lo(ra[m].y, rb[m].y), h2_sum_hi(ra[m].y, rb[m].y)};
                else v = v * a.inv_scale;
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                amax = sat_amax4(amax, v);
                unsigned p0, q0, p1, q1;
                h2_split_pair(v.x, v.y, p0, q0);
                h2_split_pair(v.z, v.w, p1, q1);
                C16_GLOBAL unsigned short *op = (C16_GLOBAL unsigned short *)a.out + o0 + (size_t)m * W * 16;
                *reinterpret_cast<C16_GLOBAL u32x2_t *>(op) = (u32x2_t){p0, p1};
                *reinterpret_cast<C16_GLOBAL u32x2_t *>(op + a.out_stride) = (u32x2_t){q0, q1};
            }
        };
        if (rh == 0) {
            epilogue();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            epilogue();
        }
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        nbuf = nbuf + 1 == NBUF ? 0 : nbuf + 1;
    }
    sat_report(a.sat, amax);
}

template <bool RES, int NBUF>
static hipError_t launch_ws_t(hipStream_t s, const ConvWsArgs &a, int abl, int grid)
{
#define WS_CASE(V) case V: hipLaunchKernelGGL((conv_ws_kernel<RES, NBUF, V>), dim3(grid), dim3(512), 0, s, a); break;
    switch (abl) {
        WS_CASE(0) WS_CASE(1) WS_CASE(2) WS_CASE(4) WS_CASE(6) WS_CASE(7) WS_CASE(8) WS_CASE(12) WS_CASE(13)
        default: return hipErrorInvalidValue;
    }
#undef WS_CASE
    return hipGetLastError();
}

hipError_t launch_conv_ws(hipStream_t s, const ConvWsArgs &a, int abl, int nbuf, int grid)
{
    if (a.N <= 0 || (a.H % WS_TH) || (a.W & 15) || !a.zeros || grid <= 0) return hipErrorInvalidValue;
    if (nbuf == 3) return a.res ? launch_ws_t<true, 3>(s, a, abl, grid) : launch_ws_t<false, 3>(s, a, abl, grid);
    return a.res ? launch_ws_t<true, 2>(s, a, abl, grid) : launch_ws_t<false, 2>(s, a, abl, grid);
}

}  // namespace pmp
