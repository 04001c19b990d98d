// rbfuse32.hip — a whole ResidualBlock of the HBM-bound 32x32 layers as ONE launch (f16x3 datapath; round 5).
//
// trunk_B3.1 (32 -> 16), trunk_B3.2 (16 -> 8, + max_pool2d) and trunk_Att2.0 (3 -> 32) of the MTT nets (Model_QBD.py:118,125,149-151) run at
// 32x32 with 32 output channels or fewer: launch per layer they move their input twice (first convolution, 1x1 shortcut) and their
// intermediate twice (written, read) at the rate a device copy sustains - bytes are their time.  Here one workgroup owns a 16x16 OUTPUT
// tile of a block and keeps everything between the block's input and its output in LDS:
//
//   input, 20x20 px halo image per 16-channel group (2 px of halo: two 3x3 convolutions deep)      global -> LDS, zero outside the map
//   t = relu(conv3x3(in) / S0) on the 18x18 px the second convolution reads (324 px = 21 MFMA columns-of-16; the ring outside the tile is
//       RECOMPUTED, x1.27 of a convolution that is far from filling the matrix pipes; t = 0 outside the map: zero padding)   -> LDS
//   out = relu((conv3x3(t) + conv1x1(in)) / S2) [2x2 max-pool]                                         -> global (split-2 planes | pooled fp32)
//
// Traffic per block and ResidualBlock: 1.56 x input + output instead of 2 x input + 2 x intermediate + output (B3.1: 270 KB for 457,
// B3.2: 118 for 276, Att2.0: 131 for 588 - its input, three channels made of logits, is built in the kernel and never exists in HBM).  BIT-IDENTICAL to the launch-per-layer path (tests/test_gpu_parity.py): the same pack_h2 weight
// streams in the same K-step order (chain16_dev.h's list, restated for these image widths), the same three products per K-step into the
// same fp32 accumulators (x0*w1, x0*w0, x1*w0), the main pass before the shortcut pass, the epilogue arithmetic of conv_f16x3.hip
// (x 1/S, ReLU, range-flag maximum, pool, two-term split with the clamp).  A value of t that two tiles both compute is the same number
// in both.  LDS 46-72 KB: two workgroups of eight waves per CU.
#include "pmp_kernels.h"
#include "split3.h"

namespace pmp {

namespace {

#define RF_GLOBAL __attribute__((address_space(1)))
constexpr int RF_IW = 20, RF_TW = 18;                        // image widths: input (tile + 2 x 2), intermediate (tile + 2 x 1)
constexpr int RF_IPLN = RF_IW * RF_IW * 32, RF_ISLOT = 2 * RF_IPLN;    // bytes per fp16 plane / per 16-channel group (both planes)
constexpr int RF_TPLN = RF_TW * RF_TW * 32, RF_TSLOT = 2 * RF_TPLN;
constexpr int RF_GRID = 512;                                 // persistent workgroups: 2 per CU x 256 CUs
constexpr int RF_NT1 = (RF_TW * RF_TW + 15) / 16;            // 21 pixel columns-of-16 cover the 18x18 region of the first convolution

// pack_h2's K-step list (pack.cpp; chain16_dev.h: c16_step_off) for an image of width IMW and group slots of SLOT bytes:
// byte offset (group + tap) of K-half `half` of step `st`.  T = 1: the offset of the tap is in the caller's pixel base.
template <int T, int CB, int IMW, int SLOT>
__device__ __forceinline__ constexpr int rf_step_off(int st, int half)
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
    return cb * SLOT + (T == 1 ? 0 : ((tap / 3) * IMW + tap % 3) * 32);
}

template <int T, int CB>
struct RfSteps {
    static constexpr bool paired = !(CB & 1) && (T & 1);
    static constexpr int NS = paired ? (CB / 2) * T : CB * ((T + 1) / 2);
    static constexpr int D = NS < 2 ? NS : 2;               // K-steps of weight lead
};

// One pass over CB source groups into KI accumulators of this wave (KI pixel columns-of-16 x ONE output group `ct` of NT).  pb[k]: this
// lane's byte offset of item k's window origin inside a group plane (+ 16 for the upper 8 channels); the LAST item is skipped unless
// `last` (wave-uniform: the 21 columns-of-16 of the first convolution do not divide evenly among the waves).  Order per accumulator: x0*w1, x0*w0, x1*w0, K-step after K-step - the launch path's.
template <int T, int CB, int NT, int KI, int IMW, int SLOT, int PLN>
__device__ __forceinline__ void rf_accumulate(const char *src, const unsigned short *wpk, int lane, int ct, const int (&pb)[KI], bool last, f32x4 (&acc)[KI])
{
    typedef RfSteps<T, CB> ST;
    constexpr int NS = ST::NS, D = ST::D;
    const int g = lane >> 4;
    const bool hi = (g >> 1) != 0;
    const RF_GLOBAL f16x8 *wl = (const RF_GLOBAL f16x8 *)wpk + lane + ct * 64;
    f16x8 wq[D][2];
#pragma unroll
    for (int st = 0; st < D; ++st) { wq[st][0] = wl[(size_t)st * (2 * NT * 64)]; wq[st][1] = wl[(size_t)st * (2 * NT * 64) + NT * 64]; }
    constexpr bool DB = false && KI <= 3;                             // pixel fragments of the next K-step on their way during this one's MFMAs (registers permitting)
    f16x8 xq[DB ? 2 : 1][2][KI];
    auto xload = [&](int st) __attribute__((always_inline)) {
        int off = hi ? rf_step_off<T, CB, IMW, SLOT>(st, 1) : rf_step_off<T, CB, IMW, SLOT>(st, 0);
        asm volatile("" : "+v"(off));      // opaque: keeps hipcc from hoisting the addresses of every K-step and item out of the pass (54 live registers)
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            if (k < KI - 1 || last) {
                xq[DB ? st & 1 : 0][0][k] = *reinterpret_cast<const f16x8 *>(src + pb[k] + off);
                xq[DB ? st & 1 : 0][1][k] = *reinterpret_cast<const f16x8 *>(src + pb[k] + off + PLN);
            }
        }
    };
    if (DB) xload(0);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        __builtin_amdgcn_sched_barrier(0);
        if (DB) { if (st + 1 < NS) xload(st + 1); }
        else xload(st);
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 w0 = wq[st % D][0], w1 = wq[st % D][1];
        f16x8 (&xa)[KI] = xq[DB ? st & 1 : 0][0], (&xb)[KI] = xq[DB ? st & 1 : 0][1];
#pragma unroll
        for (int k = 0; k < KI; ++k) if (k < KI - 1 || last) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xa[k], acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < KI; ++k) if (k < KI - 1 || last) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xa[k], acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < KI; ++k) if (k < KI - 1 || last) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xb[k], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + D < NS) {
            wq[st % D][0] = wl[(size_t)(st + D) * (2 * NT * 64)];
            wq[st % D][1] = wl[(size_t)(st + D) * (2 * NT * 64) + NT * 64];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

struct RbFuse32Dev {
    const unsigned short *x; size_t x_stride;
    const unsigned short *w0, *w2, *wsc;
    float s0, s2;
    unsigned short *out; size_t out_stride;
    float *out_f32;
    unsigned *sat;
    int H, W;
    const float *q, *bt, *dire;       // ATT: the attention trunk's input is built here, from the logits (conv_misc.hip: att_input_kernel)
    int layer;
};

// CB_IN input groups, NT output groups (= groups of the intermediate), POOLF: 2x2 max-pool and plain fp32 output (trunk_B3.2)
// ATT (trunk_Att2.0): no input tensor at all - cat[up(q), up(bt[layer]), up(dire[layer])] (Model_QBD.py:147), three channels of a 16-channel
// group, is computed from the logits into the halo image, with att_input_kernel's arithmetic (two-term split, clamp, range flag).
template <int CB_IN, int NT, bool POOLF, bool ATT = false>
__global__ __launch_bounds__(512, 4) void rbfuse32_kernel(RbFuse32Dev a, int total)
{
    __shared__ __attribute__((aligned(16))) char img[CB_IN * RF_ISLOT];
    __shared__ __attribute__((aligned(16))) char timg[NT * RF_TSLOT];
    int tid = threadIdx.x;
    const int tiles_x = a.W >> 4, tiles = tiles_x * (a.H >> 4);
    const int H = a.H, W = a.W;
    // PERSISTENT: a workgroup walks a contiguous run of tiles (the tiles of a block - they share halo rows and columns - stay on one
    // workgroup, hence on one XCD's L2) and requests tile i+1's input into registers before it starts on tile i's convolutions: the first
    // form of this kernel (one tile per workgroup: load, wait, compute, store) spent more time waiting for its 51 KB than computing.
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = (int)blockIdx.x * per, t_end = min(t_begin + per, total);
    if (t_begin >= t_end) return;

    float amax = 0.f;
    constexpr int PIECES = CB_IN * 2 * RF_IW * RF_IW * 2, NLD = ATT ? 1 : (PIECES + 511) / 512;
    unsigned inmask = 0;
    u32x4 r[NLD];          // the next tile's input on its way (ATT: the three logit values of this thread's pixel, and whether it is inside the map)
    auto fetch = [&](int tl) __attribute__((always_inline)) {
        const int n = tl / tiles, t = tl - n * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
        if (ATT) {
            const int px = min(tid, RF_IW * RF_IW - 1);
            const int row = px / RF_IW, col = px - row * RF_IW, gy = ty * 16 - 2 + row, gx = tx * 16 - 2 + col;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            const int sq = H / 8, sh = H / 16;
            const size_t o = ((size_t)n * 3 + a.layer) * 256 + (cy / sh) * 16 + (cx / sh);
            r[0] = (u32x4){__float_as_uint(a.q[(size_t)n * 64 + (cy / sq) * 8 + (cx / sq)]), __float_as_uint(a.bt[o]), __float_as_uint(a.dire[o]), in ? 1u : 0u};
        } else {
            inmask = 0;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int i = min(tid + k * 512, PIECES - 1);
                const int half = i & 1, pix = (i >> 1) % (RF_IW * RF_IW), sp = ((i >> 1) / (RF_IW * RF_IW)) & 1, cb = (i >> 1) / (2 * RF_IW * RF_IW);
                const int row = pix / RF_IW, col = pix - row * RF_IW, gy = ty * 16 - 2 + row, gx = tx * 16 - 2 + col;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
                r[k] = *reinterpret_cast<const u32x4 *>(a.x + sp * a.x_stride + (((size_t)n * CB_IN + cb) * H + cy) * W * 16 + (size_t)cx * 16 + half * 8);
                if (in) inmask |= 1u << k;     // applied when the data is used (stash): nothing here waits for the loads
            }
        }
    };
    // registers -> LDS halo images (zero outside the map)
    auto stash = [&]() __attribute__((always_inline)) {
        if (ATT) {
            if (tid < RF_IW * RF_IW) {
                u32x4 lo = {0u, 0u, 0u, 0u}, hi = {0u, 0u, 0u, 0u};
                if (r[0].w) {
                    const f32x4 v = {__uint_as_float(r[0].x), __uint_as_float(r[0].y), __uint_as_float(r[0].z), 0.f};
                    amax = sat_amax4(amax, v);
                    unsigned p0, q0, p1, q1;
                    h2_split_pair(v.x, v.y, p0, q0);
                    h2_split_pair(v.z, v.w, p1, q1);
                    lo.x = p0; lo.y = p1; hi.x = q0; hi.y = q1;
                }
                const u32x4 z = {0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4 *>(img + tid * 32) = lo;
                *reinterpret_cast<u32x4 *>(img + tid * 32 + 16) = z;
                *reinterpret_cast<u32x4 *>(img + RF_IPLN + tid * 32) = hi;
                *reinterpret_cast<u32x4 *>(img + RF_IPLN + tid * 32 + 16) = z;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int i = tid + k * 512;
                if (i < PIECES) {
                    const int half = i & 1, pix = (i >> 1) % (RF_IW * RF_IW), sp = ((i >> 1) / (RF_IW * RF_IW)) & 1, cb = (i >> 1) / (2 * RF_IW * RF_IW);
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4 *>(img + cb * RF_ISLOT + sp * RF_IPLN + pix * 32 + half * 16) = ((inmask >> k) & 1u) ? r[k] : z;
                }
            }
        }
    };
    static_assert(!ATT || CB_IN == 1, "the attention input is one channel group");
    fetch(t_begin);
  for (int tl = t_begin; tl < t_end; ++tl) {
    // opaque per tile: otherwise hipcc hoists every lane-derived address of the loop body (piece decomposition, window origins, weight
    // pointers: ~100 registers of loop invariants) in front of the loop and spills them
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6, xl = lane & 15, g = lane >> 4;
    const int n = tl / tiles, t = tl - n * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
    stash();
    __syncthreads();
    fetch(min(tl + 1, t_end - 1));        // (the last tile is requested twice: the loads stay unconditional)

    const int ct = wave % NT, wsub = wave / NT;
    constexpr int WPG = 8 / NT;                                   // waves per output group
    // ---- first convolution on the 18x18 region: wave (ct, wsub) takes the columns-of-16 wsub, wsub + WPG, ...
    {
        constexpr int KI = (RF_NT1 + WPG - 1) / WPG;
        static_assert((KI - 1) * WPG < RF_NT1, "only a wave's last item may fall outside the region");
        int pb[KI], pix[KI];
        const bool last = wsub + (KI - 1) * WPG < RF_NT1;
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int tile = wsub + k * WPG;
            const int p = min(tile * 16 + xl, RF_TW * RF_TW - 1);
            pix[k] = tile * 16 + xl;
            pb[k] = ((p / RF_TW) * RF_IW + p % RF_TW) * 32 + (g & 1) * 16;
        }
        f32x4 acc[KI];
#pragma unroll
        for (int k = 0; k < KI; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rf_accumulate<9, CB_IN, NT, KI, RF_IW, RF_ISLOT, RF_IPLN>(img, a.w0, lane, ct, pb, last, acc);
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            if ((k == KI - 1 && !last) || pix[k] >= RF_TW * RF_TW) continue;
            const int py = pix[k] / RF_TW, px = pix[k] - py * RF_TW, gy = ty * 16 - 1 + py, gx = tx * 16 - 1 + px;
            f32x4 v = acc[k] * a.s0;
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) v = (f32x4){0.f, 0.f, 0.f, 0.f};      // the second convolution's zero padding
            amax = sat_amax4(amax, v);
            unsigned p0, q0, p1, q1;
            h2_split_pair(v.x, v.y, p0, q0);
            h2_split_pair(v.z, v.w, p1, q1);
            char *dp = timg + ct * RF_TSLOT + pix[k] * 32 + g * 8;
            *reinterpret_cast<u32x2_t *>(dp) = (u32x2_t){p0, p1};
            *reinterpret_cast<u32x2_t *>(dp + RF_TPLN) = (u32x2_t){q0, q1};
        }
    }
    __syncthreads();

    // ---- second convolution + 1x1 shortcut on the 16x16 tile: wave (ct, wsub) takes the KI consecutive rows KI * wsub ..
    {
        constexpr int KI = 16 / WPG;
        int pb[KI], pbs[KI];
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int row = wsub * KI + k;
            pb[k] = (row * RF_TW + xl) * 32 + (g & 1) * 16;
            pbs[k] = ((row + 2) * RF_IW + xl + 2) * 32 + (g & 1) * 16;
        }
        f32x4 acc[KI];
#pragma unroll
        for (int k = 0; k < KI; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rf_accumulate<9, NT, NT, KI, RF_TW, RF_TSLOT, RF_TPLN>(timg, a.w2, lane, ct, pb, true, acc);
        rf_accumulate<1, CB_IN, NT, KI, RF_IW, RF_ISLOT, RF_IPLN>(img, a.wsc, lane, ct, pbs, true, acc);       // ResidualBlock, Model_QBD.py:33-38
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            f32x4 v = acc[k] * a.s2;
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            if (!POOLF) amax = sat_amax4(amax, v);      // a plain fp32 output is not clamped: not the range flag's business (conv_f16x3.hip)
            acc[k] = v;
        }
        const int row0 = wsub * KI;
        if (POOLF) {
            const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
            for (int k = 0; k < KI; k += 2) {
                f32x4 v = acc[k], u = acc[k + 1];
                v.x = fmaxf(v.x, u.x); v.y = fmaxf(v.y, u.y); v.z = fmaxf(v.z, u.z); v.w = fmaxf(v.w, u.w);
                f32x4 o;
                o.x = __shfl_xor(v.x, 1); o.y = __shfl_xor(v.y, 1); o.z = __shfl_xor(v.z, 1); o.w = __shfl_xor(v.w, 1);
                v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
                if ((xl & 1) == 0) {
                    const int yo = ty * 8 + ((row0 + k) >> 1), xo = tx * 8 + (xl >> 1);
                    *reinterpret_cast<f32x4 *>(a.out_f32 + ((((size_t)n * NT + ct) * Ho + yo) * Wo + xo) * 16 + g * 4) = v;
                }
            }
        } else {
            // 16-byte stores as conv_f16x3.hip's epilogue: one v_permlane16_swap per register turns {rows m, m+1} x {couts 4g..} into the 8
            // consecutive channels 8(g>>1).. of row m + (g&1)
#pragma unroll
            for (int k = 0; k < KI; k += 2) {
                u32x4 p, q;
                split2_rows(acc[k], acc[k + 1], p, q);
                rows16_swap(p);
                rows16_swap(q);
                const int gy = ty * 16 + row0 + k + (g & 1), gx = tx * 16 + xl;
                unsigned short *op = a.out + ((((size_t)n * NT + ct) * H + gy) * W + gx) * 16 + 8 * (g >> 1);
                __builtin_nontemporal_store(p, reinterpret_cast<u32x4 *>(op));
                __builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(op + a.out_stride));
            }
        }
    }
    __syncthreads();      // every wave is done with this tile's images: the next tile's input may land
  }
    sat_report(a.sat, amax);
}

}  // namespace

hipError_t launch_rbfuse32(hipStream_t s, const RbFuse32Args &h)
{
    if ((h.H & 15) || (h.W & 15) || h.N <= 0) return hipErrorInvalidValue;
    RbFuse32Dev a{h.x, h.x_stride, h.w0, h.w2, h.wsc, h.s0, h.s2, h.out, h.out_stride, h.out_f32, h.sat, h.H, h.W, h.q, h.bt, h.dire, h.att_layer};
    const int total = h.N * ((h.H >> 4) * (h.W >> 4));
    const unsigned grid = (unsigned)(total < RF_GRID ? total : RF_GRID);       // persistent: two workgroups per CU, each a contiguous run of tiles
    if (!h.x) {
        if (!h.q || !h.bt || !h.dire || h.cin_groups != 1 || h.cout_groups != 2 || h.pool_f32) return hipErrorInvalidValue;
        hipLaunchKernelGGL((rbfuse32_kernel<1, 2, false, true>), dim3(grid), dim3(512), 0, s, a, total);
        return hipGetLastError();
    }
    if (h.cin_groups == 2 && h.cout_groups == 1 && !h.pool_f32) hipLaunchKernelGGL((rbfuse32_kernel<2, 1, false>), dim3(grid), dim3(512), 0, s, a, total);
    else if (h.cin_groups == 1 && h.cout_groups == 1 && h.pool_f32) hipLaunchKernelGGL((rbfuse32_kernel<1, 1, true>), dim3(grid), dim3(512), 0, s, a, total);
    else if (h.cin_groups == 1 && h.cout_groups == 2 && !h.pool_f32) hipLaunchKernelGGL((rbfuse32_kernel<1, 2, false>), dim3(grid), dim3(512), 0, s, a, total);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace pmp
