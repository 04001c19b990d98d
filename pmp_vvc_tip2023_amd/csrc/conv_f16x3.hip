// conv_f16x3.hip — fp32-accurate convolution on the fp16 matrix cores with HALF the MFMAs of the bf16x6 path.
//
// Every fp32 value v is carried as two fp16 values v = h0 + h1 (2 x 11 significand bits = 22).  A product x*w is
//      x0w0 + (x0w1 + x1w0)                                       (the dropped x1w1 is <= 2^-22 relative)
// i.e. three v_mfma_f32_16x16x32_f16 per 32 channels x taps instead of six bf16 ones, accumulated in fp32.  fp16 has only
// 5 exponent bits, so the low terms of small values would fall into the subnormal range and lose bits; two facts make the
// scheme as accurate as fp32 arithmetic itself (tools/precision_study.py: 1.0e-4 on the real Luma_Q_22 logits, bf16x6
// 9.2e-5, fp64-vs-fp32 8.2e-5; without the weight scaling 4.6e-4):
//   * weights are multiplied by a per-launch power of two S (max |S*w| in [4096, 8192)) before the split, so both weight
//     terms are normal numbers for every weight down to 2^-16 of the largest; the epilogue multiplies by 1/S (exact);
//   * activations are O(1)..O(1e3) in these nets (ReLU outputs of 8-bit pixels); an activation's low term is subnormal
//     only below ~0.1, where its absolute error (< 2^-25) is far below the rounding of the sums it enters.  The MFMA
//     honours fp16 subnormals (tools/probe/f16_denorm.hip).  Values beyond +-65504 are clamped when split (split3.h).
//
// Activation format "split-2": two fp16 planes, each blocked channels-last [n][C/16][H][W][16] (32 B per pixel and
// group), plane stride = N*C*H*W elements.  Tiling, LDS image, tap pairing and the weight stream order are those of
// conv_bf16x6.hip (one workgroup = 16x16 pixels x all Cout, K-step = 16 channels x a pair of taps); the wave tile is 8 rows x
// 32 couts at Cout = 64 (WaveTile, split3.h) and the deferred tap of a tap pair crosses the group barrier in registers.
//
// K-step schedule (48 MFMAs at Cout = 64):
//   top: request the NEXT K-step's w0 into a second register set, read this K-step's x1 fragments, issue a slice of the halo
//   phase A: x0*w1 -> request the next w1 into the same registers     phase B1: x0*w0 -> read the next K-step's x0
//   phase B2: x1*w0 -> move the prefetched w0 over
// Measurements: DESIGN.md 4.1, EXPERIMENTS.md.  The forms of this kernel that were built, parity-tested and measured no better (persistent, loader-wave,
// 512-thread, 32x16-tile, Winograd-x, two-workgroup forms) and its timing-only builds live in abl/ (measurement library, `make abl`).
#include <cstdio>
#include <type_traits>

#include "pmp_kernels.h"
#include "split3.h"

namespace pmp {

template <int KH, int KW, int NTHR = 256>
struct StagePlanH {
    unsigned off[GeoH<KH, KW, 16, NTHR>::NLD];
    unsigned valid;
};

template <int KH, int KW, int NTHR = 256>
__device__ __forceinline__ void h2_plan(StagePlanH<KH, KW, NTHR> &p, size_t plane_stride, int H, int W, int ty, int tx)
{
    typedef GeoH<KH, KW, 16, NTHR> G;
    constexpr int PY = KH / 2, PX = KW / 2;
    p.valid = 0;
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
        const int i = min((int)threadIdx.x + k * NTHR, G::PIECES - 1);
        const int sp = i / G::PLANE, j = i - sp * G::PLANE, pix = j >> 1, half = j & 1;
        const int row = pix / G::TW, col = pix - row * G::TW;
        const int gy = ty * 16 + row - PY, gx = tx * 16 + col - PX;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W && (int)threadIdx.x + k * NTHR < G::PIECES;
        if (in) p.valid |= 1u << k;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);   // clamped: loads stay unconditional
        p.off[k] = (unsigned)(sp * plane_stride + ((size_t)cy * W + cx) * 16 + half * 8);
    }
}

template <int KH, int KW, int NTHR = 256>
__device__ __forceinline__ void h2_stage_load(const StagePlanH<KH, KW, NTHR> &p, const unsigned short *__restrict__ grp,
                                              u32x4 (&r)[GeoH<KH, KW, 16, NTHR>::NLD], int k0 = 0, int k1 = 1 << 20)
{
#pragma unroll
    for (int k = 0; k < GeoH<KH, KW, 16, NTHR>::NLD; ++k) {
        if (k < k0 || k >= k1) continue;   // folds away: callers pass constants into unrolled code
        r[k] = *reinterpret_cast<const u32x4 *>(grp + p.off[k]);   // default cache policy: the non-temporal hint measured 1 % slower
    }
}

template <int KH, int KW, int NTHR = 256>
__device__ __forceinline__ void h2_stage_store(const StagePlanH<KH, KW, NTHR> &p, u32x4 *lds, const u32x4 (&r)[GeoH<KH, KW, 16, NTHR>::NLD],
                                               int k0 = 0, int k1 = 1 << 20)
{
    typedef GeoH<KH, KW, 16, NTHR> G;
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
        if (k < k0 || k >= k1) continue;
        const int i = min((int)threadIdx.x + k * NTHR, G::PIECES);   // pieces past the tile all land in one spare slot
        const u32x4 z = {0u, 0u, 0u, 0u};
        lds[i] = ((p.valid >> k) & 1u) ? r[k] : z;   // LDS image: [split][pixel][2 halves], linear
    }
}

template <int KH, int KW, int NT, bool LEAN = false>
__device__ __forceinline__ void h2_accumulate(const unsigned short *__restrict__ x, size_t plane_stride,
                                              const unsigned short *__restrict__ wpk, int C, int H, int W, int n, int ty,
                                              int tx, u32x4 *lds, f32x4 (&acc)[WaveTile<NT>::RW][WaveTile<NT>::CW])
{
    constexpr int NTHR = 256;
    typedef GeoH<KH, KW, 16, NTHR> G;
    typedef WaveTile<NT> WT;
    constexpr int RW = WT::RW, CW = WT::CW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, xl = lane & 15, g = lane >> 4;
    const int rh = wave % WT::RSPLIT, ch = wave / WT::RSPLIT;   // this wave: rows RW*rh.., cout groups CW*ch..
    const int CB = C >> 4;
    const size_t grp_sz = (size_t)H * W * 16;
    const unsigned short *grp0 = x + (size_t)n * CB * grp_sz;
    // Halo staging registers: ONE set, a distance of one group (the next group's tile is requested in slices behind this group's
    // K-steps and written to the partner LDS buffer LAG K-steps later).
    // W0DB: a K-step's w0 fragments are requested one K-step ahead into a second register set and moved over between its
    // phases A and B1 (8 v_mov) - a full K-step of lead for the split that is used last and needed first.
    constexpr bool W0DB = NT == 4;
    constexpr bool W2 = W0DB && G::TAPS == 9 && !LEAN;
    u32x4 r[G::NLD];
    StagePlanH<KH, KW, NTHR> plan;
    // Tap pairing as in conv_bf16x6.hip: mode 0 plain (last pair zero-padded), mode 1 even group of a pair (its last tap
    // is deferred and carried in registers), mode 2 odd group (first K-step = the deferred tap + its own last tap).
    const bool paired = (CB & 1) == 0 && (G::TAPS & 1);
    const f16x8 *wl = reinterpret_cast<const f16x8 *>(wpk) + lane + ch * CW * 64;
    const int last = paired ? (CB / 2) * G::TAPS - 1 : CB * G::NKS - 1;   // last K-step of the weight stream
    f16x8 w0[CW], w1[CW];   // ONE weight set, refilled in place as soon as the last MFMA that reads a split has issued
    f16x8 w0n[CW];          // W0DB: the K-step's w0 fragments land here and move to w0 between its phases A and B1
    // W2: weight fragments are requested TWO K-steps ahead (16 more registers).  The L1 returns data in order for the whole
    // CU, so a weight hit queued behind a halo request that went to HBM - this workgroup's or its neighbour's - waits for
    // it; one K-step of lead (0.8-1.5 k cycles) is less than that round trip, two are more.
    f16x8 w1n[CW], w0nn[CW];
    h2_plan<KH, KW, NTHR>(plan, plane_stride, H, W, ty, tx);
#pragma unroll
    for (int nt = 0; nt < CW; ++nt) {
        if (W0DB) w0n[nt] = wl[(0 * NT + nt) * 64]; else w0[nt] = wl[(0 * NT + nt) * 64];
        w1[nt] = wl[(1 * NT + nt) * 64];
        if (W2) {
            const f16x8 *w2 = wl + (size_t)min(1, last) * (2 * NT * 64);
            w0nn[nt] = w2[(0 * NT + nt) * 64]; w1n[nt] = w2[(1 * NT + nt) * 64];
        }
    }
    __syncthreads();
    h2_stage_load<KH, KW, NTHR>(plan, grp0, r);
    h2_stage_store<KH, KW, NTHR>(plan, lds, r);
    __syncthreads();
    const int pb = ((rh * RW * G::TW + xl) * 2 + (g & 1)) * 16;   // bytes inside a split plane, tap (0,0)
    int stream = 0;
    int tapsel = g >> 1;
    constexpr int O_LAST = (((G::TAPS - 1) / KW) * G::TW + (G::TAPS - 1) % KW) * 32;   // byte offset of the last tap
    f16x8 x0[RW], x1[RW];   // pixel fragments of the current K-step (split 0 / split 1)

    auto group = [&](auto mode_tag, auto tail_tag, int cb) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int TAIL = decltype(tail_tag)::value;     // 2: last channel group of the pass, 1: the one before, 0: any other
        constexpr bool LAST = TAIL == 2;
        constexpr int NK = MODE == 0 ? G::NKS : (MODE == 1 ? (G::TAPS - 1) / 2 : (G::TAPS - 1) / 2 + 1);
        constexpr int PER = (G::NLD + (NK > 0 ? NK : 1) - 1) / (NK > 0 ? NK : 1);   // staging loads issued per K-step
        constexpr int LAG = G::TAPS > 9 ? 3 : 2;             // K-steps between a halo slice's request and its LDS store
        constexpr bool FETCH = TAIL < 2;                      // is there a group cb+1 to request
        u32x4 (&rl)[G::NLD] = r;                              // requested during this group ...
        u32x4 (&rs)[G::NLD] = r;                              // ... and written to LDS before its end (for group cb+1)
        constexpr bool more = !LAST;
        const unsigned short *nxt_grp = grp0 + (size_t)min(cb + 1, CB - 1) * grp_sz;   // clamped: loads stay unconditional
        const int bcur = cb & 1, bprev = (cb + 1) & 1;
        const char *buf = reinterpret_cast<const char *>(lds + bcur * G::BUF);
        auto xaddr = [&](int ks) -> const char * {   // in-group tap pair of K-step ks
            const int j = MODE == 2 ? ks - 1 : ks;
            int tA = 2 * j, tB = 2 * j + 1;
            if (MODE == 0 && tB >= G::TAPS) tB = tA;
            const int oA = ((tA / KW) * G::TW + tA % KW) * 32, oB = ((tB / KW) * G::TW + tB % KW) * 32;
            return buf + (tapsel ? oB : oA) + pb;
        };
        if constexpr (LEAN) {
            // 168-VGPR form for a third workgroup per CU: pixel fragments of RS rows at a time, single-buffered (the other two
            // waves of the SIMD cover the LDS round trip), both weight splits double-buffered one K-step ahead, and the
            // cross-group tap read from the partner buffer instead of carried in 64 registers - with a barrier before that
            // buffer's first rolling store.
            constexpr int RS = 2, SUB = RW / RS;   // 2-row sub-steps: 4-row ones spill (-15 %), 1-row ones expose more LDS round trips (-1 %)
            const char *part = reinterpret_cast<const char *>(lds + bprev * G::BUF);
            if (NK == 0 && FETCH) h2_stage_load<KH, KW, NTHR>(plan, nxt_grp, rl);   // 1x1 source, even group: only fetch the partner group
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                asm volatile("" : "+v"(tapsel));
                const char *px = (MODE == 2 && ks == 0) ? (g < 2 ? part : buf) + O_LAST + pb : xaddr(ks);
                ++stream;
                const f16x8 *wf = wl + (size_t)min(stream, last) * (2 * NT * 64);
                f16x8 xa[RS], xb[RS];
                auto rd_a = [&](int h) __attribute__((always_inline)) {
#pragma unroll
                    for (int m = 0; m < RS; ++m) xa[m] = *reinterpret_cast<const f16x8 *>(px + (h * RS + m) * G::TW * 32);
                };
                auto rd_b = [&](int h) __attribute__((always_inline)) {
#pragma unroll
                    for (int m = 0; m < RS; ++m) xb[m] = *reinterpret_cast<const f16x8 *>(px + G::PLANE * 16 + (h * RS + m) * G::TW * 32);
                };
#pragma unroll
                for (int h = 0; h < SUB; ++h) {
                    rd_a(h); rd_b(h);   // nothing is prefetched across sub-steps: requesting the next fragments as soon as their registers are free measured 0.9 % slower
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < RS; ++m)
#pragma unroll
                        for (int nt = 0; nt < CW; ++nt)
                            acc[h * RS + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[nt], xa[m], acc[h * RS + m][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (h == 0) {
#pragma unroll
                        for (int nt = 0; nt < CW; ++nt) w0[nt] = w0n[nt];
#pragma unroll
                        for (int nt = 0; nt < CW; ++nt) { w1n[nt] = wf[(1 * NT + nt) * 64]; w0n[nt] = wf[(0 * NT + nt) * 64]; }
                        if (FETCH) h2_stage_load<KH, KW, NTHR>(plan, nxt_grp, rl, ks * PER, (ks + 1) * PER);   // slices: one burst per group measured 1 % slower here
                        if (more && ks >= LAG)
                            h2_stage_store<KH, KW, NTHR>(plan, lds + ((cb + 1) & 1) * G::BUF, rs, (ks - LAG) * PER, (ks - LAG + 1) * PER);
                    } else if (h == SUB - 1) {
#pragma unroll
                        for (int nt = 0; nt < CW; ++nt) w1[nt] = w1n[nt];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < RS; ++m)
#pragma unroll
                        for (int nt = 0; nt < CW; ++nt)
                            acc[h * RS + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], xa[m], acc[h * RS + m][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < RS; ++m)
#pragma unroll
                        for (int nt = 0; nt < CW; ++nt)
                            acc[h * RS + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], xb[m], acc[h * RS + m][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (MODE == 2 && ks == 0) h2_lds_barrier();   // every wave has read the partner buffer's last tap
            }
        } else {
            if (NK == 0) {   // 1x1 source, even group: nothing to compute yet, only fetch the partner group
                if (FETCH) h2_stage_load<KH, KW, NTHR>(plan, nxt_grp, rl);
            } else if (MODE == 2) {
                // cross-group pair: lanes g < 2 still hold the even group's last tap (picked up before the barrier that ended
                // it - that buffer is being overwritten by now), lanes g >= 2 read this group's last tap
                if (g >= 2) {
                    const char *pl = buf + O_LAST + pb;
#pragma unroll
                    for (int m = 0; m < RW; ++m) {
                        x0[m] = *reinterpret_cast<const f16x8 *>(pl + m * G::TW * 32);
                        x1[m] = *reinterpret_cast<const f16x8 *>(pl + G::PLANE * 16 + m * G::TW * 32);
                    }
                }
            } else {
                const char *p0x = xaddr(0);
#pragma unroll
                for (int m = 0; m < RW; ++m) x0[m] = *reinterpret_cast<const f16x8 *>(p0x + m * G::TW * 32);
            }
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                asm volatile("" : "+v"(tapsel));   // keeps hipcc from hoisting every K-step's tap offset out of the group loop
                const char *px = (MODE == 2 && ks == 0) ? buf : xaddr(ks);
                ++stream;
                if (!(MODE == 2 && ks == 0)) {
#pragma unroll
                    for (int m = 0; m < RW; ++m) x1[m] = *reinterpret_cast<const f16x8 *>(px + G::PLANE * 16 + m * G::TW * 32);
                }
                // next K-step's fragments (L2-resident)
                const f16x8 *wf = wl + (size_t)min(stream, last) * (2 * NT * 64);
                __builtin_amdgcn_sched_barrier(0);
                // phase A: x0*w1, then w1 is free for the next K-step's fragments.  No vector-memory request precedes it inside
                // the K-step: hipcc loses the exact outstanding-load count across the group loop's back edge and waits for
                // vmcnt(0) at a group's first use of a weight fragment - which must not cover a halo request issued just before.
#pragma unroll
                for (int m = 0; m < RW; ++m)
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[nt], x0[m], acc[m][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);   // keep the refill behind the MFMAs that read the old fragments (same registers)
                if (W2) {   // everything moves up one place, the fragments of K-step k+2 are requested
                    const f16x8 *wf2 = wl + (size_t)min(stream + 1, last) * (2 * NT * 64);
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) { w1[nt] = w1n[nt]; w0[nt] = w0n[nt]; w0n[nt] = w0nn[nt]; }
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) w1n[nt] = wf2[(1 * NT + nt) * 64];
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) w0nn[nt] = wf2[(0 * NT + nt) * 64];
                } else {
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) w1[nt] = wf[(1 * NT + nt) * 64];
                    if (W0DB) {   // this K-step's w0 was requested a K-step ago; its successor goes out at once
#pragma unroll
                        for (int nt = 0; nt < CW; ++nt) w0[nt] = w0n[nt];
#pragma unroll
                        for (int nt = 0; nt < CW; ++nt) w0n[nt] = wf[(0 * NT + nt) * 64];
                    }
                }
                // The halo requests follow the weight requests: the counter that orders vector-memory operations is in-order, so
                // a weight fragment requested after an HBM load cannot be used before that load has landed.  Here the next such
                // fragment is the w1 request of the NEXT K-step, used two K-steps from now.
                if (FETCH) h2_stage_load<KH, KW, NTHR>(plan, nxt_grp, rl, ks * PER, (ks + 1) * PER);
                // ... and the slice requested LAG K-steps ago goes to the partner LDS buffer, which nobody reads during this group
                if (more && ks >= LAG)
                    h2_stage_store<KH, KW, NTHR>(plan, lds + ((cb + 1) & 1) * G::BUF, rs, (ks - LAG) * PER, (ks - LAG + 1) * PER);
                __builtin_amdgcn_sched_barrier(0);
                // phase B1: x0*w0, then x0 is free for the next K-step's pixels
#pragma unroll
                for (int m = 0; m < RW; ++m)
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], x0[m], acc[m][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < NK) {
                    const char *pn = xaddr(ks + 1);
#pragma unroll
                    for (int m = 0; m < RW; ++m) x0[m] = *reinterpret_cast<const f16x8 *>(pn + m * G::TW * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
                // phase B2: x1*w0, then w0 is free
#pragma unroll
                for (int m = 0; m < RW; ++m)
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], x1[m], acc[m][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!W0DB) {
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) w0[nt] = wf[(0 * NT + nt) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the partner halo buffer is only overwritten here, after the last K-step that may read the previous group from it
            if (MODE == 1 && g < 2) {   // deferred last tap of the even group, carried in registers across the barrier
                const char *pl = buf + O_LAST + pb;
#pragma unroll
                for (int m = 0; m < RW; ++m) {
                    x0[m] = *reinterpret_cast<const f16x8 *>(pl + m * G::TW * 32);
                    x1[m] = *reinterpret_cast<const f16x8 *>(pl + G::PLANE * 16 + m * G::TW * 32);
                }
            }
        }
        // the store goes to the buffer nobody reads during this group (the deferred tap travels in registers)
        if (more) h2_stage_store<KH, KW, NTHR>(plan, lds + ((cb + 1) & 1) * G::BUF, rs, NK > LAG ? (NK - LAG) * PER : 0);
        __syncthreads();
    };

    typedef std::integral_constant<int, 0> M0;
    typedef std::integral_constant<int, 1> M1;
    typedef std::integral_constant<int, 2> M2;
    typedef std::integral_constant<int, 0> T0;
    typedef std::integral_constant<int, 1> T1;
    typedef std::integral_constant<int, 2> T2;
    if (paired) {   // the last two groups are peeled: they have nothing (or less) to request (peeling all four groups of a
                    // 64-channel input, to lose the register shuffles at the loop's back edge, gains nothing and spills more)
        for (int cb = 0; cb + 2 < CB; cb += 2) {
            group(M1{}, T0{}, cb);
            group(M2{}, T0{}, cb + 1);
        }
        group(M1{}, T1{}, CB - 2);
        group(M2{}, T2{}, CB - 1);
    } else {
        for (int cb = 0; cb + 1 < CB; ++cb) group(M0{}, T0{}, cb);
        group(M0{}, T2{}, CB - 1);
    }
}

template <int NT>
__device__ __forceinline__ void h2_epilogue(const ConvX6Args &a, f32x4 (&acc)[WaveTile<NT>::RW][WaveTile<NT>::CW], int n, int ty, int tx);

// The 1x1 shortcut source of a Cout = 64 block has 32 channels in every net (RB(32,64,k)): two channel groups = ONE K-step.
// Instead of the general staging pipeline (five barriers, its own plan, weights and staging registers) the whole 16x16 x 32 ch
// tile goes to LDS at once - no halo - and each wave runs that K-step in 2-row sub-steps.  Same MFMA order per accumulator as
// the general pass (x0*w1, x0*w0, x1*w0 after the main pass), so the results are bit-identical.
template <int NT>
__device__ __forceinline__ void h2_shortcut32(const ConvX6Args &a, int n, int ty, int tx, u32x4 *lds,
                                              f32x4 (&acc)[WaveTile<NT>::RW][WaveTile<NT>::CW])
{
    typedef WaveTile<NT> WT;
    constexpr int RW = WT::RW, CW = WT::CW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, xl = lane & 15, g = lane >> 4;
    const int rh = wave % WT::RSPLIT, ch = wave / WT::RSPLIT;
    const int W = a.W;
    const size_t grp_sz = (size_t)a.H * W * 16;
    const unsigned short *base = a.x_sc + (size_t)n * 2 * grp_sz + ((size_t)(ty * 16) * W + tx * 16) * 16;
    __syncthreads();   // every wave is done with the main pass's halo tiles
    u32x4 r[8];        // piece i = [group][plane][pixel][half]: 2 x 2 x 256 x 2 pieces of 16 B
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + k * 256, grp = i >> 10, plane = (i >> 9) & 1, j = i & 511, px = j >> 1, half = j & 1;
        r[k] = *reinterpret_cast<const u32x4 *>(base + plane * a.sc_stride + grp * grp_sz + ((size_t)(px >> 4) * W + (px & 15)) * 16 + half * 8);
    }
    const f16x8 *wl = reinterpret_cast<const f16x8 *>(a.w_sc) + lane + ch * CW * 64;
    f16x8 w0[CW], w1[CW];
#pragma unroll
    for (int nt = 0; nt < CW; ++nt) { w0[nt] = wl[(0 * NT + nt) * 64]; w1[nt] = wl[(1 * NT + nt) * 64]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) lds[tid + k * 256] = r[k];
    __syncthreads();
    const char *px = reinterpret_cast<const char *>(lds) + (g >> 1) * 16384 + ((rh * RW * 16 + xl) * 2 + (g & 1)) * 16;
#pragma unroll
    for (int h = 0; h < RW / 2; ++h) {
        f16x8 xa[2], xb[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            xa[m] = *reinterpret_cast<const f16x8 *>(px + (h * 2 + m) * 16 * 32);
            xb[m] = *reinterpret_cast<const f16x8 *>(px + 8192 + (h * 2 + m) * 16 * 32);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < CW; ++nt) acc[h * 2 + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[nt], xa[m], acc[h * 2 + m][nt], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < CW; ++nt) acc[h * 2 + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], xa[m], acc[h * 2 + m][nt], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < CW; ++nt) acc[h * 2 + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], xb[m], acc[h * 2 + m][nt], 0, 0, 0);
    }
}

template <int KH, int KW, int NT, int SC, bool LEAN = false>   // SC: 0 none, 1 general 1x1 shortcut pass, 2 the 32-channel one
__global__ __launch_bounds__(256, NT == 4 && !LEAN ? 2 : 3) void conv_h2_kernel(ConvX6Args a)
{
    typedef GeoH<KH, KW> G;
    __shared__ u32x4 lds[2 * G::BUF];
    const int tiles_x = a.W >> 4, tiles = tiles_x * (a.H >> 4);
    // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so consecutive ids would put
    // neighbouring tiles - which share halo columns/rows - on different L2s.  Give every XCD a contiguous run of tiles instead.
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int n = bid / tiles, t = bid - n * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
    typedef WaveTile<NT> WT;
    constexpr int RW = WT::RW, CW = WT::CW;

    f32x4 acc[RW][CW];
#pragma unroll
    for (int m = 0; m < RW; ++m)
#pragma unroll
        for (int nt = 0; nt < CW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    h2_accumulate<KH, KW, NT, LEAN>(a.x, a.x_stride, a.w, a.Cin, a.H, a.W, n, ty, tx, lds, acc);
    if (SC == 2) h2_shortcut32<NT>(a, n, ty, tx, lds, acc);
    else if (SC) h2_accumulate<1, 1, NT, LEAN>(a.x_sc, a.sc_stride, a.w_sc, a.Csc, a.H, a.W, n, ty, tx, lds, acc);
    h2_epilogue<NT>(a, acc, n, ty, tx);
}

template <int NT>
__device__ __forceinline__ void h2_epilogue(const ConvX6Args &a, f32x4 (&acc)[WaveTile<NT>::RW][WaveTile<NT>::CW], int n, int ty, int tx)
{
    typedef WaveTile<NT> WT;
    constexpr int RW = WT::RW, CW = WT::CW;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // opaque: keeps hipcc from sharing lane arithmetic with the K-loop (fewer live registers there)
    const int lane = tid & 63, wave = tid >> 6, xl = lane & 15, g = lane >> 4;
    const int rh = wave % WT::RSPLIT, ch = wave / WT::RSPLIT;
    const int H = a.H, W = a.W;
    const size_t grp = (size_t)H * W * 16;

    const float inv_scale = a.out_scale;
    // element offset of (row 0, cout group 0) of this wave inside a [n][NT][H][W][16] tensor: < 2^32 for every chunk size
    const unsigned off0 = (unsigned)(((size_t)n * NT + ch * CW) * grp + ((size_t)(ty * 16 + rh * RW) * W + tx * 16 + xl) * 16 + g * 4);
    const unsigned row_el = (unsigned)W * 16;
    // (A straight-line special case for ReLU + split-2 output without gate/pool measured 2 % SLOWER than this general path
    // with its wave-uniform branches per cout group: the branches keep one group's stores ahead of the next group's conversions.)
    // All residual (then gate) fragments are requested before the first store: the weight, pixel and staging registers are
    // dead by now, and a load issued after a store to `out` would otherwise have to wait for it (possible aliasing).
    // Residual loads and output stores move 16 bytes per lane: lane (xl, g) handles the 8 consecutive channels 8(g>>1).. of pixel
    // (row m + (g&1), xl) - one v_permlane16_swap per register (gfx950) exchanges that form with the accumulator layout (4
    // couts 4g.. of rows m and m+1).  Half the vector-memory instructions of 8-byte accesses: the epilogue is issue-bound.
    const unsigned off0w = off0 - (unsigned)(g * 4) + (unsigned)(8 * (g >> 1)) + (unsigned)(g & 1) * row_el;
    u32x4 ra[RW / 2][CW], rbv[RW / 2][CW];
    auto res_load = [&](int nt, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < RW; m += 2) {
            const unsigned off = off0w + (unsigned)m * row_el + (unsigned)nt * (unsigned)grp;
            // (non-temporal: a residual tile is read once; keeping it out of the L2's way measured -0.2 % on the step, same-box A/B)
            ra[m >> 1][slot] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.res + off));
            rbv[m >> 1][slot] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.res + off + a.res_stride));
        }
    };
    auto res_add = [&](int nt, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < RW; m += 2) {
            u32x4 p = ra[m >> 1][slot], q = rbv[m >> 1][slot];
            rows16_swap(p);
            rows16_swap(q);
            acc[m][nt] = acc[m][nt] * inv_scale + h2_sum4_lo(p, q);        // undo the power-of-two weight scaling (exact)
            acc[m + 1][nt] = acc[m + 1][nt] * inv_scale + h2_sum4_hi(p, q);
        }
    };
    if (a.res) {
#pragma unroll
        for (int nt = 0; nt < CW; ++nt) res_load(nt, nt);
#pragma unroll
        for (int nt = 0; nt < CW; ++nt) res_add(nt, nt);
    } else {
#pragma unroll
        for (int nt = 0; nt < CW; ++nt)
#pragma unroll
            for (int m = 0; m < RW; ++m) acc[m][nt] = acc[m][nt] * inv_scale;
    }
    float amax = 0.f;   // largest activation about to be stored: the split clamps beyond +-65504 (split3.h) and the flag says so
#pragma unroll
    for (int nt = 0; nt < CW; ++nt) {
#pragma unroll
        for (int m = 0; m < RW; ++m) {
            const unsigned off = off0 + (unsigned)m * row_el + (unsigned)nt * (unsigned)grp;
            f32x4 v = acc[m][nt];
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (a.gate) v *= load_split2_4(a.gate + off, a.gate_stride);
            amax = sat_amax4(amax, v);
            acc[m][nt] = v;
        }
        if (!a.pool) {
            if (a.out_f32) {
#pragma unroll
                for (int m = 0; m < RW; ++m)
                    *reinterpret_cast<f32x4 *>(a.out_f32 + off0 + (unsigned)m * row_el + (unsigned)nt * (unsigned)grp) = acc[m][nt];
            } else {
#pragma unroll
                for (int m = 0; m < RW; m += 2) {
                    const unsigned off = off0w + (unsigned)m * row_el + (unsigned)nt * (unsigned)grp;
                    u32x4 p, q;
                    split2_rows(acc[m][nt], acc[m + 1][nt], p, q);
                    rows16_swap(p);
                    rows16_swap(q);
                    // (non-temporal: the next layer reads this tile gigabytes later; with the residual loads non-temporal too the stores measure
                    // -0.65 % on the step, 3x3 class -1.3 %, same-box A/B - in rounds 1-2, with default-policy residual loads, they measured +-0)
                    __builtin_nontemporal_store(p, reinterpret_cast<u32x4 *>(a.out + off));
                    __builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(a.out + off + a.out_stride));
                }
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
                    const int yo = ty * 8 + rh * (RW / 2) + (m >> 1), xo = tx * 8 + (xl >> 1);
                    const size_t off = (((size_t)n * NT + ch * CW + nt) * Ho + yo) * Wo * 16 + (size_t)xo * 16 + g * 4;
                    if (a.out_f32) *reinterpret_cast<f32x4 *>(a.out_f32 + off) = v;
                    else store_split2_4(a.out + off, a.out_stride, v);
                }
            }
        }
    }
    if (!a.out_f32) sat_report(a.sat, amax);   // fp32 outputs are not clamped
}

// The 1x1 shortcut source is a separate instantiation: its extra live state would spill in the common kernel.
#define PMP_H2_LAUNCH(NT)                                                                                       \
    if (a.x_sc) hipLaunchKernelGGL((conv_h2_kernel<KH, KW, NT, 1>), dim3(grid), dim3(256), 0, s, a);           \
    else hipLaunchKernelGGL((conv_h2_kernel<KH, KW, NT, 0>), dim3(grid), dim3(256), 0, s, a)

template <int KH, int KW>
static hipError_t launch_h2(hipStream_t s, const ConvX6Args &a)
{
    const int grid = a.N * (a.H >> 4) * (a.W >> 4);
    switch (a.Cout >> 4) {
    case 1: PMP_H2_LAUNCH(1); break;
    case 2: PMP_H2_LAUNCH(2); break;
    case 4:
        if constexpr (KH > 1) {
            if (a.x_sc && a.Csc == 32) {
                // RB(32,64,k): the whole 32-channel shortcut tile goes to LDS at once (h2_shortcut32); two workgroups per CU - in the
                // 168-VGPR form these kernels measure the same (5x5 class 5.94 vs 5.95 ms per 1024 blocks)
                hipLaunchKernelGGL((conv_h2_kernel<KH, KW, 4, 2>), dim3(grid), dim3(256), 0, s, a);
                break;
            }
            if (!a.x_sc) {
                // the Cout = 64 layers without a shortcut source: the 168-VGPR form, three workgroups per CU (3x3: -5.6 %, 5x5: -1.9 %
                // against the two-workgroup form; the shortcut instantiations would spill 70 registers in this form)
                hipLaunchKernelGGL((conv_h2_kernel<KH, KW, 4, 0, true>), dim3(grid), dim3(256), 0, s, a);
                break;
            }
        }
        PMP_H2_LAUNCH(4);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#undef PMP_H2_LAUNCH

hipError_t launch_conv_h2(hipStream_t s, const ConvX6Args &a_in)
{
    const ConvX6Args &a = a_in;

    if ((a.H & 15) || (a.W & 15) || (a.Cin & 15) || (a.Cout & 15) || (a.x_sc && (a.Csc & 15)) || a.N <= 0)
        return hipErrorInvalidValue;
    if (a.pool && a.gate) return hipErrorInvalidValue;
    if (!(a.out_scale > 0.f)) return hipErrorInvalidValue;
    if (a.KH == 3 && a.KW == 3) return launch_h2<3, 3>(s, a);
    if (a.KH == 5 && a.KW == 5) return launch_h2<5, 5>(s, a);
    if (a.KH == 1 && a.KW == 1) return launch_h2<1, 1>(s, a);
    return hipErrorInvalidValue;
}

// ---- format converters ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f32_to_split2_kernel(const float *__restrict__ x, unsigned short *__restrict__ out,
                                                            size_t n4, size_t plane_stride, unsigned *sat)
{
    float amax = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + i * 4);
        amax = sat_amax4(amax, v);
        store_split2_4(out + i * 4, plane_stride, v);
    }
    sat_report(sat, amax);
}

__global__ __launch_bounds__(256) void split2_to_f32_kernel(const unsigned short *__restrict__ x, float *__restrict__ out,
                                                            size_t n4, size_t plane_stride)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        *reinterpret_cast<f32x4 *>(out + i * 4) = load_split2_4(x + i * 4, plane_stride);
}

hipError_t launch_f32_to_split2(hipStream_t s, const float *x, unsigned short *out, size_t n, size_t plane_stride, unsigned *sat)
{
    const size_t n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256);
    if (n4) hipLaunchKernelGGL(f32_to_split2_kernel, dim3(grid), dim3(256), 0, s, x, out, n4, plane_stride, sat);
    return hipGetLastError();
}

hipError_t launch_split2_to_f32(hipStream_t s, const unsigned short *x, float *out, size_t n, size_t plane_stride)
{
    const size_t n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256);
    if (n4) hipLaunchKernelGGL(split2_to_f32_kernel, dim3(grid), dim3(256), 0, s, x, out, n4, plane_stride);
    return hipGetLastError();
}

}  // namespace pmp
