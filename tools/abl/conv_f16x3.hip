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
// Measurements, ablations and the variants that did not pay: DESIGN.md 4.1a.
#include <cstdio>
#include <type_traits>

#include "abl_kernels.h"
#include "split3.h"

namespace pmp {

// In-kernel stamps (diagnostic build only, ABL bit 128): shader-clock ticks of wave 0 at phase boundaries, written to a debug
// buffer nothing else reads.
__device__ __forceinline__ unsigned long long h2_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

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

// State that outlives one tile in the persistent kernel (CHAIN): the weight fragments of the next K-step - the stream wraps
// around to the first K-step at the end of a tile - and the staging plan of the tile whose first halo group is in LDS.
template <int KH, int KW, int NT, int W8 = 0>
struct H2Carry {
    static constexpr int CW = WaveTile<NT, W8>::CW;
    f16x8 w0[CW], w1[CW], w0n[CW], w1n[CW], w0nn[CW];
    StagePlanH<KH, KW, W8 == 1 ? 512 : 256> plan;
};

template <int KH, int KW, int NT, int ABL = 0, bool CHAIN = false, bool LEAN = false, int W8 = 0>
__device__ __forceinline__ void h2_accumulate(const unsigned short *__restrict__ x, size_t plane_stride,
                                              const unsigned short *__restrict__ wpk, int C, int H, int W, int n, int ty,
                                              int tx, u32x4 *lds, f32x4 (&acc)[WaveTile<NT, W8>::RW][WaveTile<NT, W8>::CW],
                                              unsigned long long *dbg, H2Carry<KH, KW, NT, W8> &c, bool first, int n2, int ty2, int tx2)
{
    constexpr int NTHR = W8 == 1 ? 512 : 256;
    constexpr bool STAGE = W8 != 3;   // W8 == 3: a fifth wave of the workgroup fills the halo buffers by LDS-DMA (conv_h2_ld_kernel); the compute
                                      // waves issue no halo request at all - only the barriers of the staging protocol remain
    unsigned long long t_pro = 0, t_k = 0, t_s = 0, t_b = 0, tmark = 0;   // diagnostic accumulators (ABL & 128)
    if (ABL & 128) tmark = h2_stamp();
    typedef GeoH<KH, KW, 16, NTHR, W8 == 3> G;
    typedef WaveTile<NT, W8> WT;
    constexpr int RW = WT::RW, CW = WT::CW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, xl = lane & 15, g = lane >> 4;
    const int rh = wave % WT::RSPLIT, ch = wave / WT::RSPLIT;   // this wave: rows RW*rh.., cout groups CW*ch..
    const int CB = C >> 4;
    const size_t grp_sz = (size_t)H * W * 16;
    const unsigned short *grp0 = x + (size_t)n * CB * grp_sz;
    // Halo staging registers.  With tap pairing (even group count) there are TWO sets: group g+2 is requested during group
    // g and written to LDS at the end of group g+1, so an HBM round trip has two groups of MFMAs to hide behind (loaded
    // latency here is longer than one 3x3 group).  Plain mode keeps one set and a distance of one group.
    // 5x5 groups (12.5 K-steps) outlast the HBM latency on their own; the Cout <= 32 kernels are latency-bound small layers
    // that gain more from a third resident workgroup (168 VGPRs) than from the second staging set
    // W0DB: a K-step's w0 fragments are requested one K-step ahead into a second register set and moved over between its
    // phases A and B1 (8 v_mov) - a full K-step of lead for the split that is used last and needed first.
    constexpr bool W0DB = NT == 4;
    constexpr bool W2 = W0DB && G::TAPS == 9 && !CHAIN && !LEAN && !(ABL & 256);   // ABL 256: A/B build with one K-step of lead
    constexpr bool DEEP = G::TAPS <= 9 && NT == 4 && !W0DB;
    u32x4 r[G::NLD], rb[DEEP ? G::NLD : 1];
    StagePlanH<KH, KW, NTHR> &plan = c.plan;
    // Tap pairing as in conv_bf16x6.hip: mode 0 plain (last pair zero-padded), mode 1 even group of a pair (its last tap
    // is deferred and carried in registers), mode 2 odd group (first K-step = the deferred tap + its own last tap).
    const bool paired = (CB & 1) == 0 && (G::TAPS & 1);
    const f16x8 *wl = reinterpret_cast<const f16x8 *>(wpk) + lane + ch * CW * 64;
    const int last = paired ? (CB / 2) * G::TAPS - 1 : CB * G::NKS - 1;   // last K-step of the weight stream
    f16x8 (&w0)[CW] = c.w0, (&w1)[CW] = c.w1;   // ONE weight set, refilled in place as soon as the last MFMA that reads a split has issued
    f16x8 (&w0n)[CW] = c.w0n;                  // W0DB: the K-step's w0 fragments land here and move to w0 between its phases A and B1
    // W2: weight fragments are requested TWO K-steps ahead (16 more registers).  The L1 returns data in order for the whole
    // CU, so a weight hit queued behind a halo request that went to HBM - this workgroup's or its neighbour's - waits for
    // it; one K-step of lead (0.8-1.5 k cycles) is less than that round trip, two are more.
    f16x8 (&w1n)[CW] = c.w1n, (&w0nn)[CW] = c.w0nn;
    if (!CHAIN || first) {   // a chained tile finds its weights in the carry and its first halo group in LDS buffer 0
        if (STAGE) h2_plan<KH, KW, NTHR>(plan, plane_stride, H, W, ty, tx);
#pragma unroll
        for (int nt = 0; nt < CW; ++nt) {
            if (W0DB) w0n[nt] = wl[(0 * NT + nt) * 64]; else w0[nt] = wl[(0 * NT + nt) * 64];
            w1[nt] = wl[(1 * NT + nt) * 64];
            if (W2) {
                const f16x8 *w2 = wl + (size_t)min(1, last) * (2 * NT * 64);
                w0nn[nt] = w2[(0 * NT + nt) * 64]; w1n[nt] = w2[(1 * NT + nt) * 64];
            }
        }
        if (STAGE) {
            __syncthreads();
            h2_stage_load<KH, KW, NTHR>(plan, grp0, r);
            if (DEEP && paired) h2_stage_load<KH, KW, NTHR>(plan, grp0 + grp_sz, reinterpret_cast<u32x4 (&)[G::NLD]>(rb));   // group 1 exists: CB is even
            h2_stage_store<KH, KW, NTHR>(plan, lds, r);
        }
        __syncthreads();   // loader-wave form: the loader has DMA'd group 0 and waited for it before this barrier
    } else if (LEAN) {
        // 168 VGPRs cannot carry the plan and four weight sets across the epilogue: a chained tile of the three-workgroup
        // form recomputes its plan and requests its first weights again (L2 hits); only the halo group in LDS is carried
        h2_plan<KH, KW, NTHR>(plan, plane_stride, H, W, ty, tx);
#pragma unroll
        for (int nt = 0; nt < CW; ++nt) { w0n[nt] = wl[(0 * NT + nt) * 64]; w1[nt] = wl[(1 * NT + nt) * 64]; }
    }
    if (ABL & 128) { const unsigned long long t = h2_stamp(); t_pro = t - tmark; tmark = t; }
    // ABL: timing-only builds (tools/conv_x6_bench.py h2 ablate): 1 no halo staging, 2 no weight refills, 4 no fragment reads, 8 no epilogue
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
        constexpr int DIST = (MODE == 0 || !DEEP) ? 1 : 2;   // groups between a halo request and its LDS store
        constexpr int LAG = G::TAPS > 9 ? 3 : 2;             // K-steps between a halo slice's request and its LDS store
        constexpr bool ROLL = DIST == 1;
        constexpr bool CHAINF = CHAIN && LAST;                // persistent kernel: the last group requests the NEXT tile's first group
        constexpr bool FETCH = CHAINF || TAIL < (DIST == 2 ? 1 : 2);    // is there a group cb+DIST to request
        u32x4 (&rbb)[G::NLD] = reinterpret_cast<u32x4 (&)[G::NLD]>(rb);
        u32x4 (&rl)[G::NLD] = (DIST == 2 && MODE == 2) ? rbb : r;   // requested during this group
        u32x4 (&rs)[G::NLD] = (DIST == 2 && MODE == 1) ? rbb : r;   // written to LDS at the end of this group (for group cb+1)
        constexpr bool more = !LAST || CHAINF;
        // the plan's last use for this tile was the request of this group during the previous one
        if (CHAINF) h2_plan<KH, KW, NTHR>(plan, plane_stride, H, W, ty2, tx2);
        const unsigned short *nxt_grp = CHAINF ? x + (size_t)n2 * CB * grp_sz
                                               : grp0 + (size_t)min(cb + DIST, CB - 1) * grp_sz;   // clamped (odd group in a deep pair before the tail)
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
            constexpr int RS = W8 == 1 ? 4 : 2, SUB = RW / RS;   // 2-row sub-steps: 4-row ones spill (-15 %), 1-row ones expose more LDS round trips (-1 %)
            const char *part = reinterpret_cast<const char *>(lds + bprev * G::BUF);
            if (STAGE && NK == 0 && FETCH) h2_stage_load<KH, KW, NTHR>(plan, nxt_grp, rl);   // 1x1 source, even group: only fetch the partner group
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                asm volatile("" : "+v"(tapsel));
                const char *px = (MODE == 2 && ks == 0) ? (g < 2 ? part : buf) + O_LAST + pb : xaddr(ks);
                ++stream;
                const f16x8 *wf = wl + (size_t)(CHAIN ? (stream > last ? 0 : stream) : min(stream, last)) * (2 * NT * 64);
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
                        if (STAGE && FETCH) h2_stage_load<KH, KW, NTHR>(plan, nxt_grp, rl, ks * PER, (ks + 1) * PER);   // slices: one burst per group measured 1 % slower here
                        if (STAGE && more && ks >= LAG)
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
                if (!(ABL & 1) && FETCH) h2_stage_load<KH, KW, NTHR>(plan, nxt_grp, rl);
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
                if ((!(ABL & 4) || ks == 0) && !(MODE == 2 && ks == 0)) {
#pragma unroll
                    for (int m = 0; m < RW; ++m) x1[m] = *reinterpret_cast<const f16x8 *>(px + G::PLANE * 16 + m * G::TW * 32);
                }
                // next K-step's fragments (L2-resident); the persistent kernel wraps around to the next tile's first K-step
                const f16x8 *wf = wl + (size_t)(CHAIN ? (stream > last ? 0 : stream) : min(stream, last)) * (2 * NT * 64);
                __builtin_amdgcn_sched_barrier(0);
                // phase A: x0*w1, then w1 is free for the next K-step's fragments.  No vector-memory request precedes it inside
                // the K-step: hipcc loses the exact outstanding-load count across the group loop's back edge and waits for
                // vmcnt(0) at a group's first use of a weight fragment - which must not cover a halo request issued just before.
#pragma unroll
                for (int m = 0; m < RW; ++m)
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[nt], x0[m], acc[m][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);   // keep the refill behind the MFMAs that read the old fragments (same registers)
                if (W2 && !(ABL & 2)) {   // everything moves up one place, the fragments of K-step k+2 are requested
                    const f16x8 *wf2 = wl + (size_t)min(stream + 1, last) * (2 * NT * 64);
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) { w1[nt] = w1n[nt]; w0[nt] = w0n[nt]; w0n[nt] = w0nn[nt]; }
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) w1n[nt] = wf2[(1 * NT + nt) * 64];
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) w0nn[nt] = wf2[(0 * NT + nt) * 64];
                } else if (!(ABL & 2)) {
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
                if (!(ABL & 1) && FETCH) {
                    if (ABL & 32) { if (ks == 0) h2_stage_load<KH, KW, NTHR>(plan, nxt_grp, rl); }
                    else h2_stage_load<KH, KW, NTHR>(plan, nxt_grp, rl, ks * PER, (ks + 1) * PER);
                }
                // ... and the slice requested LAG K-steps ago goes to the partner LDS buffer, which nobody reads during this group
                if (ROLL && more && !(ABL & 1) && ks >= LAG)
                    h2_stage_store<KH, KW, NTHR>(plan, lds + ((cb + 1) & 1) * G::BUF, rs, (ks - LAG) * PER, (ks - LAG + 1) * PER);
                __builtin_amdgcn_sched_barrier(0);
                // phase B1: x0*w0, then x0 is free for the next K-step's pixels
#pragma unroll
                for (int m = 0; m < RW; ++m)
#pragma unroll
                    for (int nt = 0; nt < CW; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], x0[m], acc[m][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < NK && !(ABL & 4)) {
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
                if (!W0DB && !(ABL & 2)) {
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
        if (ABL & 128) { const unsigned long long t = h2_stamp(); t_k += t - tmark; tmark = t; }
        // the store goes to the buffer nobody reads during this group (the deferred tap travels in registers)
        if (STAGE && more && !(ABL & 1))
            h2_stage_store<KH, KW, NTHR>(plan, lds + ((cb + 1) & 1) * G::BUF, rs, ROLL && NK > LAG ? (NK - LAG) * PER : 0);
        if (ABL & 128) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t = h2_stamp(); t_s += t - tmark; tmark = t; }
        __syncthreads();
        if (ABL & 128) { const unsigned long long t = h2_stamp(); t_b += t - tmark; tmark = t; }
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
    if ((ABL & 128) && dbg && threadIdx.x == 0) { dbg[0] = t_pro; dbg[1] = t_k; dbg[2] = t_s; dbg[3] = t_b; }
}

template <int NT, int ABL, int W8 = 0>
__device__ __forceinline__ void h2_epilogue(const ConvX6Args &a, f32x4 (&acc)[WaveTile<NT, W8>::RW][WaveTile<NT, W8>::CW], int n, int ty, int tx);

// The 1x1 shortcut source of a Cout = 64 block has 32 channels in every net (RB(32,64,k)): two channel groups = ONE K-step.
// Instead of the general staging pipeline (five barriers, its own plan, weights and staging registers) the whole 16x16 x 32 ch
// tile goes to LDS at once - no halo - and each wave runs that K-step in 2-row sub-steps.  Same MFMA order per accumulator as
// the general pass (x0*w1, x0*w0, x1*w0 after the main pass), so the results are bit-identical.
template <int NT, int W8>
__device__ __forceinline__ void h2_shortcut32(const ConvX6Args &a, int n, int ty, int tx, u32x4 *lds,
                                              f32x4 (&acc)[WaveTile<NT, W8>::RW][WaveTile<NT, W8>::CW])
{
    typedef WaveTile<NT, W8> WT;
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

template <int KH, int KW, int NT, int SC, int ABL = 0, bool LEAN = false, int W8 = 0>   // SC: 0 none, 1 general 1x1 shortcut pass, 2 the 32-channel one
__global__ __launch_bounds__(W8 == 1 ? 512 : 256, W8 == 1 ? 4 : (NT == 4 && !LEAN ? 2 : 3)) void conv_h2_kernel(ConvX6Args a)
{
    typedef GeoH<KH, KW, 16, W8 == 1 ? 512 : 256> G;
    __shared__ u32x4 lds[2 * G::BUF];
    const int tiles_x = a.W >> 4, tiles = tiles_x * (a.H >> 4);
    // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so consecutive ids would put
    // neighbouring tiles - which share halo columns/rows - on different L2s.  Give every XCD a contiguous run of tiles instead.
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int n0 = bid / tiles, t = bid - n0 * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
    const int n = (ABL & 64) ? 0 : n0;   // timing-only build: every block's addresses collapse onto block 0 (L2-resident working set)
    // ABL 512: only the READS collapse onto block 0 (input and residual L2-resident, the output still goes to its own addresses): what a
    // layer costs when its inputs come from cache and only its output travels to HBM
    const int n_in = (ABL & 512) ? 0 : n;
    if ((ABL & 512) && a.res) a.res -= (size_t)n0 * NT * a.H * a.W * 16;
    typedef WaveTile<NT, W8> WT;
    constexpr int RW = WT::RW, CW = WT::CW;

    f32x4 acc[RW][CW];
#pragma unroll
    for (int m = 0; m < RW; ++m)
#pragma unroll
        for (int nt = 0; nt < CW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned long long t_begin = (ABL & 128) ? h2_stamp() : 0;
    {
        H2Carry<KH, KW, NT, W8> carry;
        h2_accumulate<KH, KW, NT, ABL, false, LEAN, W8>(a.x, a.x_stride, a.w, a.Cin, a.H, a.W, n_in, ty, tx, lds, acc,
                                       a.abl.dbg ? a.abl.dbg + (size_t)blockIdx.x * 16 : nullptr, carry, true, n_in, ty, tx);
    }
    const unsigned long long t_acc = (ABL & 128) ? h2_stamp() : 0;
    if (SC == 2) {
        h2_shortcut32<NT, W8>(a, n, ty, tx, lds, acc);
    } else if (SC) {
        H2Carry<1, 1, NT, W8> carry;
        h2_accumulate<1, 1, NT, 0, false, LEAN, W8>(a.x_sc, a.sc_stride, a.w_sc, a.Csc, a.H, a.W, n, ty, tx, lds, acc, nullptr, carry, true, n, ty, tx);
    }
    h2_epilogue<NT, ABL, W8>(a, acc, n, ty, tx);
    if ((ABL & 128) && a.abl.dbg && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // include the store acknowledgements in the epilogue span
        const unsigned long long t_end = h2_stamp();
        unsigned long long *d = a.abl.dbg + (size_t)blockIdx.x * 16;
        d[4] = t_acc - t_begin; d[5] = t_end - t_acc; d[6] = t_begin; d[7] = t_end;
        unsigned hw, xcc;   // where the workgroup ran: HW_ID (cu/sh/se in bits 8..15) and XCC_ID
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        d[8] = hw; d[9] = xcc;
    }
}

#ifdef PMP_ABLATION   // A/B forms that lost their measurement: built into libpmp_hip_abl.so only (make abl)
// Persistent form of the same kernel for the Cout = 64 layers without a shortcut source and with an even group count: 2
// workgroups per CU walk the tiles of their XCD's contiguous share.  The last channel group of a tile requests the first
// group of the workgroup's next tile and the weight stream wraps around, so a tile starts at its first MFMA: no dispatch
// gap, no prologue (halo round trip + barriers) between tiles.  It is NOT faster (the layer is bound by what the K-loop and
// the epilogue move, not by the gaps), so launch_h2 only takes it on request.
template <int KH, int KW, int NT, int ABL = 0, bool LEAN = false>
__global__ __launch_bounds__(256, LEAN ? 3 : 2) void conv_h2_persist_kernel(ConvX6Args a)
{
    typedef GeoH<KH, KW> G;
    __shared__ u32x4 lds[2 * G::BUF];
    typedef WaveTile<NT> WT;
    constexpr int RW = WT::RW, CW = WT::CW;
    const int tiles_x = a.W >> 4, tiles = tiles_x * (a.H >> 4);
    const int per_xcd = (a.N * tiles) >> 3, xcd = blockIdx.x & 7, S = gridDim.x >> 3;   // launch_h2 guarantees the divisibility
    H2Carry<KH, KW, NT> carry;
    bool first = true;
    for (int t = blockIdx.x >> 3; t < per_xcd; t += S) {
        const int tile = xcd * per_xcd + t, tile2 = t + S < per_xcd ? tile + S : tile;   // the last tile re-requests itself (unused)
        const int n = tile / tiles, tt = tile - n * tiles, ty = tt / tiles_x, tx = tt - ty * tiles_x;
        const int n2 = tile2 / tiles, tt2 = tile2 - n2 * tiles, ty2 = tt2 / tiles_x, tx2 = tt2 - ty2 * tiles_x;
        f32x4 acc[RW][CW];
#pragma unroll
        for (int m = 0; m < RW; ++m)
#pragma unroll
            for (int nt = 0; nt < CW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        h2_accumulate<KH, KW, NT, ABL, true, LEAN>(a.x, a.x_stride, a.w, a.Cin, a.H, a.W, n, ty, tx, lds, acc, nullptr, carry, first, n2, ty2, tx2);
        first = false;
        h2_epilogue<NT, ABL>(a, acc, n, ty, tx);
    }
}

// ---- layer-pipelined persistent form of a TRUNK of 3x3 64->64 layers (round 4, tools/trunk_probe.py).  With one launch per layer every
// tensor of the trunk crosses HBM two or three times (out, in, residual): 3.2-3.7 GB per 1024-block launch at 3.7-4.1 TB/s.  With its
// reads served from cache the same kernel runs 10-16 % faster (ABL 512: 0.806 -> 0.680 ms with residual, 0.718 -> 0.649 without).  This
// kernel tries to get there for real: resident workgroups take tiles (layer, block, tile) from ONE work counter in wave-front order -
// at block step s layer l works on block s - l * D - so that a layer's output block is consumed by the next layer ~D * L * 16 tiles
// later, while it is still in the 256 MB Infinity Cache.  A tile of layer l waits until all 16 tiles of block b of layer l - 1 are
// published (a counter per layer and block, release / acquire at agent scope); dependencies only point at smaller work ids, so the
// scheme cannot deadlock whatever the residency.  The tile body is conv_h2_kernel's: bit-identical results.
struct TrunkPipeArgs {
    const ConvX6Args *layers;     // L descriptors in global memory: same N, H, W; 3x3, Cin = Cout = 64, no shortcut source
    int L, N, D;                  // layers, blocks, delay in blocks between consecutive layers
    unsigned *work, *done, *err;  // zeroed: work counter, tile counters [L][N], error flag (spin limit hit)
};

// SYNC (timing-only switches, results may be wrong): 1 no release fence, 2 relaxed flag loads (no cache invalidate), 4 no waiting at all
template <int SYNC>
__global__ __launch_bounds__(256, 3) void trunk_pipe_kernel(TrunkPipeArgs p)
{
    typedef GeoH<3, 3> G;
    __shared__ u32x4 lds[2 * G::BUF];
    __shared__ unsigned s_wk;
    typedef WaveTile<4> WT;
    constexpr int RW = WT::RW, CW = WT::CW;
    const int tiles_x = p.layers[0].W >> 4, T = tiles_x * (p.layers[0].H >> 4), per_step = p.L * T;
    const unsigned total = (unsigned)(p.N + (p.L - 1) * p.D) * (unsigned)per_step;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_wk = atomicAdd(p.work, 1u);
        __syncthreads();
        const unsigned wk = s_wk;
        if (wk >= total) break;
        const int s = (int)(wk / (unsigned)per_step), r = (int)(wk - (unsigned)s * (unsigned)per_step), l = r / T, t = r - l * T, b = s - l * p.D;
        if (b < 0 || b >= p.N) continue;
        if (l > 0 && !(SYNC & 4)) {
            if (threadIdx.x == 0) {
                unsigned *flag = p.done + (size_t)(l - 1) * p.N + b;
                unsigned spins = 0;
                while (((SYNC & 2) ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                   : __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < (unsigned)T) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1u << 22)) { atomicExch(p.err, 1u); break; }      // never hang the box: give up, report
                }
            }
            __syncthreads();
        }
        const ConvX6Args &a = p.layers[l];       // (a reference: the fields are fetched where they are used - a local copy costs 40 SGPRs for the whole tile)
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        f32x4 acc[RW][CW];
#pragma unroll
        for (int m = 0; m < RW; ++m)
#pragma unroll
            for (int nt = 0; nt < CW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            H2Carry<3, 3, 4> carry;
            h2_accumulate<3, 3, 4, 0, false, true>(a.x, a.x_stride, a.w, 64, a.H, a.W, b, ty, tx, lds, acc, nullptr, carry, true, b, ty, tx);
        }
        h2_epilogue<4, 0>(a, acc, b, ty, tx);
        __syncthreads();          // every wave's stores are complete (vmcnt(0) in front of the barrier)
        if (threadIdx.x == 0) {
            if (!(SYNC & 1)) __threadfence();      // ... and visible at agent scope before the tile is published
            atomicAdd(p.done + (size_t)l * p.N + b, 1u);
        }
    }
}

hipError_t launch_trunk_pipe(hipStream_t s, const TrunkPipeArgs &p, int grid, int sync)
{
    if (p.L < 1 || p.N < 1 || p.D < 1 || grid < 1) return hipErrorInvalidValue;
    switch (sync) {
    case 0: hipLaunchKernelGGL((trunk_pipe_kernel<0>), dim3(grid), dim3(256), 0, s, p); break;
    case 1: hipLaunchKernelGGL((trunk_pipe_kernel<1>), dim3(grid), dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL((trunk_pipe_kernel<2>), dim3(grid), dim3(256), 0, s, p); break;
    case 3: hipLaunchKernelGGL((trunk_pipe_kernel<3>), dim3(grid), dim3(256), 0, s, p); break;
    case 7: hipLaunchKernelGGL((trunk_pipe_kernel<7>), dim3(grid), dim3(256), 0, s, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- loader-wave form (Cout = 64, no shortcut source, even number of channel groups; opt-in, pmp_debug_set_conv_variant(6)).
// A wave's loads return IN ORDER: in the kernels above every halo request a compute wave issues stands in front of that wave's own
// weight stream (L1 hits) for a whole HBM round trip - the timing-only builds of conv_f16x3_t32.hip put 27 % of a launch on exactly
// that.  Here a FIFTH wave of the workgroup (threads 256..319) does nothing but fill the halo buffers by LDS-DMA, a whole channel
// group ahead; the four compute waves run the LEAN K-loop with no halo request in their queues.  MEASURED SLOWER than the default:
// 327 against 385 TFLOP/s on the 3x3 64->64 class (5x5: 480 against 520) at 146 VGPRs, where 12 wave slots per CU hold two five-wave
// workgroups = 8 compute waves instead of 12; 310-320 at 128 VGPRs with three workgroups resident (occupancy API), where one round
// of workgroups takes 43.2 us against the default's 38.7: at equal residency a workgroup lives LONGER with the loader.  Together
// with the counters (matrix pipes busy 69 % of the cycles at an in-kernel clock of 1.67 GHz: 0.69 x 1.67 / 2.4 = the 0.48 of the
// roofline) the reading is that this class is no longer latency-bound: stalls removed come back as a lower clock.
// The loader takes part in every barrier of the staging protocol:
//   B0: group 0 is in buffer 0        E(g): group g is consumed, group g+1 has landed        X(g), odd g: the cross step has read
//   the even group's last tap, its buffer may be refilled
// Same K order, same products per accumulator: bit-identical results.
__device__ __forceinline__ void h2_dma16(const void *src, unsigned lds_byte_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_byte_base) : "memory");
}

template <int KH, int KW>
__device__ __forceinline__ void h2_loader(const ConvX6Args &a, int n, int ty, int tx, u32x4 *lds)
{
    typedef GeoH<KH, KW, 16, 256, true> G;
    constexpr int PY = KH / 2, PX = KW / 2;
    const int lane = threadIdx.x & 63, H = a.H, W = a.W, CB = a.Cin >> 4;
    const size_t grp_sz = (size_t)H * W * 16;
    const unsigned short *grp0 = a.x + (size_t)n * CB * grp_sz;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    // the per-lane source offsets are the same for every channel group: worked out once (21 / 25 registers - the loader has nothing
    // else to keep), so that a group's DMAs go out back to back
    unsigned off[G::NQ];
    unsigned in_image = 0;
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
        const int i = q * 64 + lane, ic = min(i, G::PIECES - 1);
        const int sp = ic >= G::PLANE ? 1 : 0, j = ic - sp * G::PLANE, pix = j >> 1, half = j & 1;
        const int row = pix / G::TW, col = pix - row * G::TW;
        const int gy = ty * 16 + row - PY, gx = tx * 16 + col - PX;
        if (i < G::PIECES && gy >= 0 && gy < H && gx >= 0 && gx < W) in_image |= 1u << q;   // else zero padding (or a dummy piece past the tile)
        off[q] = (unsigned)((size_t)sp * a.x_stride + ((size_t)max(gy, 0) * W + max(gx, 0)) * 16 + half * 8);
    }
    auto fill = [&](int cb, int buf) __attribute__((always_inline)) {
        const unsigned short *grp = grp0 + (size_t)cb * grp_sz;
        const unsigned lb = lds_base + (unsigned)(buf * G::BUF * 16);
#pragma unroll
        for (int q = 0; q < G::NQ; ++q) {
            const void *src = ((in_image >> q) & 1u) ? (const void *)(grp + off[q]) : a.abl.zeros;
            h2_dma16(src, __builtin_amdgcn_readfirstlane(lb + (unsigned)(q * 1024)));
        }
    };
    // (A third buffer with the loader two groups ahead measured no better: 313 against 327 TFLOP/s - the waits are not what binds.)
    fill(0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                       // B0
    for (int cb = 0; cb < CB; ++cb) {
        if (cb & 1) asm volatile("s_barrier" ::: "memory");                             // X(cb): buffer (cb + 1) & 1 is free now
        if (cb + 1 < CB) fill(cb + 1, (cb + 1) & 1);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                   // E(cb)
    }
}

template <int KH, int KW>
__global__ __launch_bounds__(320, 3) void conv_h2_ld_kernel(ConvX6Args a)
{
    typedef GeoH<KH, KW, 16, 256, true> G;
    __shared__ u32x4 lds[2 * G::BUF];
    const int tiles_x = a.W >> 4, tiles = tiles_x * (a.H >> 4);
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int n = bid / tiles, t = bid - n * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
    if (threadIdx.x >= 256) {   // wave 4: the loader
        h2_loader<KH, KW>(a, n, ty, tx, lds);
        return;
    }
    typedef WaveTile<4, 3> WT;
    f32x4 acc[WT::RW][WT::CW];
#pragma unroll
    for (int m = 0; m < WT::RW; ++m)
#pragma unroll
        for (int nt = 0; nt < WT::CW; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        H2Carry<KH, KW, 4, 3> carry;
        h2_accumulate<KH, KW, 4, 0, false, true, 3>(a.x, a.x_stride, a.w, a.Cin, a.H, a.W, n, ty, tx, lds, acc, nullptr, carry, true, n, ty, tx);
    }
    h2_epilogue<4, 0, 3>(a, acc, n, ty, tx);
}

#endif   // PMP_ABLATION

template <int NT, int ABL, int W8>
__device__ __forceinline__ void h2_epilogue(const ConvX6Args &a, f32x4 (&acc)[WaveTile<NT, W8>::RW][WaveTile<NT, W8>::CW], int n, int ty, int tx)
{
    typedef WaveTile<NT, W8> WT;
    constexpr int RW = WT::RW, CW = WT::CW;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // opaque: in the persistent kernels nothing lane-dependent of the epilogue is hoisted out of the tile loop
    const int lane = tid & 63, wave = tid >> 6, xl = lane & 15, g = lane >> 4;
    const int rh = wave % WT::RSPLIT, ch = wave / WT::RSPLIT;
    const int H = a.H, W = a.W;
    const size_t grp = (size_t)H * W * 16;

    const float inv_scale = a.out_scale;
    // element offset of (row 0, cout group 0) of this wave inside a [n][NT][H][W][16] tensor: < 2^32 for every chunk size
    const unsigned off0 = (unsigned)(((size_t)n * NT + ch * CW) * grp + ((size_t)(ty * 16 + rh * RW) * W + tx * 16 + xl) * 16 + g * 4);
    const unsigned row_el = (unsigned)W * 16;
    if (ABL & 8) {  // timing-only build: skip the epilogue but keep the accumulators live
        float sacc = 0.f;
#pragma unroll
        for (int nt = 0; nt < CW; ++nt)
#pragma unroll
            for (int m = 0; m < RW; ++m) sacc += acc[m][nt].x + acc[m][nt].y + acc[m][nt].z + acc[m][nt].w;
        if (sacc == 123.456f) a.out[0] = 1;
        return;
    }
    // (A straight-line special case for ReLU + split-2 output without gate/pool measured 2 % SLOWER than this general path
    // with its wave-uniform branches per cout group: the branches keep one group's stores ahead of the next group's conversions.)
    // All residual (then gate) fragments are requested before the first store: the weight, pixel and staging registers are
    // dead by now, and a load issued after a store to `out` would otherwise have to wait for it (possible aliasing).
    // Residual loads and output stores move 16 bytes per lane: lane (xl, g) handles the 8 consecutive channels 8(g>>1).. of pixel
    // (row m + (g&1), xl) - one v_permlane16_swap per register (gfx950) exchanges that form with the accumulator layout (4
    // couts 4g.. of rows m and m+1).  Half the vector-memory instructions of 8-byte accesses: the epilogue is issue-bound.
    const unsigned off0w = off0 - (unsigned)(g * 4) + (unsigned)(8 * (g >> 1)) + (unsigned)(g & 1) * row_el;
    // CHUNK (loader-wave form, 128 VGPRs): one cout group's residual fragments at a time - the next group's are requested after this
    // group's values are final and BEFORE its stores - instead of all of them up front (64 registers next to 64 accumulators).
    constexpr bool CHUNK = W8 == 3;
    u32x4 ra[RW / 2][CHUNK ? 1 : CW], rbv[RW / 2][CHUNK ? 1 : CW];
    auto res_load = [&](int nt, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < RW; m += 2) {
            const unsigned off = off0w + (unsigned)m * row_el + (unsigned)nt * (unsigned)grp;
            ra[m >> 1][slot] = *reinterpret_cast<const u32x4 *>(a.res + off);
            rbv[m >> 1][slot] = *reinterpret_cast<const u32x4 *>(a.res + off + a.res_stride);
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
        if (CHUNK) { res_load(0, 0); res_add(0, 0); }
        else {
#pragma unroll
            for (int nt = 0; nt < CW; ++nt) res_load(nt, CHUNK ? 0 : nt);
#pragma unroll
            for (int nt = 0; nt < CW; ++nt) res_add(nt, CHUNK ? 0 : nt);
        }
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
        if (CHUNK && a.res && nt + 1 < CW) res_load(nt + 1, 0);   // before this group's stores (the output may alias the residual tensor)
        if (!a.pool) {
            if (a.out_f32) {
#pragma unroll
                for (int m = 0; m < RW; ++m)
                    *reinterpret_cast<f32x4 *>(a.out_f32 + off0 + (unsigned)m * row_el + (unsigned)nt * (unsigned)grp) = acc[m][nt];
            } else {
#pragma unroll
                for (int m = 0; m < RW; m += 2) {
                    const unsigned off = off0w + (unsigned)m * row_el + (unsigned)nt * (unsigned)grp;
                    if ((ABL & 16) && a.N > 0) { f32x4 q = acc[m][nt] + acc[m + 1][nt]; _Float16 h0, h1; split2(q.x + q.y + q.z + q.w, h0, h1); if ((float)h0 + (float)h1 == 123.456f) a.out[off] = 1; continue; }   // timing-only: conversion work without the stores
                    u32x4 p, q;
                    split2_rows(acc[m][nt], acc[m + 1][nt], p, q);
                    rows16_swap(p);
                    rows16_swap(q);
                    *reinterpret_cast<u32x4 *>(a.out + off) = p;
                    *reinterpret_cast<u32x4 *>(a.out + off + a.out_stride) = q;
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
        if (CHUNK && a.res && nt + 1 < CW) res_add(nt + 1, 0);
    }
    if (!a.out_f32) sat_report(a.sat, amax);   // fp32 outputs are not clamped
}

// The 1x1 shortcut source is a separate instantiation: its extra live state would spill in the common kernel.
#define PMP_H2_LAUNCH(NT)                                                                                          \
    if (a.x_sc) hipLaunchKernelGGL((conv_h2_kernel<KH, KW, NT, true>), dim3(grid), dim3(256), 0, s, a);           \
    else hipLaunchKernelGGL((conv_h2_kernel<KH, KW, NT, false>), dim3(grid), dim3(256), 0, s, a)

#ifdef PMP_ABLATION
// Measurement library only (make abl -> libpmp_hip_abl.so; tools/conv_ab.py, tools/variants_agree.py, tools/conv_x6_bench.py):
// the forms of the Cout = 64 kernels that were built, parity-tested and measured slower than or equal to the shipped ones
// (g_conv_variant 1, 3..8: bit-identical results), and the timing-only builds (>= 10: WRONG results).  Returns true if it launched.
//   1 = the 32-channel-shortcut instantiations at three workgroups per CU     3 = two workgroups per CU (236-256 VGPRs) / general shortcut pass
//   4 = the two-workgroup form made persistent    5 = the default form made persistent (3x3)    6 = loader-wave form (LDS-DMA halo)
//   7 = 512-thread workgroups                     8 = 16-row x 1-cout-group wave tiles (3x3)
template <int KH, int KW>
static bool launch_h2_variant(hipStream_t s, const ConvX6Args &a, int grid)
{
    if constexpr (KH > 1) {
        if (a.Cout != 64) return false;
        const int v = g_conv_variant;
        if (!a.x_sc && !((a.Cin >> 4) & 1) && a.abl.zeros && v == 6) {
            { static bool once = false; if (!once) { once = true; int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_h2_ld_kernel<KH, KW>, 320, 0);
              fprintf(stderr, "conv_h2_ld_kernel<%d,%d>: occupancy API says %d workgroups of 320 threads per CU\n", KH, KW, nb); } }
            hipLaunchKernelGGL((conv_h2_ld_kernel<KH, KW>), dim3(grid), dim3(320), 0, s, a);
            return true;
        }
        if (a.x_sc && a.Csc == 32 && v == 1) { hipLaunchKernelGGL((conv_h2_kernel<KH, KW, 4, 2, 0, true>), dim3(grid), dim3(256), 0, s, a); return true; }
        if (v == 3) { PMP_H2_LAUNCH(4); return true; }     // general shortcut pass / two-workgroup form
        if (a.x_sc) return false;
        if (KH == 5 && v >= 10) {   // timing-only ablation builds, 5x5
            switch (v - 10) {
            case 1: hipLaunchKernelGGL((conv_h2_kernel<5, 5, 4, false, 1>), dim3(grid), dim3(256), 0, s, a); return true;
            case 2: hipLaunchKernelGGL((conv_h2_kernel<5, 5, 4, false, 2>), dim3(grid), dim3(256), 0, s, a); return true;
            case 4: hipLaunchKernelGGL((conv_h2_kernel<5, 5, 4, false, 4>), dim3(grid), dim3(256), 0, s, a); return true;
            case 8: hipLaunchKernelGGL((conv_h2_kernel<5, 5, 4, false, 8>), dim3(grid), dim3(256), 0, s, a); return true;
            case 9: hipLaunchKernelGGL((conv_h2_kernel<5, 5, 4, false, 9>), dim3(grid), dim3(256), 0, s, a); return true;
            case 15: hipLaunchKernelGGL((conv_h2_kernel<5, 5, 4, false, 15>), dim3(grid), dim3(256), 0, s, a); return true;
            case 32: hipLaunchKernelGGL((conv_h2_kernel<5, 5, 4, false, 32>), dim3(grid), dim3(256), 0, s, a); return true;
            default: PMP_H2_LAUNCH(4); return true;
            }
        }
        if (KH == 3 && v >= 10) {   // timing-only ablation builds, 3x3
            switch (v - 10) {
            case 1: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 1>), dim3(grid), dim3(256), 0, s, a); return true;
            case 2: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 2>), dim3(grid), dim3(256), 0, s, a); return true;
            case 4: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 4>), dim3(grid), dim3(256), 0, s, a); return true;
            case 8: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 8>), dim3(grid), dim3(256), 0, s, a); return true;
            case 9: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 9>), dim3(grid), dim3(256), 0, s, a); return true;
            case 15: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 15>), dim3(grid), dim3(256), 0, s, a); return true;
            case 16: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 16>), dim3(grid), dim3(256), 0, s, a); return true;
            case 32: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 32>), dim3(grid), dim3(256), 0, s, a); return true;
            case 64: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 64>), dim3(grid), dim3(256), 0, s, a); return true;
            case 128: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 128>), dim3(grid), dim3(256), 0, s, a); return true;
            case 256: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 256>), dim3(grid), dim3(256), 0, s, a); return true;
            case 1152: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 128, true>), dim3(grid), dim3(256), 0, s, a); return true;   // stamps of the default (three-workgroup) form
            case 1088: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 64, true>), dim3(grid), dim3(256), 0, s, a); return true;    // three-workgroup form, every address on block 0
            case 1536: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 512, true>), dim3(grid), dim3(256), 0, s, a); return true;   // three-workgroup form, reads on block 0, writes where they belong
            case 129: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 129>), dim3(grid), dim3(256), 0, s, a); return true;
            case 130: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 130>), dim3(grid), dim3(256), 0, s, a); return true;
            case 131: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 131>), dim3(grid), dim3(256), 0, s, a); return true;
            case 135: hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 135>), dim3(grid), dim3(256), 0, s, a); return true;
            default: PMP_H2_LAUNCH(4); return true;
            }
        }
        if (KH == 3 && v == 8) { hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 0, true, 2>), dim3(grid), dim3(256), 0, s, a); return true; }
        if (KH == 3 && v == 7) { hipLaunchKernelGGL((conv_h2_kernel<3, 3, 4, false, 0, true, 1>), dim3(grid), dim3(512), 0, s, a); return true; }
        if (!((a.Cin >> 4) & 1) && !(grid & 7) && v == 4) {      // persistent, 2 workgroups per CU, 64 per XCD: 2 % slower
            hipLaunchKernelGGL((conv_h2_persist_kernel<KH, KW, 4>), dim3(8 * min(64, grid >> 3)), dim3(256), 0, s, a);
            return true;
        }
        if (KH == 3 && !((a.Cin >> 4) & 1) && !(grid & 7) && v == 5) {   // persistent three-workgroup form: spills, 17 % slower
            hipLaunchKernelGGL((conv_h2_persist_kernel<3, 3, 4, 0, true>), dim3(8 * min(96, grid >> 3)), dim3(256), 0, s, a);
            return true;
        }
    }
    return false;
}
#endif   // PMP_ABLATION

template <int KH, int KW>
static hipError_t launch_h2(hipStream_t s, const ConvX6Args &a)
{
    const int grid = a.N * (a.H >> 4) * (a.W >> 4);
#ifdef PMP_ABLATION
    if (launch_h2_variant<KH, KW>(s, a, grid)) return hipGetLastError();
#endif
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
                hipLaunchKernelGGL((conv_h2_kernel<KH, KW, 4, false, 0, true>), dim3(grid), dim3(256), 0, s, a);
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
#ifdef PMP_ABLATION
    if (conv_h2_wx_applicable(a)) return launch_conv_h2_wx(s, a);     // Winograd-x form of the 3x3 64->64 blocks (the caller set w_wx)
    if ((g_conv_variant == 9 || (g_conv_variant >= 90 && g_conv_variant < 200)) && conv_h2_t32_applicable(a)) return launch_conv_h2_t32(s, a);   // 32x16 tiles, LDS-DMA (A/B: opt-in)
#endif
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
