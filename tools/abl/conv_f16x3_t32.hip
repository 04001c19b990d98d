// conv_f16x3_t32.hip — the 3x3 64->64 convolution of the f16x3 datapath (55 % of a luma pass) in a second, structurally different
// form.  MEASURED SLOWER than the 16x16-tile kernel of conv_f16x3.hip (372 vs 404-412 TFLOP/s on the same box,
// profiles/r02_t32_experiment.txt), so launch_conv_h2 only takes it on request (pmp_debug_set_conv_variant(9)); it is kept, tested
// bit-identical, because its timing-only builds (tools/t32_ablate.py) are what shows WHY this layer sits where it sits:
//
//   * workgroup tile 32 rows x 16 columns, a wave owns 8 rows x ALL 64 couts (128 accumulator registers, 2 waves per SIMD): every
//     pixel fragment read from LDS feeds 4 cout groups instead of 2, the per-visit costs (first halo round trip, barriers,
//     epilogue) are paid half as often per pixel, the weight bytes per MFMA are those of the 16x16 kernel;
//   * halo tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, out-of-image pixels from a zero line): no staging
//     registers, no LDS store instructions;
//   * EVERY vector-memory operation of the K-loop is issued from inline asm and the in-order vmcnt counter is counted by hand
//     (hipcc either drains the counter around the DMA builtin or, next to asm loads, does not count them);
//   * a K-step runs as six half-phases of 16 MFMAs (x0*w1, x0*w0, x1*w0 over rows 0-3 / 4-7 x 4 cout groups); pixel fragments
//     travel in four-row sets that are re-read as soon as their last MFMA has issued, a weight split is refilled in place (ONE
//     register set) right after the last half-phase that reads it; the 18 K-steps are fully unrolled.
//
// What the builds say (one 3x3 64->64 layer, 1024 blocks of 64x64, ms per launch; 16x16 kernel: 0.750):
//   halo DMA sliced into the K-steps (4+3+3 pieces behind each K-step's weight requests)        0.940
//     ... without those DMAs 0.683 | without the epilogue 0.891 | without both 0.518 (597 TFLOP/s) | MFMAs only 0.496
//   both halo buffers refilled in ONE burst at the pair boundary, nothing but weights in the K-steps   0.832 (this file)
//     ... without the epilogue 0.665
// A wave's loads return in order: a weight fragment requested behind a halo request cannot be used before that HBM round trip
// has ended, however the waits are counted - and every K-step requests weights.  Taking the halo requests out of the compute
// waves needs loader waves, which at 256 VGPRs per wave (2 waves per SIMD are all the register file holds) would replace
// compute waves; taking the weights off the vector-memory path needs them in LDS, where two 40 KB halo buffers per workgroup
// leave no room.  With two waves per SIMD nothing covers a workgroup's prologue, refill and epilogue; the 16x16 kernel's third
// wave (168 VGPRs, 64 accumulators) does, and wins.  DESIGN.md 4.1a has the full account.
//
// Same arithmetic as conv_h2_kernel: same weight stream (pack_h2, paired tap order), same product order per accumulator
// (x0*w1, x0*w0, x1*w0), so the results are bit-identical to the other forms of the kernel (tests/test_gpu_parity.py).
//
// LDS: two halo buffers of one 16-channel group each, [plane][34 x 18 pixels][2 halves] 16-byte pieces = 39 168 B, padded to
// 40 960 B per buffer (dummy pieces, see T32_BUF): 81 920 B per workgroup -> exactly two workgroups (8 waves) per CU.
//
// Synchronisation: groups 0, 1 -> buffers 0, 1 in the prologue; K-steps 0..8 (pair 0; r = 4 is the cross step: tap 8 of the even
// group for lanes g < 2, tap 8 of the odd group for lanes g >= 2); barrier, burst refill with groups 2, 3, vmcnt(0), barrier;
// K-steps 9..17.
#include "abl_kernels.h"
#include "split3.h"

namespace pmp {

namespace {

constexpr int T32_TH = 34, T32_TW = 18;
constexpr int T32_PLANE = T32_TH * T32_TW * 2;      // 16-byte pieces per split plane
constexpr int T32_PIECES = 2 * T32_PLANE;           // per buffer (2448)
constexpr int T32_NDMA = (T32_PIECES + 255) / 256;  // DMA instructions per thread and group (10)
constexpr int T32_BUF = T32_NDMA * 256;             // pieces reserved per buffer (2560 = 40 960 B): EVERY lane issues all 10 DMAs - the
                                                    // 112 pieces past the tile are dummies fed from the zero line - so that all waves
                                                    // have the same number of operations in flight and one vmcnt count fits them all
constexpr int T32_NKS = 18;                         // K-steps per tile at Cin = 64

// DMA instructions issued at the end of K-step ks, behind both weight requests (the wait counts of the K-steps are written for
// any such schedule; the sliced forms measured in the header used 4+3+3 / 5+5 here).  Final form: none - see the pair boundary.
constexpr int t32_nd(int) { return 0; }

// LDS-DMA of one 16-byte piece per lane: LDS destination = M0 + lane * 16 (wave-uniform base), source per lane.
__device__ __forceinline__ void t32_dma16(const void *src, unsigned lds_byte_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_byte_base) : "memory");
}

// One weight fragment (16 bytes per lane) from the K-step block at `base`; counted by hand (t32_wait_weights).
template <int OFF>
__device__ __forceinline__ void t32_wload(f16x8 &dst, const char *ptr)
{
    asm volatile("global_load_dwordx4 %0, %1, off offset:%c2" : "=v"(dst) : "v"(ptr), "i"(OFF) : "memory");
}

// Wait until at most N vector-memory operations of this wave are outstanding; names the weight registers so that no use of
// them is scheduled above the wait.
template <int N>
__device__ __forceinline__ void t32_wait4(f16x8 (&w)[4])
{
    asm volatile("s_waitcnt vmcnt(%c4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "i"(N) : "memory");
}

__device__ __forceinline__ void t32_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int N>
__device__ __forceinline__ void t32_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(0)\n\ts_barrier" :: "i"(N) : "memory");
}

}  // namespace

// ABL: timing-only builds (libpmp_hip_abl.so only): 2 no weight refills, 4 no fragment reads, 8 no epilogue
template <int ABL>
__global__ __launch_bounds__(256, 2) void conv_h2_t32_kernel(ConvX6Args a)
{
    __shared__ u32x4 lds[2 * T32_BUF];
    const int tiles_x = a.W >> 4, tiles = tiles_x * (a.H >> 5);
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);   // XCD-contiguous tile runs (conv_f16x3.hip)
    const int n = bid / tiles, t = bid - n * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, xl = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const size_t grp_sz = (size_t)H * W * 16;
    const unsigned short *xg0 = a.x + (size_t)n * 4 * grp_sz;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;

    // ---- halo DMA: piece i = (4k + wave) * 64 + lane of [plane][pixel][half].  The source address is worked out where the piece is
    // issued (k is a compile-time constant at every call, ~20 integer instructions that hide under the MFMAs) instead of being
    // kept in 11 registers for the whole tile: the kernel sits at the 256-VGPR line.
    auto dma = [&](int grp, int buf, int k) __attribute__((always_inline)) {
        int ln = lane;
        asm volatile("" : "+v"(ln));   // opaque here: hipcc would otherwise hoist the address arithmetic of all 40 DMAs to the kernel's top
        const int i = (4 * k + wave) * 64 + ln;
        const int ic = min(i, T32_PIECES - 1);
        const int sp = ic >= T32_PLANE ? 1 : 0, j = ic - sp * T32_PLANE, pix = j >> 1, half = j & 1;
        const int row = pix / T32_TW, col = pix - row * T32_TW;
        const int gy = ty * 32 + row - 1, gx = tx * 16 + col - 1;
        const bool in_image = i < T32_PIECES && gy >= 0 && gy < H && gx >= 0 && gx < W;   // else: zero padding, or a dummy piece
        const size_t off = (size_t)sp * a.x_stride + (size_t)grp * grp_sz + ((size_t)(gy * W + gx) * 16 + half * 8);
        const void *src = in_image ? (const void *)(xg0 + off) : a.abl.zeros;
        t32_dma16(src, lds_base + (unsigned)((buf * T32_BUF + (4 * k + wave) * 64) * 16));
    };

    // ---- weights: per K-step [2 splits][4 cout groups][64 lanes][16 B] = 8 KiB; pointer centred for the 13-bit offsets.
    // ONE register set, refilled in place: a split's fragments for K-step k+1 are requested right after the last MFMA of K-step k
    // that reads them (w1 after phase A, w0 after phase B2).
    const char *wp = reinterpret_cast<const char *>(a.w) + lane * 16 + 4096;
    f16x8 w0[4], w1[4];
    auto req_w0 = [&](int ks) __attribute__((always_inline)) {
        const char *p = wp + (size_t)ks * 8192;
        t32_wload<0 * 1024 - 4096>(w0[0], p); t32_wload<1 * 1024 - 4096>(w0[1], p);
        t32_wload<2 * 1024 - 4096>(w0[2], p); t32_wload<3 * 1024 - 4096>(w0[3], p);
    };
    auto req_w1 = [&](int ks) __attribute__((always_inline)) {
        const char *p = wp + (size_t)ks * 8192;
        t32_wload<4 * 1024 - 4096>(w1[0], p); t32_wload<5 * 1024 - 4096>(w1[1], p);
        t32_wload<6 * 1024 - 4096>(w1[2], p); t32_wload<7 * 1024 - 4096>(w1[3], p);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: groups 0 and 1 -> buffers 0 and 1, weights of K-step 0
    req_w1(0);
    req_w0(0);
#pragma unroll
    for (int k = 0; k < T32_NDMA; ++k) dma(0, 0, k);
#pragma unroll
    for (int k = 0; k < T32_NDMA; ++k) dma(1, 1, k);
    t32_wait_barrier<0>();

    // per-lane byte offset of (row 8*wave, column xl, half g&1) at tap (0,0) inside a plane
    const unsigned lane_off = (unsigned)(((wave * 8) * T32_TW + xl) * 32 + (g & 1) * 16);
    const bool tapsel = (g >> 1) != 0;
    const char *ldsb = reinterpret_cast<const char *>(lds);
    // Pixel fragments travel in four-row sets of 16 registers; three sets are live at any time:
    //   P, Q = high terms (plane 0) of rows 0-3 / 4-7,  S, T = low terms (plane 1) of rows 0-3 / 4-7.
    // A set is re-read as soon as its last MFMA has issued (S before phase A, T when P dies, the next K-step's P when Q dies, its
    // Q when S dies), so every LDS read has two or more 16-MFMA half-phases to land.
    f16x8 P[4], Q[4];

    // LDS address of this lane's fragment of row 0 for K-step ks (cross step: lanes g < 2 read the EVEN group's last tap from the
    // other buffer)
    auto frag_addr = [&](auto ks_tag) __attribute__((always_inline)) -> const char * {
        constexpr int KS = decltype(ks_tag)::value;
        constexpr int R = KS % 9, GRP = 2 * (KS / 9) + (R >= 4 ? 1 : 0), BUF = GRP & 1;
        constexpr bool CROSS = R == 4;
        constexpr int TA = CROSS ? 8 : 2 * (R < 4 ? R : R - 5), TB = CROSS ? 8 : TA + 1;
        constexpr unsigned CA = (unsigned)(((CROSS ? BUF ^ 1 : BUF) * T32_BUF) * 16 + ((TA / 3) * T32_TW + TA % 3) * 32);
        constexpr unsigned CB = (unsigned)((BUF * T32_BUF) * 16 + ((TB / 3) * T32_TW + TB % 3) * 32);
        return ldsb + lane_off + (tapsel ? CB : CA);
    };
    auto read4 = [&](f16x8 (&x)[4], const char *px, int plane, int row0) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m) x[m] = *reinterpret_cast<const f16x8 *>(px + plane * (T32_PLANE * 16) + (row0 + m) * T32_TW * 32);
    };
    auto mma4 = [&](f16x8 (&w)[4], f16x8 (&x)[4], int row0) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[row0 + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[nt], x[m], acc[row0 + m][nt], 0, 0, 0);
    };
    {
        const char *px0 = frag_addr(std::integral_constant<int, 0>{});
        read4(P, px0, 0, 0);
        read4(Q, px0, 0, 4);
    }

    // One K-step = six half-phases of 16 MFMAs: x0*w1 (rows 0-3, 4-7), x0*w0 (0-3, 4-7), x1*w0 (0-3, 4-7).  Vector-memory operations
    // in program order per K-step k:  W1(k+1) x4 | W0(k+1) x4 | DMA slice x nd(k)   - the wait counts below follow from that order
    // (a slice comes LAST so that neither weight wait of the next K-step covers it; the K-step after that still does).
    auto kstep = [&](auto ks_tag) __attribute__((always_inline)) {
        constexpr int KS = decltype(ks_tag)::value;
        constexpr int PAIR = KS / 9, R = KS % 9;
        constexpr int GRP = 2 * PAIR + (R >= 4 ? 1 : 0);
        constexpr bool REQ = KS + 1 < T32_NKS;                           // there is a next K-step to request weights for
        constexpr int ND = t32_nd(KS), K0 = 0;
        constexpr bool BARRIER_DMA = R == 8 && PAIR == 0;               // pair boundary: both buffers are refilled (groups 2, 3)
        typedef std::integral_constant<int, (REQ ? KS + 1 : KS)> NextTag;
        const char *px = frag_addr(ks_tag);
        f16x8 S[4], T[4], Pn[4], Qn[4];
        // phases A need W1(KS): younger operations are W0(KS) and DMA(KS-1)
        if (KS > 0 && !(ABL & 2)) t32_wait4<4 + t32_nd(KS - 1)>(w1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 4) || KS == 0) read4(S, px, 1, 0); else { for (int m = 0; m < 4; ++m) S[m] = P[m]; }
        __builtin_amdgcn_sched_barrier(0);
        mma4(w1, P, 0);
        mma4(w1, Q, 4);
        __builtin_amdgcn_sched_barrier(0);
        if (REQ && !(ABL & 2)) req_w1(KS + 1);
        // phases B need W0(KS): younger operations are DMA(KS-1) and W1(KS+1)
        if (KS > 0 && !(ABL & 2)) t32_wait4<t32_nd(KS - 1) + (REQ ? 4 : 0)>(w0);
        __builtin_amdgcn_sched_barrier(0);
        mma4(w0, P, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 4) || KS == 0) read4(T, px, 1, 4); else { for (int m = 0; m < 4; ++m) T[m] = Q[m]; }   // P is dead
        __builtin_amdgcn_sched_barrier(0);
        mma4(w0, Q, 4);
        __builtin_amdgcn_sched_barrier(0);
        if (ABL & 4) { for (int m = 0; m < 4; ++m) { Pn[m] = P[m]; Qn[m] = Q[m]; } }
        if (REQ && !BARRIER_DMA && !(ABL & 4)) read4(Pn, frag_addr(NextTag{}), 0, 0);   // Q is dead; the next K-step's buffer is readable already
        __builtin_amdgcn_sched_barrier(0);
        mma4(w0, S, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (REQ && !BARRIER_DMA && !(ABL & 4)) read4(Qn, frag_addr(NextTag{}), 0, 4);   // S is dead
        __builtin_amdgcn_sched_barrier(0);
        mma4(w0, T, 4);
        __builtin_amdgcn_sched_barrier(0);
        if (REQ && !(ABL & 2)) req_w0(KS + 1);
        if (ND) {
#pragma unroll
            for (int k = K0; k < K0 + ND; ++k) dma(GRP + 1, (GRP + 1) & 1, k);
        }
        if (BARRIER_DMA) {
            // Pair boundary.  No vector-memory operation other than the weight stream runs inside the K-steps: a wave's loads return
            // in order, so a weight fragment requested behind a halo request cannot be used before that HBM round trip has ended.
            // Both buffers are refilled here in one burst instead, and the workgroup waits for it once - while the other
            // workgroup of the CU computes.
            t32_lds_barrier();                       // every wave is done reading both buffers
            {
#pragma unroll
                for (int k = 0; k < T32_NDMA; ++k) dma(GRP + 1, 0, k);
#pragma unroll
                for (int k = 0; k < T32_NDMA; ++k) dma(GRP + 2, 1, k);
            }
            t32_wait_barrier<0>();
            if (!(ABL & 4)) { read4(Pn, frag_addr(NextTag{}), 0, 0); read4(Qn, frag_addr(NextTag{}), 0, 4); }
        }
        if (REQ) {
#pragma unroll
            for (int m = 0; m < 4; ++m) { P[m] = Pn[m]; Q[m] = Qn[m]; }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

#define T32_STEP(KS) kstep(std::integral_constant<int, KS>{})
    T32_STEP(0);  T32_STEP(1);  T32_STEP(2);  T32_STEP(3);  T32_STEP(4);  T32_STEP(5);  T32_STEP(6);  T32_STEP(7);  T32_STEP(8);
    T32_STEP(9);  T32_STEP(10); T32_STEP(11); T32_STEP(12); T32_STEP(13); T32_STEP(14); T32_STEP(15); T32_STEP(16); T32_STEP(17);
#undef T32_STEP

    if (ABL & 8) {   // timing-only: keep the accumulators live, skip the epilogue
        float sacc = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) sacc += acc[m][nt].x + acc[m][nt].y + acc[m][nt].z + acc[m][nt].w;
        if (sacc == 123.456f) a.out[0] = 1;
        return;
    }
    // ---- epilogue: x 1/S, + residual, ReLU, (2x2 max-pool), split, 16-byte accesses through v_permlane16_swap (split3.h)
    const float inv_scale = a.out_scale;
    const unsigned grp = (unsigned)grp_sz, row_el = (unsigned)W * 16;
    const unsigned off0 = (unsigned)((size_t)n * 4 * grp_sz + ((size_t)(ty * 32 + wave * 8) * W + tx * 16 + xl) * 16);
    const unsigned off0w = off0 + (unsigned)(8 * (g >> 1)) + (unsigned)(g & 1) * row_el;
    float amax = 0.f;
    // Residual fragments are requested two cout groups ahead of their use, always BEFORE the stores of the group in between (the
    // output may alias the residual tensor: hipcc cannot move a load above such a store by itself, and a load issued after the
    // stores would wait for them).  The weight and pixel registers are dead here: 64 of them hold the two groups in flight.
    u32x4 ra[2][4], rb[2][4];
    auto res_load = [&](int nt, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 8; m += 2) {
            const unsigned off = off0w + (unsigned)m * row_el + (unsigned)nt * grp;
            ra[slot][m >> 1] = *reinterpret_cast<const u32x4 *>(a.res + off);
            rb[slot][m >> 1] = *reinterpret_cast<const u32x4 *>(a.res + off + a.res_stride);
        }
    };
    if (a.res) { res_load(0, 0); res_load(1, 1); }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        if (a.res) {
#pragma unroll
            for (int m = 0; m < 8; m += 2) {
                u32x4 p = ra[nt & 1][m >> 1], q = rb[nt & 1][m >> 1];
                rows16_swap(p);
                rows16_swap(q);
                acc[m][nt] = acc[m][nt] * inv_scale + (h2_lo4(p) + h2_lo4(q));
                acc[m + 1][nt] = acc[m + 1][nt] * inv_scale + (h2_hi4(p) + h2_hi4(q));
            }
            if (nt + 2 < 4) res_load(nt + 2, nt & 1);
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m][nt] = acc[m][nt] * inv_scale;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            f32x4 v = acc[m][nt];
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            amax = sat_amax4(amax, v);
            acc[m][nt] = v;
        }
        if (!a.pool) {
#pragma unroll
            for (int m = 0; m < 8; m += 2) {
                const unsigned off = off0w + (unsigned)m * row_el + (unsigned)nt * grp;
                u32x4 p, q;
                split2_rows(acc[m][nt], acc[m + 1][nt], p, q);
                rows16_swap(p);
                rows16_swap(q);
                *reinterpret_cast<u32x4 *>(a.out + off) = p;
                *reinterpret_cast<u32x4 *>(a.out + off + a.out_stride) = q;
            }
        } else {
            const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
            for (int m = 0; m < 8; m += 2) {
                f32x4 v = acc[m][nt], u = acc[m + 1][nt];
                v.x = fmaxf(v.x, u.x); v.y = fmaxf(v.y, u.y); v.z = fmaxf(v.z, u.z); v.w = fmaxf(v.w, u.w);
                f32x4 o;
                o.x = __shfl_xor(v.x, 1); o.y = __shfl_xor(v.y, 1); o.z = __shfl_xor(v.z, 1); o.w = __shfl_xor(v.w, 1);
                v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
                if ((xl & 1) == 0) {
                    const int yo = ty * 16 + wave * 4 + (m >> 1), xo = tx * 8 + (xl >> 1);
                    const size_t off = (((size_t)n * 4 + nt) * Ho + yo) * Wo * 16 + (size_t)xo * 16 + g * 4;
                    store_split2_4(a.out + off, a.out_stride, v);
                }
            }
        }
    }
    sat_report(a.sat, amax);
}

// 3x3, Cin = Cout = 64, no shortcut source, no gate, split-2 output, H a multiple of 32: the trunk layers M1.1-5 / M2.0-3.
bool conv_h2_t32_applicable(const ConvX6Args &a)
{
    return a.KH == 3 && a.KW == 3 && a.Cin == 64 && a.Cout == 64 && !a.x_sc && !a.gate && !a.out_f32 && a.out && (a.H & 31) == 0 &&
           (a.W & 15) == 0 && a.abl.zeros != nullptr && a.N > 0;
}

hipError_t launch_conv_h2_t32(hipStream_t s, const ConvX6Args &a)
{
    const int grid = a.N * (a.H >> 5) * (a.W >> 4);
#ifdef PMP_ABLATION   // timing-only builds (wrong results): measurement library only
    if (g_conv_variant >= 90) {
        switch (g_conv_variant - 90) {
        case 2: hipLaunchKernelGGL(conv_h2_t32_kernel<2>, dim3(grid), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(conv_h2_t32_kernel<4>, dim3(grid), dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL(conv_h2_t32_kernel<8>, dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(conv_h2_t32_kernel<0>, dim3(grid), dim3(256), 0, s, a); break;
        }
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(conv_h2_t32_kernel<0>, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace pmp
