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
// streams in the same K-step order (tail16_dev.h's list, restated for these image widths), the same three products per K-step into the
// same fp32 accumulators (x0*w1, x0*w0, x1*w0), the main pass before the shortcut pass, the epilogue arithmetic of conv_f16x3.hip
// (x 1/S, ReLU, range-flag maximum, pool, two-term split with the clamp).  A value of t that two tiles both compute is the same number
// in both.  LDS 46-72 KB: two workgroups of eight waves per CU.
#include "pmp_kernels.h"
#include "split3.h"

namespace pmp {

namespace {

#define RF_GLOBAL __attribute__((address_space(1)))
constexpr int RF_IW = 20, RF_TW = 18;                        // image widths: input (tile + 2 x 2), intermediate (tile + 2 x 1)
// Inside a plane: [channel half][pixel][8 channels = 16 B] - NOT the [pixel][32 B] of the launch kernels' halo tiles: 16 lanes that read 16
// neighbouring pixels of one half cover 256 contiguous bytes.  ds_read_b128 serves the wave in four 16-lane groups that are NOT the four
// quarters: lanes {0-3, 12-15} of one quarter go with lanes {4-11} of the NEXT (MI355X_MICROARCH.md, LDS) - the other channel half of the same
// pixels - so the two halves must lie a whole number of 256-byte bank rows apart to complement each other.  The 20x20 input image does
// (6400 B); the 18x18 intermediate did not until round 6 (5184 B: 4 of 16 slots collided) and is padded to 5376 now.  Round 5's PMC (r06d
// on the round-5 build): 40-45 % of these kernels' LDS cycles were bank conflicts, the LDS array 72 % busy.
constexpr int RF_IHH = RF_IW * RF_IW * 16, RF_THH = ((RF_TW * RF_TW * 16 + 255) / 256) * 256;  // bytes per channel half of a plane
static_assert(RF_IHH % 256 == 0 && RF_THH % 256 == 0, "channel halves a whole number of bank rows apart");
constexpr int RF_IPLN = 2 * RF_IHH, RF_ISLOT = 2 * RF_IPLN;            // bytes per fp16 plane / per 16-channel group (both planes)
constexpr int RF_TPLN = 2 * RF_THH, RF_TSLOT = 2 * RF_TPLN;
constexpr int RF_GRID = 512;                                 // persistent workgroups: 2 per CU x 256 CUs
constexpr int RF_NT1 = (RF_TW * RF_TW + 15) / 16;            // 21 pixel columns-of-16 cover the 18x18 region of the first convolution

// pack_h2's K-step list (pack.cpp; tail16_dev.h: t16_step_off) for an image of width IMW and group slots of SLOT bytes:
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
    return cb * SLOT + (T == 1 ? 0 : ((tap / 3) * IMW + tap % 3) * 16);
}

template <int T, int CB>
struct RfSteps {
    static constexpr bool paired = !(CB & 1) && (T & 1);
    static constexpr int NS = paired ? (CB / 2) * T : CB * ((T + 1) / 2);
    static constexpr int D = NS < 2 ? NS : 2;               // K-steps of weight lead
};

// One pass over CB source groups into KI accumulators of this wave (KI pixel columns-of-16 x ONE output group).  Item k's window origin
// for this lane: pb[k] bytes inside a group plane (+ the half-plane size for the upper 8 channels) if ISTR == 0, else pb[0] + k * ISTR (consecutive rows:
// the offsets become instruction immediates).  The LAST item is skipped unless `last` (wave-uniform: the 21 columns-of-16 of the first
// convolution do not divide evenly among the waves).  wbase: the pass's weight stream at this wave's output group (uniform), wv: lane * 16.
// Order per accumulator: x0*w1, x0*w0, x1*w0, K-step after K-step - the launch path's.
template <int T, int CB, int NT, int KI, int IMW, int SLOT, int PLN, int ISTR, int NPB>
__device__ __forceinline__ void rf_accumulate(const char *src, const char *wbase, unsigned wv, bool hi, const int (&pb)[NPB], bool last, f32x4 (&acc)[KI])
{
    typedef RfSteps<T, CB> ST;
    constexpr int NS = ST::NS, D = ST::D, WSTEP = 2 * NT * 64 * 16;      // bytes of weight stream per K-step
    auto wld = [&](int st, int sp) __attribute__((always_inline)) {
        return *reinterpret_cast<const RF_GLOBAL f16x8 *>((const RF_GLOBAL char *)wbase + (st * WSTEP + sp * (NT * 64 * 16)) + wv);
    };
    f16x8 wq[D][2];
#pragma unroll
    for (int st = 0; st < D; ++st) { wq[st][0] = wld(st, 0); wq[st][1] = wld(st, 1); }
    // Rolling pixel fragments, no second register set: a K-step's high-term fragments xa are re-requested for the NEXT K-step as soon as the
    // two MFMA groups that read them have issued, its low-term fragments xb after the third group - the LDS round trip of one runs behind
    // the MFMAs of the other (eight waves per workgroup, two workgroups per CU: nobody else hides it).
    f16x8 xa[KI], xb[KI];
    auto soff = [&](int st) __attribute__((always_inline)) {
        int off = hi ? rf_step_off<T, CB, IMW, SLOT>(st, 1) : rf_step_off<T, CB, IMW, SLOT>(st, 0);
        asm volatile("" : "+v"(off));      // opaque: keeps hipcc from hoisting the addresses of every K-step and item out of the pass
        return off;
    };
    auto rd = [&](f16x8 (&x)[KI], int off, int plane) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KI; ++k)
            if (k < KI - 1 || last) x[k] = *reinterpret_cast<const f16x8 *>(src + (ISTR ? pb[0] + k * ISTR : pb[ISTR ? 0 : k]) + off + plane * PLN);
    };
    int off = soff(0);
    rd(xa, off, 0);
    rd(xb, off, 1);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 w0 = wq[st % D][0], w1 = wq[st % D][1];
#pragma unroll
        for (int k = 0; k < KI; ++k) if (k < KI - 1 || last) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xa[k], acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < KI; ++k) if (k < KI - 1 || last) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xa[k], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < NS) { off = soff(st + 1); rd(xa, off, 0); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KI; ++k) if (k < KI - 1 || last) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xb[k], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < NS) rd(xb, off, 1);
        if (st + D < NS) { wq[st % D][0] = wld(st + D, 0); wq[st % D][1] = wld(st + D, 1); }
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
    const float *q, *bt, *dire;       // ATT: the attention trunk's input is built here, from the logits (conv_misc.hip: att_input_kernel)
    int layer;
    float att_scale;
};

// CB_IN input groups, NT output groups (= groups of the intermediate), POOLF: 2x2 max-pool and plain fp32 output (trunk_B3.2).
// ATT (trunk_Att2.0): no input tensor at all - cat[up(q), up(bt[layer]), up(dire[layer])] (Model_QBD.py:147), three channels of a 16-channel
// group, is computed from the logits into the halo image, with att_input_kernel's arithmetic (two-term split, clamp, range flag).
// The map is 32 x 32 (S), four tiles per block.
//
// PERSISTENT: a workgroup walks a contiguous run of tiles (the tiles of a block - they share halo rows and columns - stay on one workgroup,
// hence on one XCD's L2) and requests tile i+1's input into registers before it starts on tile i's convolutions.
// VALU diet (the second form of this kernel spent 65 % of its cycles issuing ~820 non-MFMA vector instructions per wave and tile - address
// arithmetic - with the matrix pipes 32 % busy): everything that does not depend on the tile is computed ONCE in front of the loop and is
// small (window origins, one store offset, one weight offset: <= 12 registers); tile coordinates are scalar (the wave index comes from
// v_readfirstlane, so output group and row band are SGPRs); global addresses are `uniform base + 32-bit lane offset` (the saddr form: no
// 64-bit vector arithmetic); a thread stages ONE pixel of the halo image (all groups, planes, halves of it: offsets become immediates).
template <int CB_IN, int NT, bool POOLF, bool ATT = false>
__global__ __launch_bounds__(512, 4) void rbfuse32_kernel(RbFuse32Dev a, int total)
{
    constexpr int S = 32, TILES = 4, NPX = RF_IW * RF_IW;
    __shared__ __attribute__((aligned(16))) char img[CB_IN * RF_ISLOT];
    __shared__ __attribute__((aligned(16))) char timg[NT * RF_TSLOT];
    const int tid = threadIdx.x, lane = tid & 63, xl = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave % NT, wsub = wave / NT;
    constexpr int WPG = 8 / NT;                                   // waves per output group
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = (int)blockIdx.x * per, t_end = min(t_begin + per, total);
    if (t_begin >= t_end) return;
    static_assert(!ATT || CB_IN == 1, "the attention input is one channel group");

    // ---- tile-independent lane state
    const bool hi = (g >> 1) != 0;
    const unsigned wv = (unsigned)lane * 16u;
    const int fpx = min(tid, NPX - 1), frow = fpx / RF_IW, fcol = fpx - frow * RF_IW;       // the halo-image pixel this thread stages
    constexpr int KI1 = (RF_NT1 + WPG - 1) / WPG, KI2 = 16 / WPG;
    static_assert((KI1 - 1) * WPG < RF_NT1, "only a wave's last item may fall outside the region");
    const bool last1 = wsub + (KI1 - 1) * WPG < RF_NT1;
    int pb1[KI1], pyx1[KI1];                                      // first convolution: window origin and (row | column << 8 | valid << 16) of each item's pixel
#pragma unroll
    for (int k = 0; k < KI1; ++k) {
        // the 21 columns-of-16 of the 18x18 region: items 0..17 = the first 16 pixels of row `item` (16 neighbouring cells: conflict-free
        // reads and writes), items 18..20 = what is left, columns 16 and 17 of the 18 rows, two pixels per row (36 pixels of 48 lanes).
        // (Round 5 cut the region's row-major pixel list into 16s: every item straddled a row end of the 20-wide image.)
        const int item = wsub + k * WPG, rest = (item - RF_TW) * 16 + xl;
        const bool whole = item < RF_TW, valid = whole || (item < RF_NT1 && rest < 2 * RF_TW);
        const int py = whole ? item : min(rest >> 1, RF_TW - 1), px = whole ? xl : RF_TW - 2 + (rest & 1);
        pb1[k] = (py * RF_IW + px) * 16 + (g & 1) * RF_IHH;
        pyx1[k] = py | (px << 8) | (valid ? 1 << 16 : 0);
    }
    const int row0 = wsub * KI2;                                  // second convolution: this wave's KI2 consecutive rows
    const int pb2[1] = {(row0 * RF_TW + xl) * 16 + (g & 1) * RF_THH}, pbs[1] = {((row0 + 2) * RF_IW + xl + 2) * 16 + (g & 1) * RF_IHH};
    const unsigned ov = POOLF ? (unsigned)(((row0 >> 1) * (S / 2) + (xl >> 1)) * 64 + g * 16)
                              : (unsigned)(((row0 + (g & 1)) * S + xl) * 32 + 16 * (g >> 1));            // output byte offset inside (block, group, tile)
    const char *w0b = reinterpret_cast<const char *>(a.w0) + ct * 1024, *w2b = reinterpret_cast<const char *>(a.w2) + ct * 1024,
               *wsb = reinterpret_cast<const char *>(a.wsc) + ct * 1024;

    float amax = 0.f;
    constexpr int NLD = ATT ? 1 : CB_IN * 4;
    unsigned fin = 0;      // is the staged pixel inside the map (the tile in flight)
    u32x4 r[NLD];          // the next tile's input on its way: [group][plane][half] of this thread's pixel (ATT: its three logit values)
    auto fetch = [&](int tl) __attribute__((always_inline)) {
        const int n = tl / TILES, t = tl % TILES, ty = t >> 1, tx = t & 1;
        const int gy = ty * 16 - 2 + frow, gx = tx * 16 - 2 + fcol;
        fin = (gy >= 0 && gy < S && gx >= 0 && gx < S) ? 1u : 0u;
        const int cy = min(max(gy, 0), S - 1), cx = min(max(gx, 0), S - 1);
        if (ATT) {
            const float *qb = a.q + (size_t)n * 64, *bb = a.bt + ((size_t)n * 3 + a.layer) * 256, *db = a.dire + ((size_t)n * 3 + a.layer) * 256;
            const unsigned o = (unsigned)((cy >> 1) * 16 + (cx >> 1)) * 4u;
            r[0] = (u32x4){*reinterpret_cast<const RF_GLOBAL unsigned *>((const RF_GLOBAL char *)qb + (unsigned)((cy >> 2) * 8 + (cx >> 2)) * 4u),
                           *reinterpret_cast<const RF_GLOBAL unsigned *>((const RF_GLOBAL char *)bb + o),
                           *reinterpret_cast<const RF_GLOBAL unsigned *>((const RF_GLOBAL char *)db + o), 0u};
        } else {
            const unsigned vo = (unsigned)(cy * S + cx) * 32u;
#pragma unroll
            for (int cb = 0; cb < CB_IN; ++cb)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const char *base = reinterpret_cast<const char *>(a.x + sp * a.x_stride + ((size_t)n * CB_IN + cb) * (S * S * 16));      // uniform
                    r[(cb * 2 + sp) * 2 + 0] = *reinterpret_cast<const RF_GLOBAL u32x4 *>((const RF_GLOBAL char *)base + vo);
                    r[(cb * 2 + sp) * 2 + 1] = *reinterpret_cast<const RF_GLOBAL u32x4 *>((const RF_GLOBAL char *)base + vo + 16);
                }
        }
    };
    // registers -> LDS halo images (zero outside the map)
    auto stash = [&]() __attribute__((always_inline)) {
        if (tid >= NPX) return;
        const u32x4 z = {0u, 0u, 0u, 0u};
        char *d = img + tid * 16;
        if (ATT) {
            u32x4 lo = z, hi4 = z;
            if (fin) {
                f32x4 v = {__uint_as_float(r[0].x), __uint_as_float(r[0].y), __uint_as_float(r[0].z), 0.f};
                v *= a.att_scale;
                amax = sat_amax4(amax, v);
                unsigned p0, q0, p1, q1;
                h2_split_pair(v.x, v.y, p0, q0);
                h2_split_pair(v.z, v.w, p1, q1);
                lo.x = p0; lo.y = p1; hi4.x = q0; hi4.y = q1;
            }
            *reinterpret_cast<u32x4 *>(d) = lo;
            *reinterpret_cast<u32x4 *>(d + RF_IHH) = z;
            *reinterpret_cast<u32x4 *>(d + RF_IPLN) = hi4;
            *reinterpret_cast<u32x4 *>(d + RF_IPLN + RF_IHH) = z;
        } else {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                *reinterpret_cast<u32x4 *>(d + (i >> 2) * RF_ISLOT + ((i >> 1) & 1) * RF_IPLN + (i & 1) * RF_IHH) = fin ? r[i] : z;
        }
    };
    fetch(t_begin);
  for (int tl = t_begin; tl < t_end; ++tl) {
    const int n = tl / TILES, t = tl % TILES, ty = t >> 1, tx = t & 1;
    // opaque per tile: `uniform base + lane offset` must stay an addressing mode; as a loop invariant hipcc materialises it - one 64-bit
    // vector address per weight fragment of every K-step - in front of the loop and spills it
    unsigned wvt = wv, ovt = ov;
    asm volatile("" : "+v"(wvt), "+v"(ovt));
    stash();
    __syncthreads();
    fetch(min(tl + 1, t_end - 1));        // (the last tile is requested twice: the loads stay unconditional)

    // ---- first convolution on the 18x18 region: wave (ct, wsub) takes the columns-of-16 wsub, wsub + WPG, ... - three at a time (six
    //      accumulators with their fragments do not fit next to the prefetched tile: the second trio re-reads the weight fragments from L1)
    {
        constexpr int KC = 3;
        static_assert(KI1 % KC == 0, "items in trios");
        const int oy = ty * 16 - 1, ox = tx * 16 - 1;
#pragma unroll
        for (int c = 0; c < KI1; c += KC) {
            f32x4 acc[KC];
            int pbc[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k) { acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f}; pbc[k] = pb1[c + k]; }
            const bool lastc = c + KC < KI1 ? true : last1;
            rf_accumulate<9, CB_IN, NT, KC, RF_IW, RF_ISLOT, RF_IPLN, 0, KC>(img, w0b, wvt, hi, pbc, lastc, acc);
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                if ((k == KC - 1 && !lastc) || !(pyx1[c + k] >> 16)) continue;
                const int py = pyx1[c + k] & 255, px = (pyx1[c + k] >> 8) & 255;
                f32x4 v = acc[k] * a.s0;
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                if ((unsigned)(oy + py) >= (unsigned)S || (unsigned)(ox + px) >= (unsigned)S) v = (f32x4){0.f, 0.f, 0.f, 0.f};      // the second convolution's zero padding
                amax = sat_amax4(amax, v);
                unsigned p0, q0, p1, q1;
                h2_split_pair(v.x, v.y, p0, q0);
                h2_split_pair(v.z, v.w, p1, q1);
                char *dp = timg + ct * RF_TSLOT + (g >> 1) * RF_THH + (py * RF_TW + px) * 16 + (g & 1) * 8;
                *reinterpret_cast<u32x2_t *>(dp) = (u32x2_t){p0, p1};
                *reinterpret_cast<u32x2_t *>(dp + RF_TPLN) = (u32x2_t){q0, q1};
            }
        }
    }
    __syncthreads();

    // ---- second convolution + 1x1 shortcut on the 16x16 tile: wave (ct, wsub) takes the KI2 consecutive rows row0 ..
    {
        f32x4 acc[KI2];
#pragma unroll
        for (int k = 0; k < KI2; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rf_accumulate<9, NT, NT, KI2, RF_TW, RF_TSLOT, RF_TPLN, RF_TW * 16, 1>(timg, w2b, wvt, hi, pb2, true, acc);
        rf_accumulate<1, CB_IN, NT, KI2, RF_IW, RF_ISLOT, RF_IPLN, RF_IW * 16, 1>(img, wsb, wvt, hi, pbs, true, acc);       // ResidualBlock, Model_QBD.py:33-38
#pragma unroll
        for (int k = 0; k < KI2; ++k) {
            f32x4 v = acc[k] * a.s2;
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            if (!POOLF) amax = sat_amax4(amax, v);      // a plain fp32 output is not clamped: not the range flag's business (conv_f16x3.hip)
            acc[k] = v;
        }
        if (POOLF) {
            char *ob = reinterpret_cast<char *>(a.out_f32 + (((size_t)n * NT + ct) * (S / 2) + ty * 8) * (S / 2) * 16 + tx * 8 * 16);     // uniform
#pragma unroll
            for (int k = 0; k < KI2; k += 2) {
                f32x4 v = acc[k], u = acc[k + 1];
                v.x = fmaxf(v.x, u.x); v.y = fmaxf(v.y, u.y); v.z = fmaxf(v.z, u.z); v.w = fmaxf(v.w, u.w);
                f32x4 o;
                o.x = __shfl_xor(v.x, 1); o.y = __shfl_xor(v.y, 1); o.z = __shfl_xor(v.z, 1); o.w = __shfl_xor(v.w, 1);
                v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
                if ((xl & 1) == 0) *reinterpret_cast<RF_GLOBAL f32x4 *>((RF_GLOBAL char *)ob + ovt + (k >> 1) * ((S / 2) * 64)) = v;
            }
        } else {
            // 16-byte stores as conv_f16x3.hip's epilogue: one v_permlane16_swap per register turns {rows m, m+1} x {couts 4g..} into the 8
            // consecutive channels 8(g>>1).. of row m + (g&1)
            char *ob = reinterpret_cast<char *>(a.out + (((size_t)n * NT + ct) * S + ty * 16) * S * 16 + tx * 16 * 16);                      // uniform
            char *ob1 = ob + a.out_stride * 2;
#pragma unroll
            for (int k = 0; k < KI2; k += 2) {
                u32x4 p, q;
                split2_rows(acc[k], acc[k + 1], p, q);
                rows16_swap(p);
                rows16_swap(q);
                __builtin_nontemporal_store(p, reinterpret_cast<RF_GLOBAL u32x4 *>((RF_GLOBAL char *)ob + ovt + k * (S * 32)));
                __builtin_nontemporal_store(q, reinterpret_cast<RF_GLOBAL u32x4 *>((RF_GLOBAL char *)ob1 + ovt + k * (S * 32)));
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
    if (h.H != 32 || h.W != 32 || h.N <= 0) return hipErrorInvalidValue;
    RbFuse32Dev a{h.x, h.x_stride, h.w0, h.w2, h.wsc, h.s0, h.s2, h.out, h.out_stride, h.out_f32, h.sat, h.q, h.bt, h.dire, h.att_layer, h.att_scale};
    const int total = h.N * 4;
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
