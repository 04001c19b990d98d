// conv_mfma.hip — im2col-free implicit-GEMM convolution on the CDNA4 fp32 matrix cores (gfx950).
//
// Replaces the cuDNN convolutions behind ResidualBlock (Model_QBD.py:23-44): ~97 % of the path's FLOPs.
//
// Why fp32 MFMA (v_mfma_f32_16x16x4_f32) and not bf16/fp16: tools/precision_study.py shows that single-pass
// bf16 moves the real Luma_Q_22 logits by 0.8 and even the 3-product bf16 split by 4.6e-3 (tolerance 1e-3);
// the f32 MFMA is an exact fmaf chain, so logits land within ~1e-5 of the reference's oneDNN result.
//
// Mapping (one workgroup = 256 threads = 4 waves = one 16x16 output tile of one block, all Cout):
//   GEMM  D[cout][pixel] += W[cout][k] * X[k][pixel],  k = (tap, channel)
//   MFMA 16x16x4:  A = weights  (lane: cout = l&15, k-group g = l>>4),
//                  B = pixels   (lane: pixel x = l&15, k-group g),
//                  D: lane holds couts 4g..4g+3 of pixel x  -> one float4 (16 B) per lane, and the 64 lanes of a
//                  wave cover 16 px * 16 ch = 1 KiB contiguous in the blocked layout: one coalesced
//                  global_store_dwordx4 per (row, cout-tile).
//   Each lane fetches 4 consecutive channels (one ds_read_b128 / one 16-B global load); register j of lane-group
//   g stands for channel 4g+j, so MFMA #j contracts channels {j, 4+j, 8+j, 12+j}: 4 MFMAs per 16 channels.
//   wave w owns tile rows 4w..4w+3 (4 pixel tiles) x Cout/16 cout tiles  -> 4*NT accumulators of 4 VGPRs.
//
// LDS: the (16+KH-1) x (16+KW-1) halo tile of ONE 16-channel group (64 B per pixel), XOR-swizzled so the
// ds_read_b128 of 16 consecutive pixels is conflict-free in every b128 lane group (slot ^= 2*((pix>>2)&1)).
// Weights never touch LDS: they are pre-packed in fragment order and streamed from L2 with 16-B lane loads.
#include "abl_kernels.h"

namespace pmp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// float4 index of channel slot s of tile pixel (row, col): the 16-B slot is XORed with 2*bit2(col), which makes the
// ds_read_b128 of 16 consecutive columns conflict-free for every row stride and tap offset (tools: brute-forced).
__device__ __forceinline__ int lds_slot(int row, int col, int s, int tw) { return (row * tw + col) * 4 + (s ^ (((col >> 2) & 1) << 1)); }

template <int KH, int KW, int NT>
__device__ __forceinline__ void accumulate(const float *__restrict__ x, const float *__restrict__ wpk, int C, int H,
                                           int W, int n, int ty, int tx, f32x4 *lds, f32x4 (&acc)[4][NT])
{
    constexpr int TH = 16 + KH - 1, TW = 16 + KW - 1, PY = KH / 2, PX = KW / 2, TAPS = KH * KW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, xl = lane & 15, g = lane >> 4;
    const int CB = C >> 4;
    for (int cb = 0; cb < CB; ++cb) {
        __syncthreads();  // everyone is done reading the previous channel group
        const float *plane = x + ((size_t)n * CB + cb) * H * W * 16;
        for (int i = tid; i < TH * TW * 4; i += 256) {
            const int row = i / (TW * 4), r = i - row * (TW * 4), px = r >> 2, s = r & 3;
            const int gy = ty * 16 + row - PY, gx = tx * 16 + px - PX;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const f32x4 *>(plane + ((size_t)gy * W + gx) * 16 + s * 4);
            lds[lds_slot(row, px, s, TW)] = v;
        }
        __syncthreads();
        const f32x4 *wl = reinterpret_cast<const f32x4 *>(wpk) + (size_t)cb * TAPS * NT * 64 + lane;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = tap / KW, dx = tap % KW;
            f32x4 wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = wl[(tap * NT + nt) * 64];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 a = lds[lds_slot(wave * 4 + m + dy, xl + dx, g, TW)];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].x, a.x, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].y, a.y, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].z, a.z, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].w, a.w, acc[m][nt], 0, 0, 0);
                }
            }
        }
    }
}

// ---- software-pipelined variant -------------------------------------------------------------------------------
// Two LDS buffers; the halo tile of channel group cb+1 is fetched into registers BEFORE the MFMAs of group cb are
// issued and written to the other buffer after them (one barrier per group), and the weight fragments of tap t+1
// are requested while tap t computes.  Rationale (profiles/r01a): with 3 co-resident workgroups running the same
// instruction stream in near lock-step, the un-pipelined kernel exposed the staging round trip and one L2 latency
// per tap on all waves of a SIMD at once - the matrix pipe sat idle 31 % of the time.
template <int KH, int KW>
struct Geo {
    static constexpr int TH = 16 + KH - 1, TW = 16 + KW - 1, PIECES = TH * TW * 4, NLD = (PIECES + 255) / 256;
};

// Staging loads are UNCONDITIONAL (coordinates clamped into the image, zero selected at store time): with
// predicated loads hipcc cannot count the outstanding requests and falls back to s_waitcnt vmcnt(0) in front of the
// first MFMA of every channel group, which exposes the full HBM round trip of the prefetch it was meant to hide.
template <int KH, int KW>
__device__ __forceinline__ void stage_load(const float *__restrict__ plane, int H, int W, int ty, int tx,
                                           f32x4 (&r)[Geo<KH, KW>::NLD])
{
    constexpr int TW = Geo<KH, KW>::TW, PY = KH / 2, PX = KW / 2;
#pragma unroll
    for (int k = 0; k < Geo<KH, KW>::NLD; ++k) {
        const int i = min((int)threadIdx.x + k * 256, Geo<KH, KW>::PIECES - 1);
        const int row = i / (TW * 4), rr = i - row * (TW * 4), px = rr >> 2, s = rr & 3;
        const int gy = min(max(ty * 16 + row - PY, 0), H - 1), gx = min(max(tx * 16 + px - PX, 0), W - 1);
        r[k] = *reinterpret_cast<const f32x4 *>(plane + ((size_t)gy * W + gx) * 16 + s * 4);
    }
}

template <int KH, int KW>
__device__ __forceinline__ void stage_store(f32x4 *lds, const f32x4 (&r)[Geo<KH, KW>::NLD], int H, int W, int ty, int tx)
{
    constexpr int TW = Geo<KH, KW>::TW, PY = KH / 2, PX = KW / 2;
#pragma unroll
    for (int k = 0; k < Geo<KH, KW>::NLD; ++k) {
        const int i = threadIdx.x + k * 256;
        const int row = i / (TW * 4), rr = i - row * (TW * 4), px = rr >> 2, s = rr & 3;
        const int gy = ty * 16 + row - PY, gx = tx * 16 + px - PX;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (i < Geo<KH, KW>::PIECES) lds[lds_slot(row, px, s, TW)] = in ? r[k] : z;
    }
}

template <int KH, int KW, int NT, int BUFSZ>
__device__ __forceinline__ void accumulate_pipe(const float *__restrict__ x, const float *__restrict__ wpk, int C, int H,
                                                int W, int n, int ty, int tx, f32x4 *lds, f32x4 (&acc)[4][NT])
{
    constexpr int TW = Geo<KH, KW>::TW, TAPS = KH * KW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, xl = lane & 15, g = lane >> 4;
    const int CB = C >> 4;
    const size_t plane_sz = (size_t)H * W * 16;
    const float *plane = x + (size_t)n * CB * plane_sz;
    f32x4 r[Geo<KH, KW>::NLD];
    __syncthreads();  // LDS may still be read by a previous source
    stage_load<KH, KW>(plane, H, W, ty, tx, r);
    stage_store<KH, KW>(lds, r, H, W, ty, tx);
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(wpk) + lane;
    const int last = CB * TAPS - 1;
    f32x4 wn[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wn[nt] = wl[nt * 64];
    __syncthreads();
    int stream = 0;  // index of the (channel group, tap) whose weights sit in wn
    for (int cb = 0; cb < CB; ++cb) {
        if (cb + 1 < CB) stage_load<KH, KW>(plane + (size_t)(cb + 1) * plane_sz, H, W, ty, tx, r);
        const f32x4 *buf = lds + (cb & 1) * BUFSZ;
        int dy = 0, dx = 0;
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap) {
            f32x4 wf[NT];
            stream = min(stream + 1, last);
            const f32x4 *wp = wl + (size_t)stream * (NT * 64);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { wf[nt] = wn[nt]; wn[nt] = wp[nt * 64]; }
            f32x4 a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = buf[lds_slot(wave * 4 + m + dy, xl + dx, g, TW)];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].x, a[m].x, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].y, a[m].y, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].z, a[m].z, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].w, a[m].w, acc[m][nt], 0, 0, 0);
                }
            if (++dx == KW) { dx = 0; ++dy; }
        }
        if (cb + 1 < CB) stage_store<KH, KW>(lds + ((cb + 1) & 1) * BUFSZ, r, H, W, ty, tx);
        __syncthreads();
    }
}

// ---- fully software-pipelined variant (variant 2) --------------------------------------------------------------
// Unrolled taps with a scheduling fence between them, so every s_waitcnt is an exact count:
//   * weight fragments are requested TWO taps ahead (3-deep register ring),
//   * the pixel fragments of tap t+1 are read from LDS while tap t computes (2-deep),
//   * the next channel group's halo tile is requested at tap 0 and written to the other LDS buffer after the last tap.
// Needs ~200 VGPRs -> 2 waves per SIMD; the explicit prefetch distance replaces the third wave.
template <int KH, int KW, int NT, int BUFSZ>
__device__ __forceinline__ void accumulate_pipe2(const float *__restrict__ x, const float *__restrict__ wpk, int C, int H,
                                                 int W, int n, int ty, int tx, f32x4 *lds, f32x4 (&acc)[4][NT])
{
    constexpr int TW = Geo<KH, KW>::TW, TAPS = KH * KW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, xl = lane & 15, g = lane >> 4;
    const int CB = C >> 4;
    const size_t plane_sz = (size_t)H * W * 16;
    const float *plane = x + (size_t)n * CB * plane_sz;
    f32x4 r[Geo<KH, KW>::NLD];
    __syncthreads();  // LDS may still be read by a previous source
    stage_load<KH, KW>(plane, H, W, ty, tx, r);
    const f32x4 *wl = reinterpret_cast<const f32x4 *>(wpk) + lane;
    const int last = CB * TAPS - 1;
    f32x4 w0[NT], w1[NT], w2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { w0[nt] = wl[nt * 64]; w1[nt] = wl[((size_t)min(1, last) * NT + nt) * 64]; }
    stage_store<KH, KW>(lds, r, H, W, ty, tx);
    __syncthreads();
    // per-lane byte offset of (row 4*wave, column xl+dx, slot g) for every dx; (m+dy)*TW*64 is an immediate offset
    int off[KW];
#pragma unroll
    for (int dx = 0; dx < KW; ++dx) off[dx] = lds_slot(wave * 4, xl + dx, g, TW) * 16;
    for (int cb = 0; cb < CB; ++cb) {
        const char *buf = reinterpret_cast<const char *>(lds + (cb & 1) * BUFSZ);
        const bool more = cb + 1 < CB;
        f32x4 a0[4], a1[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a0[m] = *reinterpret_cast<const f32x4 *>(buf + off[0] + m * TW * 64);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int s2 = min(cb * TAPS + tap + 2, last);
            const f32x4 *wp = wl + (size_t)s2 * (NT * 64);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w2[nt] = wp[nt * 64];
            if (tap == 0 && more) stage_load<KH, KW>(plane + (size_t)(cb + 1) * plane_sz, H, W, ty, tx, r);
            if (tap + 1 < TAPS) {
                const int dyn = (tap + 1) / KW, dxn = (tap + 1) % KW;
#pragma unroll
                for (int m = 0; m < 4; ++m) a1[m] = *reinterpret_cast<const f32x4 *>(buf + off[dxn] + (m + dyn) * TW * 64);
            }
            // (A second fence here, forcing the LDS reads ahead of the MFMA block, measured 9 % SLOWER: the register
            //  rotation below then needs the just-requested fragments at the end of the same tap.)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[nt].x, a0[m].x, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[nt].y, a0[m].y, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[nt].z, a0[m].z, acc[m][nt], 0, 0, 0);
                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[nt].w, a0[m].w, acc[m][nt], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { w0[nt] = w1[nt]; w1[nt] = w2[nt]; }
#pragma unroll
            for (int m = 0; m < 4; ++m) a0[m] = a1[m];
        }
        if (more) stage_store<KH, KW>(lds + ((cb + 1) & 1) * BUFSZ, r, H, W, ty, tx);
        __syncthreads();
    }
}

template <int KH, int KW, int NT, int PIPE, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_mfma_kernel(ConvMfmaArgs a)
{
    constexpr int BUFSZ = Geo<KH, KW>::PIECES;
    __shared__ f32x4 lds[BUFSZ * (PIPE ? 2 : 1)];  // PIPE: 0 plain, 1 pipelined, 2 fully pipelined
    const int tiles_x = a.W >> 4, tiles = tiles_x * (a.H >> 4);
    const int n = blockIdx.x / tiles, t = blockIdx.x - n * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, xl = lane & 15, g = lane >> 4;

    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (PIPE == 2) {
        accumulate_pipe2<KH, KW, NT, BUFSZ>(a.x, a.w, a.Cin, a.H, a.W, n, ty, tx, lds, acc);
        if (a.x_sc) accumulate_pipe2<1, 1, NT, BUFSZ>(a.x_sc, a.w_sc, a.Csc, a.H, a.W, n, ty, tx, lds, acc);
    } else if constexpr (PIPE == 1) {
        accumulate_pipe<KH, KW, NT, BUFSZ>(a.x, a.w, a.Cin, a.H, a.W, n, ty, tx, lds, acc);
        if (a.x_sc) accumulate_pipe<1, 1, NT, BUFSZ>(a.x_sc, a.w_sc, a.Csc, a.H, a.W, n, ty, tx, lds, acc);
    } else {
        accumulate<KH, KW, NT>(a.x, a.w, a.Cin, a.H, a.W, n, ty, tx, lds, acc);
        if (a.x_sc) accumulate<1, 1, NT>(a.x_sc, a.w_sc, a.Csc, a.H, a.W, n, ty, tx, lds, acc);
    }

    // ---- epilogue: lane owns pixel (row 4*wave+m, x = xl), channels 16*nt + 4g .. +3
    const int H = a.H, W = a.W;
    const size_t plane = (size_t)H * W * 16;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int y = ty * 16 + wave * 4 + m, x = tx * 16 + xl;
            const size_t off = ((size_t)n * NT + nt) * plane + ((size_t)y * W + x) * 16 + g * 4;
            f32x4 v = acc[m][nt];
            if (a.res) v += *reinterpret_cast<const f32x4 *>(a.res + off);
            if (a.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (a.gate) v *= *reinterpret_cast<const f32x4 *>(a.gate + off);
            acc[m][nt] = v;
        }
        if (!a.pool) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int y = ty * 16 + wave * 4 + m, x = tx * 16 + xl;
                const size_t off = ((size_t)n * NT + nt) * plane + ((size_t)y * W + x) * 16 + g * 4;
                *reinterpret_cast<f32x4 *>(a.out + off) = acc[m][nt];
            }
        } else {
            // 2x2 max-pool: rows (m, m+1) live in this lane, columns (x, x^1) in the neighbouring lane.
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
                    *reinterpret_cast<f32x4 *>(a.out + off) = v;
                }
            }
        }
    }
}

#ifdef PMP_ABLATION
int g_conv_variant = 2;  // measurement library only: 0: un-pipelined; 1: pipelined, 3 waves/SIMD; 2: the shipped form (A/B measurements)
#endif

template <int KH, int KW, int PIPE, int OCC>
static hipError_t launch_k(hipStream_t s, const ConvMfmaArgs &a)
{
    const int grid = a.N * (a.H >> 4) * (a.W >> 4);
    switch (a.Cout >> 4) {
    case 1: hipLaunchKernelGGL((conv_mfma_kernel<KH, KW, 1, PIPE, OCC>), dim3(grid), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((conv_mfma_kernel<KH, KW, 2, PIPE, OCC>), dim3(grid), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((conv_mfma_kernel<KH, KW, 4, PIPE, OCC>), dim3(grid), dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_mfma(hipStream_t s, const ConvMfmaArgs &a)
{
    if ((a.H & 15) || (a.W & 15) || (a.Cin & 15) || (a.Cout & 15) || (a.x_sc && (a.Csc & 15)) || a.N <= 0)
        return hipErrorInvalidValue;
    if (a.pool && a.gate) return hipErrorInvalidValue;
#ifdef PMP_ABLATION
    const int v = g_conv_variant > 2 ? 2 : g_conv_variant;
    if (v < 2) {
        if (a.KH == 3 && a.KW == 3) return v == 0 ? launch_k<3, 3, 0, 1>(s, a) : launch_k<3, 3, 1, 3>(s, a);
        if (a.KH == 5 && a.KW == 5) return v == 0 ? launch_k<5, 5, 0, 1>(s, a) : launch_k<5, 5, 1, 3>(s, a);
    }
#endif
    // shipped form: fully software-pipelined, two waves per SIMD
    if (a.KH == 3 && a.KW == 3) return launch_k<3, 3, 2, 2>(s, a);
    if (a.KH == 5 && a.KW == 5) return launch_k<5, 5, 2, 2>(s, a);
    if (a.KH == 1 && a.KW == 1) return launch_k<1, 1, 0, 1>(s, a);
    return hipErrorInvalidValue;
}

}  // namespace pmp
