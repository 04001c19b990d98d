// conv_f16x3_wx.hip — the 3x3 64->64 trunk convolution (57.8 % of the MTT-net FLOPs) with a ONE-DIMENSIONAL Winograd transform
// F(2, 3) along x on the f16x3 datapath: two output pixels of a row cost 4 multiplications per vertical tap instead of 6, i.e.
// 1.5x fewer MFMAs than the direct form (conv_f16x3.hip) for the same result within fp32-equivalent accuracy
// (tools/precision_winograd.py: the logits move by less than the direct form's own distance to fp32, 1e-4 at worst).
//
//   input tile of an output pair (x = 2j, 2j+1):  d0..d3 = columns 2j-1 .. 2j+2
//   V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3            (formed in fp32, then split into two fp16 terms)
//   U_p[ky] = sum_kx G[p][kx] w[ky][kx],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   (host, fp64, scaled by 2^k, two fp16 terms)
//   M_p = sum over (ky, cin) of U_p[ky] * V_p[row + ky]                  (4 "positions" p, each a vertical 3-tap convolution)
//   y(2j) = M0 + M1 + M2      y(2j+1) = M1 - M2 - M3                     (fp32, in the epilogue)
//
// A workgroup = 16x16 output pixels x 64 couts, 4 waves; wave (rh, ch) = rows 8rh.., couts 32ch.. .  One MFMA tile
// (v_mfma_f32_16x16x32_f16) covers 8 pairs x 2 rows of one position; a wave holds 4 positions x 4 row pairs x 2 cout groups =
// 128 accumulator registers, so two workgroups per CU.  K-step = 16 channels x a pair of vertical taps; the odd tap ky = 2 of an
// even channel group is paired with ky = 2 of the following odd group (both V images are resident, one per LDS buffer), as the
// direct kernel pairs its odd tap: no zero-padded K-steps.
// LDS image of one channel group: V[plane 2][row 18][pos 4][pair 8][half 2] x 16 B, rows padded by 16 B so that the two rows of an
// MFMA tile fall into different banks (ds_read_b128 conflict-free).
#include <cstdio>

#include "pmp_kernels.h"
#include "split3.h"

namespace pmp {

namespace {

constexpr int WX_ROWB = 4 * 8 * 32 + 16;    // bytes per V row and plane
constexpr int WX_PLANEB = 18 * WX_ROWB;     // 18720
constexpr int WX_BUFB = 2 * WX_PLANEB;      // 37440 per channel group; two buffers per workgroup

struct WxItem {          // one staging item: 4 consecutive input pixels (8 channels each) of one row -> the 4 V values of one pair
    unsigned off[4];     // element offsets of the pixels inside a channel group (clamped into the image)
    unsigned valid;      // bit k: pixel k lies inside the image (zero padding otherwise)
    unsigned lds;        // byte offset inside a V buffer: row, pair, half (position and plane are added)
};

__device__ __forceinline__ void wx_plan(WxItem &it, int item, int H, int W, int ty, int tx)
{
    const int row = item >> 4, rem = item & 15, j = rem >> 1, chalf = rem & 1;
    const int gy = ty * 16 + row - 1;
    const int cy = min(max(gy, 0), H - 1);
    it.valid = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gx = tx * 16 - 1 + 2 * j + k;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) it.valid |= 1u << k;
        const int cx = min(max(gx, 0), W - 1);
        it.off[k] = (unsigned)(((size_t)cy * W + cx) * 16 + chalf * 8);
    }
    it.lds = (unsigned)(row * WX_ROWB + j * 32 + chalf * 16);
}

__device__ __forceinline__ void wx_load(const WxItem &it, const unsigned short *__restrict__ grp, size_t plane_stride, u32x4 (&r)[8])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[2 * k] = *reinterpret_cast<const u32x4 *>(grp + it.off[k]);
        r[2 * k + 1] = *reinterpret_cast<const u32x4 *>(grp + it.off[k] + plane_stride);
    }
}

// r -> V0..V3 of 8 channels -> two fp16 terms each -> LDS.  Returns the largest |V| (range guard: V is up to twice an activation).
__device__ __forceinline__ float wx_transform_store(const WxItem &it, const u32x4 (&r)[8], char *buf)
{
    float d[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f16x8 a = __builtin_bit_cast(f16x8, r[2 * k]), b = __builtin_bit_cast(f16x8, r[2 * k + 1]);
        const bool in = (it.valid >> k) & 1u;
#pragma unroll
        for (int i = 0; i < 8; ++i) d[k][i] = in ? (float)a[i] + (float)b[i] : 0.f;
    }
    float amax = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f16x8 h0, h1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = p == 0 ? d[0][i] - d[2][i] : p == 1 ? d[1][i] + d[2][i] : p == 2 ? d[2][i] - d[1][i] : d[1][i] - d[3][i];
            amax = fmaxf(amax, fabsf(v));
            const float c = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
            const _Float16 a = (_Float16)c;
            h0[i] = a;
            h1[i] = (_Float16)(c - (float)a);
        }
        *reinterpret_cast<u32x4 *>(buf + it.lds + p * 256) = __builtin_bit_cast(u32x4, h0);
        *reinterpret_cast<u32x4 *>(buf + it.lds + p * 256 + WX_PLANEB) = __builtin_bit_cast(u32x4, h1);
    }
    return amax;
}

}  // namespace

__global__ __launch_bounds__(256, 2) void conv_h2_wx_kernel(ConvX6Args a)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * WX_BUFB];
    const int tiles_x = a.W >> 4, tiles = tiles_x * (a.H >> 4);
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);     // XCD-aware: neighbouring tiles share an L2
    const int n = bid / tiles, t = bid - n * tiles, ty = t / tiles_x, tx = t - ty * tiles_x;
    const int H = a.H, W = a.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, xl = lane & 15, g4 = lane >> 4;
    const int rh = wave & 1, ch = wave >> 1;
    const size_t grp_sz = (size_t)H * W * 16;
    const unsigned short *grp0 = a.x + (size_t)n * 4 * grp_sz;

    f32x4 acc[4][4][2];     // [position][row pair][cout group]
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int rp = 0; rp < 4; ++rp)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[p][rp][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: 18 rows x 16 (pair, half) items = 288: rows 0..15 one item per thread, rows 16, 17 (32 items) by the lower half of one
    // wave - wave g & 3 for channel group g, so that the extra half round rotates over the waves
    WxItem itA, itB;
    wx_plan(itA, tid, H, W, ty, tx);
    wx_plan(itB, 256 + (lane & 31), H, W, ty, tx);
    u32x4 rA[8], rB[8];
    float amax = 0.f;
    auto stage_load = [&](int g) __attribute__((always_inline)) {
        const unsigned short *grp = grp0 + (size_t)g * grp_sz;
        wx_load(itA, grp, a.x_stride, rA);
        if (wave == (g & 3) && lane < 32) wx_load(itB, grp, a.x_stride, rB);
    };
    auto stage_store = [&](int g) __attribute__((always_inline)) {
        char *buf = lds + (g & 1) * WX_BUFB;
        amax = fmaxf(amax, wx_transform_store(itA, rA, buf));
        if (wave == (g & 3) && lane < 32) amax = fmaxf(amax, wx_transform_store(itB, rB, buf));
    };

    // MFMA operand addressing.  B operand (V): column xl = (pair j = xl & 7, row r = xl >> 3 of the row pair); K = lane group g4:
    // channels 8 (g4 & 1).. of vertical tap g4 >> 1.
    const int tapsel = g4 >> 1;
    const unsigned vbase = (unsigned)((rh * 8 + (xl >> 3)) * WX_ROWB + (xl & 7) * 32 + (g4 & 1) * 16);
    const f16x8 *wl = reinterpret_cast<const f16x8 *>(a.w) + lane + ch * 2 * 64;
    // one K-step: 4 positions x (4 row pairs x 2 cout groups x 3 products).  kind 0: taps (ky0, ky1) of the group in buffer b;
    // kind 1: the cross step, ky2 of the even group (buffer b ^ 1) and ky2 of the odd group (buffer b)
    auto kstep = [&](int step, int kind, int b) __attribute__((always_inline)) {
        const char *vb = kind == 0 ? lds + b * WX_BUFB + vbase + tapsel * WX_ROWB
                                   : lds + (tapsel ? b : (b ^ 1)) * WX_BUFB + vbase + 2 * WX_ROWB;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f16x8 *wf = wl + (size_t)((step * 4 + p) * 2) * 4 * 64;
            f16x8 w0[2], w1[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) { w0[nt] = wf[nt * 64]; w1[nt] = wf[(4 + nt) * 64]; }
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const f16x8 xa = *reinterpret_cast<const f16x8 *>(vb + p * 256 + rp * 2 * WX_ROWB);
                const f16x8 xb = *reinterpret_cast<const f16x8 *>(vb + p * 256 + rp * 2 * WX_ROWB + WX_PLANEB);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[p][rp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[nt], xa, acc[p][rp][nt], 0, 0, 0);
                    acc[p][rp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], xa, acc[p][rp][nt], 0, 0, 0);
                    acc[p][rp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], xb, acc[p][rp][nt], 0, 0, 0);
                }
            }
        }
    };

    stage_load(0);
    stage_store(0);
    stage_load(1);
    __syncthreads();
#pragma unroll
    for (int P = 0; P < 2; ++P) {
        // even group 2P (buffer 0)
        kstep(3 * P + 0, 0, 0);
        stage_store(2 * P + 1);                     // -> buffer 1 (the odd group of the previous pair is consumed)
        if (P == 0) stage_load(2);
        __syncthreads();
        // odd group 2P + 1 (buffer 1): the cross step reads both buffers
        kstep(3 * P + 1, 1, 1);
        kstep(3 * P + 2, 0, 1);
        if (P == 0) {
            __syncthreads();                        // every wave is done with buffer 0
            stage_store(2);
            stage_load(3);
            __syncthreads();
        }
    }

    // ---- epilogue: output transform, 1/S, residual, ReLU, split, store
    const float inv_scale = a.out_scale;
    float omax = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int ct = ch * 2 + nt;
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            const int row = ty * 16 + rh * 8 + 2 * rp + (xl >> 3), x0 = tx * 16 + 2 * (xl & 7);
            const size_t off = (((size_t)n * 4 + ct) * H + row) * W * 16 + (size_t)x0 * 16 + g4 * 4;
            f32x4 y0 = (acc[0][rp][nt] + acc[1][rp][nt] + acc[2][rp][nt]) * inv_scale;
            f32x4 y1 = (acc[1][rp][nt] - acc[2][rp][nt] - acc[3][rp][nt]) * inv_scale;
            if (a.res) {
                y0 += load_split2_4(a.res + off, a.res_stride);
                y1 += load_split2_4(a.res + off + 16, a.res_stride);
            }
            if (a.relu) {
                y0.x = fmaxf(y0.x, 0.f); y0.y = fmaxf(y0.y, 0.f); y0.z = fmaxf(y0.z, 0.f); y0.w = fmaxf(y0.w, 0.f);
                y1.x = fmaxf(y1.x, 0.f); y1.y = fmaxf(y1.y, 0.f); y1.z = fmaxf(y1.z, 0.f); y1.w = fmaxf(y1.w, 0.f);
            }
            omax = sat_amax4(sat_amax4(omax, y0), y1);
            store_split2_4(a.out + off, a.out_stride, y0);
            store_split2_4(a.out + off + 16, a.out_stride, y1);
        }
    }
    // range guard: a clamped V (|V| can reach twice an activation) or a clamped output
    sat_report(a.sat, amax);
    sat_report(a.sat, omax);
}

bool conv_h2_wx_applicable(const ConvX6Args &a)
{
    return a.w_wx && a.KH == 3 && a.KW == 3 && a.Cin == 64 && a.Cout == 64 && !a.x_sc && !a.gate && !a.out_f32 && !a.pool && a.out &&
           !(a.H & 15) && !(a.W & 15) && a.N > 0;
}

hipError_t launch_conv_h2_wx(hipStream_t s, const ConvX6Args &a_in)
{
    ConvX6Args a = a_in;
    a.w = a.w_wx;
    a.out_scale = a.wx_out_scale;
    const int grid = a.N * (a.H >> 4) * (a.W >> 4);
    hipLaunchKernelGGL(conv_h2_wx_kernel, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace pmp
