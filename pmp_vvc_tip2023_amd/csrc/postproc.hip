// postproc.hip — Map2Partition on the GPU: one wavefront (64 lanes) per 64x64 block.
//
// Replaces, bit-exactly:
//   eli_structual_error / check_square_unity   Metrics.py:612-637
//   Map_to_Partition                           Map2Partition.py:98-373
//   output_block_yuv (block cutter)            Inference_QBD.py:104-149
//
// Work decomposition (wave64): the block's 16x16 grid of 4x4-pixel cells is spread over the lanes, 4 cells per
// lane in row-major order (lane l: row l>>2, columns 4*(l&3)..+3).  The reference's candidate-tree search
// (Map2Partition.py:203-266) is a depth-3 DFS whose control flow is identical for all lanes, so it runs as
// wave-uniform scalar control with
//   * region counts  (can_split_mode_list, :140-201)  = ballot + popcount over the lanes' cell predicates,
//   * small "arrays indexed by a uniform index" (CU lists, candidate lists) kept lane-distributed in VGPRs and
//     read with v_readlane,
//   * the three tree levels unrolled at compile time (template recursion), so every level's maps are registers.
// The float32 L1 error (:307-312) must reproduce numpy's pairwise summation order bit for bit: cell values are
// scattered to LDS in region-flattened order and 8 (or 2x8) lanes run the interleaved accumulators exactly as
// numpy's FLOAT_pairwise_sum does; the combine tree is done with xor-shuffles (fp add is commutative, so the
// tree ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) is reproduced exactly).  This file is compiled with
// -ffp-contract=off so `s + 0.8f*d` is not fused.
#include "pmp_kernels.h"

namespace pmp {

namespace {

__device__ __forceinline__ int rlane(int v, int lane)
{
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane));
}
__device__ __forceinline__ float rlanef(float v, int lane) { return __int_as_float(rlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int cnt(bool p) { return __popcll(__ballot(p)); }

__device__ __forceinline__ int pack_cu(int x, int y, int h, int w) { return x | (y << 4) | ((h - 1) << 8) | ((w - 1) << 12); }

struct Search {
    // per-lane constants of this block
    int row, col0;
    int mb[3][4];     // np.round(bt) (Map2Partition.py:104) as an integer saturated to +-100; NaN -> +100.  The reference keeps the float and
                      // has no clamp, but the rounded map is only ever compared (`== 0`, `< 0` of msbt - candidate, candidate depths 0..6,
                      // :142,189-190): a value beyond 100 - or NaN, on which both comparisons are False - behaves like 100, one below
                      // -100 like -100.  Pinned by tests/golden/g3b_m2p_range.npz (|bt| up to 3e38, +-inf, NaN; made by the reference)
    int md[3][4];     // th_round(dire, 0.5) (Map2Partition.py:30-35,105)
    float ob[3][4];   // raw MTT depth logits
    float od[3][4];   // raw direction logits
    int cf;           // chroma_factor
    // tree levels 0..3 (Map_Node, Map2Partition.py:89-96)
    int bt[4][4], dr[4][4];
    int cu[4];        // lane c holds CU c of the level: x | y<<4 | (h-1)<<8 | (w-1)<<12
    int ncu[4];       // uniform
    float sb[4], sd[4];
    // QT-leaf region and best leaf so far
    int rx, ry, rh, rw;
    float best_err;
    int have_best;
    int best_dr[3][4];
    int best_cu, best_ncu;
    float *valb, *vald;  // LDS scratch, 256 floats each
};

// np.sum(np.abs(map - ori)) over the region for the depth map (lanes 0..15) and the direction map (16..31).
__device__ __forceinline__ void region_sums(Search &s, const int (&btm)[4], const int (&drm)[4], const float (&ob)[4],
                                            const float (&od)[4], float &out_b, float &out_d)
{
    const int lane = threadIdx.x & 63;
    const int n = s.rh * s.rw;
    // wave-local ordering only (the scratch is private to the wave and the four waves of a block search different quadrants,
    // so a workgroup barrier here would not even be reached the same number of times)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = s.row - s.rx, c = s.col0 + k - s.ry;
        if (r >= 0 && r < s.rh && c >= 0 && c < s.rw) {
            const int f = r * s.rw + c;
            s.valb[f] = fabsf((float)btm[k] - ob[k]);
            s.vald[f] = fabsf((float)drm[k] - od[k]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float *v = (lane & 16) ? s.vald : s.valb;
    const int j = lane & 15, half = j >> 3, k = j & 7;
    float r = 0.f;
    if (n >= 8) {
        const int per = n > 128 ? 128 : n;          // numpy PW_BLOCKSIZE = 128; n = 256 splits into 128 + 128
        const float *p = v + half * 128 + k;
        if (half == 0 || n > 128) {
            r = p[0];
            for (int i = 8; i < per; i += 8) r = r + p[i];
        }
        r = r + __shfl_xor(r, 1);
        r = r + __shfl_xor(r, 2);
        r = r + __shfl_xor(r, 4);
        if (n > 128) r = r + __shfl_xor(r, 8);
    } else {                                         // n < 8: sequential from 0.
        for (int i = 0; i < n; ++i) r = r + v[i];
    }
    out_b = rlanef(r, 0);
    out_d = rlanef(r, 16);
}

// can_split_mode_list (Map2Partition.py:140-201): returns modes packed 3 bits each (first entry 0), count in `n`.
template <int L>
__device__ __forceinline__ int can_split(const Search &s, int cu, int &n)
{
    const int x = cu & 15, y = (cu >> 4) & 15, h = ((cu >> 8) & 15) + 1, w = ((cu >> 12) & 15) + 1;
    bool in[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) in[k] = s.row >= x && s.row < x + h && s.col0 + k >= y && s.col0 + k < y + w;
    int zero = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) zero += cnt(in[k] && s.mb[2][k] == s.bt[L][k]);
    n = 1;
    if ((double)zero >= 0.7 * h * w) return 0;  // lamb1
    int hor = 0, ver = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        hor += cnt(in[k] && s.md[L][k] == 1);
        ver += cnt(in[k] && s.md[L][k] == -1);
    }
    int direction = 0;
    if ((double)(ver + hor) >= 0.7 * h * w) {  // lamb2, lamb3
        if ((double)hor >= 1.5 * ver) direction = 1;
        else if ((double)ver >= 1.5 * hor) direction = 2;
    }
    const int cf = s.cf;
    int list = 0;
    for (int mode = 1; mode <= 4; ++mode) {
        const bool horiz = (mode & 1) != 0;      // 1 BT-H, 3 TT-H
        const int ext = horiz ? h : w;
        const int div = (mode <= 2 ? 2 : 4) * cf;
        if (ext / div == 0 || ext % div != 0) continue;
        if (horiz && direction == 2) continue;
        if (!horiz && direction == 1) continue;
        const int parts = mode <= 2 ? 2 : 3;
        int ok = 0;
        for (int p = 0; p < parts; ++p) {
            // sub-part extent along the split axis
            int o0, o1, inc;
            if (mode <= 2) { o0 = p * (ext / 2); o1 = o0 + ext / 2; inc = 1; }
            else if (p == 0) { o0 = 0; o1 = ext / 4; inc = 2; }
            else if (p == 1) { o0 = ext / 4; o1 = o0 + ext / 2; inc = 1; }
            else { o0 = (ext * 3) / 4; o1 = ext; inc = 2; }
            int minus = 0, zer = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = horiz ? (s.row - x) : (s.col0 + k - y);
                const bool ins = in[k] && t >= o0 && t < o1;
                const int tgt = s.bt[L][k] + inc;
                minus += cnt(ins && s.mb[L][k] < tgt);
                zer += cnt(ins && s.mb[L][k] == tgt);
            }
            const int np_ = (o1 - o0) * (horiz ? w : h);
            if ((double)minus < np_ * 0.3 && (double)zer > np_ * 0.7) ++ok;  // lamb4, lamb5
        }
        if (ok == parts) { list |= mode << (3 * n); ++n; }
    }
    return list;
}

// get_candidate_map_tree (Map2Partition.py:203-266) for a node at level L; leaves (level 3) are scored in DFS
// order with a strict '<' so the FIRST minimum wins, as list.index(min(list)) does (:315).
template <int L>
__device__ __forceinline__ void expand(Search &s)
{
    const int lane = threadIdx.x & 63;
    const int ncu = s.ncu[L];
    int my_list = 0, my_n = 1;
    for (int c = 0; c < ncu; ++c) {
        int n;
        const int list = can_split<L>(s, rlane(s.cu[L], c), n);
        if (lane == c) { my_list = list; my_n = n; }
    }
    // mixed-radix combination index, first CU slowest (Search, Map2Partition.py:53-87)
    int my_p = 1, total = 1;
    for (int c = ncu - 1; c >= 0; --c) {
        if (lane == c) my_p = total;
        total *= rlane(my_n, c);
    }
    total = uni(total);
    for (int t = 0; t < total; ++t) {
        const int my_mode = (my_list >> (3 * ((t / my_p) % my_n))) & 7;
        int nb[4], nd[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { nb[k] = s.bt[L][k]; nd[k] = 0; }  // child dire map starts from zeros (:240)
        int child_cu = 0, nchild = 0;
        for (int c = 0; c < ncu; ++c) {
            const int mode = rlane(my_mode, c), cu = rlane(s.cu[L], c);
            const int x = cu & 15, y = (cu >> 4) & 15, h = ((cu >> 8) & 15) + 1, w = ((cu >> 12) & 15) + 1;
            if (mode == 0) {
                if (lane == nchild) child_cu = cu;
                nchild += 1;
                continue;
            }
            const bool horiz = (mode & 1) != 0;
            const int ext = horiz ? h : w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = s.row - x, cc = s.col0 + k - y;
                if (r >= 0 && r < h && cc >= 0 && cc < w) {
                    nd[k] = horiz ? 1 : -1;
                    const int tt = horiz ? r : cc;
                    nb[k] += (mode >= 3 && (tt < ext / 4 || tt >= (ext * 3) / 4)) ? 2 : 1;
                }
            }
            // split_cur_map (:124-138): sub-CUs appended in order
            if (mode <= 2) {
                const int e = ext / 2;
                const int c0 = horiz ? pack_cu(x, y, e, w) : pack_cu(x, y, h, e);
                const int c1 = horiz ? pack_cu(x + e, y, e, w) : pack_cu(x, y + e, h, e);
                if (lane == nchild) child_cu = c0;
                if (lane == nchild + 1) child_cu = c1;
                nchild += 2;
            } else {
                const int q = ext / 4, e = ext / 2, o2 = (ext * 3) / 4;
                const int c0 = horiz ? pack_cu(x, y, q, w) : pack_cu(x, y, h, q);
                const int c1 = horiz ? pack_cu(x + q, y, e, w) : pack_cu(x, y + q, h, e);
                const int c2 = horiz ? pack_cu(x + o2, y, q, w) : pack_cu(x, y + o2, h, q);
                if (lane == nchild) child_cu = c0;
                if (lane == nchild + 1) child_cu = c1;
                if (lane == nchild + 2) child_cu = c2;
                nchild += 3;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { s.bt[L + 1][k] = nb[k]; s.dr[L + 1][k] = nd[k]; }
        s.cu[L + 1] = child_cu;
        s.ncu[L + 1] = uni(nchild);
        region_sums(s, nb, nd, s.ob[L], s.od[L], s.sb[L + 1], s.sd[L + 1]);
        if constexpr (L == 2) {
            // error of the leaf and its two ancestors, float32 left to right (Map2Partition.py:307-312)
            const float eb = (s.sb[1] + s.sb[2]) + s.sb[3];
            const float ed = (s.sd[1] + s.sd[2]) + s.sd[3];
            const float err = eb + 0.8f * ed;
            if (!s.have_best || err < s.best_err) {
                s.have_best = 1;
                s.best_err = err;
#pragma unroll
                for (int k = 0; k < 4; ++k) { s.best_dr[0][k] = s.dr[1][k]; s.best_dr[1][k] = s.dr[2][k]; s.best_dr[2][k] = s.dr[3][k]; }
                s.best_cu = s.cu[3];
                s.best_ncu = s.ncu[3];
            }
        } else {
            expand<L + 1>(s);
        }
    }
}

}  // namespace

// One wave per block.  LDS per workgroup: 2 x 256 floats (sum scratch) + 2 x 256 bytes (edge planes).
// Four waves per block: the QT leaves under the four 32x32 quadrants are independent searches writing disjoint cells, so wave
// w takes the nodes of quadrant w (a block that is one 64x64 leaf is searched by wave 0 alone).  Every wave keeps the whole
// block's maps - the search code is the single-wave one - and has its own pairwise-sum scratch.
__global__ __launch_bounds__(256, 4) void postprocess_kernel(const float *__restrict__ qt, const float *__restrict__ bt,
                                                          const float *__restrict__ dire, int64_t N, int cf,
                                                          uint8_t *__restrict__ hor_o, uint8_t *__restrict__ ver_o,
                                                          uint8_t *__restrict__ qt_o, int8_t *__restrict__ dire_o,
                                                          int s_edge, int s_qt, int s_dire)   // bytes between consecutive blocks
{
    __shared__ float valb_all[4][256], vald_all[4][256];
    __shared__ uint32_t horw[64], verw[64];
    uint8_t *hor = reinterpret_cast<uint8_t *>(horw), *ver = reinterpret_cast<uint8_t *>(verw);
    const int64_t b = blockIdx.x;
    if (b >= N) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *valb = valb_all[wv], *vald = vald_all[wv];

    // ---- eli_structual_error (Metrics.py:630-637): lanes 0..15 own the 4x4 pooled map
    float pv = 0.f;
    {
        const int i = lane & 15, pr = i >> 2, pc = i & 3;
        const float *q = qt + b * 64 + (2 * pr) * 8 + 2 * pc;
        pv = q[0];                                        // F.max_pool2d propagates NaN (v_max_f32 would drop it)
        if (q[1] > pv || q[1] != q[1]) pv = q[1];
        if (q[8] > pv || q[8] != q[8]) pv = q[8];
        if (q[9] > pv || q[9] != q[9]) pv = q[9];
        pv = rintf(pv);                                   // torch.round: half to even
        pv = pv < 0.f ? 0.f : (pv > 3.f ? 3.f : pv);      // clamp(0, 3); -0.0 compares equal to 0; NaN stays NaN (torch.clamp)
    }
    // a NaN depth travels as -1: every comparison the reference makes on it (== 0, == 1, == depth, > depth) is False, and so is
    // every comparison below; its quadrant's sum is NaN (Metrics.py:618-619: no rule applies); it is emitted as 0 (.astype(np.uint8))
    int m = pv != pv ? -1 : (int)pv;
    {
        const unsigned long long z = __ballot(m == 0) & 0xFFFFull;
        const int num0 = __popcll(z);
        if (num0 <= 12) {                                 // check_square_unity (Metrics.py:612-628)
            if (m == 0) m = 1;
            int sum = m + __shfl_xor(m, 1);
            sum += __shfl_xor(sum, 4);
            int one = (m == 1) ? 1 : 0;
            int n1 = one + __shfl_xor(one, 1);
            n1 += __shfl_xor(n1, 4);
            int nnan = (m < 0) ? 1 : 0;
            nnan += __shfl_xor(nnan, 1);
            nnan += __shfl_xor(nnan, 4);
            if (nnan == 0 && sum >= 5 && sum <= 10) {
                if (n1 < 3) { if (m == 1) m = 2; }
                else m = 1;
            }
        } else if (num0 < 16) {
            m = 0;
        }
    }
    // nearest x2 (Metrics.py:635): lane l holds the 8x8 value at (l>>3, l&7)
    const int qt8 = rlane(m, 0) * 0 + __shfl(m, ((lane >> 4) << 2) + ((lane & 7) >> 1));
    if (wv == 0) qt_o[b * s_qt + lane] = (uint8_t)(qt8 < 0 ? 0 : qt8);

    // ---- Map_to_Partition.__init__ (Map2Partition.py:100-122)
    Search s;
    s.row = lane >> 2;
    s.col0 = (lane & 3) << 2;
    s.cf = cf;
    s.valb = valb;
    s.vald = vald;
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3) {
        const float4 vb = *reinterpret_cast<const float4 *>(bt + (b * 3 + k3) * 256 + lane * 4);
        const float4 vd = *reinterpret_cast<const float4 *>(dire + (b * 3 + k3) * 256 + lane * 4);
        const float fb[4] = {vb.x, vb.y, vb.z, vb.w}, fd[4] = {vd.x, vd.y, vd.z, vd.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s.ob[k3][k] = fb[k];
            s.od[k3][k] = fd[k];
            float r = rintf(fb[k]);                       // np.round: half to even
            r = !(r <= 100.f) ? 100.f : (r < -100.f ? -100.f : r);   // see Search::mb; NaN -> 100
            s.mb[k3][k] = (int)r;
            s.md[k3][k] = fd[k] >= 0.5f ? 1 : (fd[k] <= -0.5f ? -1 : 0);
        }
    }
    if (wv == 0) { horw[lane] = 0; verw[lane] = 0; }
    int outd[3][4];
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3)
#pragma unroll
        for (int k = 0; k < 4; ++k) outd[k3][k] = 0;
    __syncthreads();

    // ---- set_partition_vector (Map2Partition.py:348-362), flattened: a node is reached iff every ancestor split
    for (int d = 0; d < 4; ++d) {
        const int sms = 8 >> d, nside = 1 << d;
        for (int node = 0; node < nside * nside; ++node) {
            const int qx = (node / nside) * sms, qy = (node % nside) * sms;
            bool reached = true;
            for (int a = 0; a < d; ++a) {
                const int am = ~((8 >> a) - 1);
                if (!(rlane(qt8, (qx & am) * 8 + (qy & am)) > a)) reached = false;
            }
            if (!reached) continue;
            if (wv != (d == 0 ? 0 : ((qx >= 4 ? 2 : 0) + (qy >= 4 ? 1 : 0)))) continue;   // another wave's quadrant
            const int c = rlane(qt8, qx * 8 + qy);
            if (c > d) {
                if (d < 3 && lane < 2 * sms) {            // paint the QT cross
                    hor[(2 * qx + sms) * 16 + 2 * qy + lane] = 1;
                    ver[(2 * qx + lane) * 16 + 2 * qy + sms] = 1;
                }
                continue;
            }
            if (c < d) continue;                          // hole: no MTT edges, dire stays 0
            // ---- set_bt_partition_vector (Map2Partition.py:287-346) on [2qx, 2qy, 2sms, 2sms]
            s.rx = 2 * qx; s.ry = 2 * qy; s.rh = 2 * sms; s.rw = 2 * sms;
#pragma unroll
            for (int k = 0; k < 4; ++k) { s.bt[0][k] = 0; s.dr[0][k] = 0; }
            s.cu[0] = pack_cu(s.rx, s.ry, s.rh, s.rw);
            s.ncu[0] = 1;
            s.have_best = 0;
            s.best_err = 0.f;
            s.best_ncu = 0;
            s.best_cu = 0;
            expand<0>(s);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = s.row - s.rx, cc = s.col0 + k - s.ry;
                if (r >= 0 && r < s.rh && cc >= 0 && cc < s.rw) {
                    outd[0][k] = s.best_dr[0][k]; outd[1][k] = s.best_dr[1][k]; outd[2][k] = s.best_dr[2][k];
                }
            }
            const int j = lane & 15, part = lane >> 4;
            for (int ci = 0; ci < s.best_ncu; ++ci) {
                const int cu = rlane(s.best_cu, ci);
                const int x = cu & 15, y = (cu >> 4) & 15, h = ((cu >> 8) & 15) + 1, w = ((cu >> 12) & 15) + 1;
                // par_vec is 17x17 in the reference; row/column 16 is cropped away (Map2Partition.py:373)
                if (part == 0 && j < w) hor[x * 16 + y + j] = 1;
                if (part == 1 && j < w && x + h < 16) hor[(x + h) * 16 + y + j] = 1;
                if (part == 2 && j < h) ver[(x + j) * 16 + y] = 1;
                if (part == 3 && j < h && y + w < 16) ver[(x + j) * 16 + y + w] = 1;
            }
        }
    }
    __syncthreads();
    if (wv == 0) {
        reinterpret_cast<uint32_t *>(hor_o + b * s_edge)[lane] = horw[lane];
        reinterpret_cast<uint32_t *>(ver_o + b * s_edge)[lane] = verw[lane];
    }
    // a lane's four cells (row lane>>2, columns 4*(lane&3)..+3) lie in one quadrant: its owner holds their directions
    const bool whole = rlane(qt8, 0) == 0;   // the block is one 64x64 leaf: wave 0 searched it
    const int quad = ((lane >> 2) >= 8 ? 2 : 0) + ((lane & 3) >= 2 ? 1 : 0);
    if (whole ? wv == 0 : wv == quad) {
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3) {
            const uint32_t pk = (uint32_t)(uint8_t)outd[k3][0] | ((uint32_t)(uint8_t)outd[k3][1] << 8) |
                                ((uint32_t)(uint8_t)outd[k3][2] << 16) | ((uint32_t)(uint8_t)outd[k3][3] << 24);
            reinterpret_cast<uint32_t *>(dire_o + b * s_dire + k3 * 256)[lane] = pk;
        }
    }
}

hipError_t launch_postprocess(hipStream_t st, const float *qt, const float *bt, const float *dire, int64_t N,
                              int chroma_factor, uint8_t *hor, uint8_t *ver, uint8_t *qt_u8, int8_t *dire_i8, int record_stride)
{
    if (N <= 0) return hipSuccess;
    // four dense arrays (strides 256 / 256 / 64 / 768 bytes per block), or one packed record per block (include/pmp.h)
    const int se = record_stride ? record_stride : 256, sq = record_stride ? record_stride : 64, sd = record_stride ? record_stride : 768;
    hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)N), dim3(256), 0, st, qt, bt, dire, N, chroma_factor, hor, ver,
                       qt_u8, dire_i8, se, sq, sd);
    return hipGetLastError();
}

// =============================================================================================== block cutter
// output_block_yuv (Inference_QBD.py:104-149): zero-pad top/left by the overlap, slice (bs+ov)^2 windows.
// 10-bit input: np.round(x / 4) in float64 = round half to even, then clip to 255 (:106-109) - done in integers.
template <typename T>
__global__ __launch_bounds__(256) void cut_plane_kernel(const T *__restrict__ plane, int F, int H, int W, int bs, int ov,
                                                        int bh, int bw, int tenbit, uint8_t *__restrict__ out)
{
    const int S = bs + ov;
    const size_t total = (size_t)F * bh * bw * S * S;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % S);
        size_t r = i / S;
        const int rr = (int)(r % S); r /= S;
        const int j = (int)(r % bw); r /= bw;
        const int bi = (int)(r % bh);
        const int f = (int)(r / bh);
        const int yy = bi * bs + rr - ov, xx = j * bs + c - ov;
        unsigned v = 0;
        if (yy >= 0 && xx >= 0) {
            v = plane[((size_t)f * H + yy) * W + xx];
            if (tenbit) {
                const unsigned qv = v >> 2, rem = v & 3;
                v = qv + (rem == 3 ? 1u : (rem == 2 ? (qv & 1u) : 0u));
                v = v > 255u ? 255u : v;
            }
        }
        out[i] = (uint8_t)v;
    }
}

hipError_t launch_cut_blocks(hipStream_t st, const void *y, const void *u, const void *v, int F, int H, int W,
                             int bitdepth, uint8_t *by, uint8_t *bu, uint8_t *bv)
{
    const int bh = H / 64, bw = W / 64;
    if (F <= 0 || bh <= 0 || bw <= 0) return hipSuccess;
    const size_t ny = (size_t)F * bh * bw * 68 * 68, nc = (size_t)F * bh * bw * 34 * 34;
    const unsigned gy = (unsigned)((ny + 255) / 256 > 8192 ? 8192 : (ny + 255) / 256);
    const unsigned gc = (unsigned)((nc + 255) / 256 > 8192 ? 8192 : (nc + 255) / 256);
    if (bitdepth == 8) {
        hipLaunchKernelGGL(cut_plane_kernel<uint8_t>, dim3(gy), dim3(256), 0, st, (const uint8_t *)y, F, H, W, 64, 4, bh, bw, 0, by);
        hipLaunchKernelGGL(cut_plane_kernel<uint8_t>, dim3(gc), dim3(256), 0, st, (const uint8_t *)u, F, H / 2, W / 2, 32, 2, bh, bw, 0, bu);
        hipLaunchKernelGGL(cut_plane_kernel<uint8_t>, dim3(gc), dim3(256), 0, st, (const uint8_t *)v, F, H / 2, W / 2, 32, 2, bh, bw, 0, bv);
    } else {
        hipLaunchKernelGGL(cut_plane_kernel<uint16_t>, dim3(gy), dim3(256), 0, st, (const uint16_t *)y, F, H, W, 64, 4, bh, bw, 1, by);
        hipLaunchKernelGGL(cut_plane_kernel<uint16_t>, dim3(gc), dim3(256), 0, st, (const uint16_t *)u, F, H / 2, W / 2, 32, 2, bh, bw, 1, bu);
        hipLaunchKernelGGL(cut_plane_kernel<uint16_t>, dim3(gc), dim3(256), 0, st, (const uint16_t *)v, F, H / 2, W / 2, 32, 2, bh, bw, 1, bv);
    }
    return hipGetLastError();
}

}  // namespace pmp
