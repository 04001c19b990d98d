// split3.h — device helpers of the "split-3" activation format (conv_bf16x6.hip): v = v0 + v1 + v2 with three bf16 terms.
#pragma once
#include <hip/hip_runtime.h>

namespace pmp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float v, __bf16 &a, __bf16 &b, __bf16 &c)
{
    a = (__bf16)v;
    const float r1 = v - (float)a;   // exact
    b = (__bf16)r1;
    const float r2 = r1 - (float)b;  // exact
    c = (__bf16)r2;                  // exact: at most 8 significant bits are left
}

// 4 consecutive channels of one pixel from the three planes, summed exactly back to fp32
__device__ __forceinline__ f32x4 load_split4(const unsigned short *p, size_t plane_stride)
{
    const bf16x4 a = *reinterpret_cast<const bf16x4 *>(p), b = *reinterpret_cast<const bf16x4 *>(p + plane_stride),
                 c = *reinterpret_cast<const bf16x4 *>(p + 2 * plane_stride);
    f32x4 v;
    v.x = ((float)a.x + (float)b.x) + (float)c.x; v.y = ((float)a.y + (float)b.y) + (float)c.y;
    v.z = ((float)a.z + (float)b.z) + (float)c.z; v.w = ((float)a.w + (float)b.w) + (float)c.w;
    return v;
}

__device__ __forceinline__ void store_split4(unsigned short *p, size_t plane_stride, f32x4 v)
{
    __bf16 a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
    split3(v.x, a0, b0, c0); split3(v.y, a1, b1, c1); split3(v.z, a2, b2, c2); split3(v.w, a3, b3, c3);
    const bf16x4 a = {a0, a1, a2, a3}, b = {b0, b1, b2, b3}, c = {c0, c1, c2, c3};
    *reinterpret_cast<bf16x4 *>(p) = a;
    *reinterpret_cast<bf16x4 *>(p + plane_stride) = b;
    *reinterpret_cast<bf16x4 *>(p + 2 * plane_stride) = c;
}

// halo-tile geometry of the split-3 MFMA kernels (conv_bf16x6.hip, conv_x6_ws.hip): 16x16 output pixels per workgroup
template <int KH, int KW>
struct GeoX {
    static constexpr int TH = 16 + KH - 1, TW = 16 + KW - 1, TAPS = KH * KW, NKS = (TAPS + 1) / 2;
    static constexpr int PLANE = TH * TW * 2;            // 16-B pieces per split plane (32 B per pixel)
    static constexpr int PIECES = 3 * PLANE;             // per buffer
    static constexpr int NLD = (PIECES + 255) / 256;
};

// ---- "split-2": v = h0 + h1 with two fp16 terms (22 significand bits; conv_f16x3.hip).  Values beyond the fp16 range are
// clamped to +-65504 (the reference's activations stay below 3e3 on every test input, tools/precision_study.py).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2(float v, _Float16 &a, _Float16 &b)
{
    v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    a = (_Float16)v;
    b = (_Float16)(v - (float)a);   // v - a is exact in fp32; b keeps its leading 11 bits
}

// ---- mixed-precision FMA (v_fma_mix_f32 / v_fma_mixlo_f16 / v_fma_mixhi_f16): an fp16 operand enters an fp32 FMA without a
// conversion instruction and the fp32 result can be written as fp16.  Two uses, both BIT-IDENTICAL to the conversion sequences they
// replace (the FMA's single rounding acts on an exactly representable value):
//   value of a split-2 element, (float)h0 + (float)h1:     one instruction instead of two conversions and an add;
//   second term of a split, h1 = fp16(v - (float)h0):       one instruction instead of a conversion back and a subtraction.
// hipcc folds fma(x, 1, y) into an add and never emits these instructions by itself, hence the asm.
__device__ __forceinline__ float h2_sum_lo(unsigned h0, unsigned h1)   // (float)lo16(h0) + (float)lo16(h1)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h0), "v"(h1));
    return d;
}
__device__ __forceinline__ float h2_sum_hi(unsigned h0, unsigned h1)   // (float)hi16(h0) + (float)hi16(h1)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h0), "v"(h1));
    return d;
}
// (v, w), both inside the fp16 range -> p = (f16(v), f16(w)) and q = (f16(v - lo(p)), f16(w - hi(p))): 3 instructions for 2 values
__device__ __forceinline__ void h2_split_pair_noclamp(float v, float w, unsigned &p, unsigned &q)
{
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(v), "v"(w));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(q) : "v"(v), "v"(w), "v"(p));
}
// the same with the clamp of split2(): values beyond +-65504 saturate (and raise the range flag where the caller tracks it)
__device__ __forceinline__ void h2_split_pair(float v, float w, unsigned &p, unsigned &q)
{
    h2_split_pair_noclamp(__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w, -65504.f, 65504.f), p, q);
}

// Saturation tracking (pmp_get_saturation, include/pmp.h): every kernel that writes split-2 planes keeps the largest |value|
// it stores and raises the context's sticky flag when the clamp above fired for any of them.
// The maximum is taken on the magnitudes' BIT PATTERNS as unsigned integers: ordered like the floats for finite values and
// infinities, and a NaN (exponent all ones, non-zero mantissa) ranks above them all - so a NaN activation raises the flag too
// (fmaxf would drop it and `NaN > 65504.f` is false).
__device__ __forceinline__ unsigned sat_bits(float x) { return __float_as_uint(x) & 0x7fffffffu; }

__device__ __forceinline__ float sat_amax4(float m, f32x4 v)
{
    const unsigned a = max(sat_bits(v.x), sat_bits(v.y)), b = max(sat_bits(v.z), sat_bits(v.w));
    return __uint_as_float(max(max(sat_bits(m), a), b));
}

__device__ __forceinline__ void sat_report(unsigned *flag, float amax)
{
    if (flag && sat_bits(amax) > 0x477fe000u) atomicOr(flag, 1u);   // 0x477fe000 = 65504.f; true for NaN as well
}

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 load_split2_4(const unsigned short *p, size_t plane_stride)
{
    const u32x2_t a = *reinterpret_cast<const u32x2_t *>(p), b = *reinterpret_cast<const u32x2_t *>(p + plane_stride);
    return (f32x4){h2_sum_lo(a.x, b.x), h2_sum_hi(a.x, b.x), h2_sum_lo(a.y, b.y), h2_sum_hi(a.y, b.y)};
}

__device__ __forceinline__ void store_split2_4(unsigned short *p, size_t plane_stride, f32x4 v)
{
    u32x2_t a, b;
    unsigned a0, b0, a1, b1;
    h2_split_pair(v.x, v.y, a0, b0);
    h2_split_pair(v.z, v.w, a1, b1);
    a.x = a0; a.y = a1; b.x = b0; b.y = b1;
    *reinterpret_cast<u32x2_t *>(p) = a;
    *reinterpret_cast<u32x2_t *>(p + plane_stride) = b;
}

// ---- 16-byte epilogue accesses of the split-2 kernels.  An MFMA accumulator leaves lane (xl, g) with the 4 couts 4g.. of one
// pixel; two rows (m, m+1) of it are 2 x 2 packed dwords per plane.  v_permlane16_swap_b32 (gfx950) swaps the odd 16-lane rows
// of one register with the even rows of another, which turns {row m, row m+1} x {couts 4g..} into the 8 consecutive channels
// 8(g>>1).. of row m + (g&1): one 16-byte access per lane instead of two 8-byte ones.  The exchange is its own inverse.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void rows16_swap(u32x4 &v)   // (x, y) = row m half, (z, w) = row m+1 half  <->  16 contiguous bytes
{
    const u32x2 a = __builtin_amdgcn_permlane16_swap(v.x, v.z, false, false);
    const u32x2 b = __builtin_amdgcn_permlane16_swap(v.y, v.w, false, false);
    v.x = a.x; v.z = a.y; v.y = b.x; v.w = b.y;
}

__device__ __forceinline__ f32x4 h2_lo4(u32x4 v)   // the 4 fp16 values in (x, y) as floats
{
    const f16x4 h = __builtin_bit_cast(f16x4, (u32x2){v.x, v.y});
    return (f32x4){(float)h.x, (float)h.y, (float)h.z, (float)h.w};
}

__device__ __forceinline__ f32x4 h2_hi4(u32x4 v)
{
    const f16x4 h = __builtin_bit_cast(f16x4, (u32x2){v.z, v.w});
    return (f32x4){(float)h.x, (float)h.y, (float)h.z, (float)h.w};
}

// values of the 4 channels in (x, y) / (z, w) of a high-term register p and a low-term register q
__device__ __forceinline__ f32x4 h2_sum4_lo(u32x4 p, u32x4 q) { return (f32x4){h2_sum_lo(p.x, q.x), h2_sum_hi(p.x, q.x), h2_sum_lo(p.y, q.y), h2_sum_hi(p.y, q.y)}; }
__device__ __forceinline__ f32x4 h2_sum4_hi(u32x4 p, u32x4 q) { return (f32x4){h2_sum_lo(p.z, q.z), h2_sum_hi(p.z, q.z), h2_sum_lo(p.w, q.w), h2_sum_hi(p.w, q.w)}; }

// two rows of one cout group -> their high-term dwords p = (row m: x, y | row m+1: z, w) and low-term dwords q
__device__ __forceinline__ void split2_rows(f32x4 r0, f32x4 r1, u32x4 &p, u32x4 &q)
{
    unsigned p0, q0, p1, q1, p2, q2, p3, q3;
    h2_split_pair(r0.x, r0.y, p0, q0);
    h2_split_pair(r0.z, r0.w, p1, q1);
    h2_split_pair(r1.x, r1.y, p2, q2);
    h2_split_pair(r1.z, r1.w, p3, q3);
    p = (u32x4){p0, p1, p2, p3};
    q = (u32x4){q0, q1, q2, q3};
}

// same, with the streaming (non-temporal) hint: activations are written once and read by the NEXT launch, far beyond L2
__device__ __forceinline__ void store_split2_4_nt(unsigned short *p, size_t plane_stride, f32x4 v)
{
    _Float16 a0, a1, a2, a3, b0, b1, b2, b3;
    split2(v.x, a0, b0); split2(v.y, a1, b1); split2(v.z, a2, b2); split2(v.w, a3, b3);
    const f16x4 a = {a0, a1, a2, a3}, b = {b0, b1, b2, b3};
    __builtin_nontemporal_store(a, reinterpret_cast<f16x4 *>(p));
    __builtin_nontemporal_store(b, reinterpret_cast<f16x4 *>(p + plane_stride));
}

// ---- geometry shared by conv_f16x3.hip and conv_f16x3_ws.hip
template <int KH, int KW, int TR = 16, int NTHR = 256, bool DMA = false>   // TR = output rows per workgroup tile (16 columns always), NTHR = threads that stage it
struct GeoH {
    static constexpr int TH = TR + KH - 1, TW = 16 + KW - 1, TAPS = KH * KW, NKS = (TAPS + 1) / 2;
    static constexpr int PLANE = TH * TW * 2;            // 16-B pieces per split plane (32 B per pixel)
    static constexpr int PIECES = 2 * PLANE;             // per buffer
    static constexpr int NLD = (PIECES + NTHR - 1) / NTHR;
    static constexpr int NQ = (PIECES + 63) / 64;        // DMA form: wave-instructions of 64 pieces per buffer
    static constexpr int BUF = DMA ? NQ * 64 : PIECES + 1;   // pieces reserved per LDS buffer.  Register staging: one spare slot, so
                                                         // that every staging thread stores all NLD of its pieces unconditionally (a
                                                         // conditional store lets hipcc sink the global load into the branch, next to
                                                         // its use, and wait for it there).  DMA form (loader wave): whole wave-instructions
};

// Wave tile: RW rows x CW cout groups of the workgroup's 16 rows x NT groups (4 waves).  At Cout >= 32 a wave takes 8 rows
// and half (or all) of the cout groups: it then streams half of the weight bytes per MFMA from L2 - the vector-memory
// path is the contended one here - and reads twice the pixel fragments from LDS, which has the headroom.
template <int NT, int W8 = 0>   // W8 = 1: 512-thread workgroups, 8 waves = 2 row halves x 4 cout groups; 2: 4 waves x (16 rows, 1 cout group) (Cout = 64 only)
struct WaveTile {
    static constexpr int WAVES = W8 == 1 ? 8 : 4;
    static constexpr int RW = W8 == 2 ? 16 : NT == 4 ? 8 : 4;   // W8 == 2: 4 waves, each all 16 rows x one cout group        // rows per wave (Cout <= 32: 4 rows - fewer registers, a third workgroup per CU)
                                                      // (W8 with 4 rows x 2 cout groups per wave - twice the weight bytes per MFMA - is 3 % slower than 8 x 1)
    static constexpr int RSPLIT = 16 / RW;            // waves along the rows
    static constexpr int CW = NT / (WAVES / RSPLIT);  // cout groups per wave (WAVES = RSPLIT x NT/CW)
};

// s_barrier after an LDS-only wait: global loads (weights, halo requests) stay in flight across it
__device__ __forceinline__ void h2_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Activation formats: 0 = plain fp32, 1 = split-3 (three bf16 planes), 2 = split-2 (two fp16 planes).
enum { FMT_F32 = 0, FMT_B3 = 1, FMT_H2 = 2 };

template <int FMT>
__device__ __forceinline__ f32x4 load_fmt4(const unsigned short *p, size_t plane_stride)
{
    if (FMT == FMT_H2) return load_split2_4(p, plane_stride);
    return load_split4(p, plane_stride);
}

template <int FMT>
__device__ __forceinline__ void store_fmt4(unsigned short *p, size_t plane_stride, f32x4 v)
{
    if (FMT == FMT_H2) store_split2_4(p, plane_stride, v);
    else store_split4(p, plane_stride, v);
}

// fp32 blocked tensor or split planes behind one store call (elements, not bytes, index both)
struct ActOut {
    float *f32;
    unsigned short *s3;   // first split plane (format `fmt`), or nullptr for fp32 output
    size_t stride;
    int fmt;
    unsigned *sat;        // split-2 only: the context's saturation flag (may be nullptr)
    __device__ __forceinline__ void store4(size_t elem, f32x4 v) const
    {
        if (!s3) *reinterpret_cast<f32x4 *>(f32 + elem) = v;
        else if (fmt == FMT_H2) { sat_report(sat, sat_amax4(0.f, v)); store_split2_4(s3 + elem, stride, v); }
        else store_split4(s3 + elem, stride, v);
    }
};

}  // namespace pmp
