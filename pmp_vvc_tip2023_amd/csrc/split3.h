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

// fp32 blocked tensor or split-3 planes behind one store call (elements, not bytes, index both)
struct ActOut {
    float *f32;
    unsigned short *s3;
    size_t stride;
    __device__ __forceinline__ void store4(size_t elem, f32x4 v) const
    {
        if (s3) store_split4(s3 + elem, stride, v);
        else *reinterpret_cast<f32x4 *>(f32 + elem) = v;
    }
};

}  // namespace pmp
