// Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs, and does the f32->f16 conversion produce them?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out, float a_val, float b_val)
{
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    a[0] = (_Float16)a_val;   // every lane: A[row][k0] = a_val
    b[0] = (_Float16)b_val;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main()
{
    float *d; hipMalloc(&d, 8);
    const float tests[][2] = {{9.5367431640625e-07f, 1024.f}, {3.0517578125e-05f, 1024.f}, {1.0f, 1024.f}, {5.9604644775390625e-08f, 16384.f}};
    for (auto &t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        // 4 lane-groups each contribute a[0]*b[0] at k = 8g -> sum over k of A[row][k]*B[k][col] = 4 * a*b
        printf("a=%.6e (as f16: %.6e) b=%g  mfma=%.6e  expected(4ab)=%.6e\n", t[0], h[1], t[1], h[0], 4.0 * t[0] * t[1]);
    }
    return 0;
}
