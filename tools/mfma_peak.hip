// mfma_peak.hip — register-resident MFMA loops: what the matrix pipes of THIS box sustain (SURVEY.md 8d asks for it).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MF32(i) d##i = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d##i, 0, 0, 0)
template <int ACC>
__global__ __launch_bounds__(256) void k_f32(float *out, int iters, float a0, float b0)
{
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0, d4 = d0, d5 = d0, d6 = d0, d7 = d0;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) { MF32(0); MF32(1); MF32(2); MF32(3); MF32(4); MF32(5); MF32(6); MF32(7); }
    const f32x4 s = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

// explicit accumulators (arrays in a rolled loop make hipcc shuffle AGPRs every iteration, which hides the pipe rate)
#define MF(i) c##i = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c##i, 0, 0, 0)
template <int ACC>
__global__ __launch_bounds__(256) void k_bf16(float *out, int iters, float a0, float b0)
{
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(a0 + ((threadIdx.x * 2654435761u + j * 40503u) & 1023) * 1e-3f - 0.5f);
        b[j] = (__bf16)(b0 - ((threadIdx.x * 40503u + j * 2654435761u) & 1023) * 1e-3f + 0.5f);
    }
    for (int it = 0; it < iters; ++it) { MF(0); MF(1); MF(2); MF(3); MF(4); MF(5); MF(6); MF(7); }
    const f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// 32x32x16 form, distinct random-ish operands per accumulator, clock stamps: cycles vs wall
__global__ __launch_bounds__(256) void k_bf16_32(float *out, int iters, float a0, float b0, unsigned long long *stamp)
{
    f32x16 acc[4];
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)(a0 * (1 + i) + ((threadIdx.x * 2654435761u + j * 40503u + i * 7919u) & 1023) * 1e-3f - 0.5f);
            b[i][j] = (__bf16)(b0 * (1 + i) - ((threadIdx.x * 40503u + j * 2654435761u + i * 104729u) & 1023) * 1e-3f + 0.5f);
        }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc[i], 0, 0, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamp[0] = t1 - t0; stamp[1] = r1 - r0; }
}

template <typename F>
static double run(F launch, double flop_per_launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return flop_per_launch * reps / (ms * 1e-3) / 1e12;
}

int main()
{
    float *out; hipMalloc(&out, 4096 * 256 * 4 * 4);
    const int iters = 4000;
    for (int wg_per_cu : {1, 2}) {   // 4 waves (1 per SIMD) or 8 waves (2 per SIMD) per CU
        const int grid = 256 * wg_per_cu;
        double f = run([&] { hipLaunchKernelGGL(k_f32<8>, dim3(grid), dim3(256), 0, 0, out, iters, 0.5f, 0.25f); },
                       2.0 * 16 * 16 * 4 * 8.0 * iters * 4 * grid, 20);
        double b = run([&] { hipLaunchKernelGGL(k_bf16<8>, dim3(grid), dim3(256), 0, 0, out, iters, 0.5f, 0.25f); },
                       2.0 * 16 * 16 * 32 * 8.0 * iters * 4 * grid, 20);
        printf("%d wave(s)/SIMD: v_mfma_f32_16x16x4_f32 %.1f TFLOP/s   v_mfma_f32_16x16x32_bf16 %.1f TFLOP/s (= %.1f per fp32-equivalent product at 6 MFMAs)\n",
               wg_per_cu, f, b, b / 6.0);
    }
    // sustained: 2 s of back-to-back launches (clock settles under load)
    double b2 = run([&] { hipLaunchKernelGGL(k_bf16<8>, dim3(512), dim3(256), 0, 0, out, iters, 0.5f, 0.25f); },
                    2.0 * 16 * 16 * 32 * 8.0 * iters * 4 * 512, 400);
    unsigned long long *stamp; hipMalloc(&stamp, 16);
    double b3 = run([&] { hipLaunchKernelGGL(k_bf16_32, dim3(512), dim3(256), 0, 0, out, iters, 0.37f, 0.21f, stamp); },
                    2.0 * 32 * 32 * 16 * 4.0 * iters * 4 * 512, 400);
    unsigned long long hs[2]; hipMemcpy(hs, stamp, 16, hipMemcpyDeviceToHost);
    printf("sustained bf16 32x32x16, varied operands (400 launches): %.1f TFLOP/s = %.1f fp32-equivalent; in-kernel clock %.2f GHz (%.1f cycles per MFMA per wave)\n",
           b3, b3 / 6.0, (double)hs[0] / (double)hs[1] * 0.1, (double)hs[0] / (4.0 * iters));
    printf("sustained bf16 (400 launches): %.1f TFLOP/s = %.1f fp32-equivalent at 6 MFMAs per product\n", b2, b2 / 6.0);
    return 0;
}
