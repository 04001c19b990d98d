// How does v_mfma_f32_16x16x32_f16 round?  One wave; A row 0 = chosen fp16 values, B column 0 = chosen fp16 values, C[0][0] = chosen fp32.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_round.hip -o tools/probe/mfma_round && tools/probe/mfma_round
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// a[k], b[k] for k = 0..31 (row 0 of A, column 0 of B), c = C[0][0]; everything else zero.  A operand: lane (i = lane & 15, kb = lane >> 4) holds
// A[i][8 kb .. 8 kb + 7]; B operand: lane (j = lane & 15, kb) holds B[8 kb ..][j]; D: lane (j = lane & 15, r = lane >> 4) holds D[4 r .. 4 r + 3][j].
__global__ void probe(const _Float16 *a, const _Float16 *b, float c, float *out)
{
    const int lane = threadIdx.x, i = lane & 15, kb = lane >> 4;
    f16x8 av, bv;
    for (int e = 0; e < 8; ++e) {
        av[e] = i == 0 ? a[8 * kb + e] : (_Float16)0.f;
        bv[e] = i == 0 ? b[8 * kb + e] : (_Float16)0.f;
    }
    f32x4 cv = {0.f, 0.f, 0.f, 0.f};
    if (lane == 0) cv[0] = c;
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, cv, 0, 0, 0);
    if (lane == 0) out[0] = d[0];
}

static float run(const float *av, const float *bv, float c)
{
    _Float16 ha[32], hb[32];
    for (int k = 0; k < 32; ++k) { ha[k] = (_Float16)av[k]; hb[k] = (_Float16)bv[k]; }
    _Float16 *da, *db; float *dout, h;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    hipMemcpy(da, ha, 64, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, c, dout);
    hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return h;
}

int main()
{
    float a[32], b[32];
    auto clear = [&]() { for (int k = 0; k < 32; ++k) { a[k] = 0.f; b[k] = 0.f; } };
    const float big = 16777216.f;   // 2^24: ulp 2 above, 1 below
    clear(); a[0] = 1.f; b[0] = 1.5f;
    printf("2^24 + 1.5        -> %.1f   (nearest 16777218, truncation 16777216)\n", run(a, b, big));
    clear(); a[0] = 1.f; b[0] = 1.f;
    printf("2^24 + 1 (tie)    -> %.1f   (ties-to-even 16777216, half-up 16777218)\n", run(a, b, big));
    clear(); a[0] = 1.f; b[0] = 3.f;
    printf("2^24 + 3 (tie)    -> %.1f   (ties-to-even 16777220, truncation 16777218)\n", run(a, b, big));
    clear(); a[0] = -1.f; b[0] = 1.5f;
    printf("-2^24 - 1.5       -> %.1f   (nearest -16777218, toward zero -16777216)\n", run(a, b, -big));
    clear(); a[0] = -1.f; b[0] = 0.75f;
    printf("2^24 - 0.75       -> %.1f   (nearest 16777215, up 16777216)\n", run(a, b, big));
    for (int j = 1; j <= 12; ++j) {      // 32 products of 2^-j each: exact sum 2^(5-j)
        clear();
        for (int k = 0; k < 32; ++k) { a[k] = 1.f; b[k] = ldexpf(1.f, -j); }
        printf("2^24 + 32 x 2^-%-2d (= %9.5f) -> %.1f\n", j, ldexpf(1.f, 5 - j), run(a, b, big));
    }
    for (int j = 0; j <= 6; ++j) {       // one big product plus 31 small ones: are the small ones summed before they meet the big one?
        clear();
        a[0] = 1.f; b[0] = 1.f;          // 1 (half an ulp of 2^24)
        for (int k = 1; k < 32; ++k) { a[k] = 1.f; b[k] = ldexpf(1.f, -j - 5); }     // 31 x 2^-(j+5)
        printf("2^24 + 1 + 31 x 2^-%-2d (= 1 + %8.6f) -> %.1f   (exact sum then round: 16777218)\n", j + 5, 31 * ldexpf(1.f, -j - 5), run(a, b, big));
    }
    // products only, C = 0: is the 32-term dot product itself exact?  1 + 31 x 2^-24-ish
    clear(); a[0] = 1.f; b[0] = 1.f; for (int k = 1; k < 32; ++k) { a[k] = ldexpf(1.f, -12); b[k] = ldexpf(1.f, -12); }
    printf("0 + 1 + 31 x 2^-24 -> %.10f   (exact: 1.0000018477, fp32 nearest 1.0000018477 = 1 + 31 x 2^-24 -> 1 + 15.5 ulp)\n", run(a, b, 0.f));
    return 0;
}
