// abl/trunk_bench.cpp — MEASUREMENT LIBRARY ONLY: a trunk of L 3x3 64->64 layers (ResidualBlocks: conv + ReLU, conv + identity residual + ReLU)
// on random data, as L launches of the product kernel and as ONE launch of the layer-pipelined persistent kernel (abl/conv_f16x3.hip:
// trunk_pipe_kernel): bit comparison of the trunk's output and interleaved timing (tools/trunk_probe.py).
#include <cmath>
#include <cstring>
#include <vector>

#include "pmp_host.h"

namespace pmp {
struct TrunkPipeArgs {
    const ConvX6Args *layers;
    int L, N, D;
    unsigned *work, *done, *err;
};
hipError_t launch_trunk_pipe(hipStream_t s, const TrunkPipeArgs &p, int grid, int sync);
}  // namespace pmp

using namespace pmp;

extern "C" int pmp_abl_trunk_bench(pmp_ctx *c, int n, int h, int w, int L, int D, int iters, int rounds, int grid, int sync,
                                   double *ms_layers, double *ms_pipe, int64_t *mismatch, int *spin_limit_hit)
{
    if (!c || n <= 0 || (h & 15) || (w & 15) || L < 2 || L > 12 || (L & 1) || D < 1 || iters <= 0 || rounds <= 0)
        return set_err(c, PMP_E_INVALID, "pmp_abl_trunk_bench: bad arguments");
    hipSetDevice(c->device);
    if (grid <= 0) {
        hipDeviceProp_t prop;
        hipGetDeviceProperties(&prop, c->device);
        grid = 3 * prop.multiProcessorCount;
    }
    const size_t ne = (size_t)n * 64 * h * w;
    unsigned long long st = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) / 16777216.0) * 2.f - 1.f; };
    std::vector<float> hx(ne);
    for (auto &v : hx) { const float r = rnd(); v = r > 0 ? r * 3.f : 0.f; }
    const float ws = 1.f / sqrtf(64.f * 9.f);
    std::vector<unsigned short *> dw(L, nullptr);
    std::vector<float> scale(L);
    hipError_t e = hipSuccess;
    auto A = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    for (int l = 0; l < L; ++l) {
        std::vector<float> hw((size_t)64 * 64 * 9);
        for (auto &v : hw) v = rnd() * ws * ((l & 1) ? 0.6f : 1.7f);      // keeps the activations of a 10-layer chain in range
        const int k = h2_scale_exp(hw.data(), hw.size());
        const std::vector<unsigned short> pk = pack_h2(hw.data(), 64, 64, 3, 3, 64, 64, k);
        A((void **)&dw[l], pk.size() * 2);
        if (e == hipSuccess) hipMemcpy(dw[l], pk.data(), pk.size() * 2, hipMemcpyHostToDevice);
        scale[l] = std::ldexp(1.f, -k);
    }
    // tensors: act[0] = input, act[l + 1] = output of layer l; two sets (reference launches, pipelined kernel) sharing act[0]
    float *dx = nullptr;
    std::vector<unsigned short *> ref(L + 1, nullptr), pip(L + 1, nullptr);
    A((void **)&dx, ne * 4);
    for (int l = 0; l <= L; ++l) { A((void **)&ref[l], ne * 4); if (l) A((void **)&pip[l], ne * 4); }
    pip[0] = ref[0];
    ConvX6Args *dlayers = nullptr;
    unsigned *dcnt = nullptr;
    const size_t ncnt = 2 + (size_t)L * n;
    A((void **)&dlayers, sizeof(ConvX6Args) * L);
    A((void **)&dcnt, ncnt * 4);
    int rc = PMP_OK;
    if (e != hipSuccess) rc = hip_fail(c, e, "hipMalloc(trunk bench)");
    if (rc == PMP_OK) {
        hipMemcpy(dx, hx.data(), ne * 4, hipMemcpyHostToDevice);
        launch_f32_to_split2(c->stream, dx, ref[0], ne, ne);
        std::vector<ConvX6Args> la(L), lp(L);
        for (int l = 0; l < L; ++l) {
            ConvX6Args a{};
            a.x = ref[l]; a.x_stride = ne; a.w = dw[l]; a.out = ref[l + 1]; a.out_stride = ne; a.N = n; a.H = h; a.W = w; a.Cin = 64; a.Cout = 64; a.KH = a.KW = 3;
            a.relu = 1; a.out_scale = scale[l]; a.sat = c->d_sat; a.abl.zeros = c->d_sat + 16;
            if (l & 1) { a.res = ref[l - 1]; a.res_stride = ne; }      // second convolution of a ResidualBlock: + the block's input
            la[l] = a;
            a.x = pip[l]; a.out = pip[l + 1];
            if (l & 1) a.res = pip[l - 1];
            lp[l] = a;
        }
        hipMemcpy(dlayers, lp.data(), sizeof(ConvX6Args) * L, hipMemcpyHostToDevice);
        TrunkPipeArgs p{dlayers, L, n, D, dcnt, dcnt + 2, dcnt + 1};
        auto layers = [&]() { hipError_t r = hipSuccess; for (int l = 0; l < L && r == hipSuccess; ++l) r = launch_conv_h2(c->stream, la[l]); return r; };
        auto pipe = [&]() { hipMemsetAsync(dcnt, 0, ncnt * 4, c->stream); return launch_trunk_pipe(c->stream, p, grid, sync); };
        e = layers();
        if (e == hipSuccess) e = pipe();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        unsigned herr = 0;
        hipMemcpy(&herr, dcnt + 1, 4, hipMemcpyDeviceToHost);
        if (spin_limit_hit) *spin_limit_hit = (int)herr;
        if (e == hipSuccess) {     // bit comparison of every layer's output
            std::vector<unsigned short> y1(ne * 2), y2(ne * 2);
            int64_t bad = 0;
            for (int l = 1; l <= L; ++l) {
                hipMemcpy(y1.data(), ref[l], ne * 4, hipMemcpyDeviceToHost);
                hipMemcpy(y2.data(), pip[l], ne * 4, hipMemcpyDeviceToHost);
                bad += memcmp(y1.data(), y2.data(), ne * 4) ? 1 : 0;
                if (l == L)
                    for (size_t i = 0; i < ne * 2; ++i) bad += y1[i] != y2[i];
            }
            if (mismatch) *mismatch = bad;
        }
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        double tl = 1e30, tp = 1e30;
        for (int r = 0; r < rounds && e == hipSuccess && !herr; ++r) {
            float ms = 0.f;
            hipEventRecord(e0, c->stream);
            for (int i = 0; i < iters; ++i) layers();
            hipEventRecord(e1, c->stream); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            tl = fmin(tl, ms / iters);
            hipEventRecord(e0, c->stream);
            for (int i = 0; i < iters; ++i) e = pipe();
            hipEventRecord(e1, c->stream); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            tp = fmin(tp, ms / iters);
        }
        if (ms_layers) *ms_layers = tl;
        if (ms_pipe) *ms_pipe = tp;
        hipEventDestroy(e0); hipEventDestroy(e1);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) rc = hip_fail(c, e, "trunk bench");
        unsigned zero = 0;
        hipMemcpy(c->d_sat, &zero, sizeof(zero), hipMemcpyHostToDevice);
    }
    for (int l = 0; l <= L; ++l) { if (ref[l]) hipFree(ref[l]); if (l && pip[l]) hipFree(pip[l]); }
    for (auto p : dw) if (p) hipFree(p);
    for (void *p : {(void *)dx, (void *)dlayers, (void *)dcnt}) if (p) hipFree(p);
    return rc;
}
