// abl/rbfuse_bench.cpp — MEASUREMENT LIBRARY ONLY: one ResidualBlock(64, 64, 3) on random data, as the launch pair of the product
// (conv_f16x3.hip twice) and as the fused prototype (rbfuse_proto.hip): bit comparison and interleaved timing (tools/rbfuse_probe.py).
#include <cmath>
#include <cstring>
#include <vector>

#include "pmp_host.h"

namespace pmp {
struct RbFuseArgs {
    const unsigned short *x; size_t x_stride;
    const unsigned short *w1, *w2; float s1, s2;
    unsigned short *out; size_t out_stride;
    int N, H, W;
    unsigned *sat;
};
hipError_t launch_rb64_fused(hipStream_t s, const RbFuseArgs &a, int abl);
}  // namespace pmp

using namespace pmp;

extern "C" int pmp_abl_rbfuse_bench(pmp_ctx *c, int n, int h, int w, int iters, int rounds, int abl, double *ms_pair, double *ms_fused,
                                    int64_t *mismatch, double *max_abs_diff, double *max_abs_ref)
{
    if (!c || n <= 0 || (h & 15) || (w & 15) || iters <= 0 || rounds <= 0) return set_err(c, PMP_E_INVALID, "pmp_abl_rbfuse_bench: bad arguments");
    hipSetDevice(c->device);
    const size_t ne = (size_t)n * 64 * h * w;
    std::vector<float> hx(ne), hw1((size_t)64 * 64 * 9), hw2(hw1.size());
    unsigned long long st = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) / 16777216.0) * 2.f - 1.f; };
    for (auto &v : hx) { const float r = rnd(); v = r > 0 ? r * 3.f : 0.f; }       // a ReLU output: half the values are zero
    const float ws = 1.f / sqrtf(64.f * 9.f);
    for (auto &v : hw1) v = rnd() * ws * 1.7f;
    for (auto &v : hw2) v = rnd() * ws * 1.7f;
    const int k1 = h2_scale_exp(hw1.data(), hw1.size()), k2 = h2_scale_exp(hw2.data(), hw2.size());
    const std::vector<unsigned short> p1 = pack_h2(hw1.data(), 64, 64, 3, 3, 64, 64, k1), p2 = pack_h2(hw2.data(), 64, 64, 3, 3, 64, 64, k2);
    float *dx = nullptr;
    unsigned short *dxs = nullptr, *dt = nullptr, *dy = nullptr, *dy2 = nullptr, *dw1 = nullptr, *dw2 = nullptr;
    hipError_t e = hipSuccess;
    auto A = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    A((void **)&dx, ne * 4); A((void **)&dxs, ne * 4); A((void **)&dt, ne * 4); A((void **)&dy, ne * 4); A((void **)&dy2, ne * 4);
    A((void **)&dw1, p1.size() * 2); A((void **)&dw2, p2.size() * 2);
    int rc = PMP_OK;
    if (e != hipSuccess) rc = hip_fail(c, e, "hipMalloc(rbfuse bench)");
    if (rc == PMP_OK) {
        hipMemcpy(dx, hx.data(), ne * 4, hipMemcpyHostToDevice);
        hipMemcpy(dw1, p1.data(), p1.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dw2, p2.data(), p2.size() * 2, hipMemcpyHostToDevice);
        hipMemset(dy, 0, ne * 4); hipMemset(dy2, 0xff, ne * 4);
        launch_f32_to_split2(c->stream, dx, dxs, ne, ne);
        ConvX6Args a1{}, a2{};
        a1.x = dxs; a1.x_stride = ne; a1.w = dw1; a1.out = dt; a1.out_stride = ne; a1.N = n; a1.H = h; a1.W = w; a1.Cin = 64; a1.Cout = 64; a1.KH = a1.KW = 3;
        a1.relu = 1; a1.out_scale = std::ldexp(1.f, -k1); a1.sat = c->d_sat; a1.abl.zeros = c->d_sat + 16;
        a2 = a1;
        a2.x = dt; a2.w = dw2; a2.out = dy; a2.res = dxs; a2.res_stride = ne; a2.out_scale = std::ldexp(1.f, -k2);
        RbFuseArgs f{dxs, ne, dw1, dw2, std::ldexp(1.f, -k1), std::ldexp(1.f, -k2), dy2, ne, n, h, w, c->d_sat};
        auto pair = [&]() { launch_conv_h2(c->stream, a1); return launch_conv_h2(c->stream, a2); };
        auto fused = [&](int ab) { return launch_rb64_fused(c->stream, f, ab); };
        e = pair();
        if (e == hipSuccess) e = fused(0);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) {     // bit comparison of the exact build
            std::vector<unsigned short> y1(ne * 2), y2(ne * 2);
            hipMemcpy(y1.data(), dy, ne * 4, hipMemcpyDeviceToHost);
            hipMemcpy(y2.data(), dy2, ne * 4, hipMemcpyDeviceToHost);
            int64_t bad = 0;
            double md = 0, mr = 0;
            auto val = [](unsigned short hbits) { _Float16 hh; memcpy(&hh, &hbits, 2); return (double)(float)hh; };
            for (size_t i = 0; i < ne; ++i) {
                if (y1[i] != y2[i] || y1[ne + i] != y2[ne + i]) ++bad;
                const double v1 = val(y1[i]) + val(y1[ne + i]), v2 = val(y2[i]) + val(y2[ne + i]);
                md = fmax(md, fabs(v1 - v2)); mr = fmax(mr, fabs(v1));
            }
            if (mismatch) *mismatch = bad;
            if (max_abs_diff) *max_abs_diff = md;
            if (max_abs_ref) *max_abs_ref = mr;
        }
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        double tp = 1e30, tf = 1e30;
        for (int r = 0; r < rounds && e == hipSuccess; ++r) {      // interleaved rounds, best of: same box, same clock state
            float ms = 0.f;
            hipEventRecord(e0, c->stream);
            for (int i = 0; i < iters; ++i) pair();
            hipEventRecord(e1, c->stream); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            tp = fmin(tp, ms / iters);
            hipEventRecord(e0, c->stream);
            for (int i = 0; i < iters; ++i) e = fused(abl);
            hipEventRecord(e1, c->stream); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            tf = fmin(tf, ms / iters);
        }
        if (ms_pair) *ms_pair = tp;
        if (ms_fused) *ms_fused = tf;
        hipEventDestroy(e0); hipEventDestroy(e1);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) rc = hip_fail(c, e, "rbfuse bench");
        unsigned zero = 0;
        hipMemcpy(c->d_sat, &zero, sizeof(zero), hipMemcpyHostToDevice);
    }
    for (void *p : {(void *)dx, (void *)dxs, (void *)dt, (void *)dy, (void *)dy2, (void *)dw1, (void *)dw2}) if (p) hipFree(p);
    return rc;
}
