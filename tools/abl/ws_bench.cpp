// abl/ws_bench.cpp — MEASUREMENT LIBRARY ONLY: one 3x3 64->64 convolution (+ identity residual, ReLU) on random data as the product's
// launch (conv_f16x3.hip) and as the weight-stationary persistent prototype (abl/conv_ws.hip): bit comparison and interleaved timing
// (tools/ws_probe.py).
#include <cmath>
#include <cstring>
#include <vector>

#include "pmp_host.h"

namespace pmp {
struct ConvWsArgs {
    const unsigned short *x; size_t x_stride;
    const unsigned short *w; float inv_scale;
    const unsigned short *res; size_t res_stride;
    unsigned short *out; size_t out_stride;
    int N, H, W;
    unsigned *sat;
    const void *zeros;
};
hipError_t launch_conv_ws(hipStream_t s, const ConvWsArgs &a, int abl, int nbuf, int grid);
}  // namespace pmp

using namespace pmp;

extern "C" int pmp_abl_ws_bench(pmp_ctx *c, int n, int h, int w, int with_res, int iters, int rounds, int abl, int nbuf, int grid,
                                double *ms_ref, double *ms_ws, int64_t *mismatch, double *max_abs_diff, double *max_abs_ref)
{
    if (!c || n <= 0 || (h & 15) || (w & 15) || iters <= 0 || rounds <= 0) return set_err(c, PMP_E_INVALID, "pmp_abl_ws_bench: bad arguments");
    hipSetDevice(c->device);
    if (grid <= 0) {
        hipDeviceProp_t prop;
        hipGetDeviceProperties(&prop, c->device);
        grid = prop.multiProcessorCount;
    }
    const size_t ne = (size_t)n * 64 * h * w;
    std::vector<float> hx(ne), hr(ne), hw((size_t)64 * 64 * 9);
    unsigned long long st = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) / 16777216.0) * 2.f - 1.f; };
    for (auto &v : hx) { const float r = rnd(); v = r > 0 ? r * 3.f : 0.f; }       // ReLU outputs: half the values are zero
    for (auto &v : hr) { const float r = rnd(); v = r > 0 ? r * 3.f : 0.f; }
    const float ws = 1.f / sqrtf(64.f * 9.f);
    for (auto &v : hw) v = rnd() * ws * 1.7f;
    const int k = h2_scale_exp(hw.data(), hw.size());
    const std::vector<unsigned short> pk = pack_h2(hw.data(), 64, 64, 3, 3, 64, 64, k);
    float *dx = nullptr;
    unsigned short *dxs = nullptr, *drs = nullptr, *dy = nullptr, *dy2 = nullptr, *dw = nullptr;
    hipError_t e = hipSuccess;
    auto A = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    A((void **)&dx, ne * 4); A((void **)&dxs, ne * 4); A((void **)&drs, ne * 4); A((void **)&dy, ne * 4); A((void **)&dy2, ne * 4);
    A((void **)&dw, pk.size() * 2);
    int rc = PMP_OK;
    if (e != hipSuccess) rc = hip_fail(c, e, "hipMalloc(ws bench)");
    if (rc == PMP_OK) {
        hipMemcpy(dw, pk.data(), pk.size() * 2, hipMemcpyHostToDevice);
        hipMemset(dy, 0, ne * 4); hipMemset(dy2, 0xff, ne * 4);
        hipMemcpy(dx, hx.data(), ne * 4, hipMemcpyHostToDevice);
        launch_f32_to_split2(c->stream, dx, dxs, ne, ne);
        hipStreamSynchronize(c->stream);
        hipMemcpy(dx, hr.data(), ne * 4, hipMemcpyHostToDevice);
        launch_f32_to_split2(c->stream, dx, drs, ne, ne);
        ConvX6Args a1{};
        a1.x = dxs; a1.x_stride = ne; a1.w = dw; a1.out = dy; a1.out_stride = ne; a1.N = n; a1.H = h; a1.W = w; a1.Cin = 64; a1.Cout = 64; a1.KH = a1.KW = 3;
        a1.relu = 1; a1.out_scale = std::ldexp(1.f, -k); a1.sat = c->d_sat; a1.abl.zeros = c->d_sat + 16;
        if (with_res) { a1.res = drs; a1.res_stride = ne; }
        ConvWsArgs f{dxs, ne, dw, std::ldexp(1.f, -k), with_res ? drs : nullptr, ne, dy2, ne, n, h, w, c->d_sat, c->d_sat + 16};
        auto ref = [&]() { return launch_conv_h2(c->stream, a1); };
        auto wsk = [&](int ab) { return launch_conv_ws(c->stream, f, ab, nbuf, grid); };
        e = ref();
        if (e == hipSuccess) e = wsk(0);
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
            for (int i = 0; i < iters; ++i) ref();
            hipEventRecord(e1, c->stream); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            tp = fmin(tp, ms / iters);
            hipEventRecord(e0, c->stream);
            for (int i = 0; i < iters; ++i) e = wsk(abl);
            hipEventRecord(e1, c->stream); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            tf = fmin(tf, ms / iters);
        }
        if (ms_ref) *ms_ref = tp;
        if (ms_ws) *ms_ws = tf;
        hipEventDestroy(e0); hipEventDestroy(e1);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) rc = hip_fail(c, e, "ws bench");
        unsigned zero = 0;
        hipMemcpy(c->d_sat, &zero, sizeof(zero), hipMemcpyHostToDevice);
    }
    for (void *p : {(void *)dx, (void *)dxs, (void *)drs, (void *)dy, (void *)dy2, (void *)dw}) if (p) hipFree(p);
    return rc;
}
