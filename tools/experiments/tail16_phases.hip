// tail16_phases.hip — EXPERIMENT (never part of the product library): branch16_kernel of csrc/chain16.hip, copied, with s_memtime stamps
// at its phase boundaries, on random tensors (timing does not depend on values).  Where do the non-MFMA cycles of a block go?
// Built and driven by tools/experiments/tail16_phases.py.  Building blocks: the product's own csrc/tail16_dev.h.
#include <hip/hip_runtime.h>
#include "tail16_dev.h"

using namespace pmp;

struct PhArgs {
    const unsigned short *x; size_t x_stride;
    float *bt, *dire;
    T16RB r[3];
    const float *head_w, *head_b;
    long long *stamps;      // [blocks][4 waves][NST]
    int layer, variant;
};
constexpr int NST = 24;
#define STAMP(k) do { if ((threadIdx.x & 63) == 0) st[k] = clock64(); } while (0)

__global__ __launch_bounds__(T16_THREADS, 2) void branch16_phases(PhArgs a)
{
    __shared__ __attribute__((aligned(16))) char slots[T16_NSLOT * T16_SLOT];
    typedef T16Tile<2> WT;
    const int n = blockIdx.x, tid = threadIdx.x;
    long long *st = a.stamps + ((size_t)n * 4 + (tid >> 6)) * NST;
    STAMP(0);
    char *A = slots, *C = slots + 2 * T16_SLOT, *D = slots + 3 * T16_SLOT;
    const unsigned short *x = a.x + (size_t)n * 4 * 4096;
    float amax = 0.f;
    T16Pass<9, 2> pb0;
    t16_wstart(pb0, a.r[0].w0, 2, WT::ct());
    T16Fetch f01;
    t16_fetch(f01, x, a.x_stride, 0);
    t16_clear(slots, T16_NSLOT * T16_SLOT);
    __syncthreads();
    STAMP(1);
    T16Pass<9, 2> pb1;
    {
        f32x4 acc[8];
        // ---- t16_rb64, inlined with stamps
        const T16RB w = a.r[0];
        const int ct = WT::ct(), row0 = WT::row0();
        t16_park(f01, A);
        T16Fetch f23;
        t16_fetch(f23, x, a.x_stride, 2);
        __syncthreads();
        STAMP(2);                                        // X01 parked (global latency of the first fetch)
        t16_zero<8>(acc);
        t16_accumulate<9, 2, 8>(A, row0, acc, pb0);
        STAMP(3);                                        // conv1a: 9 K-steps x 24 MFMAs
        T16Pass<9, 2> p1b;
        t16_wstart(p1b, w.w0 + 9 * 2 * T16_KSTEP, 2, ct);
        __syncthreads();
        t16_park(f23, A);
        __syncthreads();
        STAMP(4);                                        // window switch
        t16_accumulate<9, 2, 8>(A, row0, acc, p1b);
        STAMP(5);                                        // conv1b
        T16Pass<9, 2> p2;
        t16_wstart(p2, w.w2, 2, ct);
        t16_fetch(f01, x, a.x_stride, 0);
        amax = t16_epilogue<8, false, false, T16_IMG>(acc, row0, T16Epi{w.s0, nullptr, nullptr, 0, C + ct * T16_SLOT, nullptr, nullptr, 0}, amax);
        __syncthreads();
        STAMP(6);                                        // epilogue 1 + barrier
        t16_zero<8>(acc);
        t16_accumulate<9, 2, 8>(C, row0, acc, p2);
        STAMP(7);                                        // conv2
        T16Pass<1, 2> p3a, p3b;
        t16_wstart(p3a, w.wsc, 2, ct);
        t16_wstart(p3b, w.wsc + 1 * 2 * T16_KSTEP, 2, ct);
        __syncthreads();
        t16_park(f01, C);
        __syncthreads();
        STAMP(8);                                        // shortcut reload
        t16_accumulate<1, 2, 8>(C, row0, acc, p3a);
        t16_accumulate<1, 2, 8>(A, row0, acc, p3b);
        t16_wstart(pb1, a.r[1].w0, 1, 0);
        __syncthreads();
        STAMP(9);                                        // shortcut passes + barrier
        amax = t16_epilogue<8, false, false, T16_IMG>(acc, WT::row0(), T16Epi{a.r[0].s2, nullptr, nullptr, 0, A + WT::ct() * T16_SLOT, nullptr, nullptr, 0}, amax);
        __syncthreads();
        STAMP(10);                                       // epilogue 2 + barrier
    }
    const int row1 = T16Tile<1>::row0();
    float *f0 = reinterpret_cast<float *>(A);
    T16Pass<9, 1> pb2;
    f32x4 acc[4];
    amax = t16_rb<1, 2, true>(a.r[1], A, C, acc, amax, pb1, [&]() __attribute__((always_inline)) { t16_wstart(pb2, a.r[2].w0, 1, 0); });
    STAMP(11);                                           // RB1 up to its last barrier
    amax = t16_epilogue<4, false, false, T16_IMG>(acc, row1, T16Epi{a.r[1].s2, nullptr, nullptr, 0, D, nullptr, nullptr, 0}, amax);
    __syncthreads();
    STAMP(12);
    amax = t16_rb<1, 1, true>(a.r[2], D, C, acc, amax, pb2, []() {});
    STAMP(13);                                           // RB2
    t16_epilogue<4, false, false, T16_F32>(acc, row1, T16Epi{a.r[2].s2, nullptr, nullptr, 0, nullptr, f0, nullptr, 0}, 0.f);
    __syncthreads();
    STAMP(14);
    {
        float acc0, acc1;
        t16_head<16>(f0, a.head_w, a.head_b, 2, tid, acc0, acc1);
        const size_t o = ((size_t)n * 3 + a.layer) * 256 + tid;
        if (a.layer > 0) acc0 += a.bt[o - 256];
        a.bt[o] = acc0;
        a.dire[o] = acc1;
    }
    STAMP(15);                                           // head
    if (amax < 0.f) a.bt[0] = amax;
}

extern "C" int phases_launch(const unsigned short *x, size_t x_stride, float *bt, float *dire, const unsigned short *w, const float *hw,
                             long long *stamps, int N, int layer)
{
    PhArgs a{};
    a.x = x; a.x_stride = x_stride; a.bt = bt; a.dire = dire; a.stamps = stamps; a.layer = layer;
    // one random stream serves every pass (the sizes only have to be large enough): r[0].w0 needs 18 * 2 * 1024 halves
    for (int i = 0; i < 3; ++i) a.r[i] = T16RB{w, w + 40000, w + 80000, 1.f / 4096, 1.f / 4096};
    a.head_w = hw; a.head_b = hw + 200;
    hipLaunchKernelGGL(branch16_phases, dim3(N), dim3(T16_THREADS), 0, 0, a);
    return (int)hipGetLastError();
}
