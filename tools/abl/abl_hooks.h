// abl/abl_hooks.h — MEASUREMENT library: the hooks the product compiles as no-ops (hooks/abl_hooks.h), implemented in abl/abl_host.cpp.
#pragma once
#include <functional>
#include <vector>

#include "abl_kernels.h"

namespace pmp {
// 3x3 64->64 only: Winograd F(2,3)-along-x form of the f16x3 stream; *scale_exp receives its power-of-two exponent
std::vector<unsigned short> pack_h2_wx(const float *w, int *scale_exp);

const char *abl_version();
void abl_on_create();
bool abl_set_conv_variant(int variant, int *rc);
bool abl_set_winograd(pmp_ctx *c, int on, int *rc);
unsigned abl_pack_mask(const pmp_ctx *c);
int abl_prepare_pass(pmp_ctx *c, NetWeights &wq, NetWeights &wb);
void abl_conv_args(const pmp_ctx *c, const RBWeights &r, bool second, ConvX6Args &a);
int abl_pack_rb(const float *w0, const float *w2, int k, int cin, int cout, unsigned mask, RBWeights &r,
                const std::function<int(const std::vector<unsigned short> &, unsigned short **)> &upload16);
struct AblBench { bool wino = false; int kexp_w = 0; unsigned short *dww = nullptr; };
void abl_bench_prepare(pmp_ctx *c, AblBench &ab, const float *w, int k, int cin, int cout, bool h2, ConvX6Args &b);
void abl_bench_report(pmp_ctx *c, AblBench &ab, bool h2, int n, int h, int w, int k, int cout, ConvX6Args &b, const std::function<hipError_t()> &launch_split);
void abl_bench_free(AblBench &ab);
}  // namespace pmp
