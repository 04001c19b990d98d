// hooks/abl_hooks.h — PRODUCT build: every hook of the measurement library is a no-op that the compiler removes (see abl_types.h).
// Include after pmp_host.h.
#pragma once
#include <functional>
#include <vector>

namespace pmp {
inline const char *abl_version() { return nullptr; }                                  // the version string of a measurement build
inline void abl_on_create() {}                                                         // environment knobs
inline bool abl_set_conv_variant(int, int *) { return false; }                         // true: handled, *rc is the answer
inline bool abl_set_winograd(pmp_ctx *, int, int *) { return false; }
inline unsigned abl_pack_mask(const pmp_ctx *) { return 0u; }                          // extra weight formats to pack at load
inline int abl_prepare_pass(pmp_ctx *, NetWeights &, NetWeights &) { return PMP_OK; }  // ... and before a pass
inline void abl_conv_args(const pmp_ctx *, const RBWeights &, bool, ConvX6Args &) {}   // extra arguments of one convolution launch
inline int abl_pack_rb(const float *, const float *, int, int, int, unsigned, RBWeights &,
                       const std::function<int(const std::vector<unsigned short> &, unsigned short **)> &) { return PMP_OK; }
struct AblBench {};                                                                    // pmp_debug_conv_bench: extra legs of the measurement library
inline void abl_bench_prepare(pmp_ctx *, AblBench &, const float *, int, int, int, bool, ConvX6Args &) {}
inline void abl_bench_report(pmp_ctx *, AblBench &, bool, int, int, int, int, int, ConvX6Args &, const std::function<hipError_t()> &) {}
inline void abl_bench_free(AblBench &) {}
}  // namespace pmp
