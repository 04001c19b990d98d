// Declarations shared by the host-only translation units (host_emit.cpp, pack.cpp) and the rest of the library.
// Nothing here needs the HIP runtime: these files also build into the CPU-only sanitizer test library (make hostasan).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pmp.h"

namespace pmp {

// the calling thread's context-less error string (pmp_last_error(NULL)); returns `code`
int set_err_global(int code, const std::string &msg);
const char *global_err();

// pack.cpp: OIHW fp32 weights -> fragment-order streams
std::vector<float> pack_mfma(const float *w, int cout, int cin, int kh, int kw, int cout_pad, int cin_pad);
std::vector<unsigned short> pack_x6(const float *w, int cout, int cin, int kh, int kw, int cout_pad, int cin_pad);
std::vector<unsigned short> pack_h2(const float *w, int cout, int cin, int kh, int kw, int cout_pad, int cin_pad, int scale_exp);
int h2_scale_exp(const float *w, size_t n);
void h2_split8(const float *v, unsigned short *h0, unsigned short *h1);   // 8 values -> their two fp16 terms (bit patterns), as pack_h2 splits them
std::vector<unsigned short> pack_stem_h2(const float *w32, int cin, int k1, int scale_exp);
std::vector<float> pack_plain(const float *w, int cout, int cin, int kh, int kw);


// pmpw_file.cpp: the product's weight container (.pmpw)
struct WeightTensor { std::string name; int ndim = 0; int shape[4] = {0, 0, 0, 0}; long long offset = 0; };
struct WeightFile { std::string net; int qp = -1; std::vector<WeightTensor> tensors; std::vector<float> payload;
                    std::vector<int> act_exp;      // optional manifest key "act_exp": the five f16x3 activation-scale exponents of an MTT net (include/pmp.h)
                    bool have_fp = false; uint64_t act_mtt_fp = 0, act_qt_fp = 0; };   // optional "act_fp": fingerprints of the tensors the exponents were calibrated ON (this file's, its QT partner's)
int read_pmpw(const char *path, WeightFile &wf);
constexpr int PMP_ACT_EXP_MAX = 30;        // largest exponent of a trunk segment (0, 2, 4): activations up to 2^42 = 4e12 stay on the datapath, beyond that the range guard takes over
constexpr int PMP_ACT_EXP_ATT_MAX = 6;     // largest exponent of an attention segment (1, 3): its input, built from O(1) logits, must stay out of fp16's subnormals
// order-independent-of-layout fingerprint of a set of tensors (names, shapes, float bits; sorted by name) - what ties "act_exp" to its nets
uint64_t fingerprint_tensors(const float *blob, const pmp_tensor_desc *descs, int ndesc);
int net_id_of(const std::string &net);

}  // namespace pmp
