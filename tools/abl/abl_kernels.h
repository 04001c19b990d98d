// abl/abl_kernels.h — MEASUREMENT library: kernel-level declarations of the forms that are not in the product.
#pragma once
#include "pmp_kernels.h"
namespace pmp {
extern int g_conv_variant;   // process-wide form selector of the convolution kernels (pmp_debug_set_conv_variant, PMP_CONV_VARIANT); defined in abl/conv_mfma.hip
// conv_f16x3_t32.hip: the 3x3 64->64 trunk convolution on 32x16 tiles (LDS-DMA halo, hand-counted vmcnt)
bool conv_h2_t32_applicable(const ConvX6Args &a);
hipError_t launch_conv_h2_t32(hipStream_t s, const ConvX6Args &a);
// conv_f16x3_wx.hip: the 3x3 64->64 convolution with a 1-D Winograd F(2,3) transform along x (1.5x fewer MFMAs)
bool conv_h2_wx_applicable(const ConvX6Args &a);
hipError_t launch_conv_h2_wx(hipStream_t s, const ConvX6Args &a);
}  // namespace pmp
