// abl/abl_types.h — MEASUREMENT library (`make abl`, -Iabl -DPMP_ABLATION): the members the product's empty structs (hooks/abl_types.h) stand for.
#pragma once
#include <stddef.h>
namespace pmp {
struct AblConvArgs {
    unsigned long long *dbg = nullptr;   // diagnostic builds: in-kernel stamps per workgroup
    const void *zeros = nullptr;         // >= 16 zero bytes in device memory: source of out-of-image halo pieces of the LDS-DMA forms
    const unsigned short *w_wx = nullptr; float wx_out_scale = 0.f;   // 3x3 64->64 only: Winograd-x weight stream (pack_h2_wx) and its 1/S
};
struct AblRB {
    unsigned short *w0w = nullptr, *w2w = nullptr; int k0w = 0, k2w = 0;   // f16x3, 3x3 64->64 blocks: Winograd-x streams (conv_f16x3_wx.hip)
};
struct AblCtx {
    int winograd = 0;                    // run the 3x3 64->64 convolutions in the Winograd-x form (pmp_debug_set_winograd)
};
}  // namespace pmp
