// hooks/abl_types.h — PRODUCT build: the measurement library's extra state is empty.  `make -C tools/abl` compiles the same sources with -Itools/abl, where
// a header of this name defines the real members (Winograd-x weight streams, diagnostic stamp buffer, ...).  The product library contains no
// measurement code and no kernel selector; this header and abl_hooks.h (no-op inlines) are the only trace of the seam.
#pragma once
namespace pmp {
struct AblConvArgs {};   // extra kernel arguments of the measurement library's convolution forms
struct AblRB {};         // extra per-ResidualBlock weight streams
struct AblCtx {};        // extra per-context switches
}  // namespace pmp
