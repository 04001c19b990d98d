// hostasan_stub.cpp — the two context entry points the host-only sanitizer library (make hostasan) needs beside
// host_emit.cpp and pack.cpp.  There is no context in that library: pmp_last_error(NULL) is the only valid call.
#include "pmp_hostonly.h"

extern "C" {

const char *pmp_version(void) { return "pmp-hip hostasan (host-only translation units, -fsanitize=address,undefined)"; }

const char *pmp_last_error(const pmp_ctx *ctx) { return ctx ? "hostasan build: no contexts" : pmp::global_err(); }

}  // extern "C"
