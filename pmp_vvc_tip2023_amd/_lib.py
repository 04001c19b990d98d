"""ctypes binding of libpmp_hip.so (C ABI: include/pmp.h).  No torch types cross this boundary: only raw
pointers and sizes.  The library is built in-tree by `make -C pmp_vvc_tip2023_amd/csrc` (see __graft_entry__.build)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpmp_hip.so")
HOSTASAN_LIB_PATH = os.path.join(_HERE, "libpmp_hostasan.so")      # make hostasan: host-only units under ASan/UBSan (tests only)

PMP_LUMA, PMP_CHROMA = 0, 1
PMP_RECORD_BYTES = 1344
NET_IDS = {"Luma_Q": 0, "Luma_MSBD": 1, "Chroma_Q": 2, "Chroma_MSBD": 3}
ERRORS = {-1: "PMP_E_INVALID", -2: "PMP_E_HIP", -3: "PMP_E_NOWEIGHTS", -4: "PMP_E_IO", -5: "PMP_E_NOMEM", -6: "PMP_E_NODEVICE", -7: "PMP_E_RANGE"}


class PmpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "PMP_E_?"), code, msg))
        self.code = code


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ndim", C.c_int), ("shape", C.c_int * 4), ("offset", C.c_int64)]


# name -> (restype, argtypes); every symbol declared in include/pmp.h
_VP, _I, _I64, _U32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32
SIGNATURES = {
    "pmp_version": (C.c_char_p, []),
    "pmp_last_error": (C.c_char_p, [_VP]),
    "pmp_create": (_I, [_I, C.POINTER(_VP)]),
    "pmp_destroy": (_I, [_VP]),
    "pmp_trim": (_I, []),
    "pmp_set_stream": (_I, [_VP, _VP]),
    "pmp_synchronize": (_I, [_VP]),
    "pmp_set_chunk": (_I, [_VP, _I]),
    "pmp_set_overlap": (_I, [_VP, _I]),
    "pmp_get_workspace_bytes": (_I64, [_VP]),
    "pmp_set_precision": (_I, [_VP, _I]),
    "pmp_get_precision": (_I, [_VP]),
    "pmp_set_saturation_policy": (_I, [_VP, _I]),
    "pmp_get_saturation": (_I, [_VP]),
    "pmp_get_saturation_reruns": (_I64, [_VP]),
    "pmp_clear_saturation": (_I, [_VP]),
    "pmp_load_weights": (_I, [_VP, _I, _I, _VP, C.POINTER(TensorDesc), _I]),
    "pmp_load_weights_file": (_I, [_VP, _I, _I, C.c_char_p]),
    "pmp_has_weights": (_I, [_VP, _I, _I]),
    "pmp_weights_fingerprint": (_I, [_VP, _I, _I, C.POINTER(C.c_uint64)]),
    "pmp_fingerprint_tensors": (_I, [_VP, C.POINTER(TensorDesc), _I, C.POINTER(C.c_uint64)]),
    "pmp_debug_read_weights_file": (_I, [C.c_char_p, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I64), C.POINTER(C.c_double)]),
    "pmp_infer": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I64, _VP, _VP, _VP]),
    "pmp_infer_device": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I64, _VP, _VP, _VP]),
    "pmp_postprocess": (_I, [_VP, _I, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP]),
    "pmp_postprocess_device": (_I, [_VP, _I, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP]),
    "pmp_infer_postprocess": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pmp_infer_postprocess_device": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pmp_postprocess_records_device": (_I, [_VP, _I, _VP, _VP, _VP, _I64, _VP]),
    "pmp_infer_postprocess_records_device": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I64, _VP]),
    "pmp_cut_blocks": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP]),
    "pmp_cut_blocks_device": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP]),
    "pmp_write_partition_file": (_I, [C.c_char_p, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "pmp_write_partition_binary": (_I, [C.c_char_p, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "pmp_tile_partition_maps": (_I, [_I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pmp_format_partition_text": (_I64, [_I, _I, _I, _VP, _VP, _VP, _VP, _VP, _I64]),
    "pmp_format_partition_rows": (_I64, [_I, _I, _VP, _VP, _VP, _VP, _VP, _I64, _VP]),
    "pmp_format_partition_rows_records": (_I64, [_I, _I, _VP, _VP, _I64, _VP]),
    "pmp_tile_partition_rows_records": (_I, [_I, _I, _VP, _VP, _VP, _VP, _VP]),
    "pmp_ktime_enable": (_I, [_VP, _U32]),
    "pmp_ktime_classes": (_I, []),
    "pmp_ktime_name": (C.c_char_p, [_I]),
    "pmp_ktime_get": (_I, [_VP, _I, C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pmp_debug_set_conv_variant": (_I, [_I]),
    "pmp_debug_set_winograd": (_I, [_VP, _I]),
    "pmp_debug_set_fusion": (_I, [_VP, _I]),
    "pmp_debug_set_activation_scales": (_I, [_VP, _I]),
    "pmp_debug_activation_report": (_I, [_VP, _I, _I, C.POINTER(_I), C.POINTER(C.c_float), C.c_char_p, _I64]),
    "pmp_debug_pack_f16x3": (C.c_int64, [C.POINTER(C.c_float), _I, _I, _I, C.POINTER(C.c_uint16), C.c_int64, C.POINTER(C.c_int)]),
    "pmp_debug_conv_bench": (_I, [_VP, _I, _I, _I, _I, _I, _I, _I] + [C.POINTER(C.c_double)] * 4),
}

_lib = None


def open_library(path, subset=False):
    """dlopen a build of the C ABI and type its entry points.  subset=True accepts a library that exports only some of
    include/pmp.h (the host-only sanitizer build); the product library must export all of it."""
    if not os.path.isfile(path):
        raise ImportError("%s is missing - build it with `make -C %s` (or __graft_entry__.build())"
                          % (path, os.path.join(_HERE, "csrc")))
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        if subset and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def _torch_first():
    """The PyTorch-ROCm wheel bundles its own libamdhip64 with the same soname as /opt/rocm's, so whichever copy is loaded first
    serves the whole process.  With ours first, torch finds "No HIP GPUs" (seen on the GPU pool: the driver then silently kept its
    blocks on the host).  Where torch is installed - it is the device allocator of the driver, bench and tests - it goes first."""
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass


def load(path=None):
    """Load libpmp_hip.so (or, for tools/ and tests, the build at `path` - before anything else loaded the library).
    Fails loudly when the extension has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        _torch_first()
        _lib = open_library(path or LIB_PATH)
    elif path is not None and os.path.abspath(path) != os.path.abspath(_lib._name):
        raise RuntimeError("a different build of the library is already loaded: %s" % _lib._name)
    return _lib


def check(rc, ctx=None):
    if rc < 0:
        msg = load().pmp_last_error(ctx)
        raise PmpError(int(rc), msg.decode() if msg else "")
    return rc
