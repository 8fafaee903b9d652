"""ctypes binding of libpypwt_amd.so (the C ABI declared in include/pypwt_amd.h).

There is NO CPU fallback: if the shared library is missing this module raises, and if no
HIP device is present every plan creation fails with the library's error message.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpypwt_amd.so")          # fp32 build (pdwt_real = float)
LIB_PATH_F64 = os.path.join(HERE, "libpypwt_amd_f64.so")  # fp64 build (-DPDWT_DOUBLE, pdwt_real = double)
# test-only fp32 build that also holds the experiment kernels the product does not ship (-DPDWT_LAB_KERNELS, build.py)
LIB_PATH_LAB = os.path.join(HERE, "libpypwt_amd_lab.so")
_f32_alias = "f32"


def use_lab_kernels(on=True):
    """Tests of the experiment kernels: every later request for the fp32 library (``load()``, ``Wavelets``,
    ``BatchedWavelets``) is served by libpypwt_amd_lab.so.  Returns the previous setting."""
    global _f32_alias
    prev = _f32_alias == "lab"
    _f32_alias = "lab" if on else "f32"
    return prev

f32p = C.POINTER(C.c_float)
handle_t = C.c_void_p


class PdwtInfo(C.Structure):
    """struct pdwt_info == the reference's w_info (pdwt/src/utils.h:9-19)."""
    _fields_ = [("ndims", C.c_int), ("Nr", C.c_int), ("Nc", C.c_int), ("nlevels", C.c_int),
                ("do_swt", C.c_int), ("hlen", C.c_int)]


# status codes (enum pdwt_status)
OK, ERR_ARG, ERR_WAVELET, ERR_HIP, ERR_STATE, ERR_FILTER_LEN, ERR_MISMATCH, ERR_UNSUPPORTED, ERR_NOMEM = \
    0, -1, -2, -3, -4, -5, -6, -7, -8
# states (enum pdwt_state == w_state, pdwt/src/wt.h:8-17)
STATE_INIT, STATE_FORWARD, STATE_INVERSE, STATE_THRESHOLD = 0, 1, 2, 3

def signatures(real=C.c_float):
    """name -> (restype, argtypes) of every symbol include/pypwt_amd.h declares, for pdwt_real = `real`."""
    realp = C.POINTER(real)
    return {
        "pdwt_create": (C.c_int, [realp, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.POINTER(handle_t)]),
        "pdwt_create_batched": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(handle_t)]),
        "pdwt_clone": (C.c_int, [handle_t, C.POINTER(handle_t)]),
        "pdwt_destroy": (C.c_int, [handle_t]),
        "pdwt_forward": (C.c_int, [handle_t]),
        "pdwt_inverse": (C.c_int, [handle_t]),
        "pdwt_soft_threshold": (C.c_int, [handle_t, real, C.c_int, C.c_int]),
        "pdwt_hard_threshold": (C.c_int, [handle_t, real, C.c_int, C.c_int]),
        "pdwt_group_soft_threshold": (C.c_int, [handle_t, real, C.c_int, C.c_int]),
        "pdwt_shrink": (C.c_int, [handle_t, real, C.c_int]),
        "pdwt_proj_linf": (C.c_int, [handle_t, real, C.c_int]),
        "pdwt_circshift": (C.c_int, [handle_t, C.c_int, C.c_int, C.c_int]),
        "pdwt_norm1": (C.c_int, [handle_t, realp]),
        "pdwt_norm2sq": (C.c_int, [handle_t, realp]),
        "pdwt_norms_async": (C.c_int, [handle_t, C.c_void_p]),
        "pdwt_norms_slot": (C.c_int, [handle_t, C.POINTER(C.c_void_p)]),
        "pdwt_soft_threshold_norms_async": (C.c_int, [handle_t, real, C.c_int, C.c_int, C.c_void_p]),
        "pdwt_add_wavelet": (C.c_int, [handle_t, handle_t, real]),
        "pdwt_get_image": (C.c_longlong, [handle_t, C.c_void_p]),
        "pdwt_get_coeff": (C.c_longlong, [handle_t, C.c_void_p, C.c_int]),
        "pdwt_get_image_at": (C.c_longlong, [handle_t, C.c_void_p, C.c_int]),
        "pdwt_get_coeff_at": (C.c_longlong, [handle_t, C.c_void_p, C.c_int, C.c_int]),
        "pdwt_set_image": (C.c_int, [handle_t, C.c_void_p, C.c_int]),
        "pdwt_set_coeff": (C.c_int, [handle_t, C.c_void_p, C.c_int, C.c_int]),
        "pdwt_coeff_count": (C.c_longlong, [handle_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "pdwt_coeff_region": (C.c_longlong, [handle_t, C.POINTER(C.c_longlong), C.c_int]),
        "pdwt_get_coeff_region": (C.c_longlong, [handle_t, C.c_void_p]),
        "pdwt_image_ptr": (C.c_ssize_t, [handle_t]),
        "pdwt_bind_image": (C.c_int, [handle_t, C.c_void_p]),
        "pdwt_copy": (C.c_int, [handle_t, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int]),
        "pdwt_coeff_ptr": (C.c_ssize_t, [handle_t, C.c_int]),
        "pdwt_set_filters_forward": (C.c_int, [handle_t, C.c_char_p, C.c_uint, realp, realp, realp, realp]),
        "pdwt_set_filters_inverse": (C.c_int, [handle_t, realp, realp, realp, realp]),
        "pdwt_get_info": (C.c_int, [handle_t, C.POINTER(PdwtInfo), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "pdwt_print_info": (C.c_int, [handle_t]),
        "pdwt_info_string": (C.c_int, [handle_t, C.c_char_p, C.c_size_t]),
        "pdwt_schedule_string": (C.c_int, [handle_t, C.c_char_p, C.c_size_t]),
        "pdwt_trim_pool": (C.c_int, []),
        "pdwt_current_shift": (C.c_int, [handle_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "pdwt_last_error": (C.c_char_p, []),
        "pdwt_version": (C.c_char_p, []),
        "pdwt_wavelet_count": (C.c_int, []),
        "pdwt_wavelet_name": (C.c_char_p, [C.c_int]),
        "pdwt_wavelet_filters": (C.c_int, [C.c_char_p, realp, C.c_int]),
        "pdwt_synchronize": (C.c_int, [handle_t]),
        "pdwt_set_stream": (C.c_int, [handle_t, C.c_void_p]),
        "pdwt_get_stream": (C.c_void_p, [handle_t]),
        "pdwt_device": (C.c_int, [handle_t]),
        "pdwt_device_count": (C.c_int, []),
        "pdwt_wait_for_stream": (C.c_int, [handle_t, C.c_void_p]),
        "pdwt_sync_producer": (C.c_int, [C.c_int, C.c_void_p, C.c_int]),
        "pdwt_device_of_pointer": (C.c_int, [C.c_void_p]),
        "pdwt_fill_image_hash": (C.c_int, [handle_t, C.c_uint32, real, C.c_longlong]),
        "pdwt_enable_kernel_timing": (C.c_int, [handle_t, C.c_int]),
        "pdwt_kernel_times": (C.c_int, [handle_t, f32p, C.c_void_p, C.c_int]),
        "pdwt_kernel_families": (C.c_int, [handle_t, C.c_void_p, C.c_int]),
        "pdwt_reset_kernel_times": (C.c_int, [handle_t]),
        "pdwt_time_level": (C.c_int, [handle_t, C.c_int, C.c_int, C.c_int, f32p]),
        "pdwt_time_copy": (C.c_int, [handle_t, C.c_longlong, C.c_int, f32p]),
        "pdwt_copy_capacity": (C.c_longlong, [handle_t]),
        "pdwt_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
        "pdwt_comm_unique_id": (C.c_int, [C.c_void_p]),
        "pdwt_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(handle_t)]),
        "pdwt_comm_destroy": (C.c_int, [handle_t]),
        "pdwt_comm_rank": (C.c_int, [handle_t]),
        "pdwt_comm_size": (C.c_int, [handle_t]),
        "pdwt_comm_exchange": (C.c_int, [handle_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong), C.POINTER(C.c_int),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.c_void_p]),
        "pdwt_comm_all_gather": (C.c_int, [handle_t, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
        "pdwt_comm_broadcast": (C.c_int, [handle_t, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
        "pdwt_comm_last_error": (C.c_char_p, []),
    }


SIGNATURES = signatures()  # the fp32 library's

_libs = {}


def load(variant="f32"):
    """Load libpypwt_amd.so ("f32") or libpypwt_amd_f64.so ("f64") and declare every prototype.
    Raises if the library is absent."""
    if variant == "f32":
        variant = _f32_alias
    if variant in _libs:
        return _libs[variant]
    path, real = {"f32": (LIB_PATH, C.c_float), "f64": (LIB_PATH_F64, C.c_double), "lab": (LIB_PATH_LAB, C.c_float)}[variant]
    if variant == "f32" and os.environ.get("PDWT_LIB_F32"):  # A/B measurements: another build of the fp32 library
        path = os.environ["PDWT_LIB_F32"]
    if variant == "f64" and os.environ.get("PDWT_LIB_F64"):  # ... of the fp64 library
        path = os.environ["PDWT_LIB_F64"]
    if not os.path.exists(path):
        raise ImportError(
            "pypwt_amd: %s is missing. Build it with `python -m pypwt_amd.build` (needs hipcc). "
            "There is no CPU implementation to fall back to." % path)
    lib = C.CDLL(path)
    for name, (res, args) in signatures(real).items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header drift apart
        fn.restype = res
        fn.argtypes = args
    lib.pdwt_real = real
    _libs[variant] = lib
    return lib


def last_error(lib=None):
    """Message of the last failure on this thread in `lib` (default: the fp32 library)."""
    msg = (lib or load()).pdwt_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


class PdwtError(RuntimeError):
    pass


def check(rc, what="", lib=None):
    """Map a negative pdwt_status to a Python exception (reference behaviour: the Cython shim raises
    ValueError on its own checks and RuntimeError on count mismatches, src/pypwt.pyx:230-234,284-285)."""
    if rc >= 0:
        return rc
    msg = "%s%s" % ((what + ": ") if what else "", last_error(lib))
    if rc in (ERR_WAVELET, ERR_ARG, ERR_FILTER_LEN, ERR_MISMATCH):
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise PdwtError(msg)
