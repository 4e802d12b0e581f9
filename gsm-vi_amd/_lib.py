"""ctypes binding of libgsmvi_hip.so (C ABI: include/gsmvi_hip.h).  Fails loudly when missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class GsmviError(RuntimeError):
    """A C-ABI call returned a non-zero gsmvi_status."""

    def __init__(self, fn, status, msg):
        super().__init__(f"{fn} failed with status {status}: {msg}")
        self.status = status


def library_path(debug=None):
    """The product library; with GSMVI_HIP_DEBUG_LIB=1 (diagnostic scripts) the debug build of the same objects, which also
    exports include/gsmvi_hip_debug.h."""
    variant = os.environ.get("GSMVI_HIP_LIB_VARIANT", "")
    if debug is None and variant:          # diagnostic builds of csrc/Makefile (e.g. `make oldwb`), debug export list
        return os.path.join(_HERE, f"libgsmvi_hip_{variant}.so")
    if debug is None:
        debug = os.environ.get("GSMVI_HIP_DEBUG_LIB", "0") not in ("", "0")
    return os.path.join(_HERE, "libgsmvi_hip_debug.so" if debug else "libgsmvi_hip.so")


_c_dp = C.c_void_p   # device pointers travel as integers
_SIGS = {
    # name: (restype, argtypes)
    "gsmvi_abi_version": (C.c_int, []),
    "gsmvi_status_string": (C.c_char_p, [C.c_int]),
    "gsmvi_last_error": (C.c_char_p, []),
    "gsmvi_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gsmvi_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "gsmvi_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "gsmvi_destroy": (C.c_int, [C.c_void_p]),
    "gsmvi_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "gsmvi_gsm_update_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, C.c_int,
                                       _c_dp, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int]),
    "gsmvi_gsm_update_general_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, C.c_int,
                                               _c_dp, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int]),
    "gsmvi_gsm_record_len": (C.c_int, [C.c_int]),
    "gsmvi_gsm_local_stage_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp,
                                            C.c_int, _c_dp, _c_dp, C.c_int, _c_dp, C.c_int]),
    "gsmvi_gsm_apply_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, _c_dp,
                                      C.c_int, _c_dp, _c_dp, C.c_int]),
    "gsmvi_gsm_update_sharded_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int,
                                               _c_dp, C.c_int, _c_dp, _c_dp, C.c_int, _c_dp, _c_dp, _c_dp, C.c_int]),
    "gsmvi_set_rccl_library": (C.c_int, [C.c_void_p]),
    "gsmvi_gsm_factor_update_sharded_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int,
                                                      _c_dp, C.c_int, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int, _c_dp, _c_dp,
                                                      _c_dp, C.c_int, _c_dp, _c_dp]),
    "gsmvi_bam_update_sharded_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp,
                                               C.c_int, _c_dp, _c_dp, C.c_int, C.c_double, C.c_double, _c_dp, _c_dp, _c_dp,
                                               C.c_int, _c_dp]),
    "gsmvi_gsm_rows_stage_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp,
                                           C.c_int, _c_dp, C.c_int]),
    "gsmvi_gsm_records_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, C.c_int,
                                        _c_dp, _c_dp, _c_dp, C.c_int]),
    "gsmvi_gsm_apply_rows_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _c_dp,
                                           C.c_int, _c_dp, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int]),
    "gsmvi_gsm_factor_update_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, C.c_int,
                                              _c_dp, C.c_int, _c_dp, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int, _c_dp,
                                              _c_dp]),
    "gsmvi_gsm_factor_local_stage_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp,
                                                   C.c_int, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int, _c_dp, C.c_int]),
    "gsmvi_gsm_factor_apply_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, C.c_int,
                                             _c_dp, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int, _c_dp, _c_dp]),
    "gsmvi_randn_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, _c_dp, _c_dp]),
    "gsmvi_randn_batch_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int64, _c_dp, _c_dp,
                                        _c_dp]),
    "gsmvi_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "gsmvi_get_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "gsmvi_last_path": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.c_int]),
    "gsmvi_bam_set_reg_source": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gsmvi_gaussian_score_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp,
                                           _c_dp, C.c_int, _c_dp, C.c_int]),
    "gsmvi_potrf_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_dp, C.c_int, _c_dp, C.c_int, _c_dp]),
    "gsmvi_gram_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_dp, C.c_int, _c_dp, C.c_int]),
    "gsmvi_gram_shift_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_dp, C.c_int, C.c_double, _c_dp, _c_dp, C.c_int]),
    "gsmvi_whiten_rows_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, C.c_int,
                                        _c_dp, _c_dp, C.c_int, _c_dp]),
    "gsmvi_sample_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, _c_dp,
                                   C.c_int, _c_dp, C.c_int]),
    "gsmvi_sample_cols_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int,
                                        _c_dp, C.c_int]),
    "gsmvi_gsm_factor_apply_cols_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _c_dp, C.c_int,
                                                  _c_dp, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int, _c_dp,
                                                  _c_dp]),
    "gsmvi_commit_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_dp, _c_dp, _c_dp, C.c_int, _c_dp, _c_dp,
                                   C.c_int, _c_dp]),
    "gsmvi_bam_update_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, C.c_int,
                                       _c_dp, _c_dp, C.c_int, C.c_double, C.c_double, _c_dp, _c_dp, C.c_int,
                                       _c_dp]),
    "gsmvi_bam_factor_update_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int, _c_dp, C.c_int,
                                              _c_dp, C.c_int, _c_dp, _c_dp, C.c_int, C.c_double, _c_dp, _c_dp, C.c_int,
                                              _c_dp, _c_dp]),
    "gsmvi_bam_factor_update_sharded_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_dp, C.c_int,
                                                      _c_dp, C.c_int, _c_dp, C.c_int, _c_dp, _c_dp, C.c_int, C.c_double,
                                                      _c_dp, _c_dp, _c_dp, C.c_int, _c_dp, _c_dp]),
}


# include/gsmvi_hip_debug.h: exported by libgsmvi_hip_debug.so only (diagnostic scripts; GSMVI_HIP_DEBUG_LIB=1)
_DEBUG_SIGS = {
    "gsmvi_debug_read_workspace": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_double), C.c_size_t]),
    "gsmvi_debug_workspace_ptr": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "gsmvi_debug_chol128": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _c_dp, _c_dp, _c_dp, _c_dp]),
    "gsmvi_debug_read_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]),
    "gsmvi_debug_stream_copy_f64": (C.c_int, [C.c_void_p, _c_dp, _c_dp, C.c_size_t]),
}


def exported_symbols(debug=False):
    """Names the library exports: exactly include/gsmvi_hip.h for the product library; the debug build adds
    include/gsmvi_hip_debug.h (csrc/exports.map / exports_debug.map are the linker's copies of these lists)."""
    return sorted(list(_SIGS) + (list(_DEBUG_SIGS) if debug else []))


def debug_build_selected():
    return os.environ.get("GSMVI_HIP_DEBUG_LIB", "0") not in ("", "0") or bool(os.environ.get("GSMVI_HIP_LIB_VARIANT", ""))


def load_library():
    """Loads the HIP library once.  Raises ImportError with build instructions if it is absent:
    the product path has no CPU fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built.  Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C gsm-vi_amd/csrc`. "
            "gsmvi_amd has no CPU fallback.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    sigs = dict(_SIGS)
    if debug_build_selected():
        sigs.update(_DEBUG_SIGS)
    variant = bool(os.environ.get("GSMVI_HIP_LIB_VARIANT", ""))
    for name, (res, args) in sigs.items():
        try:
            fn = getattr(lib, name)     # AttributeError here = ABI mismatch, surface it
        except AttributeError:
            if variant:                 # a diagnostic build of an OLDER tree (A/B runs against a previous round's library)
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if lib.gsmvi_abi_version() != 1:
        raise ImportError(f"{path}: ABI version {lib.gsmvi_abi_version()} != 1")
    _LIB = lib
    return lib


def check(fn_name, status):
    if status != 0:
        lib = load_library()
        raise GsmviError(fn_name, status, (lib.gsmvi_last_error() or b"").decode() or
                         lib.gsmvi_status_string(status).decode())
