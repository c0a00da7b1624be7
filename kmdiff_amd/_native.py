"""ctypes binding of libkmdiff_hip.so (include/kmdiff_hip.h).

There is no CPU fallback: importing the package works without a GPU (so that the symbol
table can be checked), but every compute call goes through the HIP library and raises
KmdError when the library or a device is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KMD_LIB") or os.path.join(_HERE, "lib", "libkmdiff_hip.so")

KMD_OK = 0
ABI_VERSION = 3           # KMD_ABI_VERSION of include/kmdiff_hip.h (tests/test_abi.py holds the two together)
KMD_E_INVALID = -1
KMD_E_OVERFLOW = -4
SIGN_CONTROL, SIGN_CASE, SIGN_NO = 0, 1, 2
CORR_NOTHING, CORR_BONFERRONI, CORR_BENJAMINI, CORR_SIDAK, CORR_HOLM = 0, 1, 2, 3, 4
LAYOUT_ROWS, LAYOUT_SOA, LAYOUT_TILED = 0, 1, 2
NCOUNTERS = 8
(CNT_TOTAL, CNT_SIG, CNT_SIG_CONTROL, CNT_SIG_CASE, CNT_CANDIDATES, CNT_DEFERRED, CNT_NEAR_THRESHOLD,
 CNT_NEAR_UNRESOLVED) = range(8)


class KmdError(RuntimeError):
    pass


class Survivors(C.Structure):
    _fields_ = [("d_row", C.c_void_p), ("d_kmer_lo", C.c_void_p), ("d_kmer_hi", C.c_void_p),
                ("d_pvalue", C.c_void_p), ("d_sign", C.c_void_p), ("d_mean_control", C.c_void_p),
                ("d_mean_case", C.c_void_p), ("capacity", C.c_size_t)]


class Tile(C.Structure):
    _fields_ = [("d_counts", C.c_void_p), ("count_bytes", C.c_int), ("layout", C.c_int),
                ("ld", C.c_size_t), ("d_kmer_lo", C.c_void_p), ("d_kmer_hi", C.c_void_p),
                ("n_rows", C.c_size_t), ("row_base", C.c_uint64)]


# kmd_transport (include/kmdiff_hip.h): the wire of kmd_correct_sharded -- two collectives over device buffers
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ABORT_FN = C.CFUNCTYPE(None, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("world", C.c_int),
                ("allreduce_u64", ALLREDUCE_FN), ("allgather", ALLGATHER_FN), ("abort", ABORT_FN)]


# name -> (restype, argtypes); the list mirrors include/kmdiff_hip.h and is what
# tests/test_abi.py checks against the header and the built library.
_vp, _sz, _u64, _i, _d = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_double
# include/kmdiff_hip_test.h: test hooks, not part of the interface a host binds
TEST_SIGNATURES = {
    "kmd_test_log_rounded": (_d, [_d]),
    "kmd_test_exp_rounded": (_d, [_d]),
    "kmd_test_igamc_half_rounded": (_d, [_d]),
    "kmd_test_row_pvalue_rounded": (_d, [_vp, _u64, _u64]),
    "kmd_test_running_sums": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "kmd_test_popstrat_linear": (_i, [_i, _vp, _vp, _vp, _vp]),
    "kmd_test_popstrat_sigmoid": (_i, [_vp, _sz, _vp]),
    "kmd_test_popstrat_predict": (_i, [_vp, _vp, _i, C.POINTER(_d), C.POINTER(_d)]),
    "kmd_test_popstrat_irls": (_i, [_vp, _vp, _i, _i, _i, _vp, C.POINTER(_i), _vp, C.POINTER(_i)]),
}

SIGNATURES = {
    "kmd_status_string": (C.c_char_p, [_i]),
    "kmd_last_error": (C.c_char_p, []),
    "kmd_abi_version": (_i, []),
    "kmd_device_count": (_i, [C.POINTER(_i)]),
    "kmd_set_device": (_i, [_i]),
    "kmd_device_name": (_i, [C.c_char_p, _sz]),
    "kmd_malloc": (_i, [C.POINTER(_vp), _sz]),
    "kmd_free": (_i, [_vp]),
    "kmd_malloc_host": (_i, [C.POINTER(_vp), _sz]),
    "kmd_free_host": (_i, [_vp]),
    "kmd_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "kmd_memcpy_h2d_async": (_i, [_vp, _vp, _sz, _vp]),
    "kmd_stream_create": (_i, [C.POINTER(_vp)]),
    "kmd_stream_destroy": (_i, [_vp]),
    "kmd_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "kmd_memset": (_i, [_vp, _i, _sz, _vp]),
    "kmd_stream_sync": (_i, [_vp]),
    "kmd_release_cache": (_i, []),
    "kmd_event_create": (_i, [C.POINTER(_vp)]),
    "kmd_event_destroy": (_i, [_vp]),
    "kmd_event_record": (_i, [_vp, _vp]),
    "kmd_stream_wait_event": (_i, [_vp, _vp]),
    "kmd_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "kmd_model_create": (_i, [C.POINTER(_vp), _i, _i, _vp, _vp, _sz]),
    "kmd_model_destroy": (_i, [_vp]),
    "kmd_model_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_u64), C.POINTER(_u64),
                            C.POINTER(_sz)]),
    "kmd_model_lf_table": (_i, [_vp, _vp, _sz]),
    "kmd_poisson_filter": (_i, [_vp, C.POINTER(Tile), _d, C.POINTER(Survivors), _vp, _vp]),
    "kmd_poisson_process": (_i, [_vp, C.POINTER(Tile), _vp, _vp, _vp, _vp, _vp]),
    "kmd_survivors_sort_by_row": (_i, [C.POINTER(Survivors), _sz, _vp]),
    "kmd_survivors_gather_counts": (_i, [C.POINTER(Tile), _i, _vp, _sz, _vp, _vp]),
    "kmd_correct": (_i, [_i, _d, _u64, _vp, _vp, _sz, _vp, C.POINTER(_u64), C.POINTER(_u64),
                         C.POINTER(_u64), _vp]),
    "kmd_pvalue_histogram": (_i, [_vp, _sz, _vp, _vp]),
    "kmd_correct_critical_bin": (_i, [_i, _d, _u64, _vp, C.POINTER(C.c_uint32), C.POINTER(_u64), _vp]),
    "kmd_correct_from_rank": (_i, [_i, _d, _u64, _u64, _vp, _vp, _sz, _vp, C.POINTER(_u64), C.POINTER(_u64),
                                   C.POINTER(_u64), _vp]),
    "kmd_correct_sharded": (_i, [C.POINTER(Transport), _i, _d, _vp, _vp, _vp, _vp, _sz, _vp, C.POINTER(_u64), C.POINTER(_u64),
                                 C.POINTER(_u64), _vp]),
    "kmd_transport_local_create": (_i, [_i, C.POINTER(Transport)]),
    "kmd_transport_local_destroy": (_i, [_i, C.POINTER(Transport)]),
    "kmd_transport_abort": (_i, [C.POINTER(Transport)]),
    "kmd_pack_block_bound": (_sz, []),
    "kmd_pack_block": (_sz, [_vp, _vp, C.c_uint32, _vp]),
    "kmd_pack_records": (_sz, [_vp, C.c_uint32, C.c_uint32, _vp]),
    "kmd_pack_stream": (_sz, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "kmd_unpack_streams": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kmd_merge_partition": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _sz, _sz, _vp, _vp, _vp, C.POINTER(_u64), _vp]),
    "kmd_merge_sums": (_i, [_i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, C.POINTER(_u64), _vp]),
    "kmd_merge_filter": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _d, C.POINTER(Survivors), _vp, C.POINTER(_u64), _vp]),
    "kmd_merge_filter_batch": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _d, C.POINTER(Survivors), _vp, C.POINTER(_u64), _vp]),
    "kmd_survivors_gather_counts_streams": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "kmd_survivors_sort_by_kmer": (_i, [C.POINTER(Survivors), _sz, _vp]),
    "kmd_pvalues_refine": (_i, [_vp, _sz, _vp, _vp, _vp, _vp]),
    "kmd_poisson_filter_sums": (_i, [_vp, _vp, _vp, _vp, _sz, _d, C.POINTER(Survivors), _vp, _vp]),
    "kmd_popstrat_create": (_i, [C.POINTER(_vp), _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i]),
    "kmd_popstrat_destroy": (_i, [_vp]),
    "kmd_popstrat_set_epsilon": (_i, [_vp, _d]),
    "kmd_popstrat_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), _vp, _vp, C.POINTER(_d)]),
    "kmd_popstrat_apply": (_i, [_vp, _vp, _i, _sz, _sz, _vp, _vp]),
    "kmd_pca_create": (_i, [C.POINTER(C.c_void_p), _i, C.c_double, _u64, _i, _sz]),
    "kmd_pca_destroy": (None, [_vp]),
    "kmd_pca_sample": (_i, [_vp, _vp, _vp]),
    "kmd_pca_sample_streams": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "kmd_pca_count": (_i, [_vp, C.POINTER(_u64)]),
    "kmd_pca_gram": (_i, [_vp, _vp, _vp]),
    "kmd_pca_eigen": (_i, [_i, _vp, _i, _vp, _vp]),
    "kmd_synth_fill": (_i, [_u64, C.c_uint32, _u64, _sz, _i, _i, _i, _i, _sz, _vp, _vp, _vp, _vp]),
    "kmd_synth_streams": (_i, [_u64, C.c_uint32, _u64, _sz, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kmd_column_sums": (_i, [_vp, _i, _i, _sz, _sz, _i, _vp, _vp]),
    "kmd_copy_probe": (_i, [_vp, _vp, _sz, _vp]),
    "kmd_read_probe": (_i, [_vp, _sz, _i, _vp, _vp]),
}

_lib = None


def lib():
    """Load libkmdiff_hip.so (once).  Raises KmdError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KmdError("libkmdiff_hip.so is not built (%s); run `python -c 'import "
                           "__graft_entry__ as g; g.build()'` -- there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64
        # (soname libamdhip64.so.7, requested by libtorch_hip as "libamdhip64.so"): if this
        # library pulled in /opt/rocm's copy first, the loader would not recognise it under
        # torch's name, a second runtime would come up and find "No HIP GPUs".  Loading torch
        # first makes its copy the one this library binds to (same soname), whatever the
        # caller's import order.  Without torch (plain C / C++ hosts) there is nothing to share.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:
            raise KmdError("cannot load %s: %s" % (LIB_PATH, e))
        # (the version first: a stale .so says so instead of failing on the first name it lacks)
        try:
            L.kmd_abi_version.restype = C.c_int
            have = int(L.kmd_abi_version())
        except AttributeError:
            have = -1
        if have != ABI_VERSION:
            raise KmdError("%s has ABI version %d, this package binds version %d: rebuild it (python -c 'import "
                           "__graft_entry__ as g; g.build()')" % (LIB_PATH, have, ABI_VERSION))
        for name, (res, args) in list(SIGNATURES.items()) + list(TEST_SIGNATURES.items()):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, what=""):
    if status != KMD_OK:
        L = lib()
        msg = L.kmd_last_error().decode() or L.kmd_status_string(status).decode()
        raise KmdError("%s: %s (status %d)" % (what or "kmd", msg, status))
    return status


# ---- libkmdiff_hip_rccl.so (include/kmdiff_hip_rccl.h): the RCCL transport, a library of its own
RCCL_LIB_PATH = os.path.join(_HERE, "lib", "libkmdiff_hip_rccl.so")
RCCL_SIGNATURES = {
    "kmd_rccl_last_error": (C.c_char_p, []),
    "kmd_rccl_unique_id": (_i, [_vp]),
    "kmd_transport_rccl_init": (_i, [C.POINTER(Transport), _i, _i, _vp]),
    "kmd_transport_rccl_wrap": (_i, [C.POINTER(Transport), _vp]),
    "kmd_transport_rccl_destroy": (_i, [C.POINTER(Transport)]),
}
_rccl = None


def rccl_lib():
    """Load libkmdiff_hip_rccl.so (once; it pulls in librccl)."""
    global _rccl
    if _rccl is None:
        lib()                                               # (the HIP runtime first, as lib() arranges it)
        if not os.path.exists(RCCL_LIB_PATH):
            raise KmdError("libkmdiff_hip_rccl.so is not built (%s)" % RCCL_LIB_PATH)
        try:
            L = C.CDLL(RCCL_LIB_PATH)
        except OSError as e:
            raise KmdError("cannot load %s: %s" % (RCCL_LIB_PATH, e))
        for name, (res, args) in RCCL_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _rccl = L
    return _rccl
