"""
ctypes binding of libgretel_hip.so (C ABI: include/gretel_hip.h).

There is NO CPU fallback: if the shared library is missing, or no HIP device is
visible when a handle is created, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("GH_LIB") or os.path.join(_HERE, "libgretel_hip.so")   # GH_LIB: alternative build (A/B experiments)

GH_OK = 0
GH_ERR_ARG, GH_ERR_HIP, GH_ERR_BAND, GH_ERR_SYMBOL, GH_ERR_NOMEM, GH_ERR_STATE = -1, -2, -3, -4, -5, -6
GH_STORAGE = {"f32": 0, "f64": 1}
GH_COND = {"A": 0, "B": 1, "C": 2, "D": 3, "E": 4}
GH_K = {"fill": 0, "marg": 1, "lt": 2, "walk": 3, "reweight": 4, "seg": 5, "rwseg": 6}


class GretelHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libgretel_hip: %s (status %d)" % (msg, code))
        self.code = code


class BandError(GretelHipError, IndexError):
    pass


class SymbolError(GretelHipError, KeyError):
    pass


class gh_config(C.Structure):
    _fields_ = [("n_snps", C.c_int32), ("band", C.c_int32), ("storage", C.c_int32),
                ("cond_mode", C.c_int32), ("marginal_term", C.c_int32), ("device", C.c_int32),
                ("offer_zero", C.c_int32), ("cand_order", C.c_uint8 * 8)]


class gh_fill_stats(C.Structure):
    _fields_ = [("n_slices", C.c_int64), ("n_crumbs", C.c_int64), ("covered_snps", C.c_int64),
                ("L", C.c_int32), ("_pad", C.c_int32)]


class gh_path_rec(C.Structure):
    _fields_ = [("hp_current", C.c_double), ("hp_original", C.c_double),
                ("ratio", C.c_double), ("magnitude", C.c_double), ("min_marginal", C.c_double)]


_lib = None


def load():
    """Load libgretel_hip.so or raise -- never falls back to anything."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "gretel_amd: %s is missing. Build it with `make -C gretel_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the hot path." % SO_PATH)
    L = C.CDLL(SO_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    P = C.POINTER
    L.gh_last_error.restype = C.c_char_p
    L.gh_last_error.argtypes = []
    sigs = {
        "gh_device_count": [P(i32)],
        "gh_device_clock_khz": [i32, P(i32)],
        "gh_log10_device": [i32, vp, vp, C.c_int64],
        "gh_debug_segment_stamps": [vp, vp, i32],
        "gh_log10_host": [vp, vp, C.c_int64],
        "gh_create": [P(gh_config), P(vp)],
        "gh_destroy": [vp],
        "gh_copy": [vp, P(vp)],
        "gh_clear": [vp],
        "gh_sync": [vp],
        "gh_set_L": [vp, C.c_int32],
        "gh_get_L": [vp, P(C.c_int32)],
        "gh_get_fill_stats": [vp, P(gh_fill_stats)],
        "gh_set_fill_stats": [vp, P(gh_fill_stats)],
        "gh_reads_upload": [vp, vp, vp, vp, i64, P(vp)],
        "gh_reads_free": [vp],
        "gh_reads_max_k": [vp, P(i32)],
        "gh_reads_info": [vp, vp, vp],
        "gh_fill": [vp, vp, i32, P(gh_fill_stats)],
        "gh_add": [vp, i32, i32, i32, i32],
        "gh_add_batch": [vp, vp, vp, vp, vp, i64],
        "gh_get": [vp, i32, i32, i32, i32, P(dbl)],
        "gh_reweight_obs": [vp, i32, i32, i32, i32, dbl, P(dbl)],
        "gh_counts_at": [vp, i32, vp],
        "gh_marginal_of_at": [vp, i32, i32, P(dbl)],
        "gh_edge_weights_at": [vp, i32, vp, vp, P(i32)],
        "gh_gap_check": [vp, P(i32)],
        "gh_export_cmask": [vp, vp],
        "gh_snapshot_original": [vp],
        "gh_generate_path": [vp, vp, vp, P(dbl), P(dbl), P(dbl), P(i32)],
        "gh_reweight_path": [vp, vp, dbl, P(dbl)],
        "gh_spin": [vp, i32, dbl, vp, vp, P(i32), P(i32)],
        "gh_batch_create": [P(vp), i32, P(vp)],
        "gh_batch_destroy": [vp],
        "gh_batch_spin": [vp, i32, dbl, vp, vp, vp, vp],
        "gh_host_alloc": [C.c_size_t, P(vp)],
        "gh_host_free": [vp],
        "gh_batch_pipe_info": [vp, P(C.c_int32)],
        "gh_batch_profile_enable": [vp, i32],
        "gh_batch_profile_get": [vp, i32, P(dbl), P(i64), P(C.c_int32), P(dbl)],
        "gh_export_band": [vp, vp],
        "gh_import_band": [vp, vp],
        "gh_export_dense": [vp, vp],
        "gh_profile_enable": [vp, i32],
        "gh_profile_reset": [vp],
        "gh_profile_get": [vp, i32, P(dbl), P(i64)],
        "gh_profile_bytes": [vp, i32, P(dbl)],
        "gh_profile_overhead": [vp, i32, vp],
        "gh_debug_walk_clock": [vp, vp],
        "gh_debug_pool_geometry": [i32, i32, i32, vp],
        "gh_coverage_sites": [i32, vp, vp, vp, i64, C.c_int32, C.c_int32, C.c_int32, vp, vp],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)          # AttributeError if the ABI drifted
        fn.restype = C.c_int
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc == GH_OK:
        return
    msg = load().gh_last_error().decode("utf-8", "replace")
    if rc == GH_ERR_BAND:
        raise BandError(rc, msg)
    if rc == GH_ERR_SYMBOL:
        raise SymbolError(rc, msg)
    raise GretelHipError(rc, msg)


def device_clock_khz(device=-1):
    k = C.c_int(0)
    check(load().gh_device_clock_khz(int(device), C.byref(k)))
    return k.value


def log10_many(x, device=None):
    """log10 as the kernels evaluate it (include/gh_detlog.h): on the GPU `device`, or (None) the same source on the host."""
    import numpy as np
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    if device is None:
        check(load().gh_log10_host(x.ctypes.data, y.ctypes.data, x.size))
    else:
        check(load().gh_log10_device(int(device), x.ctypes.data, y.ctypes.data, x.size))
    return y


def device_count():
    n = C.c_int(0)
    check(load().gh_device_count(C.byref(n)))
    return n.value
