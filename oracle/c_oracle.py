"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/c/gretel_oracle.c
(the plain-C restatement; see that file's header for the reference file:line map
and the parity status).  Built by `make -C oracle` / __graft_entry__.build().
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

SYMS = "ACGTN-_"
COND_MODES = {"A": 0, "B": 1, "C": 2, "D": 3, "E": 4}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
        L.orc_create.restype = vp
        L.orc_create.argtypes = [i32] * 6
        L.orc_destroy.argtypes = [vp]
        L.orc_set_L.argtypes = [vp, i32]
        L.orc_get_L.argtypes = [vp]
        L.orc_set_full_enum.argtypes = [vp, i32]
        L.orc_set_candidates.argtypes = [vp, vp, i32]
        L.orc_reweight_calls.restype = i64
        L.orc_reweight_calls.argtypes = [vp]
        L.orc_get_stats.argtypes = [vp, C.POINTER(i64)]
        L.orc_snapshot_original.argtypes = [vp]
        L.orc_add.argtypes = [vp, i32, i32, i32, i32]
        L.orc_get.restype = dbl
        L.orc_get.argtypes = [vp, i32, i32, i32, i32]
        L.orc_reweight_obs.restype = dbl
        L.orc_reweight_obs.argtypes = [vp, i32, i32, i32, i32, dbl]
        L.orc_fill.argtypes = [vp, vp, vp, vp, i64, i32]
        L.orc_counts_at.argtypes = [vp, i32, vp]
        L.orc_marginal.restype = dbl
        L.orc_marginal.argtypes = [vp, i32, i32]
        L.orc_conditional.restype = dbl
        L.orc_conditional.argtypes = [vp, i32, i32, i32, i32]
        L.orc_edge_weights.argtypes = [vp, i32, vp, vp]
        L.orc_generate_path.argtypes = [vp, vp, vp, vp, vp]
        L.orc_reweight_path.restype = dbl
        L.orc_reweight_path.argtypes = [vp, vp, dbl]
        L.orc_spin.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
        L.orc_gap_check.argtypes = [vp]
        L.orc_export_band.argtypes = [vp, vp]
        L.orc_audit_begin.argtypes = [vp]
        L.orc_audit_get.argtypes = [vp, vp]
        L.orc_log10.restype = dbl
        L.orc_log10.argtypes = [dbl, i32]
        L.orc_log10_many.argtypes = [vp, vp, i64, i32]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


AUDIT_BINS = ("tie", "<4", "<16", "<64", "<256", "<1024", "<2^20", "rest")


class _Audit(C.Structure):
    # struct orc_audit of oracle/c/gretel_oracle.c
    _fields_ = [("steps", C.c_int64), ("flips", C.c_int64), ("order_diffs", C.c_int64), ("nan_steps", C.c_int64),
                ("margin_bins", C.c_int64 * 8), ("first_flip_path", C.c_int64), ("first_flip_snp", C.c_int64),
                ("min_margin_ulps", C.c_double), ("max_abs_dhp_cur", C.c_double), ("max_abs_dhp_orig", C.c_double),
                ("max_abs_dw", C.c_double), ("paths", C.c_int64)]


class COracle:
    def __init__(self, n, band, storage="f32", cond_mode="A", marginal_term=False, use_libm=True,
                 cand_order="ACGT-", offer_zero=False):
        self.n, self.band = n, max(1, band)
        self._h = lib().orc_create(n, self.band, 0 if storage == "f32" else 1,
                                   COND_MODES[cond_mode], int(marginal_term), int(use_libm))
        if not self._h:
            raise MemoryError
        order = np.array([SYMS.index(c) for c in cand_order], dtype=np.int32)
        if len(order) != 5 or lib().orc_set_candidates(self._h, _p(order), int(bool(offer_zero))):
            raise ValueError("cand_order must be a permutation of 'ACGT-' (got %r)" % (cand_order,))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    @property
    def L(self):
        return lib().orc_get_L(self._h)

    @L.setter
    def L(self, v):
        lib().orc_set_L(self._h, int(v))

    def set_full_enum(self, v):
        lib().orc_set_full_enum(self._h, int(v))

    def reweight_calls(self):
        return lib().orc_reweight_calls(self._h)

    def fill(self, table, use_end_sentinels=False):
        rank = np.ascontiguousarray(table.rank, dtype=np.int32)
        off = np.ascontiguousarray(table.off, dtype=np.int64)
        bases = np.ascontiguousarray(table.bases, dtype=np.uint8)
        rc = lib().orc_fill(self._h, _p(rank), _p(off), _p(bases), len(rank), int(use_end_sentinels))
        if rc:
            raise RuntimeError("orc_fill rc=%d" % rc)
        return self.stats()

    def stats(self):
        out = (C.c_int64 * 3)()
        lib().orc_get_stats(self._h, out)
        return tuple(out)          # slices, crumbs, covered

    def snapshot_original(self):
        lib().orc_snapshot_original(self._h)

    def add(self, a, b, i, j):
        if lib().orc_add(self._h, a, b, i, j):
            raise IndexError("outside band")

    def get(self, a, b, i, j):
        return lib().orc_get(self._h, a, b, i, j)

    def reweight_obs(self, a, b, i, j, ratio):
        return lib().orc_reweight_obs(self._h, a, b, i, j, ratio)

    def counts_at(self, p):
        out = np.zeros(8)
        lib().orc_counts_at(self._h, p, _p(out))
        return out

    def marginal(self, s, p):
        return lib().orc_marginal(self._h, s, p)

    def conditional(self, a, b, i, j):
        return lib().orc_conditional(self._h, a, b, i, j)

    def edge_weights(self, p, path):
        path = np.ascontiguousarray(path, dtype=np.uint8)
        w = np.zeros(7)
        mask = lib().orc_edge_weights(self._h, p, _p(path), _p(w))
        return mask, w

    def generate_path(self):
        path = np.zeros(self.n + 1, dtype=np.uint8)
        hc, ho, mn = C.c_double(), C.c_double(), C.c_double()
        hole = lib().orc_generate_path(self._h, _p(path), C.byref(hc), C.byref(ho), C.byref(mn))
        if hole:
            return None, hole
        return path, (hc.value, ho.value, mn.value)

    def reweight_path(self, path, ratio):
        path = np.ascontiguousarray(path, dtype=np.uint8)
        return lib().orc_reweight_path(self._h, _p(path), ratio)

    def spin(self, max_paths):
        n1 = self.n + 1
        paths = np.zeros((max_paths, n1), dtype=np.uint8)
        hc, ho, ra, mg = (np.zeros(max_paths) for _ in range(4))
        hole = C.c_int()
        done = lib().orc_spin(self._h, max_paths, _p(paths), _p(hc), _p(ho), _p(ra), _p(mg), C.byref(hole))
        return dict(n=done, hole_at=hole.value, paths=paths[:done], hp_current=hc[:done],
                    hp_original=ho[:done], ratio=ra[:done], magnitude=mg[:done])

    def audit_begin(self):
        """From here on every generate_path evaluates BOTH log10s (libm's = the reference's math.log10, and
        include/gh_detlog.h's = the kernels') on the same state and counts the steps they would decide differently."""
        if lib().orc_audit_begin(self._h):
            raise MemoryError

    def audit(self):
        a = _Audit()
        if lib().orc_audit_get(self._h, C.byref(a)):
            raise RuntimeError("audit_begin() was not called")
        d = {k: getattr(a, k) for k, _ in _Audit._fields_ if k != "margin_bins"}
        d["margin_ulps"] = dict(zip(AUDIT_BINS, list(a.margin_bins)))
        return d

    def gap_check(self):
        return lib().orc_gap_check(self._h)

    def export_band(self):
        out = np.zeros((self.n + 2, self.band, 7, 7))
        lib().orc_export_band(self._h, _p(out))
        return out


def paths_to_str(paths):
    lut = np.frombuffer(SYMS.encode(), dtype=np.uint8)
    return [lut[p].tobytes().decode() for p in np.atleast_2d(paths)]
