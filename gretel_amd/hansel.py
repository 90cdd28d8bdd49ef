"""
Device-backed `Hansel`: the drop-in for `hansel.Hansel` (hanselx==0.0.92, reference
setup.py:8) as Gretel consumes it.  The tensor lives in HBM behind libgretel_hip.so;
this class only marshals arguments (ctypes) and re-exposes the reference's names:

    Hansel.init_matrix(symbols, unsymbols, n)      gretel/util.py:83
    add_observation / get_observation              gretel/util.py:266-286, tests/test_test.py:41-52
    reweight_observation                           gretel/gretel.py:84,96
    get_counts_at / get_marginal_of_at             gretel/cmd.py:86,127, gretel/gretel.py:182,186
    get_edge_weights_at                            gretel/gretel.py:155
    copy / save_hansel_dump                        gretel/cmd.py:79,82
    symbols_d, L, n_slices, n_crumbs               gretel/gretel.py:138, gretel/util.py:329-333

plus the fused fast paths used by gretel_amd.gretel / gretel_amd.util
(`fill_from_support`, `generate_path`, `reweight_from_path`, `spin`).

There is no CPU implementation here: without libgretel_hip.so or without a GPU
every operation raises (gretel_amd._lib).
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import check

SYMBOLS = ['A', 'C', 'G', 'T', 'N', '-', '_']
UNSYMBOLS = ['N', '_']
_SYM_LUT = np.frombuffer("".join(SYMBOLS).encode(), dtype=np.uint8)


class HanselSymbol:
    """A symbol object: str() gives its character (gretel/cmd.py:128,164,211), it equals the same symbol object
    (gretel/cmd.py:201) and is hashable (a key of get_counts_at / get_edge_weights_at).  It is NOT a str and does not
    compare or hash equal to one: the reference has to coerce the keys of get_counts_at with str() before it can look
    "A" up (gretel/cmd.py:128; SURVEY.md App. A-3 [R]), so hanselx's symbols are not strings either."""
    __slots__ = ("c", "i")

    def __init__(self, char, i):
        self.c = char
        self.i = i

    def __str__(self):
        return self.c

    def __repr__(self):
        return self.c

    def __eq__(self, other):
        return isinstance(other, HanselSymbol) and other.i == self.i

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(("HanselSymbol", self.i))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Hansel:
    def __init__(self, n_snps, band=None, storage="f32", cond_mode="A", marginal_term=False, device=-1,
                 cand_order="ACGT-", offer_zero=False):
        """storage / cond_mode / marginal_term / cand_order / offer_zero: the switches of the Hansel arithmetic that the
        reference leaves to hanselx (include/gretel_hip.h: gh_config; same names as oracle.hansel_ref.HanselSpec)."""
        self._lib = _lib.load()
        self.n = int(n_snps)
        if sorted(cand_order) != sorted("ACGT-"):
            raise ValueError("cand_order must be a permutation of 'ACGT-' (got %r)" % (cand_order,))
        self._cfg = dict(storage=storage, cond_mode=cond_mode, marginal_term=bool(marginal_term), device=device,
                         cand_order=str(cand_order), offer_zero=bool(offer_zero))
        self.symbols = [HanselSymbol(c, i) for i, c in enumerate(SYMBOLS)]
        self.unsymbols = list(UNSYMBOLS)
        self.symbols_d = {str(s): s for s in self.symbols}
        self.is_weighted = False
        self._h = None
        self._band = None
        self._L = 1
        # observations staged on the host until the band width is known (init_matrix gives no hint)
        self._staged = []
        if band is not None:
            self._create(int(band))

    # -- construction ----------------------------------------------------------------------
    @staticmethod
    def init_matrix(symbols, unsymbols, n_snps, band=None, **kw):
        """gretel/util.py:83.  The reference's fixed alphabet is the only one supported."""
        if list(symbols) != SYMBOLS or list(unsymbols) != UNSYMBOLS:
            raise ValueError("gretel_amd.Hansel supports the alphabet %r / unsymbols %r only" % (SYMBOLS, UNSYMBOLS))
        return Hansel(n_snps, band=band, **kw)

    def _create(self, band):
        order = (C.c_uint8 * 8)(*[SYMBOLS.index(c) for c in self._cfg["cand_order"]], 0, 0, 0)
        cfg = _lib.gh_config(self.n, max(1, band), _lib.GH_STORAGE[self._cfg["storage"]],
                             _lib.GH_COND[self._cfg["cond_mode"]], int(self._cfg["marginal_term"]),
                             self._cfg["device"], int(self._cfg["offer_zero"]), order)
        h = C.c_void_p()
        check(self._lib.gh_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._band = max(1, band)
        check(self._lib.gh_set_L(self._h, self._L))

    def _destroy(self):
        if getattr(self, "_h", None):
            self._lib.gh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _ensure(self, need_band=1):
        """Make the device tensor exist with band >= need_band and no staged observations."""
        st, self._staged = self._staged, []
        if st:
            need_band = max(need_band, max(j - i for (_, _, i, j) in st))
        if self._h is None:
            self._create(need_band)
        elif need_band > self._band:
            self._reband(need_band)
        if st:
            a = np.array([s[0] for s in st], dtype=np.uint8)
            b = np.array([s[1] for s in st], dtype=np.uint8)
            i = np.array([s[2] for s in st], dtype=np.int32)
            j = np.array([s[3] for s in st], dtype=np.int32)
            check(self._lib.gh_add_batch(self._h, _p(a), _p(b), _p(i), _p(j), len(st)))

    def _reband(self, band):
        old = self.export_band()
        stats = self._get_stats()
        self._destroy()
        self._create(band)
        new = np.zeros((self.n + 2, band, 7, 7))
        new[:, :old.shape[1]] = old
        check(self._lib.gh_import_band(self._h, _p(new)))
        check(self._lib.gh_set_fill_stats(self._h, C.byref(stats)))

    def copy(self):
        """gretel/cmd.py:79"""
        self._ensure()
        o = Hansel(self.n, **self._cfg)
        h = C.c_void_p()
        check(self._lib.gh_copy(self._h, C.byref(h)))
        o._h, o._band, o._L = h, self._band, self._L
        o.is_weighted = self.is_weighted
        return o

    # -- attributes ------------------------------------------------------------------------
    def _get_stats(self):
        st = _lib.gh_fill_stats()
        check(self._lib.gh_get_fill_stats(self._h, C.byref(st)))
        return st

    @property
    def L(self):
        return self._L

    @L.setter
    def L(self, v):
        self._L = int(v)
        if self._h is not None:
            check(self._lib.gh_set_L(self._h, self._L))

    @property
    def n_slices(self):
        return int(self._get_stats().n_slices) if self._h is not None else getattr(self, "_n_slices", 0)

    @n_slices.setter
    def n_slices(self, v):
        self._ensure()
        st = self._get_stats()
        st.n_slices = int(v)
        check(self._lib.gh_set_fill_stats(self._h, C.byref(st)))

    @property
    def n_crumbs(self):
        return int(self._get_stats().n_crumbs) if self._h is not None else len(self._staged)

    @n_crumbs.setter
    def n_crumbs(self, v):
        self._ensure()
        st = self._get_stats()
        st.n_crumbs = int(v)
        check(self._lib.gh_set_fill_stats(self._h, C.byref(st)))

    @property
    def band(self):
        return self._band

    # -- symbols ---------------------------------------------------------------------------
    def _num(self, sym):
        if isinstance(sym, HanselSymbol):
            return sym.i
        if isinstance(sym, (int, np.integer)):
            return int(sym)
        return self.symbols_d[sym].i        # KeyError like the reference for an unknown symbol

    def _path_indices(self, path, upto=None):
        n = len(path) if upto is None else upto
        return np.fromiter((self._num(path[q]) for q in range(n)), dtype=np.uint8, count=n)

    # -- one-cell API (compatibility; a host round trip per call) ---------------------------
    def add_observation(self, symbol_from, symbol_to, pos_from, pos_to):
        a, b = self._num(symbol_from), self._num(symbol_to)
        if not (0 <= pos_from < pos_to <= self.n + 1):
            raise IndexError("observation positions (%d,%d) outside 0 <= i < j <= %d" % (pos_from, pos_to, self.n + 1))
        self._staged.append((a, b, int(pos_from), int(pos_to)))
        if len(self._staged) >= (1 << 20):
            self._ensure()

    def get_observation(self, symbol_from, symbol_to, pos_from, pos_to):
        self._ensure()
        out = C.c_double()
        check(self._lib.gh_get(self._h, self._num(symbol_from), self._num(symbol_to), int(pos_from), int(pos_to), C.byref(out)))
        return out.value

    def reweight_observation(self, symbol_from, symbol_to, pos_from, pos_to, ratio):
        self._ensure()
        out = C.c_double()
        check(self._lib.gh_reweight_obs(self._h, self._num(symbol_from), self._num(symbol_to),
                                        int(pos_from), int(pos_to), float(ratio), C.byref(out)))
        return out.value

    # -- lookups ---------------------------------------------------------------------------
    def counts_array(self, at_pos):
        self._ensure()
        out = np.zeros(8)
        check(self._lib.gh_counts_at(self._h, int(at_pos), _p(out)))
        return out

    def get_counts_at(self, at_pos):
        """gretel/cmd.py:86,127 -- keys: symbol objects with a positive count, plus "total"."""
        c = self.counts_array(at_pos)
        marg = {"total": float(c[7])}
        for s in self.symbols:
            if c[s.i] > 0:
                marg[s] = float(c[s.i])
        return marg

    def get_marginal_of_at(self, of_symbol, at_pos):
        """gretel/gretel.py:182,186"""
        self._ensure()
        out = C.c_double()
        check(self._lib.gh_marginal_of_at(self._h, self._num(of_symbol), int(at_pos), C.byref(out)))
        return out.value

    def get_edge_weights_at(self, at_pos, current_path, debug=False):
        """gretel/gretel.py:155"""
        self._ensure()
        path = self._path_indices(current_path, upto=at_pos)
        w = np.zeros(7)
        mask = C.c_int()
        check(self._lib.gh_edge_weights_at(self._h, int(at_pos), _p(path), _p(w), C.byref(mask)))
        # (keys in the order the candidates are offered in: the tie-break of gretel/gretel.py:166-174)
        offered = [self.symbols_d[c] for c in self._cfg["cand_order"]]
        return {s: float(w[s.i]) for s in offered if (mask.value >> s.i) & 1}

    def candidate_masks(self):
        """uint32[N+1]: bit s set <=> valid symbol s is a candidate at that position."""
        self._ensure()
        out = np.zeros(self.n + 1, dtype=np.uint32)
        check(self._lib.gh_export_cmask(self._h, _p(out)))
        return out

    def gap_check(self):
        """gretel/cmd.py:85-118: first position in [0,N] without evidence, or -1."""
        self._ensure()
        g = C.c_int()
        check(self._lib.gh_gap_check(self._h, C.byref(g)))
        return g.value

    # -- fused fast paths --------------------------------------------------------------------
    def fill_from_support(self, rank, off, bases, use_end_sentinels=False, reads_handle=None, max_k=None):
        """The pair loop of gretel/util.py:226-286 for a whole support table at once.
        Returns (n_slices, n_crumbs, covered_snps) and sets L like util.py:333.
        max_k: the longest row of the table, where the caller knows it (the native decoder reports it: gio_stats.max_row_len) --
        a pass over off[] less; gh_reads_upload finds it again on the device and the band is checked against that."""
        if reads_handle is None:
            reads_handle = DeviceReads(self, rank, off, bases, max_k=max_k)
        else:
            self._ensure(max(1, reads_handle.max_k - 1))
        st = _lib.gh_fill_stats()
        check(self._lib.gh_fill(self._h, reads_handle._r, int(bool(use_end_sentinels)), C.byref(st)))
        self._L = int(st.L)
        return int(st.n_slices), int(st.n_crumbs), int(st.covered_snps)

    def clear(self):
        self._staged = []
        if self._h is not None:
            check(self._lib.gh_clear(self._h))
        self._L = 1

    def snapshot_original(self):
        self._ensure()
        check(self._lib.gh_snapshot_original(self._h))

    def generate_path(self, original=None):
        """gretel/gretel.py:102-189 as one kernel.  Returns (indices uint8[N+1], hp_current,
        hp_original, min_marginal) or (None, hole_at, prefix walked)."""
        self._ensure()
        oh = None
        if original is not None and original is not self:
            original._ensure()
            oh = original._h
        path = np.zeros(self.n + 1, dtype=np.uint8)
        hc, ho, mn, hole = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        check(self._lib.gh_generate_path(self._h, oh, _p(path), C.byref(hc), C.byref(ho), C.byref(mn), C.byref(hole)))
        if hole.value:
            return None, hole.value, path[:hole.value]
        return path, hc.value, ho.value, mn.value

    def reweight_from_path(self, path_indices, ratio):
        """gretel/gretel.py:79-98 as one kernel."""
        self._ensure()
        p = np.ascontiguousarray(path_indices, dtype=np.uint8)
        if p.shape[0] != self.n + 1:
            raise ValueError("path must have N+1 = %d entries" % (self.n + 1))
        out = C.c_double()
        check(self._lib.gh_reweight_path(self._h, _p(p), float(ratio), C.byref(out)))
        self.is_weighted = True
        return out.value

    def spin(self, max_paths=100, min_remove=0.01, out_paths=None, out_recs=None):
        """gretel/cmd.py:148-179 on the device.  Returns dict(n, hole_at, paths uint8[n][N+1],
        hp_current, hp_original, ratio, magnitude).  out_paths uint8[max_paths][N+1] / out_recs float64[max_paths][5]: write the
        results there (C-contiguous; e.g. views of a pinned buffer that a gather sends on, gretel_amd.dist.ResultExchange) --
        `paths` of the returned dict is then a view of out_paths."""
        self._ensure()
        # (gh_spin writes rows 0 .. n-1 of both; gh_path_rec is five doubles -- include/gretel_hip.h)
        nrec = C.sizeof(_lib.gh_path_rec) // 8
        if out_paths is not None:
            if out_paths.dtype != np.uint8 or out_paths.shape != (max_paths, self.n + 1) or not out_paths.flags.c_contiguous:
                raise ValueError("out_paths must be a C-contiguous uint8[%d][%d]" % (max_paths, self.n + 1))
            paths = out_paths
        else:
            paths = np.empty((max_paths, self.n + 1), dtype=np.uint8)
        if out_recs is not None:
            if out_recs.dtype != np.float64 or out_recs.shape != (max(1, max_paths), nrec) or not out_recs.flags.c_contiguous:
                raise ValueError("out_recs must be a C-contiguous float64[%d][%d]" % (max(1, max_paths), nrec))
            recs = out_recs
        else:
            recs = np.empty((max(1, max_paths), nrec), dtype=np.float64)
        n, hole = C.c_int(), C.c_int()
        check(self._lib.gh_spin(self._h, int(max_paths), float(min_remove), _p(paths), _p(recs), C.byref(n), C.byref(hole)))
        k = n.value
        if k:
            self.is_weighted = True
        r = recs[:k]
        return dict(n=k, hole_at=hole.value, paths=paths[:k], hp_current=r[:, 0].copy(), hp_original=r[:, 1].copy(),
                    ratio=r[:, 2].copy(), magnitude=r[:, 3].copy(), min_marginal=r[:, 4].copy())

    def path_symbols(self, indices):
        return [self.symbols[int(q)] for q in indices]

    @staticmethod
    def path_str(indices):
        return _SYM_LUT[np.asarray(indices, dtype=np.uint8)].tobytes().decode()

    # -- export ----------------------------------------------------------------------------
    def export_band(self):
        self._ensure()
        out = np.zeros((self.n + 2, self._band, 7, 7))
        check(self._lib.gh_export_band(self._h, _p(out)))
        return out

    def export_dense(self):
        """[7][7][N+2][N+2] like the reference tensor (gretel/cmd.py:76-77); small N only."""
        self._ensure()
        out = np.zeros((7, 7, self.n + 2, self.n + 2))
        check(self._lib.gh_export_dense(self._h, _p(out)))
        return out

    def save_hansel_dump(self, path):
        """gretel/cmd.py:82 (--dumpmatrix).  hanselx's dump format is not in the reference
        tree; this writes an .npz with the banded tensor and the attributes."""
        np.savez_compressed(path, band=self.export_band(), n_snps=self.n, L=self.L,
                            n_slices=self.n_slices, n_crumbs=self.n_crumbs, symbols=np.array(SYMBOLS))

    # -- profiling (bench.py) ----------------------------------------------------------------
    def profile_enable(self, on=True):
        self._ensure()
        check(self._lib.gh_profile_enable(self._h, int(on)))

    def profile_reset(self):
        check(self._lib.gh_profile_reset(self._h))

    def profile_get(self):
        out = {}
        for name, k in _lib.GH_K.items():
            ms, n, by = C.c_double(), C.c_int64(), C.c_double()
            check(self._lib.gh_profile_get(self._h, k, C.byref(ms), C.byref(n)))
            check(self._lib.gh_profile_bytes(self._h, k, C.byref(by)))
            out[name] = dict(ms=ms.value, launches=n.value, bytes_per_launch=by.value)
        return out

    def profile_overhead(self, reps=20):
        """(ms of an empty HIP-event bracket, ms of a bracket around an empty kernel) on this handle's stream."""
        out = (C.c_double * 2)()
        check(self._lib.gh_profile_overhead(self._h, int(reps), out))
        return float(out[0]), float(out[1])

    def walk_clock(self):
        """(shader cycles, 100 MHz ticks, steps, variant) of the walker wave in the last path-extension launch."""
        out = (C.c_uint64 * 4)()
        check(self._lib.gh_debug_walk_clock(self._h, out))
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def sync(self):
        if self._h is not None:
            check(self._lib.gh_sync(self._h))


class _PinnedBlock:
    """One gh_host_alloc block; freed (gh_host_free) when the last numpy array made over it has gone."""

    def __init__(self, lib, nbytes):
        self._lib = lib
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib.gh_host_alloc(self.nbytes, C.byref(p)))
        self._p = p

    def array(self, dtype, count):
        dtype = np.dtype(dtype)
        assert count * dtype.itemsize <= self.nbytes
        raw = (C.c_uint8 * max(1, count * dtype.itemsize)).from_address(self._p.value)
        raw._gh_owner = self            # numpy keeps `raw` as the array's base, `raw` keeps the block
        return np.frombuffer(raw, dtype=dtype, count=count)

    def __del__(self):
        try:
            if getattr(self, "_p", None):
                self._lib.gh_host_free(self._p)
                self._p = None
        except Exception:
            pass


class PinnedTableArena:
    """Page-locked homes for a support table's three arrays (rank, off, bases), kept and grown from window to window: the native
    decoder writes the table straight into them (gretel_amd.bamio.native_support_table(arena=...), include/gretel_io.h:
    gio_support_table_from_bam_alloc) and gh_reads_upload reads them by DMA -- no 17 MB of fresh pages per million reads for the
    decoder to fault in, no staging copy for the upload.  The arrays handed out are views: their contents stand until the arena
    takes the next table, the memory for as long as a view refers to it (_PinnedBlock)."""

    def __init__(self):
        self._lib = _lib.load()
        self._blocks = [None, None, None]

    def alloc(self, which, nbytes):
        b = self._blocks[which]
        if b is None or b.nbytes < nbytes:
            # (half as much again: the next window of a contig is about as deep, not exactly)
            b = self._blocks[which] = _PinnedBlock(self._lib, max(4096, nbytes + nbytes // 2))
        return b._p.value

    def view(self, which, dtype, count):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        return self._blocks[which].array(dtype, count)


_table_arena = None
# (one table at a time in the process-wide arena: load_from_bam holds this from the decode until the upload has read the table --
# ctypes releases the GIL inside both, and a second thread's decode would write into the blocks the first one is uploading from)
table_arena_lock = threading.RLock()


def table_arena():
    """The process's PinnedTableArena (made at first use; None where page-locked memory cannot be had: no GPU runtime)."""
    global _table_arena
    if _table_arena is None:
        try:
            a = PinnedTableArena()
            a.alloc(2, 4096)
            _table_arena = a
        except Exception:
            _table_arena = False
    return _table_arena or None


class HanselBatch:
    """Many windows of one shape recovered together (gh_batch_*): every kernel of the spin loop
    (gretel/cmd.py:148-179) is launched over all windows at once, one path-extension workgroup per
    window.  The Hansels must already be filled and agree on n_snps, band, storage, modes and L."""

    def __init__(self, hansels):
        self.hansels = list(hansels)
        if not self.hansels:
            raise ValueError("empty batch")
        for h in self.hansels:
            h._ensure()
        self._lib = self.hansels[0]._lib
        arr = (C.c_void_p * len(self.hansels))(*[h._h for h in self.hansels])
        b = C.c_void_p()
        check(self._lib.gh_batch_create(arr, len(self.hansels), C.byref(b)))
        self._b = b

    def __del__(self):
        try:
            self._free_host_buffers()
            if getattr(self, "_b", None):
                self._lib.gh_batch_destroy(self._b)
                self._b = None
        except Exception:
            pass

    def pipe_info(self):
        """The last spin(): windows the window pipeline (csrc/wpipe.hpp) carried, windows it handed back to gh_spin, its
        threads per workgroup and positions per chunk."""
        out = (C.c_int32 * 4)()
        check(self._lib.gh_batch_pipe_info(self._b, out))
        return dict(windows=out[0], handed_back=out[1], threads=out[2], chunk=out[3])

    def profile_enable(self, every=10):
        """HIP events around the batched extension and the batched reweight of every `every`-th path (first window group)."""
        check(self._lib.gh_batch_profile_enable(self._b, int(every)))

    def profile_get(self):
        out = {}
        for name in ("walk", "reweight"):
            ms, n, w, by = C.c_double(), C.c_int64(), C.c_int32(), C.c_double()
            check(self._lib.gh_batch_profile_get(self._b, _lib.GH_K[name], C.byref(ms), C.byref(n), C.byref(w), C.byref(by)))
            out[name] = dict(ms=ms.value, launches=n.value, windows=w.value, bytes_per_launch=by.value)
        return out

    def _host_buffers(self, n, max_paths, n1):
        """Page-locked result buffers, kept from call to call (gh_host_alloc): the copies of 256 windows x 100 paths run at the
        link's rate and no 256 MB array is faulted in per call.  Every block is owned by a _PinnedBlock that the arrays made over
        it keep alive: a view handed out by spin(copy=False) stays valid memory for as long as anything refers to it, also when
        the batch has moved on to buffers of another shape or is gone (they used to point at freed pinned memory)."""
        key = (n, max_paths, n1)
        if getattr(self, "_hb_key", None) != key:
            self._free_host_buffers()
            sizes = (n * max_paths * n1, n * max_paths * 5 * 8)
            blocks = [_PinnedBlock(self._lib, max(1, sz)) for sz in sizes]
            self._hb_paths = blocks[0].array(np.uint8, sizes[0]).reshape(n, max_paths, n1)
            self._hb_recs = blocks[1].array(np.float64, sizes[1] // 8).reshape(n, max_paths, 5)
            self._hb_key = key
        return self._hb_paths, self._hb_recs

    def _free_host_buffers(self):
        # (the blocks free themselves when the last array over them is gone)
        self._hb_key, self._hb_paths, self._hb_recs = None, None, None

    def spin(self, max_paths=100, min_remove=0.01, copy=True):
        """copy=False: the returned arrays are views of the batch's page-locked buffers (no 256 MB copy on the host); their
        CONTENTS stand until the next spin() of this batch overwrites them, the memory itself for as long as a view refers to it.
        copy=True (default): every window gets its own arrays."""
        n = len(self.hansels)
        n1 = self.hansels[0].n + 1
        paths, recs = self._host_buffers(n, max_paths, n1)
        n_out = np.zeros(n, dtype=np.int32)
        hole = np.zeros(n, dtype=np.int32)
        check(self._lib.gh_batch_spin(self._b, int(max_paths), float(min_remove), _p(paths), _p(recs), _p(n_out), _p(hole)))
        out = []
        for w in range(n):
            k = int(n_out[w])
            if k:
                self.hansels[w].is_weighted = True
            pw, rw = paths[w, :k], recs[w, :k]
            if copy:
                pw, rw = pw.copy(), rw.copy()
            out.append(dict(n=k, hole_at=int(hole[w]), paths=pw, hp_current=rw[:, 0], hp_original=rw[:, 1], ratio=rw[:, 2],
                            magnitude=rw[:, 3], min_marginal=rw[:, 4]))
        return out


class DeviceReads:
    """A support table resident in HBM (gh_reads_upload)."""

    def __init__(self, hansel, rank, off, bases, max_k=None):
        rank = np.ascontiguousarray(rank, dtype=np.int32)
        off = np.ascontiguousarray(off, dtype=np.int64)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        self.n_reads = len(rank)
        if max_k is None:
            max_k = int(np.diff(off).max()) if self.n_reads else 0
        hansel._ensure(max(1, int(max_k) - 1))
        self._lib = hansel._lib
        r = C.c_void_p()
        check(self._lib.gh_reads_upload(hansel._h, _p(rank), _p(off), _p(bases), self.n_reads, C.byref(r)))
        self._r = r
        # (what the device found: a caller's max_k that is too small shows as GH_ERR_BAND in the fill, as any read too long for the band)
        mk = C.c_int()
        check(self._lib.gh_reads_max_k(r, C.byref(mk)))
        self.max_k = int(mk.value)

    def info(self):
        """What the upload found out about the table (gh_reads_info): dict(max_k, sorted, span_pos, dens128, first_at)."""
        inf = np.zeros(5, dtype=np.int64)
        check(self._lib.gh_reads_info(self._r, _p(inf), None))
        fa = np.zeros(int(inf[4]), dtype=np.int64)
        if len(fa):
            check(self._lib.gh_reads_info(self._r, _p(inf), _p(fa)))
        return dict(max_k=int(inf[0]), sorted=bool(inf[1]), span_pos=int(inf[2]), dens128=int(inf[3]), first_at=fa)

    def __del__(self):
        try:
            if getattr(self, "_r", None):
                self._lib.gh_reads_free(self._r)
                self._r = None
        except Exception:
            pass
