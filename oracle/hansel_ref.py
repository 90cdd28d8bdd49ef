"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under gretel_amd/ may import this.

CPU restatement (pure Python + NumPy scalars, one Python-level call per cell,
like the reference) of the `hansel.Hansel` data structure that the Gretel hot
path drives.

PARITY STATUS: **parity unpinned** for everything except the fill.
  The arithmetic of this class lives in the third-party package
  `hanselx==0.0.92` (reference `setup.py:8`, imported at `gretel/gretel.py:7`
  and `gretel/util.py:4`).  That package is NOT under /root/reference and is
  not installable here (no network).  What IS pinned by the reference's own
  tests (`tests/test_test.py:36-52`): init_matrix / add_observation /
  get_observation, n_slices, n_crumbs, L>0 -- see tests/test_oracle_golden.py.
  Everything else below follows SURVEY.md Appendix A; every disputed choice is
  a field of `HanselSpec` so it can be flipped if the real source turns up.

Call sites this class must satisfy (reference file:line):
  init_matrix            gretel/util.py:83
  add_observation        gretel/util.py:266,267,274,275,280,286
  get_observation        tests/test_test.py:41-52
  reweight_observation   gretel/gretel.py:84,96
  get_counts_at          gretel/cmd.py:86,127
  get_marginal_of_at     gretel/gretel.py:182,186
  get_edge_weights_at    gretel/gretel.py:155
  copy                   gretel/cmd.py:79
  symbols_d              gretel/gretel.py:138, gretel/cmd.py:201
  L / n_slices / n_crumbs (attributes)  gretel/util.py:329-333, gretel/cmd.py:227-229

Frozen arithmetic (the "spec"; the C oracle and the HIP kernels implement the
same thing):
  * storage dtype float32 (spec.storage="f32") or float64 ("f64");
    cells are H[sym_from, sym_to, pos_from, pos_to] (gretel/cmd.py:76-77).
  * row sums  c_s(p) = sum_t H[s,t,p,p+1]  accumulate sequentially, t ascending,
    IN THE STORAGE DTYPE (what np.sum does on a 7-element slice); everything
    downstream is float64.
  * total(p) = sum over symbols (index order) of float64(c_s) for c_s > 0.
  * marginal(s,p) = float64(c_s)/total.
  * V(p) = number of valid symbols (not in unsymbols) with c_s(p) > 0.
  * conditional of (a at i) -> (b at j):
        mode "A": (1 + H[a,b,i,j]) / (V(j) + sum_x H[a,x,i,j])     [frozen default]
        mode "B": (1 + H[a,b,i,j]) / (V(i) + c_a(i))
        mode "C": (1 + H[a,b,i,j]) / (V(i) + sum_x H[x,b,i,j])
        mode "D": (1 + H[a,b,i,j]) / (V(i) + sum_x H[a,x,i,j])
        mode "E": (1 + H[a,b,i,j]) / (V(j) + sum_x H[x,b,i,j])
      (C and E read the cell as P(a at i | b at j) -- the earlier variant given the candidate,
      the naive-Bayes form of the published method, README.md:79-94 -- with the "unique
      variants" term of gretel/gretel.py:10's TODO at i-l (C) or at i (E); A and D read it
      as P(b at j | a at i); B conditions on the marginal count of a.)
    row/col sums in the storage dtype, x ascending; the quotient in float64
    with IEEE semantics (x/0 -> inf).
  * edge weight of candidate b at p given path:
        w = 0.0 (+ log10(marginal(b,p)) if spec.marginal_term)
        for l = 1..min(L,p):  w += log10(conditional(path[p-l]@p-l -> b@p))
    candidates = valid symbols with c_b(p) > 0 (spec.offer_zero: every valid symbol, a
    zero count included), offered in spec.cand_order (default: symbol index order
    A C G T -).  The order is the insertion order of the returned dict and therefore the
    tie-break of gretel/gretel.py:166-174 (first key wins, later keys on strict >).
  * reweight_observation: old=H[..]; new = f64(old) - ratio*f64(old);
    H[..] = storage(new);  return f64(old) - new.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

SYMBOLS = ['A', 'C', 'G', 'T', 'N', '-', '_']      # gretel/util.py:83
UNSYMBOLS = ['N', '_']                             # gretel/util.py:83


@dataclass(frozen=True)
class HanselSpec:
    storage: str = "f32"          # "f32" | "f64"      
    cond_mode: str = "A"          # "A" | "B" | "C" | "D" | "E"   (D = V(pos_from) + row sum; E = V(pos_to) + column sum)
    marginal_term: bool = False   # add log10(marginal) to edge weights (App. A-7)
    cand_order: str = "ACGT-"     # order in which get_edge_weights_at offers the candidates (dict insertion order = tie-break)
    offer_zero: bool = False      # offer every valid symbol, also those never observed at the position (App. A-4/5 [M])

    def __post_init__(self):
        if sorted(self.cand_order) != sorted("ACGT-"):
            raise ValueError("cand_order must be a permutation of 'ACGT-' (got %r)" % (self.cand_order,))
        if self.cond_mode not in ("A", "B", "C", "D", "E"):
            raise ValueError("cond_mode %r" % (self.cond_mode,))

    @property
    def np_dtype(self):
        return np.float32 if self.storage == "f32" else np.float64


DEFAULT_SPEC = HanselSpec()


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


def _log10(x):
    # math.log10 raises on 0/inf-like inputs differently from IEEE; give IEEE answers.
    x = float(x)
    if x == 0.0:
        return -math.inf
    if math.isinf(x):
        return math.inf if x > 0 else math.nan
    if x < 0 or math.isnan(x):
        return math.nan
    return math.log10(x)


class _DenseStore:
    """H[a,b,i,j] as the reference allocates it: 7*7*(N+2)^2 cells (gretel/util.py:83)."""

    def __init__(self, n_sym, n, dtype):
        self.a = np.zeros((n_sym, n_sym, n + 2, n + 2), dtype=dtype)

    def get(self, a, b, i, j):
        return self.a[a, b, i, j]

    def set(self, a, b, i, j, v):
        self.a[a, b, i, j] = v

    def copy(self):
        o = _DenseStore.__new__(_DenseStore)
        o.a = self.a.copy()
        return o

    def dense(self):
        return self.a


class _BandStore:
    """Same cells, but only those with 1 <= j-i <= band are backed by memory.
    Cells outside the band are zero and can only be 'written' with zero --
    results-identical to the dense tensor whenever no observation falls outside
    the band (zero cells are fixed points of reweight).  Lets the oracle run the
    10k-SNP configs without a 19.6 GB allocation."""

    def __init__(self, n_sym, n, dtype, band):
        self.n = n
        self.band = band
        self.dtype = dtype
        self.zero = dtype(0)
        self.a = np.zeros((n + 2, band, n_sym, n_sym), dtype=dtype)

    def get(self, a, b, i, j):
        d = j - i
        if 1 <= d <= self.band and 0 <= i and j <= self.n + 1:
            return self.a[i, d - 1, a, b]
        return self.zero

    def set(self, a, b, i, j, v):
        d = j - i
        if 1 <= d <= self.band and 0 <= i and j <= self.n + 1:
            self.a[i, d - 1, a, b] = v
        elif v != 0:
            raise IndexError("observation (%d,%d) outside band %d" % (i, j, self.band))

    def copy(self):
        o = _BandStore.__new__(_BandStore)
        o.n, o.band, o.dtype, o.zero = self.n, self.band, self.dtype, self.zero
        o.a = self.a.copy()
        return o

    def dense(self):
        n_sym = self.a.shape[2]
        out = np.zeros((n_sym, n_sym, self.n + 2, self.n + 2), dtype=self.dtype)
        for i in range(self.n + 2):
            for d in range(1, self.band + 1):
                if i + d <= self.n + 1:
                    out[:, :, i, i + d] = self.a[i, d - 1]
        return out


class Hansel:
    def __init__(self, store, symbols, unsymbols, n, spec=DEFAULT_SPEC, L=1):
        self._s = store
        self.spec = spec
        self.n = n
        self.symbols = [HanselSymbol(c, i) for i, c in enumerate(symbols)]
        self.unsymbols = list(unsymbols)
        self.symbols_d = {str(s): s for s in self.symbols}
        self._valid = [s.i for s in self.symbols if str(s) not in self.unsymbols]
        self.is_weighted = False
        self.n_slices = 0
        self.n_crumbs = 0
        self.L = L

    # -- construction -----------------------------------------------------
    @staticmethod
    def init_matrix(symbols, unsymbols, n, spec=DEFAULT_SPEC, band=None):
        """gretel/util.py:83.  band=None -> dense like the reference."""
        dt = spec.np_dtype
        if band is None:
            store = _DenseStore(len(symbols), n, dt)
        else:
            store = _BandStore(len(symbols), n, dt, band)
        return Hansel(store, symbols, unsymbols, n, spec)

    def copy(self):
        """gretel/cmd.py:79"""
        o = Hansel(self._s.copy(), [str(s) for s in self.symbols], self.unsymbols,
                   self.n, self.spec, self.L)
        o.n_slices, o.n_crumbs, o.is_weighted = self.n_slices, self.n_crumbs, self.is_weighted
        return o

    def _num(self, sym):
        if isinstance(sym, HanselSymbol):
            return sym.i
        if isinstance(sym, (int, np.integer)):
            return int(sym)
        return self.symbols_d[sym].i

    # -- per-cell API -------------------------------------------------------
    def add_observation(self, symbol_from, symbol_to, pos_from, pos_to):
        a, b = self._num(symbol_from), self._num(symbol_to)
        dt = self.spec.np_dtype
        self._s.set(a, b, pos_from, pos_to, dt(self._s.get(a, b, pos_from, pos_to) + dt(1)))

    def get_observation(self, symbol_from, symbol_to, pos_from, pos_to):
        return self._s.get(self._num(symbol_from), self._num(symbol_to), pos_from, pos_to)

    def reweight_observation(self, symbol_from, symbol_to, pos_from, pos_to, ratio):
        a, b = self._num(symbol_from), self._num(symbol_to)
        old = float(self._s.get(a, b, pos_from, pos_to))
        new = old - float(ratio) * old
        self._s.set(a, b, pos_from, pos_to, self.spec.np_dtype(new))
        return old - new

    # -- lookups ------------------------------------------------------------
    def _row_sum(self, a, i, j):
        dt = self.spec.np_dtype
        acc = dt(0)
        for x in range(len(self.symbols)):
            acc = dt(acc + self._s.get(a, x, i, j))
        return acc

    def _col_sum(self, b, i, j):
        dt = self.spec.np_dtype
        acc = dt(0)
        for x in range(len(self.symbols)):
            acc = dt(acc + self._s.get(x, b, i, j))
        return acc

    def _counts(self, p):
        return [self._row_sum(s.i, p, p + 1) for s in self.symbols]

    def get_counts_at(self, at_pos):
        """gretel/cmd.py:86,127 -- keys are symbol objects plus the str "total"."""
        marg = {"total": 0.0}
        for s, c in zip(self.symbols, self._counts(at_pos)):
            if c > 0:
                marg[s] = c
                marg["total"] += float(c)
        return marg

    def get_marginal_of_at(self, of_symbol, at_pos):
        """gretel/gretel.py:182,186"""
        marg = self.get_counts_at(at_pos)
        sym = self.symbols[self._num(of_symbol)]
        if sym not in marg or marg["total"] == 0.0:
            return 0.0
        return float(marg[sym]) / marg["total"]

    def _n_valid_at(self, p):
        c = self._counts(p)
        return sum(1 for v in self._valid if c[v] > 0)

    def get_conditional_of_at(self, symbol_from, symbol_to, pos_from, pos_to):
        a, b = self._num(symbol_from), self._num(symbol_to)
        obs = float(self._s.get(a, b, pos_from, pos_to))
        mode = self.spec.cond_mode
        if mode == "A":
            den = float(self._n_valid_at(pos_to)) + float(self._row_sum(a, pos_from, pos_to))
        elif mode == "B":
            den = float(self._n_valid_at(pos_from)) + float(self._row_sum(a, pos_from, pos_from + 1))
        elif mode == "C":
            den = float(self._n_valid_at(pos_from)) + float(self._col_sum(b, pos_from, pos_to))
        elif mode == "D":
            den = float(self._n_valid_at(pos_from)) + float(self._row_sum(a, pos_from, pos_to))
        elif mode == "E":
            den = float(self._n_valid_at(pos_to)) + float(self._col_sum(b, pos_from, pos_to))
        else:
            raise ValueError(mode)
        num = 1.0 + obs
        if den == 0.0:
            return math.inf
        return num / den

    def get_edge_weights_at(self, at_pos, current_path, debug=False):
        """gretel/gretel.py:155"""
        counts = self._counts(at_pos)
        out = {}
        for c in self.spec.cand_order:
            v = self.symbols_d[c].i
            if not self.spec.offer_zero and not counts[v] > 0:
                continue
            sym = self.symbols[v]
            w = 0.0
            if self.spec.marginal_term:
                w += _log10(self.get_marginal_of_at(sym, at_pos))
            for l in range(1, min(self.L, at_pos) + 1):
                w += _log10(self.get_conditional_of_at(current_path[at_pos - l], sym, at_pos - l, at_pos))
            out[sym] = w
        return out

    # -- test helper --------------------------------------------------------
    def dense(self):
        return self._s.dense()
