"""Edge cases of the hot path on the GPU against the oracle: tiny windows, L larger than the window,
bands wider than the window, every read at the boundaries, storage/mode switches at the edges."""
import os

import numpy as np
import pytest

from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu


class _T:
    def __init__(self, n, reads):
        self.n_snps = n
        self.rank = np.array([r for r, _ in reads], dtype=np.int32)
        self.off = np.concatenate([[0], np.cumsum([len(s) for _, s in reads])]).astype(np.int64)
        self.bases = np.frombuffer("".join(s for _, s in reads).encode(), dtype=np.uint8).copy()
        self.band = max(1, max(len(s) for _, s in reads) - 1)
        self.n_reads = len(reads)


class _walk_mode:
    """GH_WALK is read when a handle is created: pin a path-extension variant for the handles made inside the block."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = os.environ.get("GH_WALK")
        if self.mode is None:
            os.environ.pop("GH_WALK", None)
        else:
            os.environ["GH_WALK"] = self.mode

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("GH_WALK", None)
        else:
            os.environ["GH_WALK"] = self.old


PINNED = bool(os.environ.get("GH_WALK") or os.environ.get("GH_WALK_THREADS"))      # A/B knobs pin a variant from outside


def _check_one(t, paths, L, want_variant, walk, **kw):
    with _walk_mode(walk):
        h = Hansel(t.n_snps, band=t.band, **kw)
    o = COracle(t.n_snps, t.band, kw.get("storage", "f32"), kw.get("cond_mode", "A"), kw.get("marginal_term", False))
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    if L is not None:
        h.L = L
        o.L = L
    assert h.gap_check() == o.gap_check()
    res, ref = h.spin(paths), o.spin(paths)
    if want_variant is not None and res["n"]:
        assert h.walk_clock()[3] == want_variant          # which path-extension variant the last launch took
    assert res["n"] == ref["n"] and res["hole_at"] == ref["hole_at"]
    assert np.array_equal(res["paths"], ref["paths"])
    assert res["hp_current"].tolist() == ref["hp_current"].tolist()
    assert res["hp_original"].tolist() == ref["hp_original"].tolist()
    assert res["ratio"].tolist() == ref["ratio"].tolist()
    assert np.allclose(res["magnitude"], ref["magnitude"], rtol=1e-12, atol=0)
    assert np.array_equal(h.export_band(), o.export_band())
    return res, h.L


def _check(t, paths=6, L=None, want_variant=None, **kw):
    """The window through the default path extension (segment-parallel for L <= 5: variant 3) and, where a test names a
    serial-walker variant (0, 1, 2), through GH_WALK=spec as well -- each against a fresh oracle."""
    if PINNED:
        return _check_one(t, paths, L, None, os.environ.get("GH_WALK"), **kw)[0]
    res, eff_L = _check_one(t, paths, L, None, None, **kw)
    if eff_L <= 5 and res["n"]:
        _check_one(t, paths, L, 3, None, **kw)
    if want_variant is not None or eff_L <= 5:
        res2, _ = _check_one(t, paths, L, want_variant, "spec", **kw)
        assert np.array_equal(res["paths"], res2["paths"]) and res["hp_current"].tolist() == res2["hp_current"].tolist()
    return res


def test_two_snp_window():
    # N=2: every pair is both "first" and "last": Sentinel->A fires (elif), B->Sentinel never does -> SNP 2 has no exit
    t = _T(2, [(0, "AC"), (0, "AC"), (0, "GT")])
    h = Hansel(2, band=1)
    o = COracle(2, 1)
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t) == (3, 3, 6)
    assert h.gap_check() == o.gap_check() == 2
    assert np.array_equal(h.export_band(), o.export_band())
    res = _check(t, paths=3)          # no symbol has outgoing evidence at SNP 2: a hole, as in the oracle
    assert res["n"] == 0 and res["hole_at"] == 2


def test_three_snp_window_all_rules():
    reads = [(0, "ACG"), (0, "ACG"), (0, "TCG"), (1, "CG"), (1, "CA"), (0, "AC")]
    res = _check(_T(3, reads), paths=5)
    assert res["n"] >= 1


@pytest.mark.parametrize("L", [1, 2, 7, 16, 17, 40])
def test_L_beyond_window_and_band(L):
    t = make_support_table(12, 400, k=3, seed=3)
    _check(t, paths=4, L=L)           # L > N, L > band, L > 16 (global-memory walker)


def test_band_wider_than_window():
    t = make_support_table(6, 200, k=6, seed=1)      # every read spans the whole window
    assert t.band == 5
    _check(t, paths=4)


@pytest.mark.parametrize("kw", [dict(storage="f64"), dict(cond_mode="B"), dict(cond_mode="C"), dict(cond_mode="D"), dict(marginal_term=True),
                                dict(storage="f64", cond_mode="C", marginal_term=True)])
def test_switches_on_small_window(kw):
    t = make_support_table(25, 700, k=4, seed=9)
    _check(t, paths=6, **kw)


def test_deletion_symbols_take_the_five_candidate_walker():
    # '-' among the candidates: the walker variant with three arg-max levels and 5 hypothesis rows
    t = make_support_table(80, 2500, k=4, seed=4)
    bases = t.bases.copy()
    rng = np.random.default_rng(1)
    bases[rng.random(len(bases)) < 0.15] = ord('-')
    t.bases = bases
    res = _check(t, paths=8)
    assert (res["paths"] == 5).any()          # '-' is actually selected somewhere


@pytest.mark.parametrize("dels", [False, True])
@pytest.mark.parametrize("L", [2, 3, 6, 13, 16])
def test_many_chunks_every_walker_variant(L, dels):
    # windows long enough for several LDS chunks at every unroll factor: without '-' the depth-2 walker on the
    # loader-derived tables (L = 2 has no resolved lag at all), with '-' the depth-1 walker on the raw blocks
    t = make_support_table(700, 12000, k=6, seed=10 + L)
    if dels:
        bases = t.bases.copy()
        bases[np.random.default_rng(L).random(len(bases)) < 0.1] = ord('-')
        t.bases = bases
    res = _check(t, paths=3, L=L, want_variant=0 if dels else 2)
    assert res["n"] == 3


@pytest.mark.parametrize("L", [2, 5, 9])
def test_deletions_with_at_most_four_candidates_take_the_depth2_walker(L):
    # three haplotypes, no sequencing errors, '-' sprinkled in: every position has <= 4 candidates, so the conditional
    # table is built over candidate RANKS and the 4-symbol depth-2 walker serves the window although '-' is selected
    t = make_support_table(500, 9000, k=6, n_haps=3, err=0.0, seed=40 + L)
    bases = t.bases.copy()
    bases[np.random.default_rng(L).random(len(bases)) < 0.2] = ord('-')
    t.bases = bases
    res = _check(t, paths=4, L=L, want_variant=2)
    assert (res["paths"] == 5).any()          # '-' is actually selected somewhere


@pytest.mark.parametrize("kw", [dict(cond_mode="B"), dict(marginal_term=True), dict(storage="f64", cond_mode="C"),
                                dict(storage="f64", cond_mode="B", marginal_term=True)])
def test_ranked_tables_under_every_switch(kw):
    # the rank layout with '-' among the candidates, for the other conditionals / the marginal term (full k_lt each
    # path) / f64 storage
    t = make_support_table(300, 6000, k=5, n_haps=3, err=0.0, seed=77)
    bases = t.bases.copy()
    bases[np.random.default_rng(7).random(len(bases)) < 0.2] = ord('-')
    t.bases = bases
    res = _check(t, paths=4, L=4, want_variant=2, **kw)
    assert (res["paths"] == 5).any()


def test_n_symbols_are_counted_but_never_selected():
    t = make_support_table(60, 2000, k=4, seed=6)
    bases = t.bases.copy()
    rng = np.random.default_rng(2)
    bases[rng.random(len(bases)) < 0.2] = ord('N')
    t.bases = bases
    res = _check(t, paths=5)
    assert not (res["paths"][:, 1:] == 4).any()


def test_repeated_spins_continue_where_the_last_stopped():
    t = make_support_table(90, 3000, k=4, seed=8)
    h = Hansel(t.n_snps, band=t.band)
    o = COracle(t.n_snps, t.band)
    h.fill_from_support(t.rank, t.off, t.bases)
    o.fill(t)
    a1, a2 = h.spin(3), h.spin(4)
    b = o.spin(7)
    assert np.array_equal(np.concatenate([a1["paths"], a2["paths"]]), b["paths"])
    assert a1["hp_current"].tolist() + a2["hp_current"].tolist() == b["hp_current"].tolist()
    assert a1["hp_original"].tolist() + a2["hp_original"].tolist() == b["hp_original"].tolist()


def test_mixing_per_cell_calls_with_fused_calls():
    t = make_support_table(50, 1500, k=4, seed=2)
    h = Hansel(t.n_snps, band=t.band)
    o = COracle(t.n_snps, t.band)
    h.fill_from_support(t.rank, t.off, t.bases)
    o.fill(t)
    p = h.generate_path()
    q = o.generate_path()
    assert np.array_equal(p[0], q[0])
    h.reweight_from_path(p[0], 0.2)
    o.reweight_path(q[0], 0.2)
    h.add_observation('A', 'C', 3, 4)           # staged on the host, flushed by the next lookup
    o.add(0, 1, 3, 4)
    assert h.reweight_observation('A', 'C', 3, 4, 0.5) == o.reweight_obs(0, 1, 3, 4, 0.5)
    p2, q2 = h.generate_path(), o.generate_path()
    assert np.array_equal(p2[0], q2[0]) and p2[1:] == q2[1]
    assert np.array_equal(h.export_band(), o.export_band())
