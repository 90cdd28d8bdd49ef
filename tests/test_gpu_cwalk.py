"""The candidate-pool segment walk (gretel_amd/csrc/cwalk.hpp: lag counts 6..128, spins) against the C oracle and the
serial walker: every lag count, pools kept across spins and dropped by a new fill, windows it must hand back (a
position with five candidates), holes, a stale table in the middle of a queue, the other conditionals and f64 storage."""
import os

import numpy as np
import pytest

from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle
from test_gpu_edges import _walk_mode, PINNED

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(PINNED, reason="GH_WALK / GH_WALK_THREADS pin another variant")]


def _pair(t, L=None, walk=None, **kw):
    with _walk_mode(walk):
        h = Hansel(t.n_snps, band=t.band, **kw)
    o = COracle(t.n_snps, t.band, kw.get("storage", "f32"), kw.get("cond_mode", "A"), kw.get("marginal_term", False))
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    if L is not None:
        h.L = L
        o.L = L
    return h, o


def _same(res, ref):
    assert res["n"] == ref["n"] and res["hole_at"] == ref["hole_at"]
    assert np.array_equal(res["paths"], ref["paths"])
    assert res["hp_current"].tolist() == ref["hp_current"].tolist()
    assert res["hp_original"].tolist() == ref["hp_original"].tolist()
    assert res["ratio"].tolist() == ref["ratio"].tolist()
    assert np.allclose(res["magnitude"], ref["magnitude"], rtol=1e-10, atol=0)


@pytest.mark.parametrize("L", [-6, 6, 7, 8, 9, 10, 12, 13, 16, 17, 18, 20, 24, 25, 31, 32, 33, 40])
def test_every_lag_count(L, monkeypatch):
    # long-read-style reads (k ~ Poisson(10)); L = 9 and up also exercises the table rows beyond the eighth lag in k_rw.
    # L = 6 in a window without a five-candidate position enumerates its 4^6 states (variant 3); -6: the pools all the same
    if L < 0:
        monkeypatch.setenv("GH_SEG6", "0")
        L = -L
        want = 4
    else:
        want = 3 if L == 6 else 4
    # (beyond 32 lags: k_cwalkg -- the table from global memory, states as bytes next to their hash)
    t = make_support_table(2500, 30000, k=None, seed=200 + L, k_max=max(21, L + 4), k_lambda=10.0 if L <= 24 else float(L))
    h, o = _pair(t, L=L)
    res, ref = h.spin(24), o.spin(24)
    assert h.walk_clock()[3] == want
    _same(res, ref)
    assert np.array_equal(h.export_band(), o.export_band())
    hs, _ = _pair(t, L=L, walk="spec")
    _same(hs.spin(24), res)


def test_pools_persist_across_spins_and_reset_on_fill():
    t = make_support_table(4000, 50000, k=None, seed=31)
    h, o = _pair(t)
    assert 6 <= h.L <= 16
    for n in (10, 7, 15):
        _same(h.spin(n), o.spin(n))
    serial_before = h.walk_clock()[1]
    _same(h.spin(20), o.spin(20))
    assert h.walk_clock()[1] - serial_before <= 6          # the pools know the tracks by now: few paths go serial
    t2 = make_support_table(4000, 50000, k=None, seed=32)
    h.clear()
    o2 = COracle(t2.n_snps, t2.band)
    assert h.fill_from_support(t2.rank, t2.off, t2.bases) == o2.fill(t2)
    _same(h.spin(12), o2.spin(12))
    assert np.array_equal(h.export_band(), o2.export_band())


@pytest.mark.parametrize("L", [6, 7, 11, 16, 19, 21, 22, 24, 25, 30, 32, 33, 37, 40, 41, 45])
def test_window_with_five_candidates(L):
    # a position that shows A, C, G, T and '-': the conditional table is over the symbols, not over candidate ranks; the
    # pools then hold 3-bit picks (k_cwalk<L, 5>: 21 lags fit a state; 22..40: k_cwalk2<24 | 28 | .. | 40, 5>, states as bytes,
    # round 6; beyond that k_cwalkg<5>)
    t = make_support_table(1500, 36000, k=None, seed=5 + L, k_max=max(24, L + 4), k_lambda=10.0 if L <= 21 else float(L))
    bases = t.bases.copy()
    bases[np.random.default_rng(1).random(len(bases)) < 0.1] = ord('-')
    t.bases = bases
    h, o = _pair(t, L=L)
    assert (h.candidate_masks()[1:] == 0x2F).any()
    res, ref = h.spin(14), o.spin(14)
    _same(res, ref)
    assert h.walk_clock()[3] == 4
    assert np.array_equal(h.export_band(), o.export_band())
    _same(h.spin(5), o.spin(5))
    hs, _ = _pair(t, L=L, walk="spec")
    _same(hs.spin(14), res)


def test_hole_and_stale_table_inside_a_queue():
    t = make_support_table(900, 6000, k=8, n_haps=1, err=0.0, seed=2)
    h, o = _pair(t, L=8)
    res, ref = h.spin(6), o.spin(6)
    _same(res, ref)
    assert res["n"] == 1 and res["hole_at"] >= 1
    t = make_support_table(2000, 30000, k=None, seed=8)
    os.environ["GH_SEG_FORCE_STALE"] = "5"
    try:
        h, o = _pair(t, L=9)
    finally:
        del os.environ["GH_SEG_FORCE_STALE"]
    _same(h.spin(14), o.spin(14))
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("kw", [dict(cond_mode="B"), dict(cond_mode="D"), dict(storage="f64"), dict(cond_mode="C"), dict(marginal_term=True)])
def test_switches(kw):
    # (conditional C / the marginal term: k_lt rebuilds the table in front of every path, the pools walk it all the same)
    t = make_support_table(1500, 20000, k=None, seed=77)
    h, o = _pair(t, L=7, **kw)
    _same(h.spin(12), o.spin(12))
    assert h.walk_clock()[3] == 4
    assert np.array_equal(h.export_band(), o.export_band())
    _same(h.spin(4), o.spin(4))


@pytest.mark.parametrize("L", [8, 18])
def test_chains_left_open_go_to_the_serial_walker_and_change_nothing(L, monkeypatch):
    # GH_CW_ROUND_CAP=1 (read when the handle is created): one round per launch, so most chains stay open, the host
    # re-queues, and the serial walker takes the paths -- its states join the pools (k_cseed).  Same results.
    # (run-on off: with it a walker follows a new track through the next segments itself and even one round closes most chains)
    t = make_support_table(3000, 36000, k=None, seed=77)
    monkeypatch.setenv("GH_CW_RUNON", "0")
    monkeypatch.setenv("GH_CW_ROUND_CAP", "1")
    h, o = _pair(t, L=L)
    monkeypatch.delenv("GH_CW_ROUND_CAP")
    res, ref = h.spin(16), o.spin(16)
    _same(res, ref)
    assert np.array_equal(h.export_band(), o.export_band())
    assert h.walk_clock()[1] >= 1                          # paths handed to the serial walker
    _same(h.spin(5), o.spin(5))


def test_the_order_of_the_reads_does_not_matter():
    # the pools start from a guess (the largest marginals); a rank-sorted and a shuffled table give the same paths
    t = make_support_table(3000, 40000, k=None, seed=91, n_haps=3, err=0.01)
    assert (np.diff(t.rank) >= 0).all()
    h, o = _pair(t)
    ref = o.spin(12)
    _same(h.spin(12), ref)
    assert h.walk_clock()[3] == 4
    perm = np.random.default_rng(2).permutation(t.n_reads)
    k = np.diff(t.off)
    off = np.concatenate([[0], np.cumsum(k[perm])]).astype(np.int64)
    idx = np.repeat(t.off[:-1][perm], k[perm]) + (np.arange(off[-1]) - np.repeat(off[:-1], k[perm]))
    h3 = Hansel(t.n_snps, band=t.band)
    h3.fill_from_support(t.rank[perm], off, t.bases[idx])
    _same(h3.spin(12), ref)


@pytest.mark.parametrize("L,dels", [(8, False), (18, False), (30, False), (9, True)])
def test_lone_paths_come_out_of_the_pools_too(L, dels):
    # gh_generate_path (gretel.py:102-189 as the reference's own loop calls it, followed by reweight_hansel_from_path):
    # the path comes out of the candidate pools, which keep what they learnt from call to call
    t = make_support_table(1500, 24000, k=None, seed=300 + L, k_max=max(21, L + 4), k_lambda=10.0 if L <= 24 else float(L))
    if dels:
        bases = t.bases.copy()
        bases[np.random.default_rng(4).random(len(bases)) < 0.1] = ord('-')
        t.bases = bases
    h, o = _pair(t, L=L)
    variants = []
    for it in range(5):
        pg, po = h.generate_path(), o.generate_path()
        if po[0] is None:
            assert pg[0] is None and pg[1] == po[1]
            break
        assert np.array_equal(pg[0], po[0])
        assert pg[1:] == po[1]
        variants.append(h.walk_clock()[3])              # (a chain the queued rounds leave open goes to the serial walker)
        ratio = max(pg[3], 0.01)
        rg, ro = h.reweight_from_path(pg[0], ratio), o.reweight_path(po[0], ratio)
        assert abs(rg - ro) <= 1e-9 * max(1.0, abs(ro))
    assert 4 in variants, variants
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("L", [7, 9, 12, 19])
@pytest.mark.parametrize("dels", [False, True])
def test_run_on_changes_nothing(L, dels, monkeypatch):
    # a walker whose exit state the next pool does not hold walks on into the next segments itself (up to CW_RUNON of them)
    # and leaves the walk with its request to join: same paths with it and without it, with one round per launch and with many
    t = make_support_table(4000, 48000, k=None, seed=400 + L, k_max=max(21, L + 4))
    if dels:
        bases = t.bases.copy()
        bases[np.random.default_rng(7).random(len(bases)) < 0.08] = ord('-')
        t.bases = bases
    h, o = _pair(t, L=L)
    ref = o.spin(30)
    _same(h.spin(30), ref)
    assert h.walk_clock()[3] == 4
    assert np.array_equal(h.export_band(), o.export_band())
    monkeypatch.setenv("GH_CW_RUNON", "0")
    monkeypatch.setenv("GH_CW_SKIP0", "0")              # ... and with the chain followed behind every round, the first included
    h0, _ = _pair(t, L=L)
    _same(h0.spin(30), ref)
    monkeypatch.delenv("GH_CW_RUNON")
    monkeypatch.delenv("GH_CW_SKIP0")
    monkeypatch.setenv("GH_CW_ROUND_CAP", "1")
    h1, _ = _pair(t, L=L)
    _same(h1.spin(30), ref)


@pytest.mark.parametrize("L", [33, 34, 36, 37, 40, 41, 44, 45, 47, 48, 49, 52, 53, 57, 60, 61, 64])
@pytest.mark.parametrize("spec", ["A", "E+mt"])
def test_33_to_48_lags_through_registers(L, spec, monkeypatch):
    """k_cwalk2<36 | 40 | ... | 64> (round 6): 33..64 lags over ranks with k_cwalk's step -- the last LC picks' row offsets in
    registers, a block of LC steps unrolled over two chunks of the slice (four beyond 48 lags), lags beyond L as blocks of zeros -- on k_cwalkg's pools
    (states as bytes next to their hash).  The oracle's paths; the same with k_cwalkg (GH_CWALK2=0), without run-on and with the
    chain followed behind every round; every lag count that shares an instantiation with another."""
    kw = dict(cond_mode="E", marginal_term=True) if spec == "E+mt" else {}
    t = make_support_table(2500, 30000, k=None, seed=700 + L, k_max=L + 4, k_lambda=float(L))
    h, o = _pair(t, L=L, **kw)
    res, ref = h.spin(16), o.spin(16)
    assert h.walk_clock()[3] == 4
    _same(res, ref)
    assert np.array_equal(h.export_band(), o.export_band())
    if L in (33, 41, 48, 53, 64):
        monkeypatch.setenv("GH_CWALK2", "0")
        hg, _ = _pair(t, L=L, **kw)
        _same(hg.spin(16), ref)
        monkeypatch.delenv("GH_CWALK2")
        monkeypatch.setenv("GH_CW_RUNON", "0")
        monkeypatch.setenv("GH_CW_SKIP0", "0")
        h0, _ = _pair(t, L=L, **kw)
        _same(h0.spin(16), ref)


def test_33_to_48_lags_short_window_and_lone_paths():
    # a window shorter than two segments (the block of LC steps never completes), then gh_generate_path out of the same pools
    t = make_support_table(70, 2500, k=None, seed=77, k_max=45, k_lambda=30.0)
    h, o = _pair(t, L=39)
    _same(h.spin(10), o.spin(10))
    t = make_support_table(1200, 20000, k=None, seed=78, k_max=46, k_lambda=38.0)
    h, o = _pair(t, L=42)
    for it in range(4):
        pg, po = h.generate_path(), o.generate_path()
        if po[0] is None:
            assert pg[0] is None and pg[1] == po[1]
            break
        assert np.array_equal(pg[0], po[0]) and pg[1:] == po[1]
        ratio = max(pg[3], 0.01)
        h.reweight_from_path(pg[0], ratio); o.reweight_path(po[0], ratio)
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("L", [65, 80])
def test_beyond_64_lags_bytes_and_ring(L):
    # k_cwalkg (states walked out of a byte ring) still serves what k_cwalk2 does not: here 65 and 80 lags over ranks
    t = make_support_table(1500, 16000, k=None, seed=900 + L, k_max=L + 4, k_lambda=float(L))
    h, o = _pair(t, L=L)
    _same(h.spin(10), o.spin(10))
    assert h.walk_clock()[3] == 4
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("L", [22, 29, 36, 40])
@pytest.mark.parametrize("spec", ["A", "E+mt"])
def test_22_to_40_lags_over_the_symbols(L, spec, monkeypatch):
    """k_cwalk2<LC, 5> (round 6): a window with deletion COLUMNS (a five-candidate position here and there, what a pileup shows)
    at 22..40 lags -- eight lanes per pool entry, the table over the symbols, a block of LC steps over one, two or four chunks of
    the slice.  The oracle's paths; the same with k_cwalkg<5> and without run-on."""
    from gretel_amd.synth import sprinkle_deletions
    kw = dict(cond_mode="E", marginal_term=True) if spec == "E+mt" else {}
    t = make_support_table(2500, 30000, k=None, seed=800 + L, k_max=L + 4, k_lambda=float(L))
    sprinkle_deletions(t, 0.02, seed=L)
    h, o = _pair(t, L=L, **kw)
    assert (h.candidate_masks()[1:] == 0x2F).any()
    res, ref = h.spin(16), o.spin(16)
    assert h.walk_clock()[3] == 4
    _same(res, ref)
    assert np.array_equal(h.export_band(), o.export_band())
    monkeypatch.setenv("GH_CWALK2", "0")
    hg, _ = _pair(t, L=L, **kw)
    _same(hg.spin(16), ref)
    monkeypatch.delenv("GH_CWALK2")
    monkeypatch.setenv("GH_CW_RUNON", "0")
    monkeypatch.setenv("GH_CW_SKIP0", "0")
    h0, _ = _pair(t, L=L, **kw)
    _same(h0.spin(16), ref)
