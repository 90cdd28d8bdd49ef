"""The segment-parallel path extension (gretel_amd/csrc/segwalk.hpp) against the C oracle and against the serial
walker it replaces: every L it serves (1..5), both state radices (ranked 4-candidate layout, 5-symbol layout with '-'
as a fifth candidate), segments longer than one LDS chunk, holes, the re-queue after a stale table, and the
one-path API."""
import os

import numpy as np
import pytest

from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle
from test_gpu_edges import _walk_mode, PINNED

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(PINNED, reason="GH_WALK / GH_WALK_THREADS pin another variant")]


def _with_dels(t, frac, seed):
    bases = t.bases.copy()
    bases[np.random.default_rng(seed).random(len(bases)) < frac] = ord('-')
    t.bases = bases
    return t


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
    assert np.allclose(res["magnitude"], ref["magnitude"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("wide", [False, True])
def test_every_lag_count_and_both_radices(L, wide):
    # wide: 10 % '-' on top of four bases with errors -> positions with five candidates -> the 5-symbol state space
    t = make_support_table(900, 30000, k=6, seed=100 + L)
    if wide:
        t = _with_dels(t, 0.1, L)
    h, o = _pair(t, L=L)
    res, ref = h.spin(6), o.spin(6)
    assert h.walk_clock()[3] == 3
    _same(res, ref)
    assert np.array_equal(h.export_band(), o.export_band())
    if wide:
        assert (h.candidate_masks()[1:] == 0x2F).any()        # some position really shows A C G T and '-'
    # and the serial walker agrees bit for bit
    hs, _ = _pair(t, L=L, walk="spec")
    _same(hs.spin(6), res)


@pytest.mark.parametrize("n,L", [(20000, 2), (30000, 1), (17000, 3)])
def test_segments_longer_than_one_chunk(n, L):
    # 256 segments at most: beyond 256 x 64 positions a segment spans several LDS chunks of k_seg
    t = make_support_table(n, 12 * n, k=4, seed=7)
    h, o = _pair(t, L=L)
    _same(h.spin(3), o.spin(3))
    assert h.walk_clock()[3] == 3


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 9, 15, 16, 17, 63, 64, 65, 129])
def test_window_lengths_around_the_segment_size(n):
    t = make_support_table(max(n, 2), 200 + 40 * n, k=min(3, max(n, 2)), seed=n)
    if n == 1:
        return
    h, o = _pair(t)
    _same(h.spin(4), o.spin(4))


def test_one_path_api_and_original_marginals():
    t = make_support_table(400, 12000, k=5, seed=5)
    h, o = _pair(t)
    h.snapshot_original()
    o.snapshot_original()
    for _ in range(5):
        pg, po = h.generate_path(), o.generate_path()
        assert np.array_equal(pg[0], po[0])
        assert pg[1:] == po[1]                         # hp_current, hp_original, min marginal (unclamped)
        r = max(pg[3], 0.01)
        assert abs(h.reweight_from_path(pg[0], r) - o.reweight_path(po[0], r)) <= 1e-12 * r * t.n_snps
    assert np.array_equal(h.export_band(), o.export_band())


def test_hole_stops_the_queue():
    t = make_support_table(300, 2000, k=3, n_haps=1, err=0.0, seed=1)
    h, o = _pair(t)
    res, ref = h.spin(6), o.spin(6)
    _same(res, ref)
    assert res["n"] == 1 and res["hole_at"] >= 1
    assert h.spin(3)["n"] == 0                         # and stays stopped


def test_requeue_after_a_stale_table_changes_nothing():
    # GH_SEG_FORCE_STALE=k: path k of the spin finds the conditional table stale (as when a candidate mask moves under
    # a reweight): the queue behind it idles, the host rebuilds the table and queues the remaining paths again
    t = make_support_table(600, 20000, k=5, seed=3)
    os.environ["GH_SEG_FORCE_STALE"] = "4"
    try:
        h, o = _pair(t)
    finally:
        del os.environ["GH_SEG_FORCE_STALE"]
    res, ref = h.spin(12), o.spin(12)
    c_ = h.walk_clock(); assert (c_[0], c_[3]) == (1, 3)          # one re-queue
    _same(res, ref)
    assert np.array_equal(h.export_band(), o.export_band())
    h2, _ = _pair(t)
    _same(h2.spin(12), res)
    assert h2.walk_clock()[0] == 0


@pytest.mark.parametrize("kw", [dict(cond_mode="B"), dict(cond_mode="C"), dict(cond_mode="D"), dict(marginal_term=True), dict(storage="f64"),
                                dict(storage="f64", cond_mode="B", marginal_term=True)])
def test_every_switch(kw):
    # conditional C and the marginal term rebuild the table in full before every path (no incremental rows)
    t = _with_dels(make_support_table(500, 15000, k=5, n_haps=4, seed=21), 0.05, 2)
    h, o = _pair(t, **kw)
    _same(h.spin(5), o.spin(5))
    assert h.walk_clock()[3] == 3
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("kw", [dict(), dict(cond_mode="C", marginal_term=True), dict(storage="f64", cond_mode="B")])
def test_three_launches_per_path(L, wide, kw, monkeypatch):
    # Spins of large windows run without k_emit: k_seg / k_scan carry the minimum marginal of every entry state's picks, and
    # k_rw chains the group maps itself (its entry states, its picks, the path's minimum marginal).  GH_FUSE=2 forces that
    # path on windows small enough for k_emit_small; GH_FUSE=0 switches it off.  Same paths, likelihoods, ratios, tensors.
    from spec_util import make_pair, same, with_dels
    t = make_support_table(2600, 60000, k=6, seed=300 + L)
    if wide:
        if L == 5:
            pytest.skip("5^5 states: the group maps do not fit k_rw's LDS; k_emit stays")
        t = with_dels(t, 0.1, L)
    monkeypatch.setenv("GH_FUSE", "2")
    h, o = make_pair(t, L=L, **kw)
    h.profile_enable(1)
    h.profile_reset()
    res, ref = h.spin(9), o.spin(9)
    same(res, ref)
    assert h.walk_clock()[3] == 3
    assert np.array_equal(h.export_band(), o.export_band())
    monkeypatch.setenv("GH_FUSE", "0")
    h2, _ = make_pair(t, L=L, **kw)
    same(h2.spin(9), ref)
    assert np.array_equal(h2.export_band(), o.export_band())


def test_three_launches_with_a_hole_and_a_stale_table(monkeypatch):
    from spec_util import make_pair, same
    monkeypatch.setenv("GH_FUSE", "2")
    t = make_support_table(300, 2000, k=3, n_haps=1, err=0.0, seed=1)
    h, o = make_pair(t)
    res, ref = h.spin(6), o.spin(6)
    same(res, ref)
    assert res["n"] == 1 and res["hole_at"] >= 1
    t = make_support_table(1500, 40000, k=5, seed=3)
    monkeypatch.setenv("GH_SEG_FORCE_STALE", "4")
    h, o = make_pair(t)
    monkeypatch.delenv("GH_SEG_FORCE_STALE")
    res, ref = h.spin(12), o.spin(12)
    c_ = h.walk_clock(); assert (c_[0], c_[3]) == (1, 3)
    same(res, ref)
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("kw", [dict(), dict(cond_mode="B"), dict(cond_mode="D", marginal_term=True), dict(storage="f64"),
                                dict(cond_mode="B", cand_order="-TGCA"), dict(offer_zero=True),
                                dict(cond_mode="C", marginal_term=True), dict(cond_mode="E"), dict(cond_mode="E", marginal_term=True, storage="f64"),
                                dict(cond_mode="C", cand_order="GA-TC")],
                         ids=lambda kw: "-".join("%s=%s" % x for x in sorted(kw.items())) or "default")
def test_reweight_rides_in_the_next_paths_launch(L, wide, kw, monkeypatch):
    # k_rwseg: the reweight of path k-1 and the k_seg of path k in one launch; every workgroup recomputes the table rows of the
    # L positions in front of its segment from the copy k_emit made of its neighbour's band blocks.  Windows whose k_scan and
    # k_emit are separate launches (the default there); GH_RWSEG=0 gives the four-launch flow.  Same everything.
    from spec_util import make_pair, same, with_dels
    t = make_support_table(2700, 60000, k=6, seed=700 + L)
    if wide:
        t = with_dels(t, 0.1, L)
    h, o = make_pair(t, L=L, **kw)
    res, ref = h.spin(11), o.spin(11)
    same(res, ref)
    assert h.walk_clock()[3] == 3
    assert np.array_equal(h.export_band(), o.export_band())
    same(h.spin(3), o.spin(3))                                 # a second spin continues from the reweighted tensor
    monkeypatch.setenv("GH_RWSEG", "0")
    h2, o2 = make_pair(t, L=L, **kw)
    same(h2.spin(11), ref)


def test_reweight_in_the_next_launch_with_a_hole_and_a_stale_table(monkeypatch):
    from spec_util import make_pair, same
    # one haplotype: the first reweight empties the matrix, the second path ends in a hole (gretel.py:176-180)
    t = make_support_table(3000, 20000, k=3, n_haps=1, err=0.0, seed=1)
    h, o = make_pair(t)
    res, ref = h.spin(6), o.spin(6)
    same(res, ref)
    assert res["n"] == 1 and res["hole_at"] >= 1
    assert h.spin(3)["n"] == 0
    # a table found stale at path 4: the queue behind idles, the host rebuilds and queues the remaining paths again
    t = make_support_table(2600, 60000, k=5, seed=3)
    monkeypatch.setenv("GH_SEG_FORCE_STALE", "4")
    h, o = make_pair(t)
    monkeypatch.delenv("GH_SEG_FORCE_STALE")
    res, ref = h.spin(12), o.spin(12)
    c_ = h.walk_clock(); assert (c_[0], c_[3]) == (1, 3)
    same(res, ref)
    assert np.array_equal(h.export_band(), o.export_band())
    # a spin of ONE path, and of two
    h, o = make_pair(t)
    same(h.spin(1), o.spin(1))
    same(h.spin(2), o.spin(2))
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("n,k,mode,storage", [(20011, 5, "C", "f64"), (12000, 9, "E", "f64"), (20011, 9, "C", "f32")])
def test_long_segments_under_the_column_conditionals(n, k, mode, storage):
    # k_rwseg stages the band blocks of a workgroup's positions in LDS under conditionals C / E: segments of 79 positions of a
    # wide band in binary64 do not fit 160 KB -- the spin must notice and keep the four launches (a launch with too much LDS
    # fails with "invalid argument": found by the fuzz), with the same paths either way
    t = make_support_table(n, n * 6, k=k, n_haps=4, seed=5)
    h = Hansel(t.n_snps, band=t.band, cond_mode=mode, storage=storage)
    o = COracle(t.n_snps, t.band, storage, mode)
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    res, ref = h.spin(5), o.spin(5)
    assert res["n"] == ref["n"] == 5 and np.array_equal(res["paths"], ref["paths"])
    assert res["hp_current"].tolist() == ref["hp_current"].tolist()
    assert np.array_equal(h.export_band(), o.export_band())
