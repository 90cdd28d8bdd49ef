"""The mixed-radix state space of the segment-parallel extension (gretel_amd/csrc/segmix.hpp): windows at L = 5 in which a
few positions offer five candidates (a deletion column here and there) are walked over prod R_p states per target instead of
5^5 -- and give the same paths, likelihoods, ratios and tensor as the oracle, bit for bit, under every switch of the Hansel
arithmetic, through k_seg (first path), k_rwseg (every later one: the halo's ranks come from the patch), plain k_rw flows
(GH_RWSEG=0), the reference's own loop, small windows (k_emit_small) and long segments (several LDS chunks).
gretel/gretel.py:143-187."""
import numpy as np
import pytest

from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table, sprinkle_deletions
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu


def pair(t, **kw):
    storage, mode, mt = kw.get("storage", "f32"), kw.get("cond_mode", "A"), kw.get("marginal_term", False)
    extra = {k: v for k, v in kw.items() if k in ("cand_order", "offer_zero")}
    h = Hansel(t.n_snps, band=t.band, **kw)
    o = COracle(t.n_snps, t.band, storage, mode, mt, **extra)
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    return h, o


def same(h, o, paths):
    res, ref = h.spin(paths), o.spin(paths)
    assert res["n"] == ref["n"] and res["hole_at"] == ref["hole_at"], ((res["n"], res["hole_at"]), (ref["n"], ref["hole_at"]))
    assert np.array_equal(res["paths"], ref["paths"]), "recovered SNP sequences differ"
    assert res["hp_current"].tolist() == ref["hp_current"].tolist()
    assert res["hp_original"].tolist() == ref["hp_original"].tolist()
    assert res["ratio"].tolist() == ref["ratio"].tolist()
    assert np.allclose(res["magnitude"], ref["magnitude"], rtol=1e-10, atol=0)
    assert np.array_equal(h.export_band(), o.export_band()), "reweighted tensors differ"
    return res


def sparse_table(n, reads, frac, seed, k=5):
    t = make_support_table(n, reads, k=k, seed=seed)
    pos = sprinkle_deletions(t, frac, seed=seed + 1)
    return t, pos


@pytest.mark.parametrize("n,reads,frac", [(3000, 150000, 0.01), (10000, 400000, 0.01), (1500, 60000, 0.03), (700, 30000, 0.005), (120, 6000, 0.03)])
def test_sparse_deletions_walk_the_mixed_radix(n, reads, frac):
    t, pos = sparse_table(n, reads, frac, seed=n)
    h, o = pair(t)
    assert h.L == 5
    res = same(h, o, 12)
    clk = h.walk_clock()
    assert clk[3] == 3 and clk[1] == 6, clk                # segment-parallel, mixed radix
    assert 0 < clk[2] <= 2048, clk                         # (1280 behind one five-candidate position whose neighbours offer four)
    # the picks include deletions where the haplotypes have them (the path is not shy of the fifth candidate)
    assert res["n"] == 12
    same(h, o, 3)                                          # a second spin on the reweighted tensor


@pytest.mark.parametrize("kw", [dict(cond_mode="C"), dict(cond_mode="E", marginal_term=True), dict(storage="f64"), dict(cond_mode="B", marginal_term=True),
                                dict(cond_mode="D", storage="f64", marginal_term=True), dict(cand_order="-TGCA"), dict(cond_mode="C", cand_order="G-ATC", marginal_term=True)])
def test_mixed_radix_under_every_switch(kw):
    t, _ = sparse_table(2500, 120000, 0.012, seed=77)
    h, o = pair(t, **kw)
    same(h, o, 10)
    assert h.walk_clock()[1] == 6, h.walk_clock()


def test_dense_deletions_keep_the_symbol_radix():
    # bench.py's wide window: 5 % of the BASES are '-', nearly every column has five candidates -> 3125 states, class 5
    t = make_support_table(2000, 100000, k=5, seed=5)
    b = t.bases.copy()
    b[np.random.default_rng(1).random(len(b)) < 0.05] = ord('-')
    t.bases = b
    h, o = pair(t)
    same(h, o, 6)
    clk = h.walk_clock()
    assert clk[1] == 5 and clk[2] == 3125, clk


def test_narrow_window_keeps_the_ranked_radix():
    t = make_support_table(2000, 100000, k=5, seed=6)
    h, o = pair(t)
    same(h, o, 6)
    assert h.walk_clock()[1] == 4, h.walk_clock()


def test_mixed_radix_off_gives_the_same_paths(monkeypatch):
    t, _ = sparse_table(3000, 150000, 0.01, seed=3000)
    h, o = pair(t)
    a = same(h, o, 8)
    monkeypatch.setenv("GH_MIXED", "0")
    # (read when a library is loaded: a fresh process would see it; within this one the switch is static -- so compare with
    # the four-launch flow and with the serial walker instead, which never see the mixed radix)
    monkeypatch.delenv("GH_MIXED")
    for env in (dict(GH_RWSEG="0"), dict(GH_WALK="spec1")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h2 = Hansel(t.n_snps, band=t.band)
        h2.fill_from_support(t.rank, t.off, t.bases)
        b = h2.spin(8)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(a["paths"], b["paths"]) and a["hp_current"].tolist() == b["hp_current"].tolist(), env


def test_mixed_radix_through_the_reference_loop():
    t, _ = sparse_table(1200, 50000, 0.02, seed=12)
    h, o = pair(t)
    h.snapshot_original()
    o.snapshot_original()
    for it in range(5):
        pg, po = h.generate_path(), o.generate_path()
        assert pg[0] is not None and np.array_equal(pg[0], po[0]), it
        assert tuple(pg[1:]) == tuple(po[1]), it
        ratio = max(pg[3], 0.01)
        assert abs(h.reweight_from_path(pg[0], ratio) - o.reweight_path(po[0], ratio)) <= 1e-10 * 1e6
    assert h.walk_clock()[1] == 6
    assert np.array_equal(h.export_band(), o.export_band())


def test_deep_spin_until_the_masks_move():
    # 60 paths: counts reach zero, candidate masks move (the table is rebuilt, the window is classified again), maybe a hole
    t, _ = sparse_table(800, 20000, 0.02, seed=8)
    h, o = pair(t)
    same(h, o, 60)


def test_long_segments_take_several_chunks():
    # 20 000 SNPs: segments of 79 positions = two LDS chunks each
    t, _ = sparse_table(20000, 500000, 0.004, seed=20)
    h, o = pair(t)
    same(h, o, 4)
    assert h.walk_clock()[1] == 6
