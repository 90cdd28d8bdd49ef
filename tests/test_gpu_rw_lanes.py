"""The reweight kernel behind the candidate pools, k_rw<T, LP, COL> (gretel/gretel.py:79-98 + the table entries a path changes),
in each of its lane-group sizes: 8 lanes per position, 16 (bands of 9..32 with at most 16 lags, row conditionals: the table
entries dealt out over the group through LDS, the unchanged row sums taken from `cnt`) and 32 (wider bands, more lags, the column
conditionals) -- against the C oracle, bit for bit, and against itself with the round-4 shortcuts switched off."""
import numpy as np
import pytest

from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table, sprinkle_deletions
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu


def _table(kmax, lam, seed, dels=0.0):
    t = make_support_table(1500, 9000, k=None, seed=seed, k_lambda=lam, k_min=2, k_max=kmax)
    if dels:
        sprinkle_deletions(t, dels, seed=seed + 7)
    return t


def _pair(t, L, **kw):
    h = Hansel(t.n_snps, band=t.band, **kw)
    o = COracle(t.n_snps, t.band, kw.get("storage", "f32"), kw.get("cond_mode", "A"), kw.get("marginal_term", False))
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
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


# (band, L): 8-lane groups need both <= 8 (the enumerated walks: k_rwseg, with a k_rw at the end of the spin); 16-lane groups a band
# of 9..32 and at most 16 lags; everything else 32
SHAPES = [(8, 5), (12, 7), (20, 11), (32, 16), (20, 17), (40, 9)]


@pytest.mark.parametrize("kmax,L", SHAPES)
@pytest.mark.parametrize("cond", ["A", "B", "D", "C"])
@pytest.mark.parametrize("storage", ["f32", "f64"])
def test_every_lane_group_against_the_oracle(kmax, L, cond, storage):
    t = _table(kmax + 1, 0.6 * kmax, seed=900 + kmax + L)
    assert t.band == kmax
    h, o = _pair(t, L, cond_mode=cond, storage=storage)
    _same(h.spin(10), o.spin(10))
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("kmax,L", [(12, 7), (20, 11), (32, 16)])
@pytest.mark.parametrize("mt", [False, True])
def test_sixteen_lanes_with_deletions_and_the_marginal_term(kmax, L, mt):
    # a few five-candidate columns: the table keeps the symbol layout (no ranks), '-' rows and columns are written
    t = _table(kmax + 1, 0.6 * kmax, seed=77 + kmax, dels=0.02)
    h, o = _pair(t, L, marginal_term=mt)
    _same(h.spin(8), o.spin(8))
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("env", [{"GH_RW_CNT": "0"}, {"GH_RW_LP16": "0"}, {"GH_RW_CNT": "0", "GH_RW_LP16": "0"}])
def test_the_shortcuts_change_nothing(env, monkeypatch):
    t = _table(21, 12.0, seed=4242)
    h, _ = _pair(t, 11)
    want = h.spin(10)
    band = h.export_band()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    h2, _ = _pair(t, 11)
    got = h2.spin(10)
    _same(got, want)
    assert got["magnitude"].tolist() == want["magnitude"].tolist() or "GH_RW_LP16" in env      # (other workgroup shapes: another fixed summation tree)
    assert np.array_equal(h2.export_band(), band)


@pytest.mark.parametrize("storage", ["f32", "f64"])
def test_the_to_major_copy_follows_every_other_writer_of_the_band(storage):
    # conditional C over a band of 20: k_rw reads its columns from the to-major copy, which only k_rw itself keeps in step --
    # whatever else writes the band between two spins (single observations, a single reweight, a reweight along a path through the
    # one-call API) must make the next spin build the copy again
    t = _table(21, 12.0, seed=515)
    h, o = _pair(t, 11, cond_mode="C", storage=storage)
    h.snapshot_original(); o.snapshot_original()            # (hp_original: against one and the same snapshot on both sides)
    _same(h.spin(6), o.spin(6))
    syms = "ACGT"
    for k, (a, b, i) in enumerate([(0, 1, 40), (2, 2, 41), (3, 0, 700), (1, 3, 701)]):
        for _ in range(3 + k):
            h.add_observation(syms[a], syms[b], i, i + 1 + k)
            o.add(a, b, i, i + 1 + k)
    _same(h.spin(5), o.spin(5))
    assert h.reweight_observation("A", "C", 40, 41, 0.5) == o.reweight_obs(0, 1, 40, 41, 0.5)
    _same(h.spin(5), o.spin(5))
    p, _ = o.generate_path()
    got = h.generate_path()
    assert np.array_equal(got[0], p)
    assert h.reweight_from_path(got[0], 0.25) == pytest.approx(o.reweight_path(p, 0.25), rel=1e-12)
    _same(h.spin(5), o.spin(5))
    assert np.array_equal(h.export_band(), o.export_band())
    # a copy starts without the to-major copy (and with its own snapshot of the original marginals: hp_original is not compared)
    h2 = h.copy()
    r2, r1 = h2.spin(4), h.spin(4)
    assert np.array_equal(r2["paths"], r1["paths"]) and r2["hp_current"].tolist() == r1["hp_current"].tolist()
    assert r2["ratio"].tolist() == r1["ratio"].tolist()
    assert np.array_equal(h2.export_band(), h.export_band())
