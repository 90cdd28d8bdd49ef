"""Spins over WIDE bands -- reads of up to 49 / 101 / 201 SNPs (long reads; C5 is "long-read-style", real ones are longer) --
at lag counts 10, 24 and 40, binary32 and binary64 storage, a row conditional (A) and a column conditional (C): the reweight
kernels with 32 lanes per position taking several rounds over the distances (k_rw<T,32>, k_marg<T,true> at W > 32), the
candidate pools with packed states (L <= 32) and with states as bytes (L = 40), all against the C oracle on
libm's log10, bit for bit (paths, likelihoods, ratios, the reweighted tensor).  gretel/gretel.py:79-98,102-189."""
import numpy as np
import pytest

from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu

SHAPES = {48: dict(n=2400, reads=5000, lam=30.0), 100: dict(n=2000, reads=2600, lam=60.0), 200: dict(n=1800, reads=1500, lam=120.0)}


def table(band, seed=5, dels=0.0):
    sh = SHAPES[band]
    t = make_support_table(sh["n"], sh["reads"], k=None, seed=seed, k_lambda=sh["lam"], k_min=2, k_max=band + 1)
    ks = np.diff(t.off)
    if int(ks.max()) < band + 1:
        # one read of the full length so that the band is exactly what the test names
        r = int(np.argmax(ks))
        need = band + 1 - int(ks[r])
        start = int(t.rank[r])
        if start + band + 1 > t.n_snps:
            t.rank[r] = t.n_snps - (band + 1)
        ext = t.haplotypes[0, t.rank[r] + ks[r]: t.rank[r] + ks[r] + need]
        t.bases = np.concatenate([t.bases[:t.off[r + 1]], ext, t.bases[t.off[r + 1]:]])
        t.off = t.off.copy()
        t.off[r + 1:] += need
        order = np.argsort(t.rank, kind="stable")
        ks2 = np.diff(t.off)
        t.bases = np.concatenate([t.bases[t.off[i]:t.off[i + 1]] for i in order])
        t.off = np.concatenate([[0], np.cumsum(ks2[order])]).astype(np.int64)
        t.rank = np.ascontiguousarray(t.rank[order])
    if dels:
        rng = np.random.default_rng(seed + 100)
        b = t.bases.copy()
        b[rng.random(len(b)) < dels] = ord('-')
        t.bases = b
    assert t.band == band, (t.band, band)
    return t


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


@pytest.mark.parametrize("mode", ["A", "C"])
@pytest.mark.parametrize("storage", ["f32", "f64"])
@pytest.mark.parametrize("L", [10, 24, 40])
@pytest.mark.parametrize("band", [48, 100, 200])
def test_spin_over_a_wide_band(band, L, storage, mode):
    t = table(band)
    h = Hansel(t.n_snps, band=t.band, storage=storage, cond_mode=mode)
    o = COracle(t.n_snps, t.band, storage, mode)
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    h.L = L
    o.L = L
    res = same(h, o, 10)
    assert res["n"] >= 3
    # a second spin on the reweighted tensor: tables and pools carried over
    same(h, o, 3)


@pytest.mark.parametrize("band,L,mode,mt", [(48, 24, "E", True), (100, 10, "D", False), (200, 24, "B", True), (100, 40, "C", True)])
def test_wide_band_with_deletions_and_the_other_switches(band, L, mode, mt):
    # '-' read at 4 % of the bases (five-candidate positions: the symbol-indexed pools), the marginal term, the remaining conditionals
    t = table(band, seed=9, dels=0.04)
    h = Hansel(t.n_snps, band=t.band, cond_mode=mode, marginal_term=mt)
    o = COracle(t.n_snps, t.band, "f32", mode, mt)
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    h.L = L
    o.L = L
    same(h, o, 8)


@pytest.mark.parametrize("band", [48, 200])
def test_wide_band_through_the_reference_loop(band):
    # one generate_path, one reweight_hansel_from_path at a time (cmd.py:148-179), L as the fill leaves it (util.py:333)
    t = table(band, seed=6)
    h = Hansel(t.n_snps, band=t.band)
    o = COracle(t.n_snps, t.band)
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    assert h.L == o.L and h.L > 8
    h.snapshot_original()
    o.snapshot_original()
    for it in range(4):
        pg, po = h.generate_path(), o.generate_path()
        assert pg[0] is not None and po[0] is not None and np.array_equal(pg[0], po[0]), it
        assert tuple(pg[1:]) == tuple(po[1]), it
        ratio = max(pg[3], 0.01)
        mg, mo = h.reweight_from_path(pg[0], ratio), o.reweight_path(po[0], ratio)
        assert abs(mg - mo) <= 1e-10 * abs(mo)
    assert np.array_equal(h.export_band(), o.export_band())
