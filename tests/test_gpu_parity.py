"""GPU parity: the HIP path (through the C ABI) against the oracle on the same seeded inputs.
Integer/byte/index results are bit-exact; the path likelihoods are bit-exact against the C
oracle AND the Python restatement, both on libm's log10 -- the kernels' log10 (include/gh_detlog.h) is that
function (the bar in BASELINE.json is 1e-6); the removed mass is a parallel sum, checked to 1e-12 relative."""
import os

import numpy as np
import pytest

from conftest import REFDATA
from gretel_amd import gretel, util
from gretel_amd.hansel import Hansel, SYMBOLS, UNSYMBOLS
from gretel_amd.synth import make_support_table
from oracle import gretel_ref as G
from oracle.c_oracle import COracle, paths_to_str
from oracle import hansel_ref

pytestmark = pytest.mark.gpu

BAM = os.path.join(REFDATA, "test.bam")
VCF = os.path.join(REFDATA, "test.vcf.gz")


def _pair(t, storage="f32", mode="A", mt=False, band=None, L=None):
    band = band or t.band
    h = Hansel(t.n_snps, band=band, storage=storage, cond_mode=mode, marginal_term=mt)
    st = h.fill_from_support(t.rank, t.off, t.bases)
    o = COracle(t.n_snps, band, storage, mode, mt, use_libm=True)
    assert o.fill(t) == st
    assert o.L == h.L
    if L is not None:
        h.L = L
        o.L = L
    return h, o


def _same_spin(res, ref):
    assert res["n"] == ref["n"]
    assert res["hole_at"] == ref["hole_at"]
    assert np.array_equal(res["paths"], ref["paths"])
    assert res["hp_current"].tolist() == ref["hp_current"].tolist()
    assert res["hp_original"].tolist() == ref["hp_original"].tolist()
    assert res["ratio"].tolist() == ref["ratio"].tolist()
    assert np.allclose(res["magnitude"], ref["magnitude"], rtol=1e-12, atol=0)


# -- the reference's own known answers, through the product path ---------------------------------
def test_reference_fixture_known_answers():
    # reference tests/test_test.py:31-52 (threads 1 and 2 give the same observations)
    for threads in (1, 2):
        v = util.process_vcf(VCF, 'hoot', 1, 20)
        h = util.load_from_bam(BAM, 'hoot', 1, 20, v, n_threads=threads)
        assert h.n_slices == 5
        assert h.n_crumbs == 9
        assert h.L > 0 and h.L == 3
        for a, b, i, j, want in [('_', 'A', 0, 1, 1), ('A', 'A', 1, 2, 1), ('A', 'A', 1, 3, 1), ('A', 'A', 1, 4, 0),
                                 ('C', 'C', 1, 2, 1), ('C', 'C', 1, 3, 1), ('C', 'C', 1, 4, 0), ('T', 'T', 1, 2, 2),
                                 ('G', 'G', 1, 2, 0), ('G', 'G', 2, 3, 0), ('G', 'G', 3, 4, 1), ('G', '_', 4, 5, 1)]:
            assert h.get_observation(a, b, i, j) == want
        assert h.export_dense().sum() == 14
        assert h.gap_check() == -1
        assert [h.get_counts_at(i)["total"] for i in range(5)] == [4, 4, 2, 1, 1]


def test_fixture_end_to_end_matches_python_oracle():
    v = util.process_vcf(VCF, 'hoot', 1, 20)
    h = util.load_from_bam(BAM, 'hoot', 1, 20, v)
    orig = h.copy()
    rank, off, bases = util.support_table_from_bam(BAM, 'hoot', 1, 20, v)
    ph = hansel_ref.Hansel.init_matrix(SYMBOLS, UNSYMBOLS, v["N"])
    G.fill_from_support(ph, [(int(rank[i]), bases[off[i]:off[i + 1]].tobytes().decode()) for i in range(len(rank))], v["N"])
    recs, _ = G.recover_paths(ph, v["N"], 10)
    for rec in recs:
        path, prob, mn = gretel.generate_path(v["N"], h, orig)
        assert "".join(str(x) for x in path) == rec["path"]
        assert prob["hp_current"] == rec["hp_current"]
        assert prob["hp_original"] == rec["hp_original"]
        mn = max(mn, 0.01)
        assert abs(mn - rec["ratio"]) < 1e-12
        mag = gretel.reweight_hansel_from_path(h, path, mn)
        assert abs(mag - rec["magnitude"]) < 1e-9
    assert np.array_equal(h.export_dense(), ph.dense().astype(np.float64))


# -- fill ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,reads,k,seed", [(60, 2000, 4, 0), (1000, 50000, 3, 1), (400, 30000, None, 2), (50, 3, 2, 3)])
def test_fill_bit_exact(n, reads, k, seed):
    t = make_support_table(n, reads, k=k, seed=seed)
    h, o = _pair(t)
    assert np.array_equal(h.export_band(), o.export_band())
    assert (h.n_slices, h.n_crumbs) == o.stats()[:2]
    assert h.gap_check() == o.gap_check()


def test_fill_with_end_sentinels_and_odd_symbols():
    t = make_support_table(80, 3000, k=4, seed=9)
    rng = np.random.default_rng(0)
    bases = t.bases.copy()
    bases[rng.random(len(bases)) < 0.05] = ord('N')
    bases[rng.random(len(bases)) < 0.05] = ord('-')
    bases[rng.random(len(bases)) < 0.01] = ord('_')
    t.bases = bases
    h = Hansel(t.n_snps, band=t.band)
    st = h.fill_from_support(t.rank, t.off, t.bases, use_end_sentinels=True)
    o = COracle(t.n_snps, t.band)
    o_st = None
    from oracle.c_oracle import lib, _p
    rc = lib().orc_fill(o._h, _p(np.ascontiguousarray(t.rank)), _p(np.ascontiguousarray(t.off)), _p(bases), t.n_reads, 1)
    assert rc == 0
    assert st == o.stats()
    assert np.array_equal(h.export_band(), o.export_band())


def test_fill_empty_and_single_snp_reads():
    h = Hansel(10, band=2)
    rank = np.array([0, 3, 5], dtype=np.int32)
    off = np.array([0, 1, 1, 2], dtype=np.int64)       # k = 1, 0, 1: none carries evidence (util.py:230)
    assert h.fill_from_support(rank, off, np.frombuffer(b"AC", dtype=np.uint8)) == (0, 0, 0)
    assert h.export_band().sum() == 0
    assert h.gap_check() == 0
    assert h.fill_from_support(np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros(0, np.uint8)) == (0, 0, 0)


def test_fill_rejects_bad_symbols_and_band_overflow():
    from gretel_amd._lib import SymbolError, BandError
    h = Hansel(10, band=1)
    with pytest.raises(SymbolError):
        h.fill_from_support([0], [0, 2], np.frombuffer(b"AX", dtype=np.uint8), reads_handle=None)
    h2 = Hansel(10, band=1)
    from gretel_amd.hansel import DeviceReads
    r = DeviceReads(Hansel(10, band=3), [0], [0, 3], np.frombuffer(b"ACG", dtype=np.uint8))
    r.max_k = 2                      # lie about the width: the kernel must notice
    with pytest.raises(BandError):
        h2.fill_from_support(None, None, None, reads_handle=r)


# -- lookups ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("storage,mode,mt", [("f32", "A", False), ("f32", "B", True), ("f32", "C", False), ("f32", "D", False), ("f64", "A", True)])
def test_lookups_match_oracle(storage, mode, mt):
    t = make_support_table(120, 5000, k=5, seed=3)
    h, o = _pair(t, storage, mode, mt)
    # make the counts ragged first
    p0 = o.generate_path()[0]
    o.reweight_path(p0, 0.37)
    h.reweight_from_path(p0, 0.37)
    rng = np.random.default_rng(1)
    for p in [0, 1, 2, 5, 60, t.n_snps - 1, t.n_snps]:
        assert h.counts_array(p).tolist() == o.counts_at(p).tolist()
        for s in range(7):
            assert h.get_marginal_of_at(s, p) == o.marginal(s, p)
    path = [6] + rng.choice([0, 1, 2, 3, 5], size=t.n_snps).tolist()
    for p in [1, 2, 3, 4, 5, 6, 50, t.n_snps]:
        mask, w = o.edge_weights(p, path)
        got = h.get_edge_weights_at(p, path)
        assert sorted(s.i for s in got) == [s for s in range(7) if (mask >> s) & 1]
        for s, val in got.items():
            assert val == w[s.i]
    for _ in range(20):
        a, b = rng.integers(0, 7, 2)
        i = int(rng.integers(0, t.n_snps))
        j = i + int(rng.integers(1, t.band + 2))
        assert h.get_observation(int(a), int(b), i, j) == o.get(int(a), int(b), i, j)


# -- path extension + reweight ------------------------------------------------------------------------
@pytest.mark.parametrize("storage,mode,mt,L", [("f32", "A", False, None), ("f32", "A", False, 1), ("f32", "A", False, 9),
                                              ("f32", "B", False, None), ("f32", "C", True, None), ("f32", "D", False, None),
                                              ("f64", "A", True, None), ("f64", "B", False, 2)])
def test_generate_and_reweight_step_by_step(storage, mode, mt, L):
    t = make_support_table(150, 6000, k=4, seed=6)
    h, o = _pair(t, storage, mode, mt, L=L)
    h.snapshot_original()
    o.snapshot_original()
    for it in range(6):
        pg = h.generate_path()
        po = o.generate_path()
        assert np.array_equal(pg[0], po[0])
        assert pg[1:] == po[1]
        ratio = max(pg[3], 0.01)
        mg = h.reweight_from_path(pg[0], ratio)
        mo = o.reweight_path(po[0], ratio)
        assert abs(mg - mo) <= 1e-12 * abs(mo)
        assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("n,reads,k,paths,seed", [(1000, 50000, 3, 100, 0), (300, 9000, 5, 40, 1), (2000, 20000, None, 30, 2)])
def test_spin_matches_oracle(n, reads, k, paths, seed):
    # first case = BASELINE.json config C2 (1k SNPs / 50k reads / L=3 / 100 paths)
    t = make_support_table(n, reads, k=k, seed=seed)
    h, o = _pair(t)
    _same_spin(h.spin(paths), o.spin(paths))
    assert np.array_equal(h.export_band(), o.export_band())


def test_spin_matches_python_oracle_small():
    t = make_support_table(40, 900, k=3, seed=8)
    h = Hansel(t.n_snps, band=t.band)
    h.fill_from_support(t.rank, t.off, t.bases)
    ph = hansel_ref.Hansel.init_matrix(SYMBOLS, UNSYMBOLS, t.n_snps)
    G.fill_from_support(ph, t.reads(), t.n_snps)
    recs, _ = G.recover_paths(ph, t.n_snps, 12)
    res = h.spin(12)
    assert [Hansel.path_str(p) for p in res["paths"]] == [r["path"] for r in recs]
    assert res["hp_current"].tolist() == [r["hp_current"] for r in recs]
    assert res["hp_original"].tolist() == [r["hp_original"] for r in recs]
    assert np.allclose(res["magnitude"], [r["magnitude"] for r in recs], rtol=1e-12)
    assert np.array_equal(h.export_dense(), ph.dense().astype(np.float64))


def test_hole_ends_recovery():
    t = make_support_table(30, 200, k=3, n_haps=1, err=0.0, seed=1)
    h, o = _pair(t)
    res, ref = h.spin(5), o.spin(5)
    _same_spin(res, ref)
    assert res["n"] == 1 and res["hole_at"] >= 1
    # the drop-in function reports it as the reference does: a None triple (gretel.py:180)
    assert gretel.generate_path(t.n_snps, h, h) == (None, None, None)


def test_gap_in_evidence_is_reported():
    t = make_support_table(40, 500, k=3, seed=2)
    keep = ~((t.rank <= 20) & (t.rank + 3 > 20))          # drop every read touching SNP 21 (pos 21)
    rank = t.rank[keep]
    ks = np.diff(t.off)[keep]
    off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int64)
    bases = np.concatenate([t.bases[t.off[i]:t.off[i + 1]] for i in np.flatnonzero(keep)])
    h = Hansel(40, band=2)
    h.fill_from_support(rank, off, bases)
    assert h.gap_check() in (20, 21)
    assert h.get_counts_at(h.gap_check()).get("total", 0) == 0


# -- per-cell compatibility API ---------------------------------------------------------------------
def test_per_cell_api_and_lazy_band():
    h = Hansel.init_matrix(SYMBOLS, UNSYMBOLS, 12)
    ph = hansel_ref.Hansel.init_matrix(SYMBOLS, UNSYMBOLS, 12)
    obs = [('_', 'A', 0, 1), ('A', 'C', 1, 2), ('A', 'C', 1, 2), ('A', 'G', 1, 5), ('C', '_', 12, 13), ('T', '-', 3, 4)]
    for ob in obs:
        h.add_observation(*ob)
        ph.add_observation(*ob)
    assert h.get_observation('A', 'C', 1, 2) == 2
    assert h.band == 4
    h.add_observation('G', 'G', 2, 9)                     # wider than the band: re-banded
    ph.add_observation('G', 'G', 2, 9)
    assert h.get_observation('G', 'G', 2, 9) == 1 and h.band == 7 and h.get_observation('A', 'C', 1, 2) == 2
    for (a, b, i, j) in obs[:3]:
        r1 = h.reweight_observation(a, b, i, j, 0.3)
        r2 = ph.reweight_observation(a, b, i, j, 0.3)
        assert r1 == r2
    assert np.array_equal(h.export_dense(), ph.dense().astype(np.float64))
    assert h.reweight_observation('A', 'A', 1, 12, 0.5) == 0.0      # outside the band: a zero cell
    c = h.copy()
    h.reweight_observation('A', 'C', 1, 2, 0.5)
    assert c.get_observation('A', 'C', 1, 2) != h.get_observation('A', 'C', 1, 2)
    sym = h.symbols_d['-']
    assert str(sym) == '-' and sym == h.symbols_d['-'] and sym != h.symbols_d['_']
    with pytest.raises(KeyError):
        h.add_observation('X', 'A', 1, 2)


def test_fill_sorted_and_shuffled_tables_agree():
    # rank-sorted tables take the LDS-counting kernel, shuffled ones the scattered-atomics kernel
    t = make_support_table(700, 40000, k=None, seed=12)
    assert (np.diff(t.rank) >= 0).all()
    h1 = Hansel(t.n_snps, band=t.band)
    st1 = h1.fill_from_support(t.rank, t.off, t.bases)
    perm = np.random.default_rng(0).permutation(t.n_reads)
    ks = np.diff(t.off)
    off2 = np.concatenate([[0], np.cumsum(ks[perm])]).astype(np.int64)
    bases2 = np.concatenate([t.bases[t.off[i]:t.off[i + 1]] for i in perm])
    h2 = Hansel(t.n_snps, band=t.band)
    st2 = h2.fill_from_support(t.rank[perm], off2, bases2)
    assert st1 == st2
    assert np.array_equal(h1.export_band(), h2.export_band())
    o = COracle(t.n_snps, t.band)
    assert o.fill(t) == st1
    assert np.array_equal(h1.export_band(), o.export_band())
    # a band narrower than the LDS slice logic expects, and the end sentinels, on the sorted path
    h3 = Hansel(t.n_snps, band=t.band)
    o3 = COracle(t.n_snps, t.band)
    from oracle.c_oracle import lib, _p
    assert lib().orc_fill(o3._h, _p(np.ascontiguousarray(t.rank)), _p(np.ascontiguousarray(t.off)), _p(t.bases), t.n_reads, 1) == 0
    assert h3.fill_from_support(t.rank, t.off, t.bases, use_end_sentinels=True) == o3.stats()
    assert np.array_equal(h3.export_band(), o3.export_band())


@pytest.mark.parametrize("k_max,lam", [(21, 10.0), (32, 24.0), (33, 24.0)])
def test_fill_of_long_reads_three_ways(k_max, lam, monkeypatch):
    # long reads over a wide band, where no LDS slice holds a workgroup's reads (C5's shape): 32 lanes per read (k_fill_pairs; reads
    # of at most 32 SNPs), one thread per read (k_fill), and the oracle -- with '-', N and '_' bases, reads at rank 0 and up to the
    # last SNP, with and without the end sentinels
    t = make_support_table(6000, 24000, k=None, seed=21, k_lambda=lam, k_max=k_max)
    rng = np.random.default_rng(3)
    bases = t.bases.copy()
    bases[rng.random(len(bases)) < 0.05] = ord('-')
    bases[rng.random(len(bases)) < 0.03] = ord('N')
    bases[rng.random(len(bases)) < 0.005] = ord('_')
    assert int(np.diff(t.off).max()) == k_max and (t.rank == 0).any() and int((t.rank + np.diff(t.off)).max()) == t.n_snps
    # (shuffled: a rank-sorted table of this size would be counted in LDS)
    perm = rng.permutation(t.n_reads)
    ks = np.diff(t.off)
    rank = np.ascontiguousarray(t.rank[perm])
    off = np.concatenate([[0], np.cumsum(ks[perm])]).astype(np.int64)
    bases = np.concatenate([bases[t.off[i]:t.off[i + 1]] for i in perm])
    from oracle.c_oracle import lib, _p
    for sentinels in (False, True):
        o = COracle(t.n_snps, t.band)
        assert lib().orc_fill(o._h, _p(rank), _p(off), _p(bases), t.n_reads, int(sentinels)) == 0
        want = o.export_band()
        for pairs in ("1", "0"):
            monkeypatch.setenv("GH_FILL_PAIRS", pairs)
            h = Hansel(t.n_snps, band=t.band)
            assert h.fill_from_support(rank, off, bases, use_end_sentinels=sentinels) == o.stats(), (sentinels, pairs)
            assert np.array_equal(h.export_band(), want), (sentinels, pairs)


def test_fill_after_spin_keeps_the_fill_accounting():
    # fill -> spin -> fill on one handle without clear(): the second fill's counters are deltas of the device
    # totals against the host mirror, which nothing but create/clear may reset (a stray reset inside gh_spin once
    # re-added the whole first fill)
    t1 = make_support_table(200, 6000, k=4, seed=21)
    t2 = make_support_table(200, 5000, k=4, seed=22)
    h = Hansel(t1.n_snps, band=t1.band)
    o = COracle(t1.n_snps, t1.band, use_libm=True)
    assert h.fill_from_support(t1.rank, t1.off, t1.bases) == o.fill(t1)
    _same_spin(h.spin(7), o.spin(7))                      # first spin allocates the result buffers
    assert h.fill_from_support(t2.rank, t2.off, t2.bases) == o.fill(t2)
    assert (h.n_slices, h.n_crumbs) == o.stats()[:2]
    assert h.L == o.L
    # (the reference never fills a reweighted tensor; here the sorted fill adds a cell's whole count at once, the
    # oracle +1 at a time, and on fractional f32 values these round differently: compare the tensors to f32 precision)
    assert np.allclose(h.export_band(), o.export_band(), rtol=1e-5, atol=0)
    h.spin(9)                                             # a larger spin re-allocates the result buffers
    o.spin(9)
    assert h.fill_from_support(t1.rank, t1.off, t1.bases) == o.fill(t1)
    assert (h.n_slices, h.n_crumbs) == o.stats()[:2]
    assert h.L == o.L
