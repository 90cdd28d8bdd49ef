"""The C restatement is bit-identical to the Python restatement (both with libm's log10),
for every switch of the frozen spec; and with the restated log10 of include/gh_detlog.h (the kernels') it gives the same doubles."""
import numpy as np
import pytest

from gretel_amd.synth import make_support_table
from oracle import gretel_ref as G
from oracle.c_oracle import COracle, paths_to_str
from oracle.hansel_ref import Hansel, HanselSpec, SYMBOLS, UNSYMBOLS

CASES = [("A", False, "f32", None, {}), ("A", False, "f32", 3, {}), ("B", False, "f32", None, {}), ("C", True, "f32", None, {}),
         ("D", False, "f32", None, {}), ("A", True, "f64", None, {}), ("B", True, "f64", 4, {}),
         ("E", False, "f32", None, {}), ("E", True, "f64", None, {}),
         ("A", False, "f32", None, dict(cand_order="-TGCA")), ("C", True, "f32", None, dict(cand_order="GA-TC")),
         ("A", False, "f32", None, dict(offer_zero=True)), ("E", True, "f32", None, dict(offer_zero=True, cand_order="TGCA-"))]


@pytest.mark.parametrize("mode,mt,storage,band,extra", CASES)
def test_c_equals_python(mode, mt, storage, band, extra):
    t = make_support_table(50, 1200, k=4, seed=5)
    spec = HanselSpec(storage=storage, cond_mode=mode, marginal_term=mt, **extra)
    h = Hansel.init_matrix(SYMBOLS, UNSYMBOLS, t.n_snps, spec, band=band)
    st = G.fill_from_support(h, t.reads(), t.n_snps)
    c = COracle(t.n_snps, t.band, storage, mode, mt, use_libm=True, **extra)
    assert c.fill(t) == st
    assert c.L == h.L
    # lookups before any reweight
    for p in (0, 1, 7, t.n_snps):
        cc = c.counts_at(p)
        m = h.get_counts_at(p)
        assert cc[7] == m["total"]
        for s in h.symbols:
            assert cc[s.i] == m.get(s, 0.0)
    # edge weights: values, candidate set AND the order they are offered in
    path = [h.symbols_d['_']] + [h.symbols_d[c] for c in "ACGTACG"]
    for p in (1, 3, 7):
        ew = h.get_edge_weights_at(p, path)
        mask, w = c.edge_weights(p, [s.i for s in path])
        assert [str(k) for k in ew] == [ch for ch in spec.cand_order if (mask >> "ACGTN-_".index(ch)) & 1]
        assert all(w[k.i] == v for k, v in ew.items())
    recs, _ = G.recover_paths(h, t.n_snps, 8)
    r = c.spin(8)
    assert r["n"] == len(recs) == 8
    assert paths_to_str(r["paths"]) == [x["path"] for x in recs]
    assert r["hp_current"].tolist() == [x["hp_current"] for x in recs]
    assert r["hp_original"].tolist() == [x["hp_original"] for x in recs]
    assert r["ratio"].tolist() == [x["ratio"] for x in recs]
    assert r["magnitude"].tolist() == [x["magnitude"] for x in recs]
    dense = h.dense()
    band_c = c.export_band()
    for i in range(t.n_snps + 2):
        for d in range(1, t.band + 1):
            if i + d <= t.n_snps + 1:
                assert np.array_equal(band_c[i, d - 1], dense[:, :, i, i + d])


@pytest.mark.parametrize("mode", ["C", "D", "E"])
def test_nan_weights_c_equals_python(mode):
    # zero-count candidates offered WITH the marginal term: at the first positions (V(0) = 0) the symbols never seen weigh
    # NaN (-inf + inf).  gretel.py:166-174 replaces the incumbent only by something that compares GREATER, so a NaN wins exactly
    # when it is offered first -- both restatements must read the loop that way (the HIP path is held to the C one)
    n_nan = 0
    for n, seed in [(2, 13), (2, 18), (3, 3), (3, 7), (3, 13), (5, 1), (5, 4)]:
        t = make_support_table(n, 44, k=2, n_haps=2, err=0.01, seed=seed, k_max=n)
        spec = HanselSpec(cond_mode=mode, marginal_term=True, offer_zero=True)
        h = Hansel.init_matrix(SYMBOLS, UNSYMBOLS, t.n_snps, spec, band=None)
        G.fill_from_support(h, t.reads(), t.n_snps)
        c = COracle(t.n_snps, t.band, "f32", mode, True, use_libm=True, offer_zero=True)
        c.fill(t)
        ew = h.get_edge_weights_at(1, [h.symbols_d['_']])
        n_nan += sum(1 for v in ew.values() if v != v)
        recs, _ = G.recover_paths(h, t.n_snps, 3)
        r = c.spin(3)
        assert r["n"] == len(recs)
        assert paths_to_str(r["paths"]) == [x["path"] for x in recs]
        assert np.array_equal(r["hp_current"], np.array([x["hp_current"] for x in recs]), equal_nan=True)
    assert mode != "C" or n_nan > 0          # (C: V(0) + the column sum of a symbol never seen = 0; D's row sum and E's V(j) are not)


def test_full_enumeration_equals_banded_enumeration():
    t = make_support_table(40, 800, k=3, seed=2)
    a = COracle(t.n_snps, t.band, use_libm=True)
    b = COracle(t.n_snps, t.band, use_libm=True)
    a.fill(t), b.fill(t)
    b.set_full_enum(1)
    ra, rb = a.spin(5), b.spin(5)
    assert np.array_equal(ra["paths"], rb["paths"])
    assert ra["magnitude"].tolist() == rb["magnitude"].tolist()
    n = t.n_snps
    assert b.reweight_calls() == 5 * (n * (n + 3) // 2 + 1)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_restated_log10_changes_nothing(seed):
    t = make_support_table(300, 9000, k=5, seed=seed)
    a = COracle(t.n_snps, t.band, use_libm=True)
    b = COracle(t.n_snps, t.band, use_libm=False)
    a.fill(t), b.fill(t)
    ra, rb = a.spin(20), b.spin(20)
    assert ra["n"] == rb["n"]
    assert np.array_equal(ra["paths"], rb["paths"])
    assert np.array_equal(ra["hp_current"], rb["hp_current"]) and np.array_equal(ra["hp_original"], rb["hp_original"])
    assert np.array_equal(a.export_band(), b.export_band())


def test_hole_terminates_recovery():
    # one haplotype only: every marginal is 1.0 -> ratio 1.0 -> the matrix is emptied
    # by the first reweight and the second generate_path hits a hole (gretel.py:176-180)
    t = make_support_table(30, 200, k=3, n_haps=1, err=0.0, seed=1)
    c = COracle(t.n_snps, t.band, use_libm=True)
    c.fill(t)
    r = c.spin(5)
    assert r["n"] == 1 and r["hole_at"] >= 1
    h = Hansel.init_matrix(SYMBOLS, UNSYMBOLS, t.n_snps)
    G.fill_from_support(h, t.reads(), t.n_snps)
    recs, _ = G.recover_paths(h, t.n_snps, 5)
    assert len(recs) == 1
