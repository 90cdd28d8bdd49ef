"""Every switch of the Hansel arithmetic the reference leaves to hanselx (gh_config / oracle.hansel_ref.HanselSpec:
conditional A..E, marginal term, storage, the order the candidates are offered in, zero-count candidates) through the
FAST paths: the segment-parallel walk, the candidate pools, the fused reweight that keeps the conditional table current
(rows for A/B/D, columns for C/E) -- bit for bit against the C oracle, and with no table rebuild between the paths of a spin."""
import itertools

import numpy as np
import pytest

from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table, SupportTable
from oracle.c_oracle import COracle
from spec_util import make_pair, same, with_dels, spec_id
from test_gpu_edges import PINNED

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(PINNED, reason="GH_WALK / GH_WALK_THREADS pin another variant")]

MATRIX = [dict(cond_mode=m, marginal_term=mt, storage=st) for m, mt, st in itertools.product("ABCDE", (False, True), ("f32", "f64"))]


@pytest.mark.parametrize("kw", MATRIX, ids=spec_id)
@pytest.mark.parametrize("wide", [False, True], ids=["ranked", "symbols"])
def test_matrix_segment_parallel(kw, wide):
    # L = 5, both table layouts (ranked: at most four candidates everywhere; symbols: '-' as a fifth somewhere)
    t = make_support_table(700, 24000, k=6, n_haps=5, seed=41)
    if wide:
        t = with_dels(t, 0.08, 3)
    h, o = make_pair(t, L=5, **kw)
    h.profile_enable(1)
    h.profile_reset()
    res, ref = h.spin(12), o.spin(12)
    assert h.walk_clock()[3] == 3
    same(res, ref)
    assert np.array_equal(h.export_band(), o.export_band())
    # the conditional table is built once in front of the spin and brought in step once behind it -- never between paths
    lt = h.profile_get()["lt"]["launches"]
    assert lt <= 2 + 2 * h.walk_clock()[0], "k_lt ran %d times in a spin of 12 paths" % lt
    h.profile_enable(0)
    # lookups on the reweighted tensor
    path = ref["paths"][0]
    for p in (1, 2, 9, t.n_snps):
        mask, w = o.edge_weights(p, path)
        ew = h.get_edge_weights_at(p, h.path_symbols(path))
        assert [s.i for s in ew] == [q for q in (0, 1, 2, 3, 5) if (mask >> q) & 1]
        assert all(w[s.i] == v for s, v in ew.items())


@pytest.mark.parametrize("kw", [dict(cond_mode="C", marginal_term=True), dict(cond_mode="E", marginal_term=True), dict(cond_mode="A", marginal_term=True),
                                dict(cond_mode="E"), dict(cond_mode="C", storage="f64"), dict(cond_mode="B", marginal_term=True)], ids=spec_id)
@pytest.mark.parametrize("L,wide", [(7, False), (12, False), (9, True), (26, False)])
def test_matrix_candidate_pools(kw, L, wide):
    t = make_support_table(1800, 24000, k=None, seed=500 + L, k_max=max(21, L + 4), k_lambda=10.0 if L <= 24 else float(L))
    if wide:
        t = with_dels(t, 0.08, 5)
    h, o = make_pair(t, L=L, **kw)
    h.profile_enable(1)
    h.profile_reset()
    res, ref = h.spin(14), o.spin(14)
    assert h.walk_clock()[3] == 4
    same(res, ref)
    assert np.array_equal(h.export_band(), o.export_band())
    handed = h.walk_clock()[1]                            # (a path the pools hand to the serial walker rebuilds the table for it, and again behind it)
    lt = h.profile_get()["lt"]["launches"]
    assert lt <= 3 + 2 * h.walk_clock()[0] + 2 * handed, "k_lt ran %d times in a spin of 14 paths" % lt
    h.profile_enable(0)
    same(h.spin(4), o.spin(4))


def _tie_table(n=40, copies=3):
    """Two haplotypes, all-A and all-C, seen equally often by identical tilings: every edge weight of A equals that of C
    until the first reweight, so the FIRST candidate offered wins every position (gretel.py:166-174)."""
    ranks, seqs = [], []
    for start in range(0, n - 3 + 1):
        for ch in "AC":
            for _ in range(copies):
                ranks.append(start)
                seqs.append(ch * 3)
    order = np.argsort(np.array(ranks), kind="stable")
    ranks = [ranks[q] for q in order]
    seqs = [seqs[q] for q in order]
    off = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.int64)
    bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()
    return SupportTable(n_snps=n, rank=np.array(ranks, dtype=np.int32), off=off, bases=bases,
                        haplotypes=np.zeros((0, n), dtype=np.uint8), abundances=np.zeros(0))


@pytest.mark.parametrize("order,first", [("ACGT-", "A"), ("-TGCA", "C"), ("GCA-T", "C"), ("TAG-C", "A")])
@pytest.mark.parametrize("L", [3, 7])
def test_the_order_the_candidates_are_offered_in_breaks_the_ties(order, first, L):
    t = _tie_table()
    h, o = make_pair(t, L=L, cand_order=order)
    res, ref = h.spin(3), o.spin(3)
    same(res, ref)
    assert Hansel.path_str(res["paths"][0]) == "_" + first * t.n_snps
    assert str(list(h.get_edge_weights_at(1, [h.symbols_d['_']]))[0]) == first      # dict order = the order offered
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("kw", [dict(cand_order="-TGCA"), dict(cand_order="GA-TC", cond_mode="C", marginal_term=True),
                                dict(cand_order="CATG-", cond_mode="E", storage="f64")], ids=spec_id)
@pytest.mark.parametrize("L", [4, 8])
def test_other_orders_on_real_windows(kw, L):
    t = with_dels(make_support_table(900, 20000, k=None, seed=61, n_haps=4), 0.04, 9)
    h, o = make_pair(t, L=L, **kw)
    same(h.spin(10), o.spin(10))
    assert np.array_equal(h.export_band(), o.export_band())
    hs, os_ = make_pair(t, L=L, walk="spec", **kw)         # and the serial walker agrees
    same(hs.spin(10), os_.spin(10))


@pytest.mark.parametrize("kw", [dict(), dict(cond_mode="C", marginal_term=True), dict(cond_mode="D", cand_order="TGCA-"),
                                dict(cond_mode="C", marginal_term=True, cand_order="-ACGT"), dict(cond_mode="D", marginal_term=True, cand_order="-TGCA")], ids=spec_id)
@pytest.mark.parametrize("L", [3, 5, 8])
@pytest.mark.parametrize("walk", [None, "spec"])
def test_zero_count_candidates(kw, L, walk):
    # every valid symbol is offered at every position; a symbol never seen there has marginal 0 (log10 -> -inf).  With the
    # marginal term and a conditional whose denominator is empty at position 0 (C, D: V(0) = 0), the first L positions
    # weigh +inf for the symbols seen and NaN (-inf + inf) for the others: a NaN offered FIRST is the incumbent of
    # gretel.py:166-174 and nothing compares greater -- '-' (never seen, offered first) is then what the reference selects.
    t = make_support_table(600, 9000, k=None, seed=71, n_haps=3)
    h, o = make_pair(t, L=L, walk=walk, offer_zero=True, **kw)
    assert (h.candidate_masks()[1:] == 0x2F).all()
    mask, w = o.edge_weights(4, np.array([6, 0, 1, 2], dtype=np.uint8))
    ew = h.get_edge_weights_at(4, [h.symbols[q] for q in (6, 0, 1, 2)])
    assert mask == 0x2F and len(ew) == 5
    assert np.array_equal(np.array([w[s.i] for s in ew]), np.array(list(ew.values())), equal_nan=True)
    res, ref = h.spin(10), o.spin(10)
    assert np.array_equal(res["paths"], ref["paths"])
    assert np.array_equal(res["hp_current"], ref["hp_current"], equal_nan=True) and np.array_equal(res["hp_original"], ref["hp_original"], equal_nan=True)
    assert res["ratio"].tolist() == ref["ratio"].tolist() and np.allclose(res["magnitude"], ref["magnitude"], rtol=1e-10, atol=0)
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("mode", ["C", "D", "E"])
def test_a_nan_behind_the_first_candidate_never_wins(mode):
    # windows of 2..5 SNPs, zero-count candidates offered with the marginal term: at the first positions the symbols never seen
    # weigh NaN (-inf + inf).  gretel.py:166-174 keeps an incumbent unless something compares GREATER: a NaN in any place but
    # the first never wins -- and must not shield the candidate behind it either (k_seg's arg-max is a tournament of pairs: a NaN
    # in front of T once kept T out; found by the fuzz)
    n_bad = 0
    for n, seed in itertools.product((2, 3, 5), range(30)):
        t = make_support_table(n, 44, k=2, n_haps=2, err=0.01, seed=seed, k_max=n)
        h, o = make_pair(t, cond_mode=mode, marginal_term=True, offer_zero=True)
        res, ref = h.spin(3), o.spin(3)
        ok = res["n"] == ref["n"] and np.array_equal(res["paths"], ref["paths"]) and \
            np.array_equal(res["hp_current"], ref["hp_current"], equal_nan=True)
        n_bad += not ok
    assert n_bad == 0


@pytest.mark.parametrize("walk", ["spec", "spec1"])
@pytest.mark.parametrize("order", ["TG-AC", "-TGCA", "CA-GT"])
@pytest.mark.parametrize("L", [1, 2, 3])
def test_serial_walkers_under_a_candidate_order(walk, order, L):
    # the depth-1 walker works on four lanes per group while no position offers the fifth candidate -- the LAST symbol of the
    # candidate order, not '-' as such (with '-' in the middle of the order the symbol in fifth place was never looked at:
    # found by the fuzz through a batched launch at L = 1, where nothing else runs the depth-1 walker on a narrow window)
    for seed in range(3):
        t = make_support_table(17 + 40 * seed, 340 + 900 * seed, k=7, n_haps=8, err=0.05, seed=seed)
        h, o = make_pair(t, L=L, walk=walk, cand_order=order)
        same(h.spin(3), o.spin(3))
        assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("order", ["TG-AC", "-TGCA"])
@pytest.mark.parametrize("L", [1, 3])
def test_batched_launches_under_a_candidate_order(order, L, monkeypatch):
    from gretel_amd.hansel import HanselBatch
    monkeypatch.setenv("GH_BATCH_STREAMS_MAX", "-1")          # kernels launched over all windows, not windows on their own streams
    ts = [make_support_table(60, 1500, k=7, n_haps=6, err=0.03, seed=90 + q) for q in range(3)]
    pairs = [make_pair(t, L=L, cand_order=order) for t in ts]
    for res, (_, o) in zip(HanselBatch([h for h, _ in pairs]).spin(4), pairs):
        same(res, o.spin(4))
