"""The window pipeline (gretel_amd/csrc/wpipe.hpp): a batch carried through all its paths by one persistent workgroup per
window -- the reweight of path s-1 (gretel/gretel.py:79-98) sweeping ahead of the walk of path s (gretel/gretel.py:143-189) --
gives, bit for bit, what the C oracle's spin loop (gretel/cmd.py:148-179) gives for every window alone."""
import numpy as np
import pytest

from gretel_amd.hansel import Hansel, HanselBatch
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _pipe_for_small_batches(monkeypatch):
    monkeypatch.setenv("GH_PIPE_MIN", "1")
    monkeypatch.delenv("GH_PIPE", raising=False)
    monkeypatch.delenv("GH_PIPE_NT", raising=False)


def _pair(seed, n, reads, k, L=None, band=None, n_haps=8, k_max=21, **kw):
    t = make_support_table(n, reads, k=k, seed=seed, n_haps=n_haps, err=0.01 if n_haps > 1 else 0.0, k_max=k_max)
    W = band if band is not None else t.band
    h = Hansel(t.n_snps, band=W, **kw)
    o = COracle(t.n_snps, W, **kw)
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    if L is not None:
        h.L = L
        o.L = L
    return h, o


def _same(r, ref, h, o):
    assert r["n"] == ref["n"] and r["hole_at"] == ref["hole_at"], (r["n"], ref["n"], r["hole_at"], ref["hole_at"])
    assert np.array_equal(r["paths"], ref["paths"])
    assert r["hp_current"].tolist() == ref["hp_current"].tolist()
    assert r["hp_original"].tolist() == ref["hp_original"].tolist()
    assert r["ratio"].tolist() == ref["ratio"].tolist()
    assert np.allclose(r["magnitude"], ref["magnitude"], rtol=1e-12, atol=0)
    assert np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("L", [2, 3, 4, 5, 6])
def test_pipeline_equals_the_oracle(L):
    wins = [_pair(100 + s, 700, 20000, 6, L=L) for s in range(4)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(12)
    info = b.pipe_info()
    assert info["windows"] == 4 and info["handed_back"] == 0 and info["threads"] == 1024, info
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(12), h, o)


@pytest.mark.parametrize("L", [7, 9, 10, 11, 14])
def test_pipeline_with_longer_memories(L, monkeypatch):
    monkeypatch.setenv("GH_PIPE_MAX_L", "14")           # (by default the pipeline stops at ten lags: beyond, the pools are faster)
    wins = [_pair(200 + s, 600, 12000, None, L=L, band=21, n_haps=6) for s in range(3)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(8)
    info = b.pipe_info()
    assert info["windows"] == 3 and info["threads"] == (768 if L <= 10 else 512), info
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(8), h, o)


@pytest.mark.parametrize("kw", [dict(storage="f64"), dict(cond_mode="B"), dict(cond_mode="D"), dict(cond_mode="D", storage="f64"),
                                dict(cand_order="TGCA-"), dict(offer_zero=True), dict(marginal_term=True),
                                dict(cond_mode="B", marginal_term=True, storage="f64"), dict(cond_mode="D", marginal_term=True, cand_order="-TGCA")],
                         ids=lambda kw: "-".join("%s=%s" % x for x in sorted(kw.items())))
def test_pipeline_under_the_row_conditionals(kw):
    wins = [_pair(300 + s, 500, 15000, 5, **kw) for s in range(3)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(10)
    # (offer_zero: every valid symbol is a candidate everywhere -> five per position, no ranked table: the batched launches)
    assert b.pipe_info()["windows"] == (0 if kw.get("offer_zero") else 3)
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(10), h, o)


@pytest.mark.parametrize("kw", [dict(cond_mode="C"), dict(cond_mode="E"), dict(cond_mode="E", marginal_term=True),
                                dict(cond_mode="C", marginal_term=True, storage="f64"), dict(cond_mode="E", storage="f64", cand_order="G-TAC")],
                         ids=lambda kw: "-".join("%s=%s" % x for x in sorted(kw.items())))
def test_pipeline_under_the_column_conditionals(kw):
    # C / E: a reweighted cell changes the sum of its COLUMN and with it the entries of every row in that column; the sweep works
    # on the to-major copy of the band and keeps it in step (E + marginal term: the published method's form)
    wins = [_pair(320 + s, 500, 15000, 5, **kw) for s in range(3)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(25)
    assert b.pipe_info()["windows"] == 3
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(25), h, o)
        more, ref = h.spin(4), o.spin(4)            # the handle goes on (the to-major copy was kept in step or is rebuilt)
        assert np.array_equal(more["paths"], ref["paths"]) and np.array_equal(h.export_band(), o.export_band())


@pytest.mark.parametrize("L", [2, 4, 7, 9, 10])
@pytest.mark.parametrize("cm", ["C", "E"])
def test_column_conditionals_across_lag_counts_and_bands(L, cm):
    wins = [_pair(340 + s, 600, 14000, None, L=L, band=21, n_haps=6, cond_mode=cm, marginal_term=bool(L & 1)) for s in range(2)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(12)
    assert b.pipe_info()["windows"] == 2
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(12), h, o)


def test_the_switches_that_leave_specs_to_the_batched_launches(monkeypatch):
    monkeypatch.setenv("GH_PIPE_COL", "0")
    monkeypatch.setenv("GH_PIPE_MT", "0")
    for kw in (dict(cond_mode="C"), dict(marginal_term=True)):
        wins = [_pair(360 + s, 300, 9000, 5, **kw) for s in range(2)]
        b = HanselBatch([h for h, _ in wins])
        res = b.spin(6)
        assert b.pipe_info()["windows"] == 0
        for (h, o), r in zip(wins, res):
            _same(r, o.spin(6), h, o)


@pytest.mark.parametrize("L", [2, 3, 6, 8, 10])
def test_marginal_term_across_lag_counts(L):
    wins = [_pair(230 + s, 600, 15000, None, L=L, band=21, n_haps=6, marginal_term=True) for s in range(2)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(25)
    assert b.pipe_info()["windows"] == 2
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(25), h, o)


@pytest.mark.parametrize("paths", [1, 2, 3])
def test_very_short_spins(paths):
    # one path: a walk and the last sweep, nothing beside each other; two: one sweep beside a walk
    wins = [_pair(250 + s, 300, 9000, 5) for s in range(2)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(paths)
    assert b.pipe_info()["windows"] == 2
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(paths), h, o)
    res = b.spin(paths)             # and again on the reweighted tensors, other lag count: the compact table is rebuilt
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(paths), h, o)


def test_beyond_ten_lags_the_pools_take_the_batch():
    wins = [_pair(260 + s, 400, 9000, None, L=12, band=21, n_haps=6) for s in range(2)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(5)
    assert b.pipe_info()["windows"] == 0
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(5), h, o)


@pytest.mark.parametrize("band", [9, 16, 17, 25, 32, 33, 40])
def test_wide_bands(band):
    # the sweep prefetches the path's element of a lane's further cells up to a band of 32 and loads in place beyond
    wins = [_pair(270 + s, 500, 10000, None, L=5, band=band, n_haps=6, k_max=min(band + 1, 34)) for s in range(2)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(8)
    assert b.pipe_info()["windows"] == 2
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(8), h, o)


@pytest.mark.parametrize("n", [7, 59, 60, 61, 64, 121, 1000, 2417])
def test_window_lengths_around_the_chunk(n):
    # chunks of 60 positions at L = 5: windows shorter than one, exactly one, one and a bit, ...
    wins = [_pair(400 + s, n, max(60, 30 * n), min(5, n), L=min(5, n)) for s in range(2)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(9)
    assert b.pipe_info()["windows"] == 2
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(9), h, o)


def test_a_window_that_empties_is_handed_back():
    # window 1 holds a single haplotype: its first reweight moves candidate masks (counts reach zero), the pipeline stops
    # there and gh_spin finds the hole; the others run through
    wins = [_pair(0, 150, 5000, 4, L=4), _pair(1, 150, 1000, 4, L=4, n_haps=1), _pair(2, 150, 5000, 4, L=4)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(8)
    info = b.pipe_info()
    assert info["windows"] == 3 and info["handed_back"] >= 1, info
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(8), h, o)
    assert res[1]["n"] == 1 and res[1]["hole_at"] >= 1 and res[0]["n"] == 8


def test_deep_spins_hand_windows_back_when_masks_move():
    # 60 paths over a small window: rare symbols run out along the way, some windows are handed back in the middle
    wins = [_pair(500 + s, 300, 2500, 5) for s in range(6)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(60)
    assert b.pipe_info()["windows"] == 6
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(60), h, o)


def test_a_window_with_five_candidates_takes_the_batched_launches():
    from spec_util import with_dels
    wins = []
    for s in range(3):
        t = make_support_table(300, 9000, k=5, seed=600 + s)
        if s == 1:
            t = with_dels(t, 0.05, 7)
        h = Hansel(t.n_snps, band=t.band)
        o = COracle(t.n_snps, t.band)
        assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
        wins.append((h, o))
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(7)
    assert b.pipe_info()["windows"] == 2
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(7), h, o)


def test_pipeline_and_batched_launches_agree(monkeypatch):
    wins = [_pair(700 + s, 800, 24000, 5) for s in range(3)]
    twins = [_pair(700 + s, 800, 24000, 5) for s in range(3)]
    res = HanselBatch([h for h, _ in wins]).spin(20)
    monkeypatch.setenv("GH_PIPE", "0")
    monkeypatch.setenv("GH_BATCH_STREAMS_MAX", "-1")
    b2 = HanselBatch([h for h, _ in twins])
    res2 = b2.spin(20)
    assert b2.pipe_info()["windows"] == 0
    for r, q in zip(res, res2):
        assert np.array_equal(r["paths"], q["paths"])
        assert r["hp_current"].tolist() == q["hp_current"].tolist()
        assert r["ratio"].tolist() == q["ratio"].tolist()
    for (h, _), (h2, _) in zip(wins, twins):
        assert np.array_equal(h.export_band(), h2.export_band())


def test_handles_stay_usable_after_the_pipeline():
    wins = [_pair(800 + s, 400, 12000, 5) for s in range(2)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(5)
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(5), h, o)
        pg, po = h.generate_path(), o.generate_path()
        assert np.array_equal(pg[0], po[0]) and pg[1:] == po[1]
        more, ref = h.spin(4), o.spin(4)
        assert np.array_equal(more["paths"], ref["paths"])
    res = b.spin(5)         # and once more through the pipeline
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(5), h, o)


@pytest.mark.parametrize("nt", [512, 768])
def test_other_workgroup_sizes_at_five_lags(nt, monkeypatch):
    monkeypatch.setenv("GH_PIPE_NT", str(nt))
    wins = [_pair(900 + s, 500, 15000, 5) for s in range(2)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(8)
    assert b.pipe_info()["threads"] == nt
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(8), h, o)


@pytest.mark.parametrize("synth", ["0", "1"])
@pytest.mark.parametrize("kw", [dict(), dict(cond_mode="D", marginal_term=True), dict(cond_mode="E", marginal_term=True), dict(cond_mode="C", storage="f64")],
                         ids=lambda kw: "-".join("%s=%s" % x for x in sorted(kw.items())) or "default")
@pytest.mark.parametrize("L,band", [(5, 4), (5, 2), (6, 3), (3, 2), (8, 5)])
def test_far_lags_made_by_the_loaders_or_read(L, band, kw, synth, monkeypatch):
    """Lags beyond the band: their table entries are constants of V and the candidate counts, which the loaders either read from
    the pipeline's table or make from the packed position words (GH_PIPE_SYNTH=0; default: made)."""
    monkeypatch.setenv("GH_PIPE_SYNTH", synth)
    wins = [_pair(900 + s, 337, 9000, band + 1, L=L, band=band, k_max=band + 1, **kw) for s in range(3)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(9)
    assert b.pipe_info()["windows"] == 3, b.pipe_info()
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(9), h, o)


# ---- windows with a few five-candidate positions (round 6: k_wpipe<.., WIDE>, wpipe.hpp) -------------------------------------------
def _wide_pair(seed, n, reads, k, frac=0.02, L=None, band=None, n_haps=8, k_max=21, **kw):
    """A window with '-' at `frac` of its POSITIONS (gretel/util.py:178-190: a deletion is an ordinary symbol): those positions offer
    five candidates, the rest at most four."""
    from gretel_amd.synth import sprinkle_deletions
    t = make_support_table(n, reads, k=k, seed=seed, n_haps=n_haps, err=0.01, k_max=k_max)
    sprinkle_deletions(t, frac, seed=seed + 1)
    W = band if band is not None else t.band
    h = Hansel(t.n_snps, band=W, **kw)
    o = COracle(t.n_snps, W, **kw)
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    if L is not None:
        h.L = L
        o.L = L
    return h, o


def _n_wide(h):
    return int((h.candidate_masks()[1:] == 0x2F).sum())


@pytest.mark.parametrize("L", [2, 3, 4, 5, 6])
def test_wide_pipeline_equals_the_oracle(L):
    wins = [_wide_pair(900 + s, 700, 20000, 6, L=L) for s in range(4)]
    assert all(_n_wide(h) > 0 for h, _ in wins)
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(12)
    info = b.pipe_info()
    assert info["windows"] == 4 and info["handed_back"] == 0, info
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(12), h, o)


@pytest.mark.parametrize("L", [7, 8, 10])
def test_wide_pipeline_with_longer_memories(L):
    wins = [_wide_pair(920 + s, 600, 12000, None, L=L, band=21, n_haps=6) for s in range(3)]
    assert all(_n_wide(h) > 0 for h, _ in wins)
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(8)
    info = b.pipe_info()
    assert info["windows"] == 3, info
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(8), h, o)


@pytest.mark.parametrize("kw", [dict(storage="f64"), dict(cond_mode="B"), dict(cond_mode="D"), dict(cond_mode="C"), dict(cond_mode="E"),
                                dict(marginal_term=True), dict(cond_mode="E", marginal_term=True), dict(cond_mode="C", marginal_term=True, storage="f64"),
                                dict(cand_order="TGCA-"), dict(cand_order="-TGCA", marginal_term=True), dict(cond_mode="E", cand_order="G-TCA"),
                                dict(cond_mode="B", marginal_term=True, storage="f64")],
                         ids=lambda kw: "-".join("%s=%s" % x for x in sorted(kw.items())))
def test_wide_pipeline_under_every_spec(kw):
    wins = [_wide_pair(940 + s, 500, 15000, 5, **kw) for s in range(3)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(10)
    assert b.pipe_info()["windows"] == 3, b.pipe_info()
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(10), h, o)


def test_wide_and_narrow_windows_in_one_batch():
    wins = [_pair(960, 700, 20000, 5), _wide_pair(961, 700, 20000, 5), _pair(962, 700, 20000, 5), _wide_pair(963, 700, 20000, 5, frac=0.05)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(15)
    assert b.pipe_info()["windows"] == 4, b.pipe_info()
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(15), h, o)


def test_wide_pipeline_in_a_deep_spin_and_afterwards():
    """Counts reach zero, masks move (a five-candidate position may lose a candidate): the window is handed back with its tensor exact,
    gh_spin finishes it; the handles stay usable."""
    wins = [_wide_pair(970 + s, 400, 5000, 5, n_haps=3) for s in range(3)]
    b = HanselBatch([h for h, _ in wins])
    res = b.spin(60)
    for (h, o), r in zip(wins, res):
        _same(r, o.spin(60), h, o)
    for h, o in wins:
        r2, ref2 = h.spin(3), o.spin(3)
        assert r2["n"] == ref2["n"] and np.array_equal(r2["paths"], ref2["paths"])


def test_too_many_wide_positions_are_left_to_the_launches():
    from spec_util import with_dels
    t = with_dels(make_support_table(600, 18000, k=5, seed=980), 0.05, 7)      # (nearly every position offers five candidates)
    h = Hansel(t.n_snps, band=t.band)
    o = COracle(t.n_snps, t.band)
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    b = HanselBatch([h])
    r = b.spin(9)[0]
    assert b.pipe_info()["windows"] == 0
    _same(r, o.spin(9), h, o)


def test_the_switch_that_leaves_wide_windows_to_the_launches(monkeypatch):
    monkeypatch.setenv("GH_PIPE_WIDE", "0")
    h, o = _wide_pair(985, 500, 15000, 5)
    b = HanselBatch([h])
    r = b.spin(9)[0]
    assert b.pipe_info()["windows"] == 0
    _same(r, o.spin(9), h, o)


def _wide_table(seed, n, reads, k, positions, frac_reads=0.4, n_haps=6, k_max=21, err=0.01):
    """'-' on `frac_reads` of the reads at exactly the given positions (1-based SNP numbers)."""
    t = make_support_table(n, reads, k=k, seed=seed, n_haps=n_haps, err=err, k_max=k_max)
    rng = np.random.default_rng(seed + 7)
    ks = np.diff(t.off)
    read_of = np.repeat(np.arange(t.n_reads, dtype=np.int64), ks)
    snp = t.rank[read_of].astype(np.int64) + (np.arange(len(t.bases), dtype=np.int64) - t.off[read_of]) + 1
    hit = np.isin(snp, np.asarray(positions)) & (rng.random(len(t.bases)) < frac_reads)
    bases = t.bases.copy()
    bases[hit] = ord('-')
    t.bases = bases
    return t


def _run_wide(t, paths, L=None, band=None, expect_pipe=True, **kw):
    W = band if band is not None else t.band
    h = Hansel(t.n_snps, band=W, **kw)
    o = COracle(t.n_snps, W, **kw)
    assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
    if L is not None:
        h.L = L
        o.L = L
    b = HanselBatch([h])
    r = b.spin(paths)[0]
    if expect_pipe is not None:
        assert (b.pipe_info()["windows"] == 1) == expect_pipe, b.pipe_info()
    _same(r, o.spin(paths), h, o)
    return h


@pytest.mark.parametrize("n", [7, 12, 30, 59, 60, 61, 64, 65, 119, 121, 241, 1203])
def test_wide_window_lengths_around_the_chunk(n):
    pos = sorted({1, 2, max(1, n // 2), n - 1, n} & set(range(1, n + 1)))
    t = _wide_table(1100 + n, n, 40 * n, min(5, n), pos)
    h = _run_wide(t, 9)
    assert _n_wide(h) > 0


def test_wide_positions_next_to_one_another_and_at_the_ends():
    # runs of adjacent five-candidate positions (the stepper's records feed one another: entry (row 4, column 4) stands in both),
    # the first and the last position of the window
    t = _wide_table(1200, 400, 16000, 5, [1, 2, 3, 50, 51, 52, 53, 54, 200, 399, 400], frac_reads=0.5)
    for kw in (dict(), dict(cond_mode="E", marginal_term=True), dict(cond_mode="B"), dict(cond_mode="C", storage="f64")):
        h = _run_wide(t, 12, **kw)
        assert _n_wide(h) >= 3


def test_more_wide_positions_than_a_chunk_holds_records_for_is_refused():
    # twenty in a row, every base seen at each of them (5 % substitutions): more five-candidate positions than PIPE_WREC in one reach
    t = _wide_table(1210, 400, 40000, 5, list(range(100, 120)), frac_reads=0.5, n_haps=8, err=0.05)
    h = _run_wide(t, 6, expect_pipe=False)
    assert _n_wide(h) >= 12


@pytest.mark.parametrize("paths", [1, 2, 3])
def test_wide_very_short_spins(paths):
    t = _wide_table(1220, 300, 9000, 5, [40, 41, 150, 299])
    _run_wide(t, paths)


@pytest.mark.parametrize("L,band", [(2, 9), (3, 12), (5, 12), (6, 20), (8, 20), (9, 12), (10, 20), (10, 9)])
def test_wide_lag_counts_and_bands(L, band):
    # bands wider than the eight lanes of a sweep's lane group, lag counts beyond eight (the sweep's further rounds), lags beyond the band
    t = _wide_table(1230 + L, 500, 12000, None, [30, 31, 100, 260, 261, 262, 470], n_haps=5, k_max=band + 1)
    for kw in (dict(), dict(cond_mode="E", marginal_term=True)):
        _run_wide(t, 7, L=L, band=band, **kw)
