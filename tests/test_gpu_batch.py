"""Batched recovery (gh_batch_*): many windows in one launch give exactly what each window gives alone."""
import numpy as np
import pytest

from gretel_amd.hansel import Hansel, HanselBatch
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu


def _window(seed, n=400, reads=12000, k=5, n_haps=8):
    t = make_support_table(n, reads, k=k, seed=seed, n_haps=n_haps, err=0.01 if n_haps > 1 else 0.0)
    h = Hansel(t.n_snps, band=t.band)
    h.fill_from_support(t.rank, t.off, t.bases)
    return t, h


def test_batch_equals_single_windows_and_oracle():
    wins = [_window(s) for s in range(6)]
    batch = HanselBatch([h for _, h in wins])
    res = batch.spin(15)
    for (t, h), r in zip(wins, res):
        o = COracle(t.n_snps, t.band)
        o.fill(t)
        ref = o.spin(15)
        assert r["n"] == ref["n"] == 15
        assert np.array_equal(r["paths"], ref["paths"])
        assert r["hp_current"].tolist() == ref["hp_current"].tolist()
        assert r["hp_original"].tolist() == ref["hp_original"].tolist()
        assert r["ratio"].tolist() == ref["ratio"].tolist()
        assert np.allclose(r["magnitude"], ref["magnitude"], rtol=1e-12, atol=0)
        assert np.array_equal(h.export_band(), o.export_band())


def test_batch_windows_may_stop_at_different_paths():
    # window 1 has a single haplotype: it empties after one path (hole) while the others go on
    wins = [_window(0, n=120, reads=4000, k=4), _window(1, n=120, reads=800, k=4, n_haps=1), _window(2, n=120, reads=4000, k=4)]
    for _, h in wins:
        h.L = 4
    res = HanselBatch([h for _, h in wins]).spin(8)
    singles = []
    for s, (n_haps) in zip(range(3), (8, 1, 8)):
        t, h = _window(s, n=120, reads=4000 if n_haps == 8 else 800, k=4, n_haps=n_haps)
        h.L = 4
        singles.append(h.spin(8))
    for r, q in zip(res, singles):
        assert r["n"] == q["n"] and r["hole_at"] == q["hole_at"]
        assert np.array_equal(r["paths"], q["paths"])
        assert r["hp_current"].tolist() == q["hp_current"].tolist()
    assert res[1]["n"] == 1 and res[1]["hole_at"] >= 1 and res[0]["n"] == 8


def test_batch_rejects_mismatched_windows():
    from gretel_amd._lib import GretelHipError
    _, a = _window(0, n=100, reads=2000, k=3)
    _, b = _window(1, n=120, reads=2000, k=3)
    with pytest.raises(GretelHipError):
        HanselBatch([a, b])
    _, c = _window(2, n=100, reads=2000, k=3)
    c.L = a.L + 1
    with pytest.raises(GretelHipError):
        HanselBatch([a, c]).spin(2)


def test_single_path_after_a_batched_spin_sees_every_reweight():
    # reweight_from_path leaves the "only this path's rows changed" hint on the handle; a batched spin then reweights
    # many paths without maintaining the single-window walker tables, so the next generate_path must rebuild them
    wins = [_window(s, n=300, reads=9000, k=5) for s in range(3)]
    oracles = []
    for t, h in wins:
        o = COracle(t.n_snps, t.band)
        o.fill(t)
        h.snapshot_original()
        o.snapshot_original()
        p = h.generate_path()
        po = o.generate_path()
        assert np.array_equal(p[0], po[0])
        h.reweight_from_path(p[0], 0.4)
        o.reweight_path(po[0], 0.4)
        oracles.append(o)
    res = HanselBatch([h for _, h in wins]).spin(6)
    for (t, h), o, r in zip(wins, oracles, res):
        ref = o.spin(6)
        assert np.array_equal(r["paths"], ref["paths"])
        pg, po = h.generate_path(), o.generate_path()
        assert np.array_equal(pg[0], po[0])
        assert pg[1:] == po[1]


@pytest.mark.parametrize("way", ["streams", "batched"])
@pytest.mark.parametrize("L", [2, 5, 9, 18])
def test_both_ways_of_running_a_batch_agree_with_the_oracle(way, L, monkeypatch):
    # up to 47 windows run on their own streams from host threads, larger batches through kernels launched over all
    # windows (one serial path extension per window); GH_BATCH_STREAMS_MAX = -1 sends this small batch the second way
    if way == "batched":
        monkeypatch.setenv("GH_BATCH_STREAMS_MAX", "-1")
    else:
        monkeypatch.delenv("GH_BATCH_STREAMS_MAX", raising=False)
    wins = []
    for s in range(5):
        t = make_support_table(700, 14000, k=None if L > 5 else 6, seed=40 + s, n_haps=6, err=0.01, k_max=21)
        h = Hansel(t.n_snps, band=21 if L > 5 else t.band)
        o = COracle(t.n_snps, 21 if L > 5 else t.band)
        assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
        h.L = L
        o.L = L
        wins.append((h, o))
    res = HanselBatch([h for h, _ in wins]).spin(10)
    for (h, o), r in zip(wins, res):
        ref = o.spin(10)
        assert r["n"] == ref["n"] and r["hole_at"] == ref["hole_at"]
        assert np.array_equal(r["paths"], ref["paths"])
        assert r["hp_current"].tolist() == ref["hp_current"].tolist()
        assert r["ratio"].tolist() == ref["ratio"].tolist()
        assert np.array_equal(h.export_band(), o.export_band())
