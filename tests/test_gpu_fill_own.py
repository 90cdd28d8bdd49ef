"""k_fill_own (rank-sorted tables: the tensor cut by owner, LDS counters, no global atomics) against the oracle's pair loop
(gretel/util.py:226-286) and against the older fills: every shape of band, the rank-0 and last-SNP special cases, the end
sentinels, N / '-' / '_' bases, bad symbols, a second fill on top of the first, both counter widths."""
import numpy as np
import pytest

from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table, make_config
from oracle.c_oracle import COracle, lib, _p

pytestmark = pytest.mark.gpu


def _dirty(t, seed):
    rng = np.random.default_rng(seed)
    b = t.bases.copy()
    b[rng.random(len(b)) < 0.05] = ord('-')
    b[rng.random(len(b)) < 0.03] = ord('N')
    b[rng.random(len(b)) < 0.004] = ord('_')
    t.bases = b
    return t


def _oracle(t, sentinels, storage="f32"):
    o = COracle(t.n_snps, t.band, storage)
    assert lib().orc_fill(o._h, _p(np.ascontiguousarray(t.rank)), _p(np.ascontiguousarray(t.off)), _p(t.bases), t.n_reads, int(sentinels)) == 0
    return o


SHAPES = [dict(n=1000, reads=50000, k=3), dict(n=10000, reads=300000, k=5), dict(n=6000, reads=24000, k=None, k_lambda=10.0, k_max=21),
          dict(n=3000, reads=9000, k=None, k_lambda=20.0, k_max=33), dict(n=64, reads=900, k=4), dict(n=9, reads=60, k=2), dict(n=500, reads=40000, k=8)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_k%s" % (s["n"], s.get("k") or s.get("k_max")))
@pytest.mark.parametrize("sentinels", [False, True])
def test_owner_fill_equals_the_pair_loop(shape, sentinels, monkeypatch):
    kw = dict(shape)
    t = _dirty(make_support_table(kw.pop("n"), kw.pop("reads"), seed=31, **kw), 5)
    assert (t.rank == 0).any() and int((t.rank + np.diff(t.off)).max()) == t.n_snps
    o = _oracle(t, sentinels)
    want = o.export_band()
    got = {}
    for own in ("1", "0"):
        monkeypatch.setenv("GH_FILL_OWN", own)
        h = Hansel(t.n_snps, band=t.band)
        assert h.fill_from_support(t.rank, t.off, t.bases, use_end_sentinels=sentinels) == o.stats(), own
        got[own] = h.export_band()
        assert np.array_equal(got[own], want), own
        assert h.L == o.L
    monkeypatch.setenv("GH_FILL_OWN", "1")
    monkeypatch.setenv("GH_FILL_OWN_WIDE", "1")                   # 4-byte counters
    h = Hansel(t.n_snps, band=t.band, storage="f64")
    o64 = _oracle(t, sentinels, "f64")
    assert h.fill_from_support(t.rank, t.off, t.bases, use_end_sentinels=sentinels) == o64.stats()
    assert np.array_equal(h.export_band(), o64.export_band())


def test_owner_fill_accumulates_and_refuses_bad_symbols():
    # a second fill on top of the first: the flush adds to what is there (the first fill of a cleared tensor only stores)
    t1 = make_support_table(400, 12000, k=5, seed=1)
    t2 = _dirty(make_support_table(400, 9000, k=5, seed=2), 9)
    h = Hansel(400, band=4)
    o = COracle(400, 4)
    assert h.fill_from_support(t1.rank, t1.off, t1.bases) == o.fill(t1)
    assert h.fill_from_support(t2.rank, t2.off, t2.bases) == o.fill(t2)
    assert np.array_equal(h.export_band(), o.export_band())
    h.clear()
    o2 = COracle(400, 4)
    assert h.fill_from_support(t2.rank, t2.off, t2.bases) == o2.fill(t2)
    assert np.array_equal(h.export_band(), o2.export_band())
    # a symbol outside the alphabet: the read is skipped and the call says so, as with the other fills
    from gretel_amd._lib import SymbolError
    b = t2.bases.copy()
    b[t2.off[17]] = ord('X')
    h.clear()
    with pytest.raises(SymbolError):
        h.fill_from_support(t2.rank, t2.off, b)


def test_owner_fill_at_c5_and_c3_sizes():
    for name in ("C3", "C5"):
        t = make_config(name, seed=4)
        o = COracle(t.n_snps, t.band)
        h = Hansel(t.n_snps, band=t.band)
        assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
        # the whole tensor at C3; at C5 (206 MB as float32, 1.6 GB as the export's doubles) a sample of positions
        if name == "C3":
            assert np.array_equal(h.export_band(), o.export_band())
        else:
            rng = np.random.default_rng(0)
            for _ in range(4000):
                i = int(rng.integers(0, t.n_snps + 1)); d = int(rng.integers(1, t.band + 1)); a = int(rng.integers(0, 7)); b = int(rng.integers(0, 7))
                if i + d <= t.n_snps + 1:
                    assert h.get_observation(a, b, i, i + d) == o.get(a, b, i, i + d)
            assert h.spin(3)["paths"].tolist() == o.spin(3)["paths"].tolist()
