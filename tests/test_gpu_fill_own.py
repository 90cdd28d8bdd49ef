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


def _table_properties(rank, off, rpb=2048):
    """What gh_reads_upload used to work out on the host (round 1-5), in numpy."""
    n = len(rank)
    ks = np.diff(off)
    max_k = int(ks.max()) if n else 0
    srt = bool(n == 0 or (np.diff(rank) >= 0).all())
    if not srt or n == 0:
        return dict(max_k=max_k, sorted=srt, span_pos=0, dens128=0, first_at=np.zeros(0, dtype=np.int64))
    q0 = np.arange(0, n, rpb)
    q1 = np.minimum(q0 + rpb, n)
    span = int((rank[q1 - 1].astype(np.int64) - rank[q0]).max()) + max_k + 1
    lo = np.searchsorted(rank, rank.astype(np.int64) - 127, side="left")
    dens = int((np.arange(n) - lo + 1).max())
    fa = np.zeros(0, dtype=np.int64)
    if rank[0] >= 0 and int(rank[-1]) + 2 <= (1 << 28):
        fa = np.searchsorted(rank, np.arange(int(rank[-1]) + 2), side="left").astype(np.int64)
    return dict(max_k=max_k, sorted=srt, span_pos=span, dens128=dens, first_at=fa)


@pytest.mark.parametrize("case", ["c3", "ragged", "gaps", "one_read", "one_rank", "unsorted", "negative_first", "empty", "exact_block"])
def test_the_tables_properties_found_on_the_device(case):
    """gh_reads_upload finds the longest read, whether the ranks ascend, and for tables whose ranks do the span of a block, the
    density and first_at ON THE DEVICE behind the copies (k_reads_meta, k_reads_first_at, k_reads_meta2): the same values as the
    host passes of rounds 1-5 (the fills pick their kernel and the width of their LDS counters by them)."""
    from gretel_amd.hansel import DeviceReads
    rng = np.random.default_rng(17)
    if case == "c3":
        t = make_config("C3", seed=1)
        rank, off, bases, n_snps = t.rank, t.off, t.bases, t.n_snps
    else:
        n_snps = 5000
        n = dict(ragged=30000, gaps=5000, one_read=1, one_rank=700, unsorted=20000, negative_first=300, empty=0, exact_block=4096)[case]
        ks = rng.integers(1, 9, size=n)
        if case == "gaps":
            rank = np.sort(np.concatenate([rng.integers(0, 40, size=n // 2), rng.integers(3000, 4900, size=n - n // 2)]))
        elif case == "one_rank":
            rank = np.full(n, 77)
        elif case == "unsorted":
            rank = rng.integers(0, 4900, size=n)
        elif case == "negative_first":
            rank = np.sort(rng.integers(0, 4000, size=n)); rank[0] = -1
        else:
            rank = np.sort(rng.integers(0, 4900, size=n))
        rank = rank.astype(np.int32)
        off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int64)
        bases = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(off[-1]))
    h = Hansel(n_snps, band=16)
    r = DeviceReads(h, rank, off, bases)
    got, want = r.info(), _table_properties(rank, off)
    if case == "negative_first":
        want["first_at"] = np.zeros(0, dtype=np.int64)             # (no first_at for ranks that start below zero; the density all the same)
    assert got["max_k"] == want["max_k"] == r.max_k and got["sorted"] == want["sorted"], (got, want)
    assert got["span_pos"] == want["span_pos"] and got["dens128"] == want["dens128"], (case, got["span_pos"], want["span_pos"], got["dens128"], want["dens128"])
    assert np.array_equal(got["first_at"], want["first_at"])


def test_upload_refuses_offsets_that_run_backwards():
    from gretel_amd.hansel import DeviceReads
    h = Hansel(100, band=4)
    rank = np.arange(50, dtype=np.int32)
    off = np.arange(0, 153, 3, dtype=np.int64)
    off[31] = off[30] - 1
    with pytest.raises(Exception) as ei:
        DeviceReads(h, rank, off, np.full(int(off[-1]), ord("A"), dtype=np.uint8), max_k=3)
    assert "not monotone at read 30" in str(ei.value)
