"""Host-side pieces that need no GPU: the synthetic generator and the BAM/VCF decoding."""
import numpy as np

from gretel_amd.synth import make_support_table, make_config, CONFIGS


def test_generator_is_seeded_and_shaped():
    a = make_support_table(500, 20000, k=3, seed=4)
    b = make_support_table(500, 20000, k=3, seed=4)
    c = make_support_table(500, 20000, k=3, seed=5)
    assert np.array_equal(a.bases, b.bases) and np.array_equal(a.rank, b.rank)
    assert not np.array_equal(a.bases, c.bases)
    assert a.n_reads == 20000 and a.max_k == 3 and a.band == 2
    assert a.rank.min() == 0 and a.rank.max() == 500 - 3
    assert set(np.unique(a.bases).tolist()) <= set(b"ACGT")
    # every SNP has >= 2 alleles among the true haplotypes
    assert all(len(set(a.haplotypes[:, s].tolist())) >= 2 for s in range(500))


def test_tiling_bridges_every_adjacent_pair():
    t = make_support_table(300, 400, k=5, seed=0)       # almost only the tiling reads
    covered = np.zeros(300, dtype=bool)
    for r in range(t.n_reads):
        k = t.off[r + 1] - t.off[r]
        covered[t.rank[r]: t.rank[r] + k - 1] = True    # pair (s, s+1) bridged
    assert covered[:299].all()


def test_variable_k_config():
    t = make_support_table(2000, 5000, k=None, seed=1)
    ks = np.diff(t.off)
    assert ks.min() >= 2 and ks.max() <= 21
    assert (t.rank + ks <= 2000).all()
    assert set(CONFIGS) == {"C2", "C3", "C5"}


def test_result_exchange_refuses_to_drop_an_uncollected_step():
    # two staging slots: the third buffers() without a collect() in between would hand out a slot whose records nobody has
    # looked at -- the class used to finish and DROP that submission silently
    import torch
    from gretel_amd.dist import ResultExchange
    ex = ResultExchange(n_snps=10, max_paths=3, device=torch.device("cpu"), world=1, rank=0)
    for step in range(2):
        pv, rv = ex.buffers()
        pv[:] = step
        ex.submit(1, 0)
    import pytest
    with pytest.raises(RuntimeError, match="never collected"):
        ex.buffers()
    got = ex.collect()
    assert got[0]["n"] == 1 and int(got[0]["paths"][0, 0]) == 0
    pv, rv = ex.buffers()           # the collected slot is free again
    ex.submit(2, 0)
    assert ex.collect()[0]["paths"][0, 0] == 1 and ex.collect()[0]["n"] == 2 and ex.collect() is None
    ex.close()
