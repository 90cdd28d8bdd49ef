"""The oracle against every known answer the reference holds for this path
(reference tests/test_test.py:17-52)."""
import os

import numpy as np
import pytest

from conftest import REFDATA
from gretel_amd import util
from oracle import gretel_ref as G
from oracle.c_oracle import COracle
from oracle.hansel_ref import Hansel, SYMBOLS, UNSYMBOLS

BAM = os.path.join(REFDATA, "test.bam")
VCF = os.path.join(REFDATA, "test.vcf.gz")
S = {c: i for i, c in enumerate(SYMBOLS)}

# reference tests/test_test.py:41-52
KNOWN = [('_', 'A', 0, 1, 1), ('A', 'A', 1, 2, 1), ('A', 'A', 1, 3, 1), ('A', 'A', 1, 4, 0),
         ('C', 'C', 1, 2, 1), ('C', 'C', 1, 3, 1), ('C', 'C', 1, 4, 0), ('T', 'T', 1, 2, 2),
         ('G', 'G', 1, 2, 0), ('G', 'G', 2, 3, 0), ('G', 'G', 3, 4, 1), ('G', '_', 4, 5, 1)]
# the full expected fill of the fixture (all other cells are 0)
FULL = {('_', 'A', 0, 1): 1, ('_', 'C', 0, 1): 1, ('_', 'T', 0, 1): 2, ('A', 'A', 1, 2): 1,
        ('C', 'C', 1, 2): 1, ('T', 'T', 1, 2): 2, ('A', 'A', 1, 3): 1, ('C', 'C', 1, 3): 1,
        ('A', 'A', 2, 3): 1, ('C', 'C', 2, 3): 1, ('G', 'G', 3, 4): 1, ('G', '_', 4, 5): 1}


def test_vcf_known_answers():
    # reference tests/test_test.py:17-29
    v = util.process_vcf(VCF, 'hoot', 1, 19)
    assert v["N"] == 3
    assert v["snp_rev"] == {0: 1, 1: 2, 2: 10}
    assert v["snp_fwd"] == {1: 0, 2: 1, 10: 2}
    assert len(v["region"]) == 20
    for i in range(20):
        assert v["region"][i] == (1 if i in (1, 2, 10) else 0)


def _fixture_table():
    v = util.process_vcf(VCF, 'hoot', 1, 20)
    rank, off, bases = util.support_table_from_bam(BAM, 'hoot', 1, 20, v)
    return v, rank, off, bases


def test_support_table_of_fixture():
    v, rank, off, bases = _fixture_table()
    assert v["N"] == 4
    seqs = [bases[off[i]:off[i + 1]].tobytes().decode() for i in range(len(rank))]
    assert list(zip(rank.tolist(), seqs)) == [(0, "AAA"), (0, "CCC"), (0, "TT"), (0, "TT"), (2, "GG")]


@pytest.mark.parametrize("storage", ["f32", "f64"])
def test_python_oracle_fill_known_answers(storage):
    from oracle.hansel_ref import HanselSpec
    v, rank, off, bases = _fixture_table()
    h = Hansel.init_matrix(SYMBOLS, UNSYMBOLS, v["N"], HanselSpec(storage=storage))
    reads = [(int(rank[i]), bases[off[i]:off[i + 1]].tobytes().decode()) for i in range(len(rank))]
    G.fill_from_support(h, reads, v["N"])
    assert h.n_slices == 5 and h.n_crumbs == 9 and h.L == 3     # tests/test_test.py:36-38, SURVEY §4
    for a, b, i, j, want in KNOWN:
        assert h.get_observation(a, b, i, j) == want
    d = h.dense()
    want = np.zeros_like(d)
    for (a, b, i, j), c in FULL.items():
        want[S[a], S[b], i, j] = c
    assert np.array_equal(d, want)
    assert [h.get_counts_at(i)["total"] for i in range(5)] == [4, 4, 2, 1, 1]
    assert G.gap_check(h, v["N"]) == -1


def test_c_oracle_fill_known_answers():
    v, rank, off, bases = _fixture_table()

    class T:
        pass
    t = T()
    t.rank, t.off, t.bases = rank, off, bases
    o = COracle(v["N"], band=2)
    assert o.fill(t) == (5, 9, 12)
    assert o.L == 3
    for a, b, i, j, want in KNOWN:
        assert o.get(S[a], S[b], i, j) == want
    band = o.export_band()
    assert band.sum() == sum(FULL.values())
    assert [o.counts_at(i)[7] for i in range(5)] == [4, 4, 2, 1, 1]
    assert o.gap_check() == -1


def test_reweight_call_sequence_golden():
    # enumeration of gretel/gretel.py:79-96 for N=3, and the call counts
    assert G.reweight_call_sequence(3) == [(0, 0), (0, 1), (0, 1), (1, 1), (1, 2), (0, 2), (1, 2),
                                           (2, 2), (2, 3), (3, 4)]
    for n, want in [(3, 10), (1000, 501501)]:
        assert len(G.reweight_call_sequence(n)) == want == n * (n + 3) // 2 + 1


def test_reweight_multiplicities():
    # SURVEY §8 a8
    n = 9
    from collections import Counter
    c = Counter(G.reweight_call_sequence(n))
    for p in range(0, n - 1):
        assert c[(p, p + 1)] == 2
    assert c[(n - 1, n)] == 1
    assert c[(n, n + 1)] == 1
    for p in range(0, n - 1):
        assert c[(p, n)] == 0
        for q in range(p + 2, n):
            assert c[(p, q)] == 1
    for p in range(0, n):
        assert c[(p, p)] == 1


def test_vcf_other_contigs_repeated_positions_and_plain_text(tmp_path):
    """process_vcf (gretel/util.py:354-414) takes the records of ONE contig inside the window, counts a position listed
    twice twice (n_snps, the reverse map) while the forward map and the region keep one entry, and reads plain text as
    well as gzip."""
    lines = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    recs = [("chr1", 5), ("chr10", 6), ("chr1", 9), ("chr1", 9), ("chr2", 9), ("chr1", 30), ("chr1", 12)]
    lines += ["%s\t%d\t.\tA\tC\t.\t.\t." % r for r in recs]
    plain = tmp_path / "v.vcf"
    plain.write_text("\n".join(lines) + "\n")
    import gzip
    gz = tmp_path / "v.vcf.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(("\n".join(lines) + "\n").encode())
    for path in (str(plain), str(gz)):
        v = util.process_vcf(path, "chr1", 5, 20)
        assert v["N"] == 4                                   # 5, 9, 9, 12 (30 lies outside, chr10 / chr2 are other contigs)
        assert v["snp_rev"] == {0: 5, 1: 9, 2: 9, 3: 12}
        assert v["snp_fwd"] == {5: 0, 9: 2, 12: 3}
        assert [int(p) for p in np.nonzero(v["region"])[0]] == [5, 9, 12]
        assert len(v["region"]) == 21
