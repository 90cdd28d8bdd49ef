"""BAM -> support table: the native decoder (libgretel_io.so) against the pure-Python restatement and
against hand-derived expectations, on the reference fixture and on BAMs written here with indels,
clips, ref-skips, filtered flags and reads that start before the window."""
import os

import numpy as np
import pytest

from conftest import REFDATA
from gretel_amd import bamio, util
from gretel_amd.synth import make_support_table

BAM = os.path.join(REFDATA, "test.bam")
VCF = os.path.join(REFDATA, "test.vcf.gz")


def _both(bam, vcf, contig, s, e, stepper="samtools"):
    v = util.process_vcf(vcf, contig, s, e)
    a = util.support_table_from_bam(bam, contig, s, e, v, stepper, decoder="native")
    b = util.support_table_from_bam(bam, contig, s, e, v, stepper, decoder="python")
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    rank, off, bases = a
    return v, [(int(rank[i]), bases[off[i]:off[i + 1]].tobytes().decode()) for i in range(len(rank))]


@pytest.mark.parametrize("window", [(1, 20), (1, 19), (10, 20), (2, 15), (5, 9)])
def test_fixture_windows_native_equals_python(window):
    _both(BAM, VCF, "hoot", *window)
    assert util.get_ref_len_from_bam(BAM, "hoot") == util.get_ref_len_from_bam(BAM, "hoot", decoder="python") == 20


def test_fixture_expected_rows():
    _, rows = _both(BAM, VCF, "hoot", 1, 20)
    assert rows == [(0, "AAA"), (0, "CCC"), (0, "TT"), (0, "TT"), (2, "GG")]
    _, rows = _both(BAM, VCF, "hoot", 2, 20)          # reads start before the window: LEFTMOST := start (util.py:165-171)
    assert rows == [(0, "AA"), (0, "CC"), (0, "T"), (0, "T"), (1, "GG")]
    _, rows = _both(BAM, VCF, "meow", 1, 20)
    assert rows == [(0, "N")]                     # read6 carries N at meow:5


def test_cigar_cases(tmp_path):
    # contig of 60 bp, SNPs at 5, 10, 15, 20, 25, 30
    bam, vcf = str(tmp_path / "c.bam"), str(tmp_path / "c.vcf.gz")
    snps = [5, 10, 15, 20, 25, 30]
    bamio.write_vcf_gz(vcf, "ctg", snps)

    def seq(n, marks):            # n bases of 'A' with letters planted at given query offsets
        s = bytearray(b"A" * n)
        for q, ch in marks.items():
            s[q] = ord(ch)
        return s.decode()
    reads = [
        # plain match covering SNPs 5,10,15 (query offsets 4,9,14)
        ("plain", 0, 0, 0, 42, "16M", seq(16, {4: "C", 9: "G", 14: "T"})),
        # soft clip 3, then 12M from pos 4 (1-based): SNPs 5 (q=3+1), 10 (q=3+6), 15 (q=3+11)
        ("clip", 0, 0, 3, 42, "3S12M", seq(15, {4: "G", 9: "T", 14: "C"})),
        # deletion over SNP 10: 6M 5D 6M from pos 3 -> covers 3..8, del 9..13, 14..19: SNP5 (q=2), SNP10 '-', SNP15 (q=7)
        ("del", 0, 0, 2, 42, "6M5D6M", seq(12, {2: "T", 7: "G"})),
        # insertion right after SNP 10: 8M 2I 8M from pos 3 -> SNP5 q=2, SNP10 q=7 (first base of the allele), SNP15 q=8+2+4=14
        ("ins", 0, 0, 2, 42, "8M2I8M", seq(18, {2: "C", 7: "C", 14: "C"})),
        # ref skip over SNPs 15 and 20: 5M 12N 5M from pos 8 -> 8..12 (SNP10 q=2), skip 13..24 ('-','-'), 25..29 (SNP25 q=5)
        ("skip", 0, 0, 7, 42, "5M12N5M", seq(10, {2: "G", 5: "T"})),
        # filtered: duplicate, secondary, qc-fail, unmapped
        ("dup", 0x400, 0, 0, 42, "16M", seq(16, {})),
        ("sec", 0x100, 0, 0, 42, "16M", seq(16, {})),
        ("qcf", 0x200, 0, 0, 42, "16M", seq(16, {})),
        # paired, not proper: dropped by "samtools", kept by "all"
        ("orph", 0x1 | 0x40, 0, 18, 42, "14M", seq(14, {1: "T", 6: "T", 11: "T"})),
        # proper pair, read 2
        ("pair", 0x1 | 0x2 | 0x80, 0, 18, 42, "14M", seq(14, {1: "G", 6: "G", 11: "G"})),
        # single SNP only
        ("one", 0, 0, 27, 42, "4M", seq(4, {2: "C"})),
    ]
    reads.sort(key=lambda r: r[3])
    bamio.write_bam(bam, [("ctg", 60)], reads)
    v, rows = _both(bam, vcf, "ctg", 1, 60)
    assert v["N"] == 6
    assert rows == [(0, "CGT"), (0, "T-G"), (0, "CCC"), (0, "GTC"), (1, "G--T"), (3, "GGG"), (5, "C")]
    _, rows_all = _both(bam, vcf, "ctg", 1, 60, stepper="all")
    assert (3, "TTT") in rows_all and len(rows_all) == len(rows) + 1
    # window that cuts reads: start at 8 -> reads starting before it get LEFTMOST = 8 (rank 0 = SNPs before 8 in the window)
    _, rows = _both(bam, vcf, "ctg", 8, 27)
    assert rows == [(0, "GT"), (0, "-G"), (0, "CC"), (0, "TC"), (0, "G--T"), (2, "GG")]


def test_synthetic_files_roundtrip(tmp_path):
    t = make_support_table(120, 1500, k=None, seed=5, k_max=6)
    bam, vcf = str(tmp_path / "s.bam"), str(tmp_path / "s.vcf.gz")
    contig, s, e = bamio.synth_to_files(t, bam, vcf)
    v, rows = _both(bam, vcf, contig, s, e)
    assert v["N"] == t.n_snps
    want = sorted(t.reads())
    assert sorted(rows) == want
    assert util.get_ref_len_from_bam(bam, contig) == e


def test_snpper_on_fixture_and_cigar_cases(tmp_path, capsys):
    from gretel_amd import snpper
    # fixture: hoot has A/C/T at 1,2 ; A/C at 10 ; only G at 20 -> sites 1,2,10 (what the reference's own VCF lists, plus none at 20)
    assert snpper.call_sites(BAM, "hoot", host=True) == [1, 2, 10]
    assert snpper.call_sites(BAM, "hoot", depth=1, host=True) == []  # no base is seen on more than one read twice
    assert snpper.call_sites(BAM, "hoot", 2, 9, host=True) == [2]
    assert snpper.main(["--bam", BAM, "--contig", "hoot", "--host"]) == 0
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "##fileformat=VCFv4.2" and out[1] == "hoot\t1\t.\tA\tC,T,G\t0\t.\tINFO" and len(out) == 4
    cov = bamio.native_count_coverage(BAM, "hoot", 0, 20)
    assert cov.shape == (4, 20) and cov[:, 0].tolist() == [1, 1, 0, 2] and cov[:, 19].tolist() == [0, 0, 1, 0]
    assert cov[:, 4].sum() == 0                                      # N bases are not counted


def test_match_runs_rebuild_the_coverage_counts(tmp_path):
    # the aligned runs handed to the GPU histogram carry exactly the coverage the host counter reports
    t = make_support_table(200, 3000, k=4, seed=5)
    bam, vcf = str(tmp_path / "s.bam"), str(tmp_path / "s.vcf.gz")
    contig, s, e = bamio.synth_to_files(t, bam, vcf)
    for (a, b) in ((0, e), (500, 1500), (0, 37)):
        ref, off, codes = bamio.native_match_runs(bam, contig, a, b)
        cov = np.zeros((4, b - a), dtype=np.int32)
        for r in range(len(ref)):
            c = codes[off[r]:off[r + 1]]
            p = ref[r] - a + np.arange(len(c))
            ok = c < 4
            np.add.at(cov, (c[ok], p[ok]), 1)
        assert np.array_equal(cov, bamio.native_count_coverage(bam, contig, a, b))


def test_debug_prints_and_depth_cap(capsys):
    """--debugreads / --debugpos (gretel/util.py:211-224) and the opt-in max_depth of the pileup the reference inherits, on the
    reference's own fixture: read keys are name_flag_1or2 (util.py:160), positions 1-based."""
    import io
    from gretel_amd import util
    v = util.process_vcf(VCF, 'hoot', 1, 20)
    out = io.StringIO()
    rank, off, bases = util.support_table_from_bam(BAM, 'hoot', 1, 20, v, debug_reads={"read1"}, debug_pos={2}, debug_out=out)
    ref = util.support_table_from_bam(BAM, 'hoot', 1, 20, v)
    assert np.array_equal(rank, ref[0]) and np.array_equal(off, ref[1]) and np.array_equal(bases, ref[2])
    lines = out.getvalue().splitlines()
    assert "read1_0_0 1 A" in lines and "read1_0_0 2 A" in lines and "read1_0_0 10 A" in lines and "RANK read1_0_0 0" in lines
    at2 = [l for l in lines if l.split()[1] == "2" and not l.startswith("RANK")]
    assert len(at2) >= 4                                  # every read covering position 2 is listed for --debugpos
    # depth cap: with one read allowed open at a time only the first of the reads starting together survives
    r1 = util.support_table_from_bam(BAM, 'hoot', 1, 20, v, max_depth=1)
    assert 0 < len(r1[0]) < len(ref[0])
    r9 = util.support_table_from_bam(BAM, 'hoot', 1, 20, v, max_depth=8000)
    assert np.array_equal(r9[2], ref[2])
