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


def _depth_bam(tmp_path, reads):
    bam = str(tmp_path / "deep.bam")
    vcf = str(tmp_path / "deep.vcf.gz")
    bamio.write_bam(bam, [("c", 400)], reads)
    bamio.write_vcf_gz(vcf, "c", [20, 30, 40, 50, 60])
    return bam, vcf


def _rows(t):
    rank, off, bases = t
    return [(int(rank[i]), bases[off[i]:off[i + 1]].tobytes().decode()) for i in range(len(rank))]


def test_max_depth_drops_what_the_pileup_buffer_would(tmp_path):
    # pysam's bam.pileup keeps at most max_depth reads in its buffer (default 8000; the reference passes none, util.py:137);
    # htslib's rule (bam_plp_push): a read is dropped when it starts where the iterator stands while the buffer holds more than
    # max_depth nodes -- the first read of a position always enters; a later one sees the reads that entered and end behind
    # position - 1, plus the list's sentinel.  Here by hand with max_depth = 3:
    #   pos 10: r0 enters (first), r1 (buffer: r0 + sentinel = 2 <= 3) enters, r2 (r0 r1 + 1 = 3) enters, r3 (3 + 1 = 4 > 3) DROPPED,
    #           r4 dropped likewise;
    #   pos 11: r5 is the first of its position: enters although four are open; r6 (r0 r1 r2 r5 + 1 = 5) dropped;
    #   pos 200: r0..r2 and r5 have ended (their ends <= 199): r7 enters (first), r8 (r7 + 1 = 2) enters.
    seq = "ACGTACGTAC" * 6
    reads = [("r%d" % i, 0, 0, 10, 60, "60M", seq) for i in range(5)]
    reads += [("r5", 0, 0, 11, 60, "60M", seq), ("r6", 0, 0, 11, 60, "60M", seq)]
    reads += [("r7", 0, 0, 200, 60, "60M", seq), ("r8", 0, 0, 200, 60, "60M", seq)]
    bam, vcf = _depth_bam(tmp_path, reads)
    v = util.process_vcf(vcf, "c", 1, 400)
    for dec in ("native", "python"):
        assert len(_rows(util.support_table_from_bam(bam, "c", 1, 400, v, decoder=dec, max_depth=0))) == 7        # r7, r8 show no SNP
        assert len(_rows(util.support_table_from_bam(bam, "c", 1, 400, v, decoder=dec))) == 7                     # 8000: nothing dropped
        got = _rows(util.support_table_from_bam(bam, "c", 1, 400, v, decoder=dec, max_depth=3))
        assert len(got) == 4, (dec, got)                                                                          # r0 r1 r2 r5
        assert got[:3] == [(0, "".join(seq[p - 11] for p in (20, 30, 40, 50, 60)))] * 3
        assert got[3] == (0, "".join(seq[p - 12] for p in (20, 30, 40, 50, 60)))
    util.support_table_from_bam(bam, "c", 1, 400, v, decoder="native", max_depth=3)
    assert bamio.native_last_stats()["depth_dropped"] == 3


def test_max_depth_native_equals_python_on_a_deep_pileup(tmp_path):
    # 3 000 reads of mixed lengths piled on a 300-base contig, caps from 5 to 400: the two decoders drop the same reads
    rng = np.random.default_rng(7)
    pos = np.sort(rng.integers(0, 250, size=3000))
    reads = []
    for i, p in enumerate(pos):
        ln = int(rng.integers(20, 120))
        cig = "%dM" % ln if rng.random() < 0.8 else "%dM%dD%dM" % (ln // 2, int(rng.integers(1, 9)), ln - ln // 2)
        flag = 0 if rng.random() < 0.9 else int(rng.choice([4, 256, 1, 3, 1024]))
        reads.append(("q%d" % i, flag, 0, int(p), 60, cig, "".join(rng.choice(list("ACGT"), size=ln))))
    bam, vcf = _depth_bam(tmp_path, reads)
    v = util.process_vcf(vcf, "c", 1, 300)
    full = _rows(util.support_table_from_bam(bam, "c", 1, 300, v, decoder="native", max_depth=0))
    last = 0
    for cap in (5, 37, 150, 400):
        a = util.support_table_from_bam(bam, "c", 1, 300, v, decoder="native", max_depth=cap)
        b = util.support_table_from_bam(bam, "c", 1, 300, v, decoder="python", max_depth=cap)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), cap
        assert last <= len(a[0]) < len(full)               # a larger buffer keeps more reads; every one of these caps drops some
        last = len(a[0])
    # a window that starts inside the pile: only records overlapping it reach the buffer
    for cap in (5, 60):
        a = util.support_table_from_bam(bam, "c", 45, 300, util.process_vcf(vcf, "c", 45, 300), decoder="native", max_depth=cap)
        b = util.support_table_from_bam(bam, "c", 45, 300, util.process_vcf(vcf, "c", 45, 300), decoder="python", max_depth=cap)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), cap


def test_the_depth_cap_refuses_an_unsorted_bam(tmp_path):
    # the cap's running sum (and pysam's pileup) needs ascending starts: both decoders say so instead of counting wrongly;
    # without the cap (max_depth=0) the decoders take records in any order, as before
    seq = "ACGTACGTAC" * 6
    reads = [("a", 0, 0, 150, 60, "60M", seq), ("b", 0, 0, 10, 60, "60M", seq), ("c", 0, 0, 10, 60, "60M", seq)]
    bam = str(tmp_path / "unsorted.bam")
    vcf = str(tmp_path / "u.vcf.gz")
    bamio.write_bam(bam, [("c", 400)], reads, index=False)
    bamio.write_vcf_gz(vcf, "c", [20, 30, 40, 50, 60])
    v = util.process_vcf(vcf, "c", 1, 400)
    for dec in ("native", "python"):
        with pytest.raises(Exception) as ei:
            util.support_table_from_bam(bam, "c", 1, 400, v, decoder=dec)
        assert "coordinate" in str(ei.value), (dec, str(ei.value))
        assert len(_rows(util.support_table_from_bam(bam, "c", 1, 400, v, decoder=dec, max_depth=0))) == 2


def test_the_depth_histogram_stays_short_over_a_long_window(tmp_path):
    # reads spread over 300 kb with a crowd at the far end: the cap is applied there as at the start (the native decoder's
    # histogram of ends is rebased as it goes instead of growing with the window)
    seq = "ACGTACGTAC" * 5
    reads = [("s%d" % i, 0, 0, 100 + 997 * i, 60, "50M", seq) for i in range(300)]
    reads += [("t%d" % i, 0, 0, 299_000 + (i // 4), 60, "50M", seq) for i in range(40)]
    reads.sort(key=lambda r: r[3])
    bam = str(tmp_path / "long.bam")
    vcf = str(tmp_path / "long.vcf.gz")
    bamio.write_bam(bam, [("c", 300_100)], reads)
    bamio.write_vcf_gz(vcf, "c", [299_010, 299_020, 299_030])
    v = util.process_vcf(vcf, "c", 1, 300_100)
    a = util.support_table_from_bam(bam, "c", 1, 300_100, v, decoder="native", max_depth=6)
    b = util.support_table_from_bam(bam, "c", 1, 300_100, v, decoder="python", max_depth=6)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert 0 < len(a[0]) < 40


def test_the_depth_cap_shown_idle_batch_by_batch_then_binding(tmp_path, monkeypatch):
    # The native decoder first SHOWS, on all threads, that a batch cannot reach the cap (reads per stretch of the longest span)
    # and sets its candidates aside; the first batch it cannot show this for takes the sequential pass over everything set aside
    # and goes on from that state.  Sparse reads over 3 Mb (several inflated windows of ~1 block each), a crowd, sparse reads
    # again, a second crowd: the same rows as the Python decoder's one sequential pass, with caps that bind nowhere / in the
    # crowds only / nearly everywhere.
    monkeypatch.setenv("GIO_WINDOW", "70000")
    rng = np.random.default_rng(11)
    seq = lambda n: "".join(rng.choice(list("ACGT"), size=n))
    reads = [("s%d" % i, 0, 0, 100 + 997 * i + int(rng.integers(0, 300)), 60, "50M", seq(50)) for i in range(1500)]
    reads += [("t%d" % i, 0, 0, 1_500_000 + (i // 5), 60, "%dM" % (30 + i % 40), seq(30 + i % 40)) for i in range(300)]
    reads += [("u%d" % i, 0, 0, 1_600_000 + 911 * i, 60, "50M", seq(50)) for i in range(1500)]
    reads += [("w%d" % i, 0, 0, 2_990_000 + (i // 3), 60, "60M", seq(60)) for i in range(200)]
    reads.sort(key=lambda r: r[3])
    bam = str(tmp_path / "crowds.bam")
    vcf = str(tmp_path / "crowds.vcf.gz")
    bamio.write_bam(bam, [("c", 3_000_100)], reads)
    snps = sorted(set([int(x) for x in rng.integers(1, 3_000_000, size=60000)] + list(range(1_500_001, 1_500_100, 7)) + list(range(2_990_001, 2_990_120, 5))))
    bamio.write_vcf_gz(vcf, "c", snps)
    v = util.process_vcf(vcf, "c", 1, 3_000_100)
    n_rows = []
    for cap in (0, 8000, 40, 6, 1):
        a = util.support_table_from_bam(bam, "c", 1, 3_000_100, v, decoder="native", max_depth=cap)
        st = bamio.native_last_stats()
        b = util.support_table_from_bam(bam, "c", 1, 3_000_100, v, decoder="python", max_depth=cap)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), cap
        assert st["blocks"] >= 5                           # (several windows: the state is carried from batch to batch)
        n_rows.append(len(a[0]))
        assert (st["depth_dropped"] > 0) == (cap in (40, 6, 1)), (cap, st)
    assert n_rows[0] == n_rows[1] > n_rows[2] > n_rows[3] > n_rows[4]


def test_the_decoders_kept_buffers_change_nothing(tmp_path, monkeypatch):
    """libgretel_io.so keeps its large working buffers between calls (include/gretel_io.h: gio_release_buffers, GIO_KEEP_MB): the same
    table with fresh buffers, with reused ones (handed out as the decode before left them) and after they were released."""
    from gretel_amd import bamio, util
    from gretel_amd.synth import make_support_table
    monkeypatch.setenv("GIO_PART_RECORDS", "512")
    t = make_support_table(2000, 120000, k=6, seed=11, n_haps=5, err=0.01)          # (large enough for blocks beyond 4 MB to be kept)
    bam, vcf = str(tmp_path / "k.bam"), str(tmp_path / "k.vcf.gz")
    contig, start, end = bamio.synth_to_files(t, bam, vcf)
    v = util.process_vcf(vcf, contig, start, end)
    bamio.native_release_buffers()
    first = [np.array(a) for a in util.support_table_from_bam(bam, contig, start, end, v)]
    for round_ in range(3):
        again = util.support_table_from_bam(bam, contig, start, end, v)
        assert all(np.array_equal(a, b) for a, b in zip(first, again)), round_
        del again
        if round_ == 1:
            bamio.native_release_buffers()
    assert np.array_equal(first[0], t.rank) and np.array_equal(first[1], t.off) and np.array_equal(first[2], t.bases)


def test_a_forked_child_decodes_with_a_pool_of_its_own(tmp_path):
    """the decoder's thread pool and buffer cache are process-wide singletons guarded by mutexes.  A child forked while the
    parent has decoded (its workers exist, its cache holds blocks) gets fresh ones (pthread_atfork): its decode returns, with the
    parent's answer -- it used to depend on no thread holding a pool mutex at the moment of the fork."""
    import os
    import threading
    from gretel_amd import bamio, util
    from gretel_amd.synth import make_support_table
    t = make_support_table(1500, 60000, k=6, seed=3, n_haps=5, err=0.01)
    bam, vcf = str(tmp_path / "f.bam"), str(tmp_path / "f.vcf.gz")
    contig, start, end = bamio.synth_to_files(t, bam, vcf)
    v = util.process_vcf(vcf, contig, start, end)
    want = [np.array(a) for a in util.support_table_from_bam(bam, contig, start, end, v)]
    # a thread of the parent keeps decoding (and so keeps taking the pool's mutexes) while the children are forked
    stop = threading.Event()

    def churn():
        while not stop.is_set():
            util.support_table_from_bam(bam, contig, start, end, v)

    th = threading.Thread(target=churn, daemon=True)
    th.start()
    try:
        for attempt in range(4):
            r, w = os.pipe()
            pid = os.fork()
            if pid == 0:
                code = 3
                try:
                    os.close(r)
                    got = util.support_table_from_bam(bam, contig, start, end, v)
                    code = 0 if all(np.array_equal(a, b) for a, b in zip(want, got)) else 2
                    os.write(w, b"k")
                finally:
                    os._exit(code)
            os.close(w)
            import select
            ready, _, _ = select.select([r], [], [], 60.0)
            if not ready:
                os.kill(pid, 9)
            _, status = os.waitpid(pid, 0)
            os.close(r)
            assert ready, "the forked child's decode never returned"
            assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    finally:
        stop.set()
        th.join(30.0)


def test_prefetch_is_taken_over_by_the_matching_decode_only(tmp_path):
    """gio_prefetch (include/gretel_io.h): the window's blocks read and inflated on the library's thread while the caller parses its
    VCF.  The decode of the same window takes the work over and gives the same table; a decode of another window, a prefetch of a
    file that is not there, a cancelled prefetch change nothing."""
    t = make_support_table(300, 6000, k=None, seed=3, k_max=6)
    bam, vcf = str(tmp_path / "p.bam"), str(tmp_path / "p.vcf.gz")
    contig, start, end = bamio.synth_to_files(t, bam, vcf)
    v = util.process_vcf(vcf, contig, start, end)
    plain = util.support_table_from_bam(bam, contig, start, end, v)
    assert bamio.native_last_stats()["prefetched"] == 0
    util.prefetch_bam(bam, contig, start, end)
    got = util.support_table_from_bam(bam, contig, start, end, v)
    st = bamio.native_last_stats()
    assert st["prefetched"] == 1 and st["records"] > 0 and st["blocks"] > 0
    for x, y in zip(plain, got):
        assert np.array_equal(x, y)
    # (taken: the next decode starts afresh)
    util.support_table_from_bam(bam, contig, start, end, v)
    assert bamio.native_last_stats()["prefetched"] == 0
    # another window than the one announced: discarded
    util.prefetch_bam(bam, contig, start, end - 7)
    got = util.support_table_from_bam(bam, contig, start, end, v)
    assert bamio.native_last_stats()["prefetched"] == 0
    for x, y in zip(plain, got):
        assert np.array_equal(x, y)
    # a file that is not there / a contig the file does not have: no error here, the decode reports
    util.prefetch_bam(str(tmp_path / "nothing.bam"), contig, start, end)
    util.prefetch_bam(bam, "no_such_contig", start, end)
    with pytest.raises(KeyError):
        util.support_table_from_bam(bam, "no_such_contig", start, end, v)
    bamio.native_prefetch(bam, contig, start, end)
    bamio.native_prefetch_cancel()
    got = util.support_table_from_bam(bam, contig, start, end, v)
    assert bamio.native_last_stats()["prefetched"] == 0
    # several windows of inflated data: the prefetch stops at the window's cap and the decode reads on
    os.environ["GIO_WINDOW"] = "70000"
    try:
        util.prefetch_bam(bam, contig, start, end)
        got = util.support_table_from_bam(bam, contig, start, end, v)
        assert bamio.native_last_stats()["prefetched"] == 1
    finally:
        del os.environ["GIO_WINDOW"]
    for x, y in zip(plain, got):
        assert np.array_equal(x, y)


def test_the_table_in_the_callers_memory(tmp_path):
    """gio_support_table_from_bam_alloc: the three arrays where the caller's allocator puts them (here: numpy buffers), the same
    table; an allocator that has nothing gives an error, not a crash; the longest row is reported (gio_stats.max_row_len)."""
    t = make_support_table(300, 6000, k=None, seed=4, k_max=6)
    bam, vcf = str(tmp_path / "a.bam"), str(tmp_path / "a.vcf.gz")
    contig, start, end = bamio.synth_to_files(t, bam, vcf)
    v = util.process_vcf(vcf, contig, start, end)
    plain = util.support_table_from_bam(bam, contig, start, end, v)
    assert bamio.native_last_stats()["max_row_len"] == int(np.diff(plain[1]).max())

    class Arena:
        def __init__(self, broke=None):
            self.bufs, self.broke, self.asked = {}, broke, []

        def alloc(self, which, nbytes):
            self.asked.append((which, nbytes))
            if which == self.broke:
                return 0
            self.bufs[which] = np.empty(nbytes + 64, dtype=np.uint8)
            return self.bufs[which].ctypes.data

        def view(self, which, dtype, count):
            return np.frombuffer(self.bufs[which], dtype=dtype, count=count)

    a = Arena()
    got = util.support_table_from_bam(bam, contig, start, end, v, arena=a)
    assert [w for w, _ in a.asked] == [0, 1, 2]
    assert a.asked[0][1] == 4 * len(plain[0]) and a.asked[1][1] == 8 * (len(plain[0]) + 1) and a.asked[2][1] == len(plain[2])
    for x, y in zip(plain, got):
        assert np.array_equal(x, y)
    with pytest.raises(IOError) as ei:
        util.support_table_from_bam(bam, contig, start, end, v, arena=Arena(broke=1))
    assert "allocator" in str(ei.value)

    class Raises(Arena):
        def alloc(self, which, nbytes):
            raise MemoryError("nothing left")

    with pytest.raises(MemoryError):
        util.support_table_from_bam(bam, contig, start, end, v, arena=Raises())


def test_a_bam_through_a_pipe(tmp_path):
    """A path that is no regular file (a FIFO: `samtools view -b ... > fifo`) is read front to back with plain reads -- no index,
    no size, no parallel slices -- and gives the table the file gives."""
    import threading
    t = make_support_table(200, 5000, k=None, seed=9, k_max=6)
    bam, vcf = str(tmp_path / "f.bam"), str(tmp_path / "f.vcf.gz")
    contig, start, end = bamio.synth_to_files(t, bam, vcf)
    v = util.process_vcf(vcf, contig, start, end)
    want = util.support_table_from_bam(bam, contig, start, end, v)
    fifo = str(tmp_path / "pipe.bam")
    os.mkfifo(fifo)

    def feed():
        with open(bam, "rb") as src, open(fifo, "wb") as dst:
            dst.write(src.read())

    th = threading.Thread(target=feed)
    th.start()
    try:
        got = util.support_table_from_bam(fifo, contig, start, end, v)
    finally:
        th.join(timeout=30)
    assert bamio.native_last_stats()["used_index"] == 0
    for x, y in zip(want, got):
        assert np.array_equal(x, y)
