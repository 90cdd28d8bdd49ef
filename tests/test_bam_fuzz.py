"""The native BAM decoder under AddressSanitizer + UBSan on the CPU: a valid BAM is damaged a few thousand ways
(truncations, byte noise, blown-up length fields, unterminated names, shifted tails, flipped bits in the compressed
stream) and every variant must come back as a status code.  Also: the index and the no-index scans agree, long-read
CIGARs in the CG tag are resolved, a CIGAR that outruns its sequence is refused."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, REFDATA
from gretel_amd import bamio, util
from gretel_amd.synth import make_support_table


def test_malformed_bams_never_read_out_of_bounds(tmp_path):
    exe = str(tmp_path / "fuzz_bam")
    src = [os.path.join(ROOT, "tests", "native", "fuzz_bam.cpp"), os.path.join(ROOT, "gretel_amd", "csrc", "bam_support.cpp")]
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                         "-I", os.path.join(ROOT, "include"), "-o", exe] + src + ["-lz", "-ldl", "-pthread"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    t = make_support_table(40, 300, k=4, seed=1)
    bam, vcf = str(tmp_path / "v.bam"), str(tmp_path / "v.vcf.gz")
    contig, s, e = bamio.synth_to_files(t, bam, vcf)
    # (two threads, a new part every 16 records: the per-thread parts and their merge run under the sanitizers too)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", GIO_THREADS="2", GIO_PART_RECORDS="16", GIO_WINDOW="70000")
    for seed, src_bam, ctg, end in ((1, bam, contig, e), (2, os.path.join(REFDATA, "test.bam"), "hoot", 20)):
        run = subprocess.run([exe, src_bam, ctg, str(end), "700", str(seed), str(tmp_path / "scratch.bam")],
                             capture_output=True, text=True, env=env, timeout=600)
        assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
        assert "fuzz_bam done" in run.stdout


def test_index_and_full_scan_agree(tmp_path):
    t = make_support_table(3000, 60000, k=4, seed=3)
    bam, vcf = str(tmp_path / "s.bam"), str(tmp_path / "s.vcf.gz")
    contig, s, e = bamio.synth_to_files(t, bam, vcf)
    assert os.path.exists(bam + ".bai")
    for (ws, we) in ((1, e), (9000, 12000), (25000, e), (1, 500), (e - 5, e)):
        v = util.process_vcf(vcf, contig, ws, we)
        a = util.support_table_from_bam(bam, contig, ws, we, v)
        st = bamio.native_last_stats()
        assert st["used_index"] == 1
        os.environ["GIO_NO_INDEX"] = "1"
        try:
            b = util.support_table_from_bam(bam, contig, ws, we, v)
            assert bamio.native_last_stats()["used_index"] == 0
            assert bamio.native_last_stats()["records"] == t.n_reads
        finally:
            del os.environ["GIO_NO_INDEX"]
        c = util.support_table_from_bam(bam, contig, ws, we, v, decoder="python")
        os.environ["GIO_WINDOW"] = "100000"                  # the inflated window capped at one or two blocks
        try:
            d = util.support_table_from_bam(bam, contig, ws, we, v)
        finally:
            del os.environ["GIO_WINDOW"]
        for x, y, z, w in zip(a, b, c, d):
            assert np.array_equal(x, y) and np.array_equal(x, z) and np.array_equal(x, w)
        if we < e // 2:
            assert st["records"] < t.n_reads                 # the indexed scan stopped behind the window


def _one_read_bam(path, cigar_ops, seq, aux=b"", n_cigar_field=None):
    """BAM with one record on contig 'c' (length 100) at position 0; cigar_ops = [(op, len)]."""
    name = b"q\x00"
    packed = bytearray((len(seq) + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= "=ACMGRSVTWYHKDBN".index(ch) << (4 if (i & 1) == 0 else 0)
    ncig = len(cigar_ops) if n_cigar_field is None else n_cigar_field
    body = struct.pack("<iiBBHHHiiii", 0, 0, len(name), 42, 4681, ncig, 0, len(seq), -1, -1, 0)
    body += name + b"".join(struct.pack("<I", (n << 4) | op) for op, n in cigar_ops) + bytes(packed) + b"\x7e" * len(seq) + aux
    text = "@SQ\tSN:c\tLN:100\n"
    data = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<ii", 1, 2) + b"c\x00" + struct.pack("<i", 100)
    bamio.bgzf_write(path, data + struct.pack("<i", len(body)) + body)


def test_long_read_cigar_in_cg_tag_and_overrunning_cigar(tmp_path):
    vcf = str(tmp_path / "v.vcf.gz")
    bamio.write_vcf_gz(vcf, "c", [3, 6, 9])
    v = util.process_vcf(vcf, "c", 1, 50)
    seq = "ACGTACGTAC"
    real = [(0, 4), (2, 3), (0, 6)]                               # 4M3D6M: SNP 3 -> G, SNP 6 deleted, SNP 9 -> query offset 5
    plain, tagged = str(tmp_path / "p.bam"), str(tmp_path / "t.bam")
    _one_read_bam(plain, real, seq)
    cg = b"CGBI" + struct.pack("<I", len(real)) + b"".join(struct.pack("<I", (n << 4) | op) for op, n in real)
    _one_read_bam(tagged, [(4, len(seq)), (3, 13)], seq, aux=b"NMC\x01" + cg)      # placeholder <l_seq>S<ref_len>N + CG:B,I
    a = util.support_table_from_bam(plain, "c", 1, 50, v)
    b = util.support_table_from_bam(tagged, "c", 1, 50, v)
    assert a[2].tobytes() == b"G-C" and b[2].tobytes() == b"G-C"
    bad = str(tmp_path / "b.bam")
    _one_read_bam(bad, [(0, 40)], seq)                             # 40M on a 10-base read
    with pytest.raises(IOError):
        util.support_table_from_bam(bad, "c", 1, 50, v)
    _one_read_bam(bad, [(4, len(seq)), (3, 13)], seq)              # placeholder without the tag
    with pytest.raises(IOError):
        util.support_table_from_bam(bad, "c", 1, 50, v)


def test_repeated_keys_and_errors_keep_file_order_across_threads(tmp_path):
    """Records are worked on by several threads; the rows, the appends to a key seen before (util.py:199-207) and the
    first error must come out as a front-to-back scan gives them."""
    rng = np.random.default_rng(5)
    n_snps, spacing = 400, 10
    length = spacing * n_snps + spacing
    reads = []
    for r in range(30000):
        k = int(rng.integers(2, 7))
        first = int(rng.integers(0, n_snps - k))
        seq = bytearray(b"A" * ((k - 1) * spacing + 1))
        for q in range(k):
            seq[q * spacing] = ord("ACGT"[int(rng.integers(0, 4))])
        # one name in three is shared with other records (same flag: the same key, so the row is appended to)
        name = "r%d" % (r if r % 3 else int(rng.integers(0, 500)))
        reads.append((name, 0, 0, spacing * (first + 1) - 1, 42, "%dM" % len(seq), seq.decode()))
    reads.sort(key=lambda x: x[3])
    bam, vcf = str(tmp_path / "d.bam"), str(tmp_path / "d.vcf.gz")
    bamio.write_bam(bam, [("c", length)], reads)
    bamio.write_vcf_gz(vcf, "c", [spacing * (s + 1) for s in range(n_snps)])
    for (ws, we) in ((1, length), (1200, 2600)):
        v = util.process_vcf(vcf, "c", ws, we)
        a = util.support_table_from_bam(bam, "c", ws, we, v)
        assert bamio.native_last_stats()["threads"] >= 1
        b = util.support_table_from_bam(bam, "c", ws, we, v, decoder="python")
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert len(a[0]) < (30000 if ws == 1 else 12000)           # keys were shared
    # a record damaged in the middle of the file: the scan fails with that record's message, whatever the thread count
    import gzip
    with gzip.open(bam, "rb") as fh:
        raw = bytearray(fh.read())
    v = util.process_vcf(vcf, "c", 1, length)
    # find the 20000th record and blow up its l_seq
    o = raw.index(b"BAM\x01")
    l_text = struct.unpack_from("<i", raw, o + 4)[0]
    o += 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, o)[0]
        o += 4 + l_name + 4
    for _ in range(20000):
        o += 4 + struct.unpack_from("<i", raw, o)[0]
    struct.pack_into("<i", raw, o + 4 + 16, 1 << 27)
    bad = str(tmp_path / "bad.bam")
    bamio.bgzf_write(bad, bytes(raw))
    with pytest.raises(Exception) as ei:
        util.support_table_from_bam(bad, "c", 1, length, v)
    assert "exceed" in str(ei.value)


def test_a_decoy_record_inside_a_tag_cannot_fool_the_threads(tmp_path):
    """The decoder's threads start in the middle of the inflated window at a place that LOOKS like a record; here every
    record carries a byte-array tag that holds three perfectly formed records.  A thread that starts on such a decoy is
    found out (the chain of the thread in front of it does not end there) and the batch is done again front to back."""
    rng = np.random.default_rng(11)
    n_snps, spacing = 60, 10
    length = spacing * n_snps + spacing

    def record(name, pos0, seq, aux=b""):
        nm = name.encode() + b"\x00"
        packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i >> 1] |= "=ACMGRSVTWYHKDBN".index(ch) << (4 if (i & 1) == 0 else 0)
        body = struct.pack("<iiBBHHHiiii", 0, pos0, len(nm), 42, 4681, 1, 0, len(seq), -1, -1, 0)
        body += nm + struct.pack("<I", (len(seq) << 4) | 0) + bytes(packed) + b"\x7e" * len(seq) + aux
        return struct.pack("<i", len(body)) + body

    decoy = b"".join(record("decoy%d" % i, 5, "ACGTACGTACGT") for i in range(3))
    aux = b"XXBC" + struct.pack("<I", len(decoy)) + decoy
    recs, rows = [], []
    for r in range(1500):
        k = int(rng.integers(2, 5))
        first = int(rng.integers(0, n_snps - k))
        seq = bytearray(b"A" * ((k - 1) * spacing + 1))
        for q in range(k):
            seq[q * spacing] = ord("ACGT"[int(rng.integers(0, 4))])
        recs.append((spacing * (first + 1) - 1, "r%d" % r, seq.decode()))
    recs.sort(key=lambda x: x[0])
    text = "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:c\tLN:%d\n" % length
    data = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<ii", 1, 2) + b"c\x00" + struct.pack("<i", length)
    data += b"".join(record(nm, pos0, seq, aux) for pos0, nm, seq in recs)
    bam, vcf = str(tmp_path / "decoy.bam"), str(tmp_path / "decoy.vcf.gz")
    bamio.bgzf_write(bam, data)
    bamio.write_vcf_gz(vcf, "c", [spacing * (s + 1) for s in range(n_snps)])
    v = util.process_vcf(vcf, "c", 1, length)
    a = util.support_table_from_bam(bam, "c", 1, length, v)
    st = bamio.native_last_stats()
    b = util.support_table_from_bam(bam, "c", 1, length, v, decoder="python")
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert len(a[0]) == 1500 and st["records"] == 1500
    if st["threads"] > 1:
        assert st["reframed"] >= 1          # three quarters of the window are decoys: some thread started on one
