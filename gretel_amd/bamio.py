"""
Minimal BAM / bgzipped-VCF *writers* (standard library only) so that the synthetic configs can be run
end to end through the same files the reference takes (`gretel <bam> <vcf.gz> <contig>`), and the
ctypes binding of the native BAM decoder (libgretel_io.so, include/gretel_io.h).
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import zlib

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
IO_SO = os.path.join(_HERE, "libgretel_io.so")
_io = None

_SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_CIGAR_OP = {c: i for i, c in enumerate("MIDNSHP=X")}


# ---------------------------------------------------------------------------------------------
# BGZF / BAM / VCF writing
# ---------------------------------------------------------------------------------------------
def _bgzf_block(payload):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(payload) + comp.flush()
    bsize = len(cdata) + 25
    head = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord('B'), ord('C'), 2, bsize)
    return head + cdata + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload))


def bgzf_write(path, data):
    with open(path, "wb") as fh:
        for o in range(0, len(data), 0xff00):
            fh.write(_bgzf_block(data[o:o + 0xff00]))
        fh.write(_bgzf_block(b""))          # EOF marker


def _parse_cigar(cigar):
    out, num = [], ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            out.append((_CIGAR_OP[ch], int(num)))
            num = ""
    return out


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def write_bam(path, refs, reads):
    """refs: [(name, length)]; reads: iterable of (qname, flag, ref_index, pos0, mapq, cigar_str, seq_str),
    already coordinate sorted."""
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    out = [b"BAM\x01", struct.pack("<i", len(text)), text.encode(), struct.pack("<i", len(refs))]
    for name, ln in refs:
        out += [struct.pack("<i", len(name) + 1), name.encode() + b"\x00", struct.pack("<i", ln)]
    for qname, flag, rid, pos, mapq, cigar, seq in reads:
        cg = _parse_cigar(cigar)
        reflen = sum(n for op, n in cg if op in (0, 2, 3, 7, 8))
        l_seq = len(seq)
        packed = bytearray((l_seq + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i >> 1] |= _SEQ_CODE[ch] << (4 if (i & 1) == 0 else 0)
        name = qname.encode() + b"\x00"
        body = struct.pack("<iiBBHHHiiii", rid, pos, len(name), mapq, _reg2bin(pos, pos + max(1, reflen)), len(cg), flag,
                           l_seq, -1, -1, 0)
        body += name + b"".join(struct.pack("<I", (n << 4) | op) for op, n in cg) + bytes(packed) + b"\x7e" * l_seq
        out.append(struct.pack("<i", len(body)) + body)
    bgzf_write(path, b"".join(out))


def write_vcf_gz(path, contig, positions):
    lines = ["##fileformat=VCFv4.2\n", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"]
    lines += ["%s\t%d\t.\tA\tC,T,G\t0\t.\tINFO\n" % (contig, p) for p in positions]
    bgzf_write(path, "".join(lines).encode())


def synth_to_files(table, bam_path, vcf_path, contig="synth", spacing=10):
    """A support table as real files: SNP s (0-based) sits at 1-based position spacing*(s+1); every read is one
    all-M alignment from its first to its last SNP with 'A' between the SNPs."""
    n = table.n_snps
    length = spacing * n + spacing
    reads = []
    bases = table.bases.tobytes()
    for r in range(table.n_reads):
        k = int(table.off[r + 1] - table.off[r])
        rk = int(table.rank[r])
        seq = bytearray(b"A" * ((k - 1) * spacing + 1))
        for q in range(k):
            seq[q * spacing] = bases[table.off[r] + q]
        pos0 = spacing * (rk + 1) - 1
        reads.append(("r%d" % r, 0, 0, pos0, 42, "%dM" % len(seq), seq.decode()))
    reads.sort(key=lambda x: x[3])
    write_bam(bam_path, [(contig, length)], reads)
    write_vcf_gz(vcf_path, contig, [spacing * (s + 1) for s in range(n)])
    return contig, 1, length


# ---------------------------------------------------------------------------------------------
# native decoder binding
# ---------------------------------------------------------------------------------------------
class _gio_table(C.Structure):
    _fields_ = [("rank", C.POINTER(C.c_int32)), ("off", C.POINTER(C.c_int64)), ("bases", C.POINTER(C.c_uint8)),
                ("n_reads", C.c_int64), ("n_bases", C.c_int64)]


def io_lib():
    global _io
    if _io is None:
        if not os.path.exists(IO_SO):
            raise ImportError("gretel_amd: %s is missing; build it with `make -C gretel_amd/csrc`" % IO_SO)
        L = C.CDLL(IO_SO)
        L.gio_last_error.restype = C.c_char_p
        L.gio_ref_len.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int64)]
        L.gio_support_table_from_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int,
                                                 C.POINTER(_gio_table)]
        L.gio_table_free.argtypes = [C.POINTER(_gio_table)]
        L.gio_count_coverage.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]
        _io = L
    return _io


def native_ref_len(bam_path, contig):
    L = io_lib()
    out = C.c_int64()
    if L.gio_ref_len(bam_path.encode(), contig.encode(), C.byref(out)):
        raise KeyError(L.gio_last_error().decode())
    return out.value


def native_support_table(bam_path, contig, start_pos, end_pos, region, stepper="samtools"):
    L = io_lib()
    reg = np.ascontiguousarray(np.asarray(region) != 0, dtype=np.uint8)
    if len(reg) < end_pos + 1:
        reg = np.concatenate([reg, np.zeros(end_pos + 1 - len(reg), dtype=np.uint8)])
    t = _gio_table()
    rc = L.gio_support_table_from_bam(bam_path.encode(), contig.encode(), int(start_pos), int(end_pos),
                                      reg.ctypes.data, int(stepper == "all"), C.byref(t))
    if rc:
        msg = L.gio_last_error().decode()
        raise (KeyError if rc == -5 else IOError)(msg)
    try:
        n, nb = t.n_reads, t.n_bases
        rank = np.ctypeslib.as_array(t.rank, shape=(max(n, 1),))[:n].copy()
        off = np.ctypeslib.as_array(t.off, shape=(n + 1,)).copy()
        bases = np.ctypeslib.as_array(t.bases, shape=(max(nb, 1),))[:nb].copy()
    finally:
        L.gio_table_free(C.byref(t))
    return rank, off, bases


def native_count_coverage(bam_path, contig, start0, stop):
    """int32[4][stop-start0]: A,C,G,T counts per position, like pysam's count_coverage(..., quality_threshold=0,
    read_callback='nofilter') as gretel/snpper.py:29 calls it."""
    L = io_lib()
    counts = np.zeros((4, max(0, stop - start0)), dtype=np.int32)
    if L.gio_count_coverage(bam_path.encode(), contig.encode(), int(start0), int(stop), counts.ctypes.data):
        msg = L.gio_last_error().decode()
        raise (KeyError if "not in" in msg else IOError)(msg)
    return counts
