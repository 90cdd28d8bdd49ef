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


BGZF_PAYLOAD = 0xff00


def bgzf_write(path, data, level=6):
    """Writes `data` as BGZF blocks of BGZF_PAYLOAD bytes; returns the file offset of every block (for write_bai)."""
    offs = []
    pos = 0
    with open(path, "wb") as fh:
        for o in range(0, len(data), BGZF_PAYLOAD):
            payload = bytes(data[o:o + BGZF_PAYLOAD])
            comp = zlib.compressobj(level, zlib.DEFLATED, -15)
            cdata = comp.compress(payload) + comp.flush()
            head = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord('B'), ord('C'), 2, len(cdata) + 25)
            blk = head + cdata + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload))
            offs.append(pos)
            fh.write(blk)
            pos += len(blk)
        offs.append(pos)
        fh.write(_bgzf_block(b""))          # EOF marker
    return offs


def write_bai(path, n_ref, rec_ref, rec_beg, rec_end, rec_uoff, rec_ulen, block_offs):
    """A .bai for a BAM written by bgzf_write: per record its reference index, 0-based [beg, end), and where its bytes
    sit in the uncompressed stream (offset of its block_size field, total length).  Bins with merged chunks and the
    16 kb linear index, as the SAM specification (section 5.2) lays them out."""
    rec_ref = np.asarray(rec_ref, dtype=np.int64)
    rec_beg = np.asarray(rec_beg, dtype=np.int64)
    rec_end = np.maximum(np.asarray(rec_end, dtype=np.int64), rec_beg + 1)
    u0 = np.asarray(rec_uoff, dtype=np.int64)
    u1 = u0 + np.asarray(rec_ulen, dtype=np.int64)
    offs = np.asarray(block_offs, dtype=np.int64)

    def voff(u):
        b = u // BGZF_PAYLOAD
        return (offs[b] << 16) | (u - b * BGZF_PAYLOAD)
    v0, v1 = voff(u0), voff(u1)
    e = rec_end - 1
    bins = np.zeros(len(rec_beg), dtype=np.int64)
    done = np.zeros(len(rec_beg), dtype=bool)
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        hit = ~done & ((rec_beg >> shift) == (e >> shift))
        bins[hit] = base + (rec_beg[hit] >> shift)
        done |= hit
    out = [b"BAI\x01", struct.pack("<i", n_ref)]
    for r in range(n_ref):
        sel = np.flatnonzero(rec_ref == r)
        chunks = {}
        for q in sel:                        # records are in file order: adjacent ones of a bin merge into one chunk
            c = chunks.setdefault(int(bins[q]), [])
            if c and c[-1][1] == int(v0[q]):
                c[-1][1] = int(v1[q])
            else:
                c.append([int(v0[q]), int(v1[q])])
        out.append(struct.pack("<i", len(chunks)))
        for b in sorted(chunks):
            out.append(struct.pack("<Ii", b, len(chunks[b])))
            out += [struct.pack("<QQ", c0, c1) for c0, c1 in chunks[b]]
        n_intv = int((rec_end[sel].max() - 1 >> 14) + 1) if len(sel) else 0
        lin = np.zeros(n_intv, dtype=np.uint64)
        for q in sel:
            w0, w1 = int(rec_beg[q] >> 14), int((rec_end[q] - 1) >> 14)
            for w in range(w0, w1 + 1):
                if lin[w] == 0 or int(v0[q]) < int(lin[w]):
                    lin[w] = int(v0[q])
        out.append(struct.pack("<i", n_intv))
        out.append(lin.astype("<u8").tobytes())
    with open(path, "wb") as fh:
        fh.write(b"".join(out))


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


def write_bam(path, refs, reads, index=True):
    """refs: [(name, length)]; reads: iterable of (qname, flag, ref_index, pos0, mapq, cigar_str, seq_str),
    already coordinate sorted.  With index=True a <path>.bai is written next to it."""
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    out = [b"BAM\x01", struct.pack("<i", len(text)), text.encode(), struct.pack("<i", len(refs))]
    for name, ln in refs:
        out += [struct.pack("<i", len(name) + 1), name.encode() + b"\x00", struct.pack("<i", ln)]
    upos = sum(len(x) for x in out)
    meta = []
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
        meta.append((rid, pos, pos + max(1, reflen), upos, len(body) + 4))
        upos += len(body) + 4
    offs = bgzf_write(path, b"".join(out))
    if index:
        m = np.array(meta, dtype=np.int64).reshape(-1, 5)
        write_bai(path + ".bai", len(refs), m[:, 0], m[:, 1], m[:, 2], m[:, 3], m[:, 4], offs)


def write_vcf_gz(path, contig, positions):
    lines = ["##fileformat=VCFv4.2\n", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"]
    lines += ["%s\t%d\t.\tA\tC,T,G\t0\t.\tINFO\n" % (contig, p) for p in positions]
    bgzf_write(path, "".join(lines).encode())


def synth_to_files(table, bam_path, vcf_path, contig="synth", spacing=10):
    """A support table as real files: SNP s (0-based) sits at 1-based position spacing*(s+1); every read is one
    all-M alignment from its first to its last SNP with 'A' between the SNPs; a .bai goes with it.
    Tables whose reads all cover the same number of SNPs (C2, C3) are assembled with NumPy -- a million records in
    seconds; the others go through write_bam record by record."""
    n = table.n_snps
    length = spacing * n + spacing
    ks = np.diff(table.off)
    if table.n_reads > 1000 and (ks == ks[0]).all() and spacing % 2 == 0 and (np.diff(table.rank) >= 0).all():
        _synth_bam_fixed_k(table, bam_path, contig, length, int(ks[0]), spacing)
    else:
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


def _synth_bam_fixed_k(table, bam_path, contig, length, k, spacing):
    """write_bam for reads of one shape: every record has the same size, so the file body is one uint8 matrix."""
    nr = table.n_reads
    l_seq = (k - 1) * spacing + 1
    l_name = 10                                            # "r%08d" + NUL
    rec = 4 + 32 + l_name + 4 + (l_seq + 1) // 2 + l_seq
    M = np.zeros((nr, rec), dtype=np.uint8)
    pos0 = (spacing * (table.rank.astype(np.int64) + 1) - 1)
    beg, end = pos0, pos0 + l_seq
    bins = np.zeros(nr, dtype=np.int64)
    done = np.zeros(nr, dtype=bool)
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        hit = ~done & ((beg >> shift) == ((end - 1) >> shift))
        bins[hit] = base + (beg[hit] >> shift)
        done |= hit

    def put(col, arr, dt):
        M[:, col:col + np.dtype(dt).itemsize] = np.ascontiguousarray(arr.astype(dt)).view(np.uint8).reshape(nr, -1)
    put(0, np.full(nr, rec - 4), "<i4")
    put(4, np.zeros(nr), "<i4")                            # refID
    put(8, pos0, "<i4")
    M[:, 12] = l_name
    M[:, 13] = 42                                          # MAPQ
    put(14, bins, "<u2")
    put(16, np.full(nr, 1), "<u2")                         # n_cigar_op
    put(18, np.zeros(nr), "<u2")                           # flag
    put(20, np.full(nr, l_seq), "<i4")
    put(24, np.full(nr, -1), "<i4")
    put(28, np.full(nr, -1), "<i4")
    put(32, np.zeros(nr), "<i4")
    ids = np.arange(nr)
    M[:, 36] = ord('r')
    for d in range(8):
        M[:, 37 + d] = ord('0') + (ids // 10 ** (7 - d)) % 10
    c0 = 36 + l_name
    put(c0, np.full(nr, (l_seq << 4) | 0), "<u4")          # <l_seq>M
    s0 = c0 + 4
    M[:, s0:s0 + (l_seq + 1) // 2] = 0x11                  # 'A','A'
    if l_seq & 1:
        M[:, s0 + (l_seq - 1) // 2] = 0x10
    code = np.zeros(256, dtype=np.uint8)
    for ch, v in _SEQ_CODE.items():
        code[ord(ch)] = v
    b = table.bases.reshape(nr, k)
    for q in range(k):
        i = q * spacing                                    # even query offset: the high nibble of byte i/2
        M[:, s0 + i // 2] = (code[b[:, q]] << 4) | (M[:, s0 + i // 2] & 0x0f)
    M[:, s0 + (l_seq + 1) // 2:] = 0x7e                    # qualities
    text = "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % (contig, length)
    head = b"".join([b"BAM\x01", struct.pack("<i", len(text)), text.encode(), struct.pack("<i", 1),
                     struct.pack("<i", len(contig) + 1), contig.encode() + b"\x00", struct.pack("<i", length)])
    offs = bgzf_write(bam_path, head + M.tobytes(), level=1)
    write_bai_sorted(bam_path + ".bai", beg, end, len(head) + np.arange(nr, dtype=np.int64) * rec, rec, offs)


def write_bai_sorted(path, beg, end, uoff, ulen, block_offs):
    """write_bai for one reference and coordinate-sorted records, without a Python loop over the records: one chunk per
    bin (first to last record of the bin -- a superset of the exact chunks, which the format allows) and the exact
    linear index."""
    offs = np.asarray(block_offs, dtype=np.int64)

    def voff(u):
        b = u // BGZF_PAYLOAD
        return (offs[b] << 16) | (u - b * BGZF_PAYLOAD)
    v0, v1 = voff(uoff), voff(uoff + ulen)
    e = end - 1
    bins = np.zeros(len(beg), dtype=np.int64)
    done = np.zeros(len(beg), dtype=bool)
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        hit = ~done & ((beg >> shift) == (e >> shift))
        bins[hit] = base + (beg[hit] >> shift)
        done |= hit
    out = [b"BAI\x01", struct.pack("<i", 1)]
    ub = np.unique(bins)
    out.append(struct.pack("<i", len(ub)))
    for b in ub:
        sel = np.flatnonzero(bins == b)
        out.append(struct.pack("<IiQQ", int(b), 1, int(v0[sel[0]]), int(v1[sel[-1]])))
    n_intv = int((e.max() >> 14) + 1)
    lin = np.zeros(n_intv, dtype=np.uint64)
    w0 = beg >> 14
    w1 = e >> 14
    # sorted by beg: the first record overlapping window w is the first with w1 >= w
    for w in range(n_intv):
        q = np.flatnonzero((w0 <= w) & (w1 >= w))
        if len(q):
            lin[w] = int(v0[q].min())
    out.append(struct.pack("<i", n_intv))
    out.append(lin.astype("<u8").tobytes())
    with open(path, "wb") as fh:
        fh.write(b"".join(out))


# ---------------------------------------------------------------------------------------------
# native decoder binding
# ---------------------------------------------------------------------------------------------
class _gio_table(C.Structure):
    _fields_ = [("rank", C.POINTER(C.c_int32)), ("off", C.POINTER(C.c_int64)), ("bases", C.POINTER(C.c_uint8)),
                ("n_reads", C.c_int64), ("n_bases", C.c_int64)]


class gio_stats(C.Structure):
    _fields_ = [("compressed_bytes", C.c_int64), ("blocks", C.c_int64), ("records", C.c_int64), ("reads_kept", C.c_int64),
                ("used_index", C.c_int32), ("libdeflate", C.c_int32), ("threads", C.c_int32), ("reframed", C.c_int32),
                ("seconds", C.c_double), ("depth_dropped", C.c_int64), ("max_row_len", C.c_int32), ("prefetched", C.c_int32)]


GIO_ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_size_t)          # include/gretel_io.h: gio_alloc_fn


def native_last_stats():
    """dict of what the last native_support_table call did (bytes, blocks, records, index use, seconds)."""
    st = gio_stats()
    io_lib().gio_last_stats(C.byref(st))
    return {k: getattr(st, k) for k, _ in gio_stats._fields_ if k != "reserved_"}


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
        L.gio_support_table_from_bam_depth.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int, C.c_int32,
                                                       C.POINTER(_gio_table)]
        L.gio_support_table_from_bam_alloc.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int, C.c_int32,
                                                       GIO_ALLOC_FN, C.c_void_p, C.POINTER(_gio_table)]
        L.gio_prefetch.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32]
        L.gio_prefetch_cancel.argtypes = []
        L.gio_prefetch_cancel.restype = None
        L.gio_table_free.argtypes = [C.POINTER(_gio_table)]
        L.gio_last_stats.argtypes = [C.POINTER(gio_stats)]
        L.gio_last_stats.restype = None
        L.gio_count_coverage.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]
        L.gio_release_buffers.argtypes = []
        L.gio_release_buffers.restype = None
        _io = L
    return _io


def native_release_buffers():
    """The decoder keeps its large working buffers from call to call (include/gretel_io.h: gio_release_buffers, GIO_KEEP_MB); this
    hands them back to the system."""
    io_lib().gio_release_buffers()


def native_prefetch(bam_path, contig, start_pos, end_pos):
    """Start reading and inflating the window's part of the BAM on a thread of the library (include/gretel_io.h: gio_prefetch) and
    return at once: the native_support_table call for the same window takes it over.  For callers that still have a VCF to parse."""
    L = io_lib()
    if L.gio_prefetch(bam_path.encode(), contig.encode(), int(start_pos), int(end_pos)):
        raise IOError(L.gio_last_error().decode())


def native_prefetch_cancel():
    io_lib().gio_prefetch_cancel()


def native_ref_len(bam_path, contig):
    L = io_lib()
    out = C.c_int64()
    if L.gio_ref_len(bam_path.encode(), contig.encode(), C.byref(out)):
        raise KeyError(L.gio_last_error().decode())
    return out.value


PYSAM_MAX_DEPTH = 8000          # bam.pileup's default, which the reference inherits (gretel/util.py:137 passes no max_depth)


def native_support_table(bam_path, contig, start_pos, end_pos, region, stepper="samtools", max_depth=PYSAM_MAX_DEPTH, arena=None):
    """max_depth: the pileup's read-buffer cap (pysam's default 8000; 0 or None: keep every read).
    arena: where the table's three arrays go instead of the library's own malloc (gio_support_table_from_bam_alloc) -- an object with
    alloc(which, nbytes) -> address (0: none) and view(which, dtype, count) -> numpy array over what alloc(which, ...) returned last;
    gretel_amd.hansel.PinnedTableArena keeps page-locked blocks from call to call.  The arrays returned are then the ARENA's:
    their contents stand until the arena is used for the next table."""
    L = io_lib()
    reg = np.ascontiguousarray(np.asarray(region) != 0, dtype=np.uint8)
    if len(reg) < end_pos + 1:
        reg = np.concatenate([reg, np.zeros(end_pos + 1 - len(reg), dtype=np.uint8)])
    t = _gio_table()
    if arena is not None:
        failed = []

        def _alloc(_ctx, which, nbytes):
            try:
                return int(arena.alloc(int(which), int(nbytes))) or None
            except Exception as exc:                    # (an exception cannot cross the C frames: NULL, and said afterwards)
                failed.append(exc)
                return None

        rc = L.gio_support_table_from_bam_alloc(bam_path.encode(), contig.encode(), int(start_pos), int(end_pos), reg.ctypes.data,
                                                int(stepper == "all"), int(max_depth or 0), GIO_ALLOC_FN(_alloc), None, C.byref(t))
        if failed:
            raise failed[0]
    else:
        rc = L.gio_support_table_from_bam_depth(bam_path.encode(), contig.encode(), int(start_pos), int(end_pos),
                                                reg.ctypes.data, int(stepper == "all"), int(max_depth or 0), C.byref(t))
    if rc:
        msg = L.gio_last_error().decode()
        raise (KeyError if rc == -5 else IOError)(msg)
    n, nb = t.n_reads, t.n_bases
    if arena is not None:
        return arena.view(0, np.int32, n), arena.view(1, np.int64, n + 1), arena.view(2, np.uint8, nb)
    # the three arrays are views of the library's buffers (no copy of 30 MB per million reads); the buffers are released
    # when the last of them is garbage-collected
    owner = _TableOwner(L, t)

    def view(ptr, ctype, count, dtype):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        buf = (ctype * count).from_address(C.addressof(ptr.contents))
        buf._owner = owner
        return np.frombuffer(buf, dtype=dtype)

    return view(t.rank, C.c_int32, n, np.int32), view(t.off, C.c_int64, n + 1, np.int64), view(t.bases, C.c_uint8, nb, np.uint8)


class _TableOwner:
    """Keeps a gio_table alive for the NumPy views made of it; gio_table_free when the last view is gone."""

    def __init__(self, lib, table):
        self._lib, self._table = lib, table

    def __del__(self):
        try:
            self._lib.gio_table_free(C.byref(self._table))
        except Exception:
            pass


class _gio_runs(C.Structure):
    _fields_ = [("ref_start", C.POINTER(C.c_int32)), ("off", C.POINTER(C.c_int64)), ("codes", C.POINTER(C.c_uint8)),
                ("n_runs", C.c_int64), ("n_bases", C.c_int64)]


def native_match_runs(bam_path, contig, start0, stop):
    """(ref_start int32[n], off int64[n+1], codes uint8[total]): the aligned runs of the contig's records clipped to
    [start0, stop), bases as A0 C1 G2 T3 / 4 = other -- the input of the GPU coverage histogram."""
    L = io_lib()
    L.gio_match_runs.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(_gio_runs)]
    L.gio_runs_free.argtypes = [C.POINTER(_gio_runs)]
    r = _gio_runs()
    if L.gio_match_runs(bam_path.encode(), contig.encode(), int(start0), int(stop), C.byref(r)):
        msg = L.gio_last_error().decode()
        raise (KeyError if "not in" in msg else IOError)(msg)
    try:
        n, nb = r.n_runs, r.n_bases
        ref = np.ctypeslib.as_array(r.ref_start, shape=(max(n, 1),))[:n].copy()
        off = np.ctypeslib.as_array(r.off, shape=(n + 1,)).copy()
        codes = np.ctypeslib.as_array(r.codes, shape=(max(nb, 1),))[:nb].copy()
    finally:
        L.gio_runs_free(C.byref(r))
    return ref, off, codes


def native_count_coverage(bam_path, contig, start0, stop):
    """int32[4][stop-start0]: A,C,G,T counts per position, like pysam's count_coverage(..., quality_threshold=0,
    read_callback='nofilter') as gretel/snpper.py:29 calls it."""
    L = io_lib()
    counts = np.zeros((4, max(0, stop - start0)), dtype=np.int32)
    if L.gio_count_coverage(bam_path.encode(), contig.encode(), int(start0), int(stop), counts.ctypes.data):
        msg = L.gio_last_error().decode()
        raise (KeyError if "not in" in msg else IOError)(msg)
    return counts
