"""
Drop-in for the ingest half of the reference's `gretel/util.py`:

    process_vcf(vcf_path, contig_name, start_pos, end_pos)      gretel/util.py:354-414
    load_from_bam(bam_path, target_contig, start_pos, end_pos, vcf_handler, ...)
                                                                 gretel/util.py:33-335
    get_ref_len_from_bam(bam_path, target_contig)                gretel/util.py:10-31

The reference reads BAM through pysam (htslib pileup) and VCF through PyVCF (tabix);
neither is available here, so the host side decodes BGZF/BAM natively (libgretel_io.so,
gretel_amd/csrc/bam_support.cpp; a pure-Python restatement of the same rules lives below) and
bgzipped VCF with the standard library, and turns every read into one row of the
*support table* `(rank, support_seq)` -- what the reference holds per read at
util.py:235-238 -- by walking the CIGAR the way htslib's pileup resolves it.  The pair
loop itself (util.py:226-286) and the counters / L (util.py:329-333) run on the GPU
(`Hansel.fill_from_support` -> k_fill).

pysam's pileup keeps at most max_depth = 8000 reads in its buffer by default and the reference passes no other value
(util.py:137): both decoders apply that cap by htslib's rule (include/gretel_io.h; `max_depth=0` / `--max-depth 0` keeps
every read).

Known divergences from pysam's pileup, all irrelevant to the synthetic configs:
  * stepper "samtools" is modelled as: drop UNMAP/SECONDARY/QCFAIL/DUP reads and
    paired reads that are not properly paired (ignore_orphans); stepper "all"
    (reference --pepper, gretel/cmd.py:39,78) only drops the four flags;
  * `n_threads` is accepted and ignored: the reference's windows (util.py:294-301) only
    partition the reads by leftmost position, the observations are the same.
"""
from __future__ import annotations

import contextlib
import gzip
import struct
import sys

import heapq

import numpy as np

from .hansel import Hansel, SYMBOLS, UNSYMBOLS

_SEQ_DECODE = "=ACMGRSVTWYHKDBN"
_FLAG_DROP = 0x4 | 0x100 | 0x200 | 0x400


# ---------------------------------------------------------------------------------------------
# VCF
# ---------------------------------------------------------------------------------------------
def process_vcf(vcf_path, contig_name, start_pos, end_pos):
    """gretel/util.py:354-414: SNP positions of `contig_name` inside [start_pos, end_pos]."""
    n_snps = 0
    snp_reverse = {}
    snp_forward = {}
    region = np.zeros(end_pos + 1, dtype=int)                 # util.py:393
    opener = gzip.open if _is_gzip(vcf_path) else open
    i = 0
    with opener(vcf_path, "rb") as fp:
        data = fp.read()                                      # (whole file at once: line iteration over gzip is the slow part)
    key = contig_name.encode() + b"\t"
    klen = len(key)
    positions = _vcf_positions(data, key)
    if positions is not None:
        # (the loop below over 10 000 records is 3 of the 3.7 ms this function took; here: the same records, the same order)
        keep = positions[(positions >= start_pos) & (positions <= end_pos)]
        region[keep] = 1
        lst = keep.tolist()
        return {"N": len(lst), "snp_fwd": dict(zip(lst, range(len(lst)))), "snp_rev": dict(enumerate(lst)), "region": region}
    for line in data.split(b"\n"):
        if not line.startswith(key):                          # header lines, other contigs
            continue
        tab = line.find(b"\t", klen)
        pos = int(line[klen:tab] if tab >= 0 else line[klen:])
        if pos < start_pos or pos > end_pos:                  # util.py:397-400
            continue
        n_snps += 1
        region[pos] = 1
        snp_reverse[i] = pos
        snp_forward[pos] = i
        i += 1
    return {"N": n_snps, "snp_fwd": snp_forward, "snp_rev": snp_reverse, "region": region}


def _vcf_positions(data, key):
    """POS of every record whose line starts with `key` (contig + tab), in file order, as an int64 array -- or None when a line
    does not look as expected (no digits behind the contig, more than 18 of them, the file's last line cut short): the caller's
    line-by-line loop then decides, with int()'s own errors.  A repeated position is kept here as the loop keeps it."""
    if not data:
        return None
    arr = np.frombuffer(data, dtype=np.uint8)
    nl = np.flatnonzero(arr == 10)
    starts = np.concatenate([[0], nl + 1])
    starts = starts[starts + len(key) < len(arr)]                 # (a line too short to hold the key and a digit is no record)
    if len(starts) == 0:
        return np.zeros(0, dtype=np.int64)
    kb = np.frombuffer(key, dtype=np.uint8)
    hit = np.ones(len(starts), dtype=bool)
    for j in range(len(kb)):
        hit &= arr[starts + j] == kb[j]
    starts = starts[hit]
    if len(starts) == 0:
        return np.zeros(0, dtype=np.int64)
    width = 19
    idx = starts[:, None] + len(key) + np.arange(width)[None, :]
    chunk = arr[np.minimum(idx, len(arr) - 1)]
    chunk = np.where(idx < len(arr), chunk, 0)
    isdig = (chunk >= 48) & (chunk <= 57)
    ndig = np.where(isdig.all(axis=1), width, np.argmin(isdig, axis=1))
    if (ndig == 0).any() or (ndig >= width).any():
        return None
    term = chunk[np.arange(len(starts)), ndig]                    # what ends the number: a tab, or the line / the file
    if not np.isin(term, (9, 10, 0)).all():
        return None                                               # (e.g. "12x": int() would raise -- let it)
    val = np.zeros(len(starts), dtype=np.int64)
    for j in range(int(ndig.max())):
        live = j < ndig
        val = np.where(live, val * 10 + (chunk[:, j].astype(np.int64) - 48), val)
    return val


def _is_gzip(path):
    with open(path, "rb") as fh:
        return fh.read(2) == b"\x1f\x8b"


# ---------------------------------------------------------------------------------------------
# BAM
# ---------------------------------------------------------------------------------------------
class BamRecord:
    __slots__ = ("ref_id", "pos", "flag", "name", "cigar", "l_seq", "seq_raw", "qual")

    def base(self, q):
        b = self.seq_raw[q >> 1]
        return _SEQ_DECODE[(b >> 4) if (q & 1) == 0 else (b & 15)]


def read_bam(bam_path):
    """Yields (header_refs, iterator of BamRecord).  header_refs = [(name, length)]."""
    with gzip.open(bam_path, "rb") as fh:
        data = fh.read()
    if data[:4] != b"BAM\x01":
        raise ValueError("%s is not a BAM file" % bam_path)
    (l_text,) = struct.unpack_from("<i", data, 4)
    o = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, o)
        o += 4
        name = data[o:o + l_name - 1].decode()
        o += l_name
        (l_ref,) = struct.unpack_from("<i", data, o)
        o += 4
        refs.append((name, l_ref))

    def records(o=o):
        n = len(data)
        while o + 4 <= n:
            (block_size,) = struct.unpack_from("<i", data, o)
            o += 4
            (ref_id, pos, l_read_name, _mapq, _bin, n_cigar, flag, l_seq, _nref, _npos, _tlen) = \
                struct.unpack_from("<iiBBHHHiiii", data, o)
            p = o + 32
            r = BamRecord()
            r.ref_id, r.pos, r.flag, r.l_seq = ref_id, pos, flag, l_seq
            r.name = data[p:p + l_read_name - 1].decode()
            p += l_read_name
            r.cigar = [(c & 15, c >> 4) for c in struct.unpack_from("<%dI" % n_cigar, data, p)]
            p += 4 * n_cigar
            r.seq_raw = data[p:p + (l_seq + 1) // 2]
            p += (l_seq + 1) // 2
            r.qual = data[p:p + l_seq]
            o += block_size
            yield r

    return refs, records()


def get_ref_len_from_bam(bam_path, target_contig, decoder="native"):
    """gretel/util.py:10-31"""
    if decoder == "native":
        from . import bamio
        return bamio.native_ref_len(bam_path, target_contig)
    refs, _ = read_bam(bam_path)
    for name, ln in refs:
        if name == target_contig:
            return ln
    raise KeyError(target_contig)


def _support_of_read(rec, region, start_pos, end_pos, positions=None):
    """Walk the CIGAR like htslib's pileup and return (leftmost_1pos, aligned_query_len,
    [support chars at the SNP columns the read covers, in reference order]).
    positions (a list, --debugreads / --debugpos): receives (1-based position, the whole captured sequence) per column --
    the base plus what an insertion right behind it carries (gretel/util.py:183-186), '-' in a deletion (util.py:180-182)."""
    ref = rec.pos            # 0-based
    q = 0
    chars = []
    qalen = 0
    hi = min(end_pos, len(region) - 1)
    cig = rec.cigar
    for ci, (op, ln) in enumerate(cig):
        if op in (0, 7, 8):                       # M = X : one pileup column per base
            lo1 = max(ref + 1, 1)
            hi1 = min(ref + ln, hi)
            if hi1 >= lo1:
                for pos1 in np.flatnonzero(region[lo1:hi1 + 1]) + lo1:
                    qi = q + (int(pos1) - 1 - ref)
                    chars.append(rec.base(qi))     # util.py:186-189: b[0] is this base
                    if positions is not None:
                        seq = rec.base(qi)
                        if int(pos1) == ref + ln and ci + 1 < len(cig) and cig[ci + 1][0] == 1:      # p_read.indel > 0
                            seq += "".join(rec.base(qi + 1 + x) for x in range(cig[ci + 1][1]))
                        positions.append((int(pos1), seq))
            ref += ln
            q += ln
            qalen += ln
        elif op in (2, 3):                        # D / N : is_del columns -> "-" (util.py:180-182)
            lo1 = max(ref + 1, 1)
            hi1 = min(ref + ln, hi)
            if hi1 >= lo1:
                chars.extend("-" * int(region[lo1:hi1 + 1].sum()))
                if positions is not None:
                    # util.py:180-182 prints "-" * (abs(p_read.indel) + 1): pysam's indel at a deleted column is the length of the
                    # insertion that follows it (only behind the deletion's last base, when the next operation inserts), else 0
                    nxt_ins = cig[ci + 1][1] if ci + 1 < len(cig) and cig[ci + 1][0] == 1 else 0
                    positions.extend((int(pos1), "-" * ((nxt_ins if int(pos1) == ref + ln else 0) + 1))
                                     for pos1 in np.flatnonzero(region[lo1:hi1 + 1]) + lo1)
            ref += ln
        elif op == 1:                             # I
            q += ln
            qalen += ln
        elif op == 4:                             # S
            q += ln
        # H, P: nothing
    return rec.pos + 1, qalen, chars


PYSAM_MAX_DEPTH = 8000


def support_table_from_bam(bam_path, target_contig, start_pos, end_pos, vcf_handler, stepper="samtools",
                           decoder="native", max_depth=PYSAM_MAX_DEPTH, debug_reads=None, debug_pos=None, debug_out=None, arena=None):
    """The pileup half of load_from_bam (gretel/util.py:137-209) -> support table arrays.
    decoder="native": libgretel_io.so (C++/zlib, include/gretel_io.h); "python": the pure-Python restatement
    below (same rules; kept as the readable specification and as a cross-check in the tests).
    max_depth: the read-buffer cap of the pileup the reference inherits (pysam's default 8000, gretel/util.py:137 passes none;
    0: keep every read), by htslib's rule in both decoders: the first read of a position always enters; a later one is dropped
    when the reads that entered and end behind position - 1, plus the list's sentinel node, number more than max_depth.
    debug_reads / debug_pos: the prints of gretel/util.py:211-224 (python decoder), to debug_out (default stdout).
    arena (native decoder): where the three arrays go (gretel_amd.hansel.PinnedTableArena; see bamio.native_support_table)."""
    if decoder == "native" and not (debug_reads or debug_pos):
        from . import bamio
        return bamio.native_support_table(bam_path, target_contig, start_pos, end_pos, vcf_handler["region"], stepper, max_depth, arena=arena)
    refs, records = read_bam(bam_path)
    names = [n for n, _ in refs]
    if target_contig not in names:
        raise KeyError("contig %r not in %s" % (target_contig, bam_path))
    tid = names.index(target_contig)
    region = np.asarray(vcf_handler["region"])
    csum = np.concatenate([[0], np.cumsum(region)])          # csum[x] = sum(region[0:x])
    reads = {}
    order = []
    dc_pos, dc_ends = None, []                       # max_depth: where the pileup iterator stands, the ends of the reads in its buffer (a heap)
    dbg = {}                                         # key -> (query name, [(pos, sequence)]) for --debugreads / --debugpos
    want_dbg = bool(debug_reads) or bool(debug_pos)
    for rec in records:
        if rec.ref_id != tid or rec.flag & _FLAG_DROP:
            continue
        if stepper == "samtools" and (rec.flag & 0x1) and not (rec.flag & 0x2):
            continue
        if max_depth and max_depth > 0:
            # htslib bam_plp_push (what pysam's bam.pileup(max_depth=...) sets): see include/gretel_io.h
            end = rec.pos + sum(ln for op, ln in rec.cigar if op in (0, 2, 3, 7, 8))
            end = end if end > rec.pos else rec.pos + 1
            if rec.pos < end_pos and end > start_pos - 1:         # the fetch only brings records that overlap the region
                if dc_pos is not None and rec.pos < dc_pos:
                    raise ValueError("records are not in coordinate order (a read at %d behind one at %d): the pileup's depth cap "
                                     "(max_depth = %d, pysam's default) needs a coordinate-sorted BAM; sort it, or pass max_depth 0"
                                     % (rec.pos, dc_pos, max_depth))
                if rec.pos != dc_pos:
                    dc_pos = rec.pos                              # the first read of a position always enters
                else:
                    while dc_ends and dc_ends[0] <= rec.pos - 1:  # a min-heap of the ends: what has expired leaves for good
                        heapq.heappop(dc_ends)
                    if len(dc_ends) + 1 > max_depth:
                        continue
                heapq.heappush(dc_ends, end)
        if rec.l_seq == 0:
            continue
        one_or_two = 0
        if rec.flag & 0x1:
            one_or_two = 1 if rec.flag & 0x40 else (2 if rec.flag & 0x80 else 0)
        key = "%s_%s_%d" % (rec.name, str(rec.flag), one_or_two)          # util.py:160
        plist = [] if want_dbg else None
        leftmost, qalen, chars = _support_of_read(rec, region, start_pos, end_pos, plist)
        if leftmost < start_pos:                                          # util.py:165-171
            if leftmost + qalen < start_pos:
                continue
            leftmost = start_pos
        if not chars:
            continue
        if key not in reads:
            lm = min(leftmost, len(region))
            rank = int(csum[lm] - csum[1]) if lm >= 1 else 0              # util.py:198 sum(region[1:LEFTMOST])
            reads[key] = [rank, []]
            order.append(key)
        reads[key][1].extend(c[0] for c in chars)
        if want_dbg:
            dbg.setdefault(key, (rec.name, []))[1].extend(plist)
    if want_dbg:
        out = debug_out if debug_out is not None else sys.stdout
        for key in sorted(k for k, (name, _) in dbg.items() if debug_reads and name in debug_reads):      # util.py:211-216
            for pos1, seq in dbg[key][1]:
                print(key, pos1, seq, file=out)
            print("RANK", key, reads[key][0], file=out)
        if debug_pos:                                                                                 # util.py:218-222
            for key in order:
                seen = [p1 for p1, _ in dbg[key][1]]
                for d_pos in set(seen) & set(debug_pos):
                    print(key, d_pos, dbg[key][1][seen.index(d_pos)][1], file=out)
    rank = np.array([reads[k][0] for k in order], dtype=np.int32)
    seqs = ["".join(reads[k][1]) for k in order]
    off = np.zeros(len(order) + 1, dtype=np.int64)
    if order:
        np.cumsum([len(s) for s in seqs], out=off[1:])
    bases = np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8).copy()
    return rank, off, bases


def prefetch_bam(bam_path, target_contig, start_pos, end_pos):
    """Tell the native decoder which window load_from_bam is going to ask for: it reads and inflates the window's part of the BAM
    on its own threads from now on (everything that does not need the SNP positions) while the caller parses the VCF -- the
    reference's order is process_vcf, then load_from_bam (gretel/cmd.py:69-78).  Never an error: a BAM that cannot be read is
    reported by load_from_bam."""
    try:
        from . import bamio
        bamio.native_prefetch(bam_path, target_contig, start_pos, end_pos)
    except Exception:
        pass


def load_from_bam(bam_path, target_contig, start_pos, end_pos, vcf_handler, use_end_sentinels=False,
                  n_threads=1, debug_reads=False, debug_pos=False, stepper="samtools", decoder="native", max_depth=PYSAM_MAX_DEPTH,
                  **hansel_kw):
    """gretel/util.py:33-335.  Returns a device-backed Hansel with n_slices, n_crumbs and L set.
    n_threads is the reference's number of BAM iterator processes (util.py:288-326): the native decoder runs its own
    threads and the fill is one kernel, so the value changes nothing here -- said once on stderr when it is not 1."""
    if n_threads not in (None, 1):
        sys.stderr.write("[NOTE] -@/--threads %s ignored: the BAM is decoded by libgretel_io.so's own threads and the matrix is filled on the GPU\n" % n_threads)
    native = decoder == "native" and not (debug_reads or debug_pos)
    arena, guard = None, contextlib.nullcontext()
    if native:
        # (the table goes straight into page-locked memory kept from window to window, which the upload reads by DMA; the arena is
        # the process's: held from the decode until the fill has taken the table to the device)
        from . import hansel as _hansel_mod
        arena = _hansel_mod.table_arena()
        if arena is not None:
            guard = _hansel_mod.table_arena_lock
    with guard:
        rank, off, bases = support_table_from_bam(bam_path, target_contig, start_pos, end_pos, vcf_handler, stepper, decoder,
                                                  max_depth=max_depth, debug_reads=debug_reads or None, debug_pos=debug_pos or None, arena=arena)
        if native:
            from . import bamio
            max_k = int(bamio.native_last_stats()["max_row_len"])           # (the decoder knows its longest row: no pass over off[])
        else:
            max_k = int(np.diff(off).max()) if len(rank) else 0
        hansel = Hansel.init_matrix(SYMBOLS, UNSYMBOLS, vcf_handler["N"], band=max(1, max_k - 1), **hansel_kw)
        n_slices, n_crumbs, covered = hansel.fill_from_support(rank, off, bases, use_end_sentinels, max_k=max_k)
    sys.stderr.write("[NOTE] Loaded %d breadcrumbs from %d bread slices.\n" % (n_crumbs, n_slices))   # util.py:331
    if n_slices == 0:
        raise ZeroDivisionError("no read carries more than one SNP (gretel/util.py:333 divides by n_reads)")
    sys.stderr.write("[NOTE] Setting Gretel.L to %d\n" % hansel.L)                                    # util.py:334
    return hansel
