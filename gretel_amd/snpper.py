"""
`gretel-snpper`: aggressively call every heterogeneous site of a contig as a SNP and print a placeholder VCF
(reference gretel/snpper.py:12-56), on top of the native BAM decoder instead of pysam.

A site is reported when more than one of A,C,G,T is seen on more than `--depth` reads (snpper.py:38-40);
records are `contig  pos  .  A  C,T,G  0  .  INFO` (snpper.py:41-50).
"""
from __future__ import annotations

import argparse
import sys

from . import bamio


def coverage_on_gpu(bam, contig, start0, end, depth=0, device=-1, want_counts=False, window=1 << 20):
    """(site mask uint8[end-start0], counts int32[4][end-start0] or None) from the GPU histogram (gh_coverage_sites)."""
    import ctypes as C

    import numpy as np

    from . import _lib
    n = max(0, end - start0)
    site = np.zeros(n, dtype=np.uint8)
    counts = np.zeros((4, n), dtype=np.int32) if want_counts else None
    # one window of the contig at a time: the decoder hands over one byte per aligned base, and a deep BAM over a whole
    # contig would not fit the host (the reference streams into a 4 x length array, snpper.py:29).  The runs are clipped to
    # the window and a position's counts only need the reads that cover it, so the windows are independent.  Each window is
    # one call of the decoder: with a .bai next to the BAM it starts at the window and stops behind it; without one it reads
    # from the start of the file and stops behind the window if the header declares coordinate order (SO:coordinate) -- an
    # unindexed, unsorted BAM is read in full once per 1 Mbp window (the reference makes one pass: index or sort such input).
    for w0 in range(start0, end, window):
        w1 = min(end, w0 + window)
        ref, off, codes = bamio.native_match_runs(bam, contig, w0, w1)
        wsite = np.zeros(w1 - w0, dtype=np.uint8)
        wcounts = np.zeros((4, w1 - w0), dtype=np.int32) if want_counts else None
        _lib.check(_lib.load().gh_coverage_sites(int(device), ref.ctypes.data, off.ctypes.data, codes.ctypes.data, len(ref),
                                                int(w0), int(w1 - w0), int(depth),
                                                wcounts.ctypes.data if want_counts else None, wsite.ctypes.data))
        site[w0 - start0:w1 - start0] = wsite
        if want_counts:
            counts[:, w0 - start0:w1 - start0] = wcounts
    return site, counts


def call_sites(bam, contig, start1=1, end=None, depth=0, host=False, device=-1):
    """1-based positions of the called sites.  Default: the coverage histogram and the site rule run on the GPU
    (k_cov / k_sites); host=True counts with the native decoder on the CPU instead (an explicit choice, `--host` on
    the command line -- never a silent fallback)."""
    if not end:
        end = bamio.native_ref_len(bam, contig)                       # snpper.py:23-24
    start0 = start1 - 1                                               # snpper.py:27
    if host:
        counts = bamio.native_count_coverage(bam, contig, start0, end)    # snpper.py:29
        sites = (counts > depth).sum(axis=0) > 1                          # snpper.py:38-39
    else:
        sites, _ = coverage_on_gpu(bam, contig, start0, end, depth, device)
    return [int(i) + 1 + start0 for i in sites.nonzero()[0]]          # snpper.py:39-43


def main(argv=None):
    ap = argparse.ArgumentParser("gretel-snpper", description="Call every heterogeneous site and print a VCF",
                                 epilog="coordinates are 1-based")
    ap.add_argument("--bam", required=True, help="reads aligned to the (pseudo-)reference")
    ap.add_argument("--contig", required=True, help="contig to call on")
    ap.add_argument("-s", type=int, default=1, help="start [1]")
    ap.add_argument("-e", type=int, help="end [length of the contig]")
    ap.add_argument("--depth", type=int, default=0, help="a base counts as a variant if more than this many reads show it [0]")
    ap.add_argument("--host", action="store_true", help="count on the CPU (native decoder) instead of the GPU histogram")
    args = ap.parse_args(argv)
    out = sys.stdout
    out.write("##fileformat=VCFv4.2\n")                               # snpper.py:33-35
    for pos in call_sites(args.bam, args.contig, args.s, args.e, args.depth, host=args.host):
        out.write("\t".join([args.contig, str(pos), ".", "A", "C,T,G", "0", ".", "INFO"]) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
