"""
Drop-in for the two hot functions of the reference's `gretel/gretel.py`, same
names, arguments, return values and stderr notes; the loops run on the GPU.

    generate_path(n_snps, hansel, original_hansel, debug_hpos=None)   gretel/gretel.py:102-189
    reweight_hansel_from_path(hansel, path, ratio)                    gretel/gretel.py:13-98

`hansel` must be a `gretel_amd.hansel.Hansel` (device tensor).  No CPU fallback.
"""
from __future__ import annotations

import sys

from .hansel import Hansel


def _need_device(h, what):
    if not isinstance(h, Hansel):
        raise TypeError("%s needs a gretel_amd.hansel.Hansel (device-backed), got %r" % (what, type(h)))


def generate_path(n_snps, hansel, original_hansel, debug_hpos=None):
    """Returns (path, {"hp_original", "hp_current"}, min_marginal) or (None, None, None)
    when no branch can be selected (gretel/gretel.py:176-180).  `path` is a list of
    N+1 Hansel symbols starting with the '_' sentinel (gretel/gretel.py:138,189)."""
    _need_device(hansel, "generate_path")
    if n_snps != hansel.n:
        raise ValueError("n_snps=%d but the Hansel was built for %d SNPs" % (n_snps, hansel.n))
    sys.stderr.write("[NOTE] *Establishing next path\n")                 # gretel.py:142
    res = hansel.generate_path(original_hansel)
    if debug_hpos:
        # gretel.py:147-150,162-164 only print the branch weights at the listed SNPs
        last = n_snps if res[0] is not None else res[1]
        walked = res[0] if res[0] is not None else res[2]
        for snp in range(1, last + 1):
            if snp in debug_hpos:
                print(hansel.get_edge_weights_at(snp, walked))
    if res[0] is None:
        snp = res[1]
        sys.stderr.write('''[NOTE] Unable to select next branch from SNP %d to %d
       By design, Gretel will attempt to recover haplotypes until a hole in the graph has been found.
       Recovery will intentionally terminate now.\n''' % (snp - 1, snp))   # gretel.py:177-179
        return None, None, None
    indices, hp_cur, hp_orig, min_marg = res
    return hansel.path_symbols(indices), {"hp_original": hp_orig, "hp_current": hp_cur}, min_marg


def reweight_hansel_from_path(hansel, path, ratio):
    """Returns the sum of removed observations (gretel/gretel.py:98)."""
    _need_device(hansel, "reweight_hansel_from_path")
    size = hansel.reweight_from_path(hansel._path_indices(path), ratio)
    sys.stderr.write("[RWGT] Ratio %.3f, Removed %.1f\n" % (ratio, size))   # gretel.py:97
    return size
