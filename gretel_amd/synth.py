"""
Seeded synthetic metahaplome generator.

Emits the *support table* -- the hand-over format between read decoding and the
Hansel fill, i.e. exactly what the reference holds per read at
gretel/util.py:235-238: `rank` (number of SNPs left of the read's first covered
SNP, util.py:198) and `support_seq` (first character of every captured allele,
util.py:238).  Table layout (CSR):

    rank  : int32[n_reads]
    off   : int64[n_reads+1]     bases[off[r]:off[r+1]] is read r's support_seq
    bases : uint8[off[-1]]       ASCII  'A','C','G','T','N','-','_'

Shapes of the BASELINE.json configs:
    C2  n_snps=1_000   n_reads=50_000     k=3 fixed
    C3  n_snps=10_000  n_reads=1_000_000  k=5 fixed
    C5  n_snps=50_000  n_reads=200_000    k ~ clip(Poisson(10), 2, 21)
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

ALPHABET = np.frombuffer(b"ACGT", dtype=np.uint8)

CONFIGS = {
    "C2": dict(n_snps=1_000, n_reads=50_000, k=3),
    "C3": dict(n_snps=10_000, n_reads=1_000_000, k=5),
    "C5": dict(n_snps=50_000, n_reads=200_000, k=None),
}


@dataclass
class SupportTable:
    n_snps: int
    rank: np.ndarray       # int32[n_reads]
    off: np.ndarray        # int64[n_reads+1]
    bases: np.ndarray      # uint8[total]
    haplotypes: np.ndarray  # uint8[K, n_snps] ASCII (ground truth; not used by the hot path)
    abundances: np.ndarray  # float64[K]

    @property
    def n_reads(self):
        return int(self.rank.shape[0])

    @property
    def max_k(self):
        return int(np.diff(self.off).max()) if self.n_reads else 0

    @property
    def band(self):
        """Largest pos_to - pos_from any observation of this table can have."""
        return max(1, self.max_k - 1)

    def reads(self):
        """Iterate (rank, support_seq:str) like gretel/util.py:227-238 (for the Python oracle)."""
        b = self.bases.tobytes()
        for r in range(self.n_reads):
            yield int(self.rank[r]), b[self.off[r]:self.off[r + 1]].decode("ascii")


def make_support_table(n_snps, n_reads, k=None, n_haps=8, err=0.01, seed=0,
                       k_lambda=10.0, k_min=2, k_max=21):
    """k: fixed SNPs per read, or None for clip(Poisson(k_lambda), k_min, k_max)."""
    rng = np.random.default_rng(seed)
    # K haplotypes, iid uniform alleles, forced >= 2 distinct alleles per SNP
    haps = rng.integers(0, 4, size=(n_haps, n_snps), dtype=np.int64)
    same = (haps == haps[0:1]).all(axis=0)
    if n_haps > 1 and same.any():
        haps[1, same] = (haps[0, same] + 1) % 4
    abund = rng.dirichlet(np.ones(n_haps))

    # deterministic tiling pass so that every adjacent SNP pair is bridged (no cmd.py:92 gap)
    kt = k if k is not None else int(round(k_lambda))
    kt = max(2, min(kt, n_snps))
    step = max(1, kt // 2)
    tile_starts = np.arange(0, max(1, n_snps - kt + 1), step, dtype=np.int64)
    if tile_starts[-1] != n_snps - kt:
        tile_starts = np.append(tile_starts, n_snps - kt)
    n_tile = len(tile_starts)
    n_rand = max(0, n_reads - n_tile)

    if k is not None:
        ks = np.full(n_rand, min(k, n_snps), dtype=np.int64)
    else:
        ks = np.clip(rng.poisson(k_lambda, size=n_rand), k_min, min(k_max, n_snps)).astype(np.int64)
    hap_of = rng.choice(n_haps, size=n_rand, p=abund)
    starts = np.floor(rng.random(n_rand) * (n_snps - ks + 1)).astype(np.int64)

    top = int(np.argmax(abund))
    all_k = np.concatenate([np.full(n_tile, kt, dtype=np.int64), ks])
    all_start = np.concatenate([tile_starts, starts])
    all_hap = np.concatenate([np.full(n_tile, top, dtype=np.int64), hap_of])

    # rank order, like the reads of a coordinate-sorted BAM (lets the fill count in LDS)
    order = np.argsort(all_start, kind="stable")
    all_k, all_start, all_hap = all_k[order], all_start[order], all_hap[order]
    is_tile = np.zeros(len(order), dtype=bool)
    is_tile[:n_tile] = True
    is_tile = is_tile[order]

    off = np.zeros(len(all_k) + 1, dtype=np.int64)
    np.cumsum(all_k, out=off[1:])
    total = int(off[-1])
    read_of = np.repeat(np.arange(len(all_k), dtype=np.int64), all_k)
    within = np.arange(total, dtype=np.int64) - off[read_of]
    snp = all_start[read_of] + within
    allele = haps[all_hap[read_of], snp]
    # substitution errors on the random reads only
    is_rand = ~is_tile[read_of]
    flip = (rng.random(total) < err) & is_rand
    shift = rng.integers(1, 4, size=total)
    allele = np.where(flip, (allele + shift) % 4, allele)

    return SupportTable(
        n_snps=n_snps,
        rank=all_start.astype(np.int32),
        off=off,
        bases=ALPHABET[allele].astype(np.uint8),
        haplotypes=ALPHABET[haps].astype(np.uint8),
        abundances=abund,
    )


def make_config(name, seed=0, **over):
    cfg = dict(CONFIGS[name])
    cfg.update(over)
    return make_support_table(seed=seed, **cfg)


def sprinkle_deletions(table, frac_positions, frac_reads=0.3, seed=0):
    """'-' at a fraction of the SNP POSITIONS (on `frac_reads` of the reads that cover them): what a real pileup shows where
    some haplotypes carry a deletion -- a few columns of the window with five candidates (A, C, G, T and '-'), the rest
    with at most four.  (Replacing a fraction of all BASES, as bench.py's wide_window does, leaves hardly a column without.)
    Returns the positions (1-based SNP numbers) that got deletions; `table.bases` is replaced."""
    rng = np.random.default_rng(seed)
    n = table.n_snps
    npos = max(1, int(round(frac_positions * n)))
    pos = np.sort(rng.choice(np.arange(1, n + 1), size=npos, replace=False))
    ks = np.diff(table.off)
    read_of = np.repeat(np.arange(table.n_reads, dtype=np.int64), ks)
    snp = table.rank[read_of].astype(np.int64) + (np.arange(len(table.bases), dtype=np.int64) - table.off[read_of]) + 1
    hit = np.isin(snp, pos) & (rng.random(len(table.bases)) < frac_reads)
    bases = table.bases.copy()
    bases[hit] = ord('-')
    table.bases = bases
    return pos
