"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under gretel_amd/ may import this.

CPU restatement of the Gretel hot path that is fully specified inside
/root/reference (control flow pinned by the reference source; the Hansel
arithmetic it calls is `oracle/hansel_ref.py`, "parity unpinned" -- see there):

  fill_from_support          gretel/util.py:226-286, 329-333  (per-read pair loop, L)
  generate_path              gretel/gretel.py:102-189
  reweight_hansel_from_path  gretel/gretel.py:79-98
  gap_check                  gretel/cmd.py:85-118
  recover_paths              gretel/cmd.py:148-179  (spin loop, 1% clamp, dedupe)

The "support table" is the canonical hand-over format between BAM decoding
(host) and the matrix fill: per read `(rank, support_seq)` exactly as
`gretel/util.py:235-238` builds them (rank = number of SNPs left of the read's
first covered SNP; support_seq = first character of every captured allele).
"""
from __future__ import annotations

import math
from math import ceil

from .hansel_ref import _log10

MIN_REMOVE = 0.01   # gretel/cmd.py:157


def fill_from_support(hansel, reads, n_snps, use_end_sentinels=False):
    """reads: iterable of (rank:int, support_seq:str).  gretel/util.py:226-286.
    Returns (slices, crumbs, covered_snps) and sets n_slices/n_crumbs/L like
    gretel/util.py:329-333."""
    slices = crumbs = covered = 0
    for rank, support_seq in reads:
        support_len = len(support_seq)
        if not support_len > 1:                       # util.py:230
            continue
        slices += 1                                   # util.py:233
        covered += len(support_seq.replace("N", "").replace("_", ""))   # util.py:239
        for i in range(0, support_len):               # util.py:242
            snp_a = support_seq[i]
            for j in range(i + 1, support_len):       # util.py:254
                snp_b = support_seq[j]
                if snp_a in ('_', 'N'):               # util.py:258
                    continue
                if i == 0 and j == 1 and rank == 0:   # util.py:262
                    hansel.add_observation('_', snp_a, 0, 1)
                    hansel.add_observation(snp_a, snp_b, 1, 2)
                    crumbs += 1
                elif (j + rank + 1) == n_snps and abs(i - j) == 1:   # util.py:271
                    hansel.add_observation(snp_a, snp_b, n_snps - 1, n_snps)
                    hansel.add_observation(snp_b, '_', n_snps, n_snps + 1)
                    crumbs += 1
                else:                                 # util.py:279
                    hansel.add_observation(snp_a, snp_b, i + rank + 1, j + rank + 1)
                    crumbs += 1
                    if use_end_sentinels:             # util.py:283
                        if j == (support_len - 1) and abs(i - j) == 1:
                            hansel.add_observation(snp_b, '_', j + rank + 1, j + rank + 2)
    hansel.n_slices = slices                          # util.py:329
    hansel.n_crumbs = crumbs                          # util.py:330
    if slices > 0:
        hansel.L = int(ceil(float(covered) / slices))  # util.py:333
    return slices, crumbs, covered


def gap_check(hansel, n_snps):
    """gretel/cmd.py:85-118: first i in [0, N] whose total is 0, else -1."""
    for i in range(0, n_snps + 1):
        if hansel.get_counts_at(i).get("total", 0) == 0:
            return i
    return -1


def generate_path(n_snps, hansel, original_hansel, debug_hpos=None):
    """gretel/gretel.py:102-189"""
    running_prob = 0.0
    running_prob_uw = 0.0
    current_path = [hansel.symbols_d['_']]            # gretel.py:138
    marginals = []
    for snp in range(1, n_snps + 1):                  # gretel.py:143
        curr_branches = hansel.get_edge_weights_at(snp, current_path)   # gretel.py:155
        next_v = 0.0
        next_m = None
        for symbol in curr_branches:                  # gretel.py:166-174
            if str(symbol) == "total":
                continue
            if next_m is None:
                next_v = curr_branches[symbol]
                next_m = symbol
            elif curr_branches[symbol] > next_v:
                next_v = curr_branches[symbol]
                next_m = symbol
        if next_m is None:                            # gretel.py:176-180
            return None, None, None
        selected_edge_weight = hansel.get_marginal_of_at(next_m, snp)   # gretel.py:182
        marginals.append(selected_edge_weight)
        running_prob += _log10(selected_edge_weight)                    # gretel.py:185
        running_prob_uw += _log10(original_hansel.get_marginal_of_at(next_m, snp))  # gretel.py:186
        current_path.append(next_m)
    return current_path, {"hp_original": running_prob_uw, "hp_current": running_prob}, min(marginals)


def reweight_hansel_from_path(hansel, path, ratio):
    """gretel/gretel.py:79-98 (the live variant; note the pair enumeration quirks)."""
    size = 0
    for i in range(0, len(path)):
        for j in range(0, i + 1 + 1):
            if i >= len(path) - 1:
                size += hansel.reweight_observation(path[i], path[j], i, i + 1, ratio)
                break
            else:
                if j < i:
                    t_i, t_j = j, i
                else:
                    t_i, t_j = i, j
                size += hansel.reweight_observation(path[t_i], path[t_j], t_i, t_j, ratio)
    return size


def reweight_call_sequence(n_snps):
    """The (pos_from, pos_to) sequence of gretel.py:79-96 for a path of N SNPs
    (pure index arithmetic; used to pin multiplicities)."""
    seq = []
    ln = n_snps + 1
    for i in range(0, ln):
        for j in range(0, i + 2):
            if i >= ln - 1:
                seq.append((i, i + 1))
                break
            seq.append((j, i) if j < i else (i, j))
    return seq


def recover_paths(hansel, n_snps, max_paths=100, original_hansel=None):
    """gretel/cmd.py:148-179.  Returns (records, PATHS) where records is the list
    of per-spin dicts in recovery order and PATHS the dedupe table of cmd.py."""
    if original_hansel is None:
        original_hansel = hansel.copy()               # cmd.py:79
    PATHS = {}
    records = []
    for i in range(0, max_paths):
        init_path, init_prob, init_min = generate_path(n_snps, hansel, original_hansel)
        if init_path is None:                         # cmd.py:153
            break
        if init_min < MIN_REMOVE:                     # cmd.py:158-160
            init_min = MIN_REMOVE
        rw_magnitude = reweight_hansel_from_path(hansel, init_path, init_min)
        current_path_str = "".join([str(x) for x in init_path])
        records.append({
            "path": current_path_str,
            "hp_current": init_prob["hp_current"],
            "hp_original": init_prob["hp_original"],
            "ratio": init_min,
            "magnitude": rw_magnitude,
        })
        if current_path_str not in PATHS:
            PATHS[current_path_str] = {
                "hp_current": [], "hp_original": [], "i": [], "i_0": i, "n": 0,
                "magnitude": 0, "hansel_path": init_path,
            }
        P = PATHS[current_path_str]
        P["n"] += 1
        P["i"].append(i)
        P["magnitude"] += rw_magnitude
        P["hp_current"].append(init_prob["hp_current"])
        P["hp_original"].append(init_prob["hp_original"])
    return records, PATHS
