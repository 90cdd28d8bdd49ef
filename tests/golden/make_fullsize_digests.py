#!/usr/bin/env python3
"""Full-depth digests of BASELINE.json's configs, written by the C ORACLE on the build container's libm
(test infrastructure: oracle/c/gretel_oracle.c restates gretel/gretel.py:79-98,143-189, gretel/util.py:226-286 and
gretel/cmd.py:148-179; its Hansel lookups are SURVEY Appendix A's reconstruction -- "parity unpinned", see the oracle's
header).  No GPU time is needed to make them and the oracle is not needed on the GPU box to compare against them:
tests/test_gpu_digests.py takes every case below through the HIP path (C ABI) and compares

    per path   sha256 of the N+1 path bytes (first 16 hex digits), hp_current / hp_original / ratio as float.hex()
               (bit for bit), magnitude as float.hex() (compared to 1e-10 relative: the kernels add the removed mass
               in a fixed tree, the oracle one cell after the other)
    per case   the fill counters, L, and the sha256 of the reweighted tensor (band layout, binary64) behind the last path

Cases:
    C5 x 1000 paths (default spec) -- "deep reweight" at its full depth
    C5 x 100 under conditional C, and under E + marginal term + f64
    C3 seeds 0..7 x 100 (C4's eight windows)
    C3 seed 0 x 100 under the 20 specs of bench.py's spec_matrix ({A..E} x marginal term x {f32, f64})
    the wide_window_sparse window of bench.py (deletions at 1 % of the positions) x 100
Round 6 (VERDICT r5 "offline-closable gaps": no digest covered deletions or L >= 6 outside C5):
    the lag-sweep window (10k SNPs, reads of ~10 SNPs, band 25) with deletions at 1 % of the positions at L = 3, 4, 6, 7, 8, 11,
      16, 22 x 100 (five-symbol enumeration, candidate pools over the symbols A C G T -: k_cwalk<L, 5>, k_cwalkg<5>), at L = 6, 8
      also under E + marginal term
    the same window with 5 % of the BASES read as deletions (nearly every position offers five candidates) at L = 4, 6, 8, 12 x 50
    C3 with 5 % of the bases read as deletions (bench.py's wide_window: 5^5 states per target) x 100, also under E + marginal term
    the lag-sweep window without deletions at L = 33, 40, 48, 56, 64 x 50 (states as bytes: k_cwalk2, k_cwalkg when first written)

Run from the repo root:   python tests/golden/make_fullsize_digests.py [--jobs 6] [--only NAME_SUBSTRING]
Writes tests/golden/fullsize_digests.json (one entry per case, keyed by name).  Inputs come from the seeded generator
gretel_amd/synth.py, which the GPU test calls with the same arguments.
"""
from __future__ import annotations

import argparse
import copy
import hashlib
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

OUT = os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")


def spec_name(kw):
    return "-".join("%s=%s" % (k, v) for k, v in sorted(kw.items())) or "default"


def cases():
    """(name, dict(config=, seed=, paths=, spec=, table=)) -- `table` names the input transform (None | 'sparse_deletions')."""
    out = []
    out.append(("C5/seed0/default/1000", dict(config="C5", seed=0, paths=1000, spec={}, table=None)))
    for kw in (dict(cond_mode="C"), dict(cond_mode="E", marginal_term=True, storage="f64")):
        out.append(("C5/seed0/%s/100" % spec_name(kw), dict(config="C5", seed=0, paths=100, spec=kw, table=None)))
    for seed in range(8):
        out.append(("C3/seed%d/default/100" % seed, dict(config="C3", seed=seed, paths=100, spec={}, table=None)))
    for st, cm, mt in itertools.product(("f32", "f64"), "ABCDE", (False, True)):
        kw = dict(storage=st, cond_mode=cm, marginal_term=mt)
        if st == "f32" and cm == "A" and not mt:
            continue                    # the default spec: C3/seed0/default/100 above
        out.append(("C3/seed0/%s/100" % spec_name(kw), dict(config="C3", seed=0, paths=100, spec=kw, table=None)))
    out.append(("C3/seed0/sparse_deletions/default/100",
                dict(config="C3", seed=0, paths=100, spec={}, table="sparse_deletions")))
    # round 6: deletions beyond L = 5, dense deletions, byte states.  `L` overrides what the fill gives (util.py:333), as
    # scratch/l_sweep.py does: the same window at every lag count
    pub = dict(cond_mode="E", marginal_term=True)       # the published method's form (README.md:79-94)
    for L in (3, 4, 6, 7, 8, 11, 16, 22, 30, 38):       # (30, 38: round 6, k_cwalk2<LC, 5>)
        out.append(("sweep/seed5/sparse_deletions/L=%d/100" % L, dict(config="sweep", seed=5, paths=100, spec={}, table="sparse_deletions", L=L)))
    for L in (6, 8):
        out.append(("sweep/seed5/sparse_deletions/L=%d/%s/100" % (L, spec_name(pub)),
                    dict(config="sweep", seed=5, paths=100, spec=pub, table="sparse_deletions", L=L)))
    for L in (4, 6, 8, 12):
        out.append(("sweep/seed5/dense_deletions/L=%d/50" % L, dict(config="sweep", seed=5, paths=50, spec={}, table="dense_deletions", L=L)))
    out.append(("C3/seed0/dense_deletions/default/100", dict(config="C3", seed=0, paths=100, spec={}, table="dense_deletions")))
    out.append(("C3/seed0/dense_deletions/%s/100" % spec_name(pub), dict(config="C3", seed=0, paths=100, spec=pub, table="dense_deletions")))
    for L in (33, 40, 48, 56, 64):                       # (56, 64: round 6, k_cwalk2's four-chunk blocks)
        out.append(("sweep/seed5/none/L=%d/50" % L, dict(config="sweep", seed=5, paths=50, spec={}, table=None, L=L)))
    return out


def make_table(case):
    """The input of a case; tests/test_gpu_digests.py calls this very function."""
    from gretel_amd.synth import make_config, make_support_table, sprinkle_deletions
    if case["config"] == "sweep":                       # scratch/l_sweep.py's window: long-read-style reads, band 25
        t = make_support_table(10000, 150000, k=None, seed=case["seed"], n_haps=8, err=0.0, k_max=26)
    else:
        t = make_config(case["config"], seed=case["seed"])
    if case.get("table") == "sparse_deletions":         # bench.py's wide_window_sparse window
        t = copy.copy(t)
        t.bases = t.bases.copy()
        sprinkle_deletions(t, 0.01, seed=4321)
    elif case.get("table") == "dense_deletions":        # bench.py's wide_window: 5 % of the BASES read as '-'
        t = copy.copy(t)
        bw = t.bases.copy()
        bw[np.random.default_rng(12345).random(len(bw)) < 0.05] = ord('-')
        t.bases = bw
    elif case.get("table"):
        raise ValueError(case["table"])
    return t


def digest_result(res, band):
    """What is compared: shared by the generator (oracle result) and the GPU test (HIP result)."""
    n = int(res["n"])
    return dict(
        n=n, hole_at=int(res["hole_at"]),
        path_sha=[hashlib.sha256(np.ascontiguousarray(res["paths"][q], dtype=np.uint8).tobytes()).hexdigest()[:16] for q in range(n)],
        hp_current=[float(x).hex() for x in res["hp_current"][:n]],
        hp_original=[float(x).hex() for x in res["hp_original"][:n]],
        ratio=[float(x).hex() for x in res["ratio"][:n]],
        magnitude=[float(x).hex() for x in res["magnitude"][:n]],
        band_sha=hashlib.sha256(np.ascontiguousarray(band, dtype=np.float64).tobytes()).hexdigest(),
    )


def run_case(item):
    name, case = item
    from oracle.c_oracle import COracle
    t0 = time.time()
    t = make_table(case)
    o = COracle(t.n_snps, t.band, **case["spec"])
    stats = o.fill(t)
    if case.get("L"):
        o.L = case["L"]
    res = o.spin(case["paths"])
    d = digest_result(res, o.export_band())
    d.update(case=case, fill_stats=[int(x) for x in stats], L=int(o.L), n_snps=int(t.n_snps), band=int(t.band),
             oracle_seconds=round(time.time() - t0, 1))
    return name, d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=max(1, min(6, (os.cpu_count() or 2) - 1)))
    ap.add_argument("--only", default=None, help="only the cases whose name contains this")
    args = ap.parse_args()
    todo = [c for c in cases() if args.only is None or args.only in c[0]]
    out = {}
    if os.path.exists(OUT) and args.only is not None:
        with open(OUT) as f:
            out = json.load(f)["cases"]
    import multiprocessing as mp
    # longest first (C5 x 1000 is five minutes of one core)
    todo.sort(key=lambda c: -(c[1]["paths"] * (25 if c[1]["config"] == "C5" else 1) * max(1, c[1].get("L", 5) // 5)))
    with mp.get_context("spawn").Pool(args.jobs) as pool:
        for name, d in pool.imap_unordered(run_case, todo):
            out[name] = d
            print("%-60s n=%d hole_at=%d  %.0f s" % (name, d["n"], d["hole_at"], d["oracle_seconds"]), flush=True)
    import platform
    meta = dict(written_by="tests/golden/make_fullsize_digests.py", oracle="oracle/c/gretel_oracle.c on libm (use_libm=1)",
                libc=" ".join(platform.libc_ver()), machine=platform.machine(),
                parity="oracle-defined (Hansel lookups: SURVEY Appendix A, parity unpinned), not hanselx-verified")
    with open(OUT, "w") as f:
        json.dump(dict(meta=meta, cases={k: out[k] for k in sorted(out)}), f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote %s (%d cases, %.1f KB)" % (OUT, len(out), os.path.getsize(OUT) / 1024))


if __name__ == "__main__":
    main()
