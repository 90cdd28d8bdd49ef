#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.json with the PYTHON oracle (oracle/hansel_ref.py +
oracle/gretel_ref.py, reference call structure, libm log10).

These vectors are ORACLE-DEFINED, not hanselx-verified: the reference cannot be imported here
(hanselx / pysam / PyVCF are absent), so apart from the fill (pinned by the
reference's own tests) they freeze the spec of SURVEY.md Appendix A.  They exist so that the
oracle itself cannot drift silently and so that the HIP path is also checked against values that
were computed once, offline.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gretel_amd.synth import make_support_table            # noqa: E402
from oracle import gretel_ref as G                         # noqa: E402
from oracle.hansel_ref import Hansel, HanselSpec, SYMBOLS, UNSYMBOLS   # noqa: E402

CASES = [
    dict(name="fixture_like", n_snps=4, reads=[(0, "AAA"), (0, "CCC"), (0, "TT"), (0, "TT"), (2, "GG")], paths=6),
    dict(name="synth_60_k4_A", n_snps=60, n_reads=1500, k=4, seed=3, paths=8, spec=dict()),
    dict(name="synth_60_k4_B_mt", n_snps=60, n_reads=1500, k=4, seed=3, paths=8, spec=dict(cond_mode="B", marginal_term=True)),
    dict(name="synth_60_k4_C_f64", n_snps=60, n_reads=1500, k=4, seed=3, paths=8, spec=dict(cond_mode="C", storage="f64")),
    dict(name="synth_90_kvar_A", n_snps=90, n_reads=1200, k=None, seed=7, paths=6, spec=dict()),
    # round 3: the switches added for the naive-Bayes readings of the published method and for the candidate dict
    dict(name="synth_60_k4_C_mt", n_snps=60, n_reads=1500, k=4, seed=3, paths=8, spec=dict(cond_mode="C", marginal_term=True)),
    dict(name="synth_60_k4_E_mt", n_snps=60, n_reads=1500, k=4, seed=3, paths=8, spec=dict(cond_mode="E", marginal_term=True)),
    dict(name="synth_60_k4_E_f64", n_snps=60, n_reads=1500, k=4, seed=3, paths=8, spec=dict(cond_mode="E", storage="f64")),
    dict(name="synth_90_kvar_D_order", n_snps=90, n_reads=1200, k=None, seed=7, paths=6, spec=dict(cond_mode="D", cand_order="-TGCA")),
    dict(name="synth_60_k4_zero", n_snps=60, n_reads=1500, k=4, seed=3, paths=8, spec=dict(offer_zero=True)),
]


def run(case):
    spec = HanselSpec(**case.get("spec", {}))
    if "reads" in case:
        reads = case["reads"]
    else:
        t = make_support_table(case["n_snps"], case["n_reads"], k=case["k"], seed=case["seed"], k_max=8)
        reads = list(t.reads())
    n = case["n_snps"]
    h = Hansel.init_matrix(SYMBOLS, UNSYMBOLS, n, spec)
    stats = G.fill_from_support(h, reads, n)
    out = dict(name=case["name"], n_snps=n, spec=case.get("spec", {}), reads=[[r, s] for r, s in reads],
               stats=list(stats), L=h.L,
               counts={str(p): [float(x) for x in h._counts(p)] for p in (0, 1, n // 2, n)},
               edge_weights_at_2={str(s): w for s, w in h.get_edge_weights_at(2, [h.symbols_d['_'], h.symbols_d['A']]).items()})
    recs, _ = G.recover_paths(h, n, case["paths"])
    out["records"] = recs
    out["final_nonzero_cells"] = int((h.dense() != 0).sum())
    out["final_sum"] = float(h.dense().astype("float64").sum())
    return out


if __name__ == "__main__":
    # cases are only ever APPENDED: every case carries its own reads, so a vector made in an earlier round keeps pinning
    # the oracle even when gretel_amd.synth (which only helped to draw them) changes
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.json")
    doc = json.load(open(path)) if os.path.exists(path) else dict(note="oracle-defined, not hanselx-verified; see make_golden.py", cases=[])
    have = {c["name"] for c in doc["cases"]}
    new = [run(c) for c in CASES if c["name"] not in have]
    doc["cases"] += new
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", path, "added", [c["name"] for c in new])
