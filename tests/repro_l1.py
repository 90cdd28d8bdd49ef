"""Test infrastructure (calls the oracle; run from the repo root on the GPU box): the serial walkers at L = 1..3 under a candidate order."""
import os, sys, itertools
sys.path.insert(0, ".")
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle
S = b"ACGTN-_"
for walk, order, L in itertools.product(("spec", "spec1", "src", None), ("ACGT-", "TG-AC"), (1, 2, 3)):
    if walk: os.environ["GH_WALK"] = walk
    else: os.environ.pop("GH_WALK", None)
    nb = 0
    for seed in range(6):
        x = make_support_table(17, 340, k=7, n_haps=8, err=0.05, seed=seed)
        h = Hansel(x.n_snps, band=x.band, cand_order=order); o = COracle(x.n_snps, x.band, cand_order=order)
        assert h.fill_from_support(x.rank, x.off, x.bases) == o.fill(x)
        h.L = L; o.L = L
        res, ref = h.spin(2), o.spin(2)
        if not np.array_equal(res["paths"], ref["paths"]):
            nb += 1
            if nb == 1: print("  ", walk, order, L, "variant", h.walk_clock()[3], [bytes(S[q] for q in p).decode() for p in res["paths"]], [bytes(S[q] for q in p).decode() for p in ref["paths"]])
    print(walk, order, L, "bad", nb)
