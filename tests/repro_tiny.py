"""Test infrastructure (calls the oracle; run from the repo root on the GPU box).  Tiny windows under every switch: first mismatch against the C oracle, with details (GPU box, repo root)."""
import sys, itertools
sys.path.insert(0, ".")
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle
SYM = b"ACGTN-_"
bad = 0
for n, seed in itertools.product((2, 3, 5), range(40)):
    t = make_support_table(n, 44, k=2, n_haps=2, err=0.01, seed=seed, k_max=min(21, n))
    for mode, mt, zero, storage in itertools.product("ABCDE", (False, True), (False, True), ("f32",)):
        sw = dict(storage=storage, cond_mode=mode, marginal_term=mt, offer_zero=zero)
        h = Hansel(t.n_snps, band=t.band, **sw)
        o = COracle(t.n_snps, t.band, storage, mode, mt, offer_zero=zero)
        assert h.fill_from_support(t.rank, t.off, t.bases) == o.fill(t)
        res, ref = h.spin(3), o.spin(3)
        ok = res["n"] == ref["n"] and np.array_equal(res["paths"], ref["paths"]) and res["hp_current"].tolist() == ref["hp_current"].tolist()
        if not ok:
            bad += 1
            if bad <= 6:
                print("MISMATCH n=%d seed=%d %s" % (n, seed, sw), "variant", h.walk_clock()[3], "L", h.L)
                print("  bases", bytes(t.bases).decode(), "rank", t.rank.tolist())
                print("  gpu   ", res["n"], [bytes(SYM[q] for q in p).decode() for p in res["paths"]], res["hp_current"].tolist())
                print("  oracle", ref["n"], [bytes(SYM[q] for q in p).decode() for p in ref["paths"]], ref["hp_current"].tolist())
                for p in range(0, n + 1):
                    print("   weights at", p + 1 if p < n else p, end=" ")
                print()
print("mismatches:", bad)
