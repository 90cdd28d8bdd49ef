import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle
def case(n, reads, k, n_haps, err, storage, mode, mt, L, paths, seed=1):
    t = make_support_table(n, reads, k=k, n_haps=n_haps, err=err, seed=seed, k_max=min(21, n))
    h = Hansel(t.n_snps, band=t.band, storage=storage, cond_mode=mode, marginal_term=mt)
    o = COracle(t.n_snps, t.band, storage, mode, mt)
    h.fill_from_support(t.rank, t.off, t.bases); o.fill(t)
    if L: h.L = L; o.L = L
    res, ref = h.spin(paths), o.spin(paths)
    print("spin1", res["n"], ref["n"], res["hole_at"], ref["hole_at"], np.array_equal(res["paths"], ref["paths"]))
    res2, ref2 = h.spin(2), o.spin(2)
    print("spin2", res2["n"], ref2["n"], res2["hole_at"], ref2["hole_at"], np.array_equal(res2["paths"], ref2["paths"]))
    res3, ref3 = h.spin(2), o.spin(2)
    print("spin3", res3["n"], ref3["n"], res3["hole_at"], ref3["hole_at"])
case(1200, 7200, None, 1, 0.0, "f64", "A", True, 4, 8)
case(1200, 7200, None, 1, 0.0, "f32", "A", False, 4, 8)
case(2, 54, 2, 8, 0.05, "f32", "B", False, None, 5)
