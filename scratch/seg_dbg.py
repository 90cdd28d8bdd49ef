"""Debug driver for the segment-parallel walk: small windows against the C oracle, step by step."""
import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle

def one(n, reads, k, seed, paths, L=None):
    t = make_support_table(n, reads, k=k, seed=seed)
    h = Hansel(t.n_snps, band=t.band)
    h.fill_from_support(t.rank, t.off, t.bases)
    o = COracle(t.n_snps, t.band)
    o.fill(t)
    if L:
        h.L = L; o.L = L
    print("case", n, reads, k, seed, "L", h.L, flush=True)
    h.snapshot_original(); o.snapshot_original()
    pg = h.generate_path(); po = o.generate_path()
    ok = np.array_equal(pg[0], po[0])
    print(" generate_path equal:", ok, "hp", pg[1:] == po[1], pg[1:], po[1], flush=True)
    if not ok:
        bad = np.flatnonzero(pg[0] != po[0])
        print("  first mismatches at", bad[:10], pg[0][bad[:10]], po[0][bad[:10]])
    res = h.spin(paths); ref = o.spin(paths)
    print(" spin n", res["n"], ref["n"], "paths equal", np.array_equal(res["paths"], ref["paths"]),
          "hp", res["hp_current"].tolist() == ref["hp_current"].tolist(), res["hp_original"].tolist() == ref["hp_original"].tolist(),
          "ratio", res["ratio"].tolist() == ref["ratio"].tolist(), "band", np.array_equal(h.export_band(), o.export_band()), flush=True)
    print(" variant", h.walk_clock()[3], flush=True)

one(40, 900, 3, 8, 5)
one(200, 6000, 4, 11, 5)
one(1000, 50000, 3, 0, 20)
one(3000, 90000, 5, 1, 10)
one(300, 9000, 5, 2, 6, L=1)
one(300, 9000, 5, 2, 6, L=2)

def long_spin(seed):
    t = make_support_table(24, 300, k=3, seed=seed, n_haps=2, err=0.0)
    h = Hansel(t.n_snps, band=t.band); h.fill_from_support(t.rank, t.off, t.bases)
    o = COracle(t.n_snps, t.band); o.fill(t)
    res = h.spin(900); ref = o.spin(900)
    print("long spin seed", seed, "n", res["n"], ref["n"], "hole", res["hole_at"], ref["hole_at"], "paths", np.array_equal(res["paths"], ref["paths"]),
          "hp", res["hp_current"].tolist() == ref["hp_current"].tolist(), "band", np.array_equal(h.export_band(), o.export_band()),
          "requeues", h.walk_clock()[0], flush=True)
for sd in range(4):
    long_spin(sd)
