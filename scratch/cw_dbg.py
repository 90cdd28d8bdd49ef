"""Debug driver for the candidate-pool walk (L = 6..16): C5-like windows against the C oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table, make_config
from oracle.c_oracle import COracle

def one(t, paths, L=None, check=True):
    h = Hansel(t.n_snps, band=t.band)
    h.fill_from_support(t.rank, t.off, t.bases)
    if L: h.L = L
    t0 = time.time(); res = h.spin(paths); dt = time.time() - t0
    wc = h.walk_clock()
    print("N", t.n_snps, "L", h.L, "paths", res["n"], "hole", res["hole_at"], "%.1f ms  %.1f us/path" % (dt * 1e3, dt / max(1, res["n"]) * 1e6),
          "variant", wc[3], "requeues", wc[0], "serial", wc[1], "rounds", wc[2], flush=True)
    if check:
        o = COracle(t.n_snps, t.band); o.fill(t)
        if L: o.L = L
        ref = o.spin(paths)
        ok = res["n"] == ref["n"] and np.array_equal(res["paths"], ref["paths"])
        print("   vs oracle: n", res["n"], ref["n"], "paths", ok, "hp", res["hp_current"].tolist() == ref["hp_current"].tolist(),
              res["hp_original"].tolist() == ref["hp_original"].tolist(), "ratio", res["ratio"].tolist() == ref["ratio"].tolist(),
              "band", np.array_equal(h.export_band(), o.export_band()), flush=True)
        if not ok:
            bad = [i for i in range(min(res["n"], ref["n"])) if not np.array_equal(res["paths"][i], ref["paths"][i])]
            print("   first bad path", bad[:5])

one(make_support_table(600, 9000, k=8, seed=1), 12, L=7)
one(make_support_table(3000, 40000, k=None, seed=2), 20)
one(make_support_table(9000, 60000, k=None, seed=3), 30)
if len(sys.argv) > 1:
    t = make_config("C5", seed=0)
    one(t, 60)
    one(t, 300, check=False)
