"""scratch: the sparse-deletion window under the mixed radix, timed; GH_MIXED=0 for the five-symbol radix on the same window."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, copy
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_config, sprinkle_deletions
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
t = make_config("C3")
if frac > 0:
    sprinkle_deletions(t, frac, seed=4321)
h = Hansel(t.n_snps, band=t.band)
h.fill_from_support(t.rank, t.off, t.bases)
h.spin(10)
for rep in range(3):
    h.clear(); h.fill_from_support(t.rank, t.off, t.bases); h.sync()
    t0 = time.perf_counter(); r = h.spin(100); dt = time.perf_counter() - t0
    print("frac %.3f: %d paths %.1f us/path  %.0f hap/s  clock %s" % (frac, r["n"], dt / r["n"] * 1e6, r["n"] / dt, h.walk_clock()))
