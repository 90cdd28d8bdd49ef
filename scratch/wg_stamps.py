"""Per-workgroup phase times of k_rwseg's LAST launch (diagnostic build: -DRWS_STAMPS_ALL -o scratch/lib_stamps_all.so; run with
GH_LIB=scratch/lib_stamps_all.so): which workgroup the launch waits for and in which phase.  argv: fraction of positions with deletions"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_config, sprinkle_deletions
frac = float(sys.argv[1])
t = make_config("C3", seed=0)
if frac > 0:
    sprinkle_deletions(t, frac, seed=4321)
h = Hansel(t.n_snps, band=t.band)
h.fill_from_support(t.rank, t.off, t.bases)
h.spin(12)
S = 250
out = np.zeros((S, 16))
from gretel_amd._lib import check
check(h._lib.gh_debug_segment_stamps(h._h, out.ctypes.data, S))
names = ["loads", "min", "rw+marg", "rows", "drain", "(entry)", "staging", "Next", "walk", "maps"]
d = np.diff(out[:, :11], axis=1)
tot = out[:, 10] - out[:, 0]
print("clock", h.walk_clock())
print("phase            median      max   (cycles; 100 MHz ticks x ? -- s_memtime)")
for i, n in enumerate(names):
    print("%-12s %10.0f %8.0f  at wg %d" % (n, np.median(d[:, i]), d[:, i].max(), int(d[:, i].argmax())))
print("total        %10.0f %8.0f  at wg %d;  start spread %.0f, end spread %.0f" % (np.median(tot), tot.max(), int(tot.argmax()),
      out[:, 0].max() - out[:, 0].min(), out[:, 10].max() - out[:, 10].min()))
worst = np.argsort(-tot)[:5]
for w in worst:
    print("wg %3d total %6.0f :" % (w, tot[w]), " ".join("%s %.0f" % (n, d[w, i]) for i, n in enumerate(names)))
print("kernel span (first start -> last end): %.0f" % (out[:, 10].max() - out[:, 0].min()))
