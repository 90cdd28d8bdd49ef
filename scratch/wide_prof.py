import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_config
t = make_config("C3", seed=0)
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
bw = t.bases.copy(); bw[np.random.default_rng(12345).random(len(bw)) < frac] = ord('-'); t.bases = bw
h = Hansel(t.n_snps, band=t.band)
h.fill_from_support(t.rank, t.off, t.bases)
print("wide positions", int((h.candidate_masks()[1:] == 0x2F).sum()))
for _ in range(2):
    t0 = time.perf_counter(); r = h.spin(100); dt = time.perf_counter() - t0
    print("%.1f us/path" % (dt / r["n"] * 1e6), h.walk_clock())
