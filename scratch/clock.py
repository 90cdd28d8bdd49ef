import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, time
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
for (n, reads, k) in [(10000, 200000, 5), (10000, 200000, 3), (10000, 200000, 8), (20000, 100000, None)]:
    t = make_support_table(n, reads, k=k, seed=1)
    h = Hansel(t.n_snps, band=t.band)
    h.fill_from_support(t.rank, t.off, t.bases)
    for rep in range(3):
        h.generate_path()
    cyc, ticks, steps, _variant = h.walk_clock()
    print("N=%d L=%d: %.1f cycles/step, %.1f ns/step, clock %.2f GHz" % (n, h.L, cyc/steps, ticks*10.0/steps, cyc/(ticks*10.0)), flush=True)
