import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
for (n, reads, k) in [(4, 20, 2), (40, 900, 3), (200, 6000, 4)]:
    t = make_support_table(n, reads, k=k, seed=1)
    h = Hansel(t.n_snps, band=t.band)
    print(n, h.fill_from_support(t.rank, t.off, t.bases), h.L, flush=True)
    print(h.generate_path()[1:], flush=True)
