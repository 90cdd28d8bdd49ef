import sys, faulthandler
faulthandler.dump_traceback_later(8, exit=True)
sys.path.insert(0, '/root/repo')
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
t = make_support_table(40, 900, k=3, seed=1)
h = Hansel(t.n_snps, band=t.band)
print("fill", h.fill_from_support(t.rank, t.off, t.bases), flush=True)
print("counts", h.counts_array(3)[:8], flush=True)
print("gen", flush=True)
p = h.generate_path()
print("done", p[1:], flush=True)
