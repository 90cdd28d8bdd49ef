import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
from oracle.c_oracle import COracle
n, reads, k = [int(x) for x in sys.argv[1:4]]
t = make_support_table(n, reads, k=k, seed=1)
h = Hansel(t.n_snps, band=t.band)
h.fill_from_support(t.rank, t.off, t.bases)
o = COracle(t.n_snps, t.band); o.fill(t)
p = h.generate_path(); q = o.generate_path()
print("L", h.L, "equal", np.array_equal(p[0], q[0]), p[1:], q[1], flush=True)
print(h.walk_clock())
