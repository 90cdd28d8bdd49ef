import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_config
t = make_config("C5", seed=0)
# reads in rank order (what a coordinate-sorted BAM gives)
order = np.argsort(t.rank, kind="stable")
k = np.diff(t.off)
off = np.concatenate([[0], np.cumsum(k[order])]).astype(np.int64)
bases = np.concatenate([t.bases[t.off[r]:t.off[r + 1]] for r in order]) if len(order) < 200000 else None
if bases is None:
    idx = np.repeat(t.off[:-1][order], k[order]) + (np.arange(off[-1]) - np.repeat(off[:-1], k[order]))
    bases = t.bases[idx]
rank = t.rank[order]
h = Hansel(t.n_snps, band=t.band)
h.fill_from_support(rank, off, bases)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.perf_counter(); res = h.spin(n); dt = time.perf_counter() - t0
print("spin(%d): %.1f ms, n %d, walk_clock %s" % (n, dt * 1e3, res["n"], h.walk_clock()))
