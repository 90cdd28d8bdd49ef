"""Windows with five-candidate positions at lag counts >= 6: candidate pools over the symbol table against the serial walker."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_support_table
t = make_support_table(10000, 150000, k=None, seed=5, n_haps=8, err=0.01, k_max=26)
bases = t.bases.copy()
bases[np.random.default_rng(1).random(len(bases)) < 0.05] = ord('-')
t.bases = bases
for L in [int(x) for x in sys.argv[1:]]:
    h = Hansel(t.n_snps, band=t.band)
    reads = DeviceReads(h, t.rank, t.off, t.bases)
    ts = []
    for it in range(2):
        h.clear(); h.fill_from_support(None, None, None, reads_handle=reads); h.L = L
        t0 = time.perf_counter(); res = h.spin(100); ts.append(time.perf_counter() - t0)
    wide = int((h.candidate_masks()[1:] == 0x2F).sum())
    print("L=%2d  %s us/path  variant %d  walk_clock %s  wide positions %d" % (L, " ".join("%.0f" % (x / res["n"] * 1e6) for x in ts), h.walk_clock()[3], h.walk_clock(), wide), flush=True)
