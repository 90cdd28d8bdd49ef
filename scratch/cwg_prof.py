import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_support_table
t = make_support_table(10000, 150000, k=None, seed=5, n_haps=8, err=0.0, k_max=26)
h = Hansel(t.n_snps, band=t.band)
h.fill_from_support(t.rank, t.off, t.bases)
h.L = int(sys.argv[1]) if len(sys.argv) > 1 else 25
t0 = time.perf_counter(); res = h.spin(100); dt = time.perf_counter() - t0
print("L=%d %.0f us/path" % (h.L, dt / res["n"] * 1e6), h.walk_clock())
