import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_support_table
t = make_support_table(10000, 150000, k=None, seed=5, n_haps=8, err=0.0, k_max=26)
h = Hansel(t.n_snps, band=t.band)
reads = DeviceReads(h, t.rank, t.off, t.bases)
for L in [int(x) for x in sys.argv[1:]]:
    ts = []
    for it in range(3):
        h.clear(); h.fill_from_support(None, None, None, reads_handle=reads); h.L = L
        t0 = time.perf_counter(); res = h.spin(200); ts.append(time.perf_counter() - t0)
    print("L=%2d  %s us/path  variant %d" % (L, " ".join("%.0f" % (x / res["n"] * 1e6) for x in ts), h.walk_clock()[3]), flush=True)
