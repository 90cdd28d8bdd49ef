import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_config
t = make_config("C5", seed=0)
h = Hansel(t.n_snps, band=t.band)
reads = DeviceReads(h, t.rank, t.off, t.bases)
for it in range(2):
    h.clear(); h.fill_from_support(None, None, None, reads_handle=reads)
    t0 = time.perf_counter(); res = h.spin(1000); dt = time.perf_counter() - t0
    print("spin(1000): %.1f ms, n %d, walk_clock %s" % (dt * 1e3, res["n"], h.walk_clock()), flush=True)
