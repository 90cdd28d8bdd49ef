"""C5 (50k SNPs, L = 11), 300 paths twice, per GH_CW_RUNON setting given in the environment."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_config
t = make_config("C5", seed=0)
h = Hansel(t.n_snps, band=t.band)
reads = DeviceReads(h, t.rank, t.off, t.bases)
for it in range(2):
    h.clear(); h.fill_from_support(None, None, None, reads_handle=reads)
    t0 = time.perf_counter(); res = h.spin(300); dt = time.perf_counter() - t0
    wc = h.walk_clock()
    print("GH_CW_RUNON=%s  %.1f us/path  n %d requeues %d serial %d rounds %d" % (os.environ.get("GH_CW_RUNON"), dt / res["n"] * 1e6, res["n"], wc[0], wc[1], wc[2]), flush=True)
