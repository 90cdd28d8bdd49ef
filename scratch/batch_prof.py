import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gretel_amd.hansel import Hansel, HanselBatch, DeviceReads
from gretel_amd.synth import make_config
t = make_config("C3", seed=0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hs = [Hansel(t.n_snps, band=t.band) for _ in range(reps)]
reads = DeviceReads(hs[0], t.rank, t.off, t.bases)
for h in hs:
    h.fill_from_support(None, None, None, reads_handle=reads)
hb = HanselBatch(hs)
for it in range(2):
    for h in hs:
        h.clear(); h.fill_from_support(None, None, None, reads_handle=reads)
    t0 = time.perf_counter(); r = hb.spin(100); dt = time.perf_counter() - t0
    print("batch", reps, "windows: %.1f ms  %.0f haplotypes/s" % (dt * 1e3, sum(x["n"] for x in r) / dt), flush=True)
