"""Throughput of a batch over the lag count: 128 replicas of the lag-sweep window of scratch/l_sweep.py (10k SNPs, long-read-style
reads, band 25), 20 paths each, through the window pipeline (csrc/wpipe.hpp: L = 2..14) and -- GH_PIPE=0 -- the ways of rounds 1-4
(up to L = 5 batched launches, beyond: the candidate pools of every window on its own stream).  argv: lag counts (default 2..14)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gretel_amd.hansel import Hansel, HanselBatch, DeviceReads
from gretel_amd.synth import make_support_table
NW, PATHS = 128, 20
t = make_support_table(10000, 150000, k=None, seed=5, n_haps=8, err=0.0, k_max=26)
h0 = Hansel(t.n_snps, band=t.band)
reads = DeviceReads(h0, t.rank, t.off, t.bases)
hs = [Hansel(t.n_snps, band=t.band) for _ in range(NW)]
hb = HanselBatch(hs)
print("N %d band %d reads %d, %d windows x %d paths, GH_PIPE=%s" % (t.n_snps, t.band, t.n_reads, NW, PATHS, os.environ.get("GH_PIPE", "1")))
for L in ([int(x) for x in sys.argv[1:]] or list(range(2, 15))):
    best = None
    for it in range(2):
        for h in hs:
            h.clear(); h.fill_from_support(None, None, None, reads_handle=reads); h.L = L
        t0 = time.perf_counter(); res = hb.spin(PATHS, copy=False); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    n = sum(r["n"] for r in res)
    assert all(np.array_equal(r["paths"], res[0]["paths"]) for r in res)
    print("L=%2d  %8.0f haplotypes/s  %7.1f us per path and window  pipe %s" % (L, n / best, best / PATHS * 1e6, hb.pipe_info()), flush=True)
