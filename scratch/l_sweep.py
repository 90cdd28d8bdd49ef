"""Per-path time of a spin over the lag count: the same 10k-SNP window (long-read-style reads, k ~ Poisson(10) up to 25)
at L = 1 .. 66 (or the lag counts given as arguments), 200 paths each (fill not included); the walker variant the last path took (3 = every state of every
segment, 4 = candidate pools, 0/2 = serial walker)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_support_table
t = make_support_table(10000, 150000, k=None, seed=5, n_haps=8, err=0.0, k_max=26)
h = Hansel(t.n_snps, band=t.band)
reads = DeviceReads(h, t.rank, t.off, t.bases)
print("N %d band %d reads %d" % (t.n_snps, t.band, t.n_reads))
for L in ([int(x) for x in sys.argv[1:]] or list(range(1, 67))):
    best = None
    for it in range(2):
        h.clear(); h.fill_from_support(None, None, None, reads_handle=reads); h.L = L
        t0 = time.perf_counter(); res = h.spin(200); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    wc = h.walk_clock()
    print("L=%2d  %7.1f us/path  n %d  variant %d  requeues %d serial %d" % (L, best / max(1, res["n"]) * 1e6, res["n"], wc[3], wc[0], wc[1]), flush=True)
