"""l_sweep.py for the lag counts of the cliff (L = 5..12), with the device state printed per queue (GH_PRINT_STATE=1; a -DCW_DIAG
build adds the histogram of the round every chain closed in)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_support_table
t = make_support_table(10000, 150000, k=None, seed=5, n_haps=8, err=0.0, k_max=26)
h = Hansel(t.n_snps, band=t.band)
reads = DeviceReads(h, t.rank, t.off, t.bases)
for L in [int(x) for x in (sys.argv[1:] or "5 6 7 8 9 10 12".split())]:
    best = None
    for it in range(2):
        h.clear(); h.fill_from_support(None, None, None, reads_handle=reads); h.L = L
        t0 = time.perf_counter(); res = h.spin(200); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    wc = h.walk_clock()
    print("L=%2d  %7.1f us/path  n %d  variant %d  requeues %d serial %d rounds %d" % (L, best / max(1, res["n"]) * 1e6, res["n"], wc[3], wc[0], wc[1], wc[2]), flush=True)
