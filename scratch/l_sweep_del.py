"""Per-path time of a spin over the lag count, with and without deletion columns: the lag-sweep window (10k SNPs, long-read-style
reads, k ~ Poisson(10) up to 25) as it is (every position offers at most four candidates) and with '-' at 1 % of the POSITIONS
on 30 % of the reads that cover them (gretel_amd.synth.sprinkle_deletions: a deletion column here and there, what a real pileup
shows, gretel/util.py:178-190).  VERDICT r5 item 1: no L in 2..16 may cost more than 1.5x its narrow figure.
Usage: l_sweep_del.py [L ...]     (default 2..16 and a dozen lag counts up to 41; 200 paths per spin, best of two spins, fill not included)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_support_table, sprinkle_deletions

def window(frac):
    t = make_support_table(10000, 150000, k=None, seed=5, n_haps=8, err=0.0, k_max=26)
    if frac > 0:
        pos = sprinkle_deletions(t, frac, seed=4321)
        print("# %d positions with '-' (%.1f %% of %d)" % (len(pos), 100.0 * frac, t.n_snps))
    h = Hansel(t.n_snps, band=t.band)
    return t, h, DeviceReads(h, t.rank, t.off, t.bases)

def per_path(h, reads, L, paths=200):
    best = None
    for it in range(2):
        h.clear(); h.fill_from_support(None, None, None, reads_handle=reads); h.L = L
        t0 = time.perf_counter(); res = h.spin(paths); dt = time.perf_counter() - t0
        v = dt / max(1, res["n"]) * 1e6
        best = v if best is None or v < best else best
    wc = h.walk_clock()
    return best, res["n"], wc

if __name__ == "__main__":
    Ls = [int(x) for x in sys.argv[1:]] or (list(range(2, 17)) + [18, 20, 21, 22, 24, 28, 32, 33, 36, 40, 41])
    tn, hn, rn = window(0.0)
    td, hd, rd = window(0.01)
    print("N %d band %d reads %d" % (tn.n_snps, tn.band, tn.n_reads))
    worst = 0.0
    for L in Ls:
        a, na, wa = per_path(hn, rn, L)
        b, nb, wb = per_path(hd, rd, L)
        worst = max(worst, b / a)
        print("L=%2d  narrow %7.1f us/path (n %d, variant %d)   1 %% deletion columns %7.1f us/path (n %d, variant %d, class/maxstates %s)   x%.2f"
              % (L, a, na, wa[3], b, nb, wb[3], "%d/%d" % (wb[1], wb[2]), b / a), flush=True)
    print("worst ratio over L = %s: x%.2f" % (Ls, worst))
