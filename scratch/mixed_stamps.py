"""Phase times of k_rwseg's workgroup 100 over a sparse-deletion window (mixed radix) or the plain one (frac 0): diagnostic build
-DRWS_STAMPS -o scratch/lib_stamps.so; run with GH_LIB=scratch/lib_stamps.so GH_PRINT_STAMPS=1.  argv: fraction of positions"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_config, sprinkle_deletions
frac = float(sys.argv[1])
t = make_config("C3", seed=0)
if frac > 0:
    sprinkle_deletions(t, frac, seed=4321)
h = Hansel(t.n_snps, band=t.band)
h.fill_from_support(t.rank, t.off, t.bases)
print("cycles: entry->loads | min-reduce | reweight+marginals | table rows | removed-reduce+fence | (seg entry) | staging | Next | walk | maps", file=sys.stderr)
for _ in range(4):
    h.spin(10)
    print(h.walk_clock(), file=sys.stderr)
