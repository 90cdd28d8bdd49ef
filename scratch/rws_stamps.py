"""Phase times of k_rwseg's workgroup 100 (diagnostic build: -DRWS_STAMPS -o scratch/lib_stamps.so; run with
GH_LIB=scratch/lib_stamps.so GH_PRINT_STAMPS=1).  argv: cond_mode storage [mt]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_config
t = make_config("C3", seed=0)
h = Hansel(t.n_snps, band=t.band, cond_mode=sys.argv[1], storage=sys.argv[2], marginal_term=len(sys.argv) > 3)
h.fill_from_support(t.rank, t.off, t.bases)
print("cycles (100 MHz ticks x ?): entry->loads | min-reduce | reweight+marginals | table rows | removed-reduce+fence | staging | Next | walk | maps", file=sys.stderr)
for _ in range(4):
    h.spin(10)
    h.walk_clock()
