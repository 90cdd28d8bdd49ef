"""Phase times of k_rw's workgroup 100 (diagnostic build: -DRW_STAMPS -o scratch/lib_stamps.so; run with
GH_LIB=scratch/lib_stamps.so GH_PRINT_STAMPS=1).  argv: cond_mode storage [config]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_config
t = make_config(sys.argv[3] if len(sys.argv) > 3 else "C3", seed=0)
h = Hansel(t.n_snps, band=t.band, cond_mode=sys.argv[1], storage=sys.argv[2])
h.fill_from_support(t.rank, t.off, t.bases)
for _ in range(3):
    h.spin(10)
    print("%s %s cycles: min-reduce / round-2 loads / reweight+marginals / table" % (sys.argv[1], sys.argv[2]), file=sys.stderr)
    h.walk_clock()
