"""Phase times of k_seg's workgroup 0 (diagnostic build scratch/lib_stamps.so, -DSEG_STAMPS): run with
GH_LIB=scratch/lib_stamps.so GH_PRINT_STAMPS=1"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_config
t = make_config(sys.argv[1] if len(sys.argv) > 1 else "C3", seed=0)
h = Hansel(t.n_snps, band=t.band)
h.fill_from_support(t.rank, t.off, t.bases)
for _ in range(3):
    h.spin(10)
    print("cycles: stage / build Next / enumerate / maps", file=sys.stderr)
    h.walk_clock()
