import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_config
t = make_config("C5", seed=0)
h = Hansel(t.n_snps, band=t.band)
h.fill_from_support(t.rank, t.off, t.bases)
res = h.spin(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
print(res["n"], h.walk_clock())
