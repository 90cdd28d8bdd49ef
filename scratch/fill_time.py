"""scratch: time the fill alone (HIP events of the library's own bracket) for a config: python scratch/fill_time.py C5"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_config
t = make_config(sys.argv[1])
h = Hansel(t.n_snps, band=t.band)
r = DeviceReads(h, t.rank, t.off, t.bases)
h.profile_enable(1)
for rep in range(4):
    h.clear(); h.profile_reset()
    h.fill_from_support(None, None, None, reads_handle=r)
    p = h.profile_get()
    print(sys.argv[1], "fill %.1f us" % (p["fill"]["ms"] * 1e3))
