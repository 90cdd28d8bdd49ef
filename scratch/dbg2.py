import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from gretel_amd.hansel import Hansel
from gretel_amd.synth import make_support_table
mode = sys.argv[1]
t = make_support_table(40, 900, k=3, seed=1)
h = Hansel(t.n_snps, band=t.band)
h.fill_from_support(t.rank, t.off, t.bases)
if mode == "orig":
    o = h.copy()
    for i in range(5):
        print(h.generate_path(o)[1:], flush=True)
elif mode == "self":
    for i in range(5):
        print(h.generate_path()[1:], flush=True)
elif mode == "spin":
    r = h.spin(10)
    print(r["n"], r["hp_current"][:3])
elif mode == "two":
    h2 = Hansel(t.n_snps, band=t.band)
    h2.fill_from_support(t.rank, t.off, t.bases)
    for i in range(5):
        print(h.generate_path()[1:], h2.generate_path()[1:], flush=True)
