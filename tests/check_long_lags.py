"""Test infrastructure (calls the oracle): C5's contig at lag counts 16, 17, 20, 24 against the C oracle, with timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_config
from oracle.c_oracle import COracle
t = make_config("C5", seed=0)
for L in (16, 17, 20, 24):
    h = Hansel(t.n_snps, band=t.band)
    reads = DeviceReads(h, t.rank, t.off, t.bases)
    h.fill_from_support(None, None, None, reads_handle=reads)
    h.L = L
    h.spin(20)
    h.clear(); h.fill_from_support(None, None, None, reads_handle=reads); h.L = L
    t0 = time.perf_counter(); res = h.spin(300); dt = time.perf_counter() - t0
    print("L=%d spin(300): %.1f ms, n %d, walk_clock %s" % (L, dt * 1e3, res["n"], h.walk_clock()), flush=True)
    if L in (17, 24):
        o = COracle(t.n_snps, t.band); o.fill(t); o.L = L
        ref = o.spin(40)
        assert np.array_equal(res["paths"][:40], ref["paths"]) and res["hp_current"][:40].tolist() == ref["hp_current"].tolist()
        print("  first 40 paths bit-exact vs oracle")
