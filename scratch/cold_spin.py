"""First spin of a fresh handle against a later one (fill not included): what a handle's first spin pays for its buffers.
argv: [sparse|narrow] ; GH_LIB picks the build."""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gretel_amd.hansel import Hansel, DeviceReads
from gretel_amd.synth import make_config, sprinkle_deletions
kind = sys.argv[1] if len(sys.argv) > 1 else "sparse"
t = make_config("C3", seed=0)
if kind == "sparse":
    t = copy.copy(t); t.bases = t.bases.copy(); sprinkle_deletions(t, 0.01, seed=4321)
h0 = Hansel(t.n_snps, band=t.band)
reads = DeviceReads(h0, t.rank, t.off, t.bases)
cold, warm = [], []
for it in range(6):
    h = Hansel(t.n_snps, band=t.band)
    h.fill_from_support(None, None, None, reads_handle=reads)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); h.spin(100); t1 = time.perf_counter()
    h.clear(); h.fill_from_support(None, None, None, reads_handle=reads); torch.cuda.synchronize()
    t2 = time.perf_counter(); h.spin(100); t3 = time.perf_counter()
    cold.append(t1 - t0); warm.append(t3 - t2)
    del h
print("%s: first spin %.3f ms (first handle of the process %.3f), a later spin %.3f ms" % (kind, np.median(cold[1:]) * 1e3, cold[0] * 1e3, np.median(warm[1:]) * 1e3))
