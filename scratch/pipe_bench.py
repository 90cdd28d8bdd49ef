"""Throughput of the window pipeline (csrc/wpipe.hpp) on replicas of a config's contig.  argv: windows [paths] [reps] [config]
env: GH_PIPE=0 -> the batched launches of rounds 1-4; GH_PIPE_NT; GH_PIPE_STAMPS=1 -> per-path times of a few windows;
PB_COND (A..E), PB_MT (0/1), PB_STORAGE (f32/f64): the spec; PB_DEL=frac -> '-' at that fraction of the POSITIONS (bench.py's
wide_window_sparse window: 0.01), the windows then go through the WIDE launch of the pipeline."""
import os, sys, time
import numpy as np
import torch
from gretel_amd.hansel import Hansel, HanselBatch, DeviceReads
from gretel_amd.synth import make_config

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 256
paths = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = sys.argv[4] if len(sys.argv) > 4 else "C3"
spec = dict(cond_mode=os.environ.get("PB_COND", "A"), marginal_term=bool(int(os.environ.get("PB_MT", "0"))),
            storage=os.environ.get("PB_STORAGE", "f32"))
tag = "%s%s %s" % (spec["cond_mode"], "+mt" if spec["marginal_term"] else "", spec["storage"])
t = make_config(cfg, seed=0)
mix = int(os.environ.get("PB_MIX", "0"))          # PB_MIX=k: only every k-th window gets the deletions (a batch of narrow and wide windows)
h0 = Hansel(t.n_snps, band=t.band, **spec)
reads_plain = DeviceReads(h0, t.rank, t.off, t.bases) if mix else None
if float(os.environ.get("PB_DEL", "0")) > 0:
    from gretel_amd.synth import sprinkle_deletions
    import copy
    t = copy.copy(t); t.bases = t.bases.copy()
    sprinkle_deletions(t, float(os.environ["PB_DEL"]), seed=4321)
    tag += " del %s%s" % (os.environ["PB_DEL"], (" in every %d-th window" % mix) if mix else "")
reads = DeviceReads(h0, t.rank, t.off, t.bases)
hs = [Hansel(t.n_snps, band=t.band, **spec) for _ in range(nw)]
hb = HanselBatch(hs)
hb.profile_enable(10)
for r in range(reps):
    for q, h in enumerate(hs):
        h.clear()
        h.fill_from_support(None, None, None, reads_handle=(reads if (not mix or q % mix == 0) else reads_plain))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = hb.spin(paths, copy=False)
    dt = time.perf_counter() - t0
    tot = sum(x["n"] for x in res)
    pg = hb.profile_get()["walk"]
    print("%s windows %d paths %d: %.1f ms, %.0f haplotypes/s; pipe %s; kernel %.2f ms (%d launches), %.2f TB/s on the pipeline's own bytes"
          % (tag, nw, paths, dt * 1e3, tot / dt, hb.pipe_info(), pg["ms"], pg["launches"],
             pg["bytes_per_launch"] / max(1e-9, pg["ms"] * 1e-3) / 1e12), flush=True)
ref = res[0]
assert all(np.array_equal(x["paths"], ref["paths"]) for q, x in enumerate(res) if not mix or q % mix == 0)
