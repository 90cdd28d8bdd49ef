"""where a bench.py --batch step spends its time: per-window clear+fill vs the batched spin"""
import sys, time
sys.path.insert(0, ".")
import torch
from gretel_amd.hansel import Hansel, HanselBatch, DeviceReads
from gretel_amd.synth import make_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tables = [make_config("C3", seed=q) for q in range(4)]
hs = [Hansel(tables[0].n_snps, band=tables[0].band, device=0) for _ in range(B)]
reads = [DeviceReads(hs[q], tables[q].rank, tables[q].off, tables[q].bases) for q in range(4)]
batch = HanselBatch(hs)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for w, h in enumerate(hs):
        h.clear()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for w, h in enumerate(hs):
        h.fill_from_support(None, None, None, reads_handle=reads[w % 4])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res = batch.spin(100)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("clear %.1f ms  fill %.1f ms  spin %.1f ms  (%d windows)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, B))
