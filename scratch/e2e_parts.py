"""Where the end-to-end leg's time goes (bench.py end_to_end_leg), piece by piece.  argv: [dir]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gretel_amd import bamio, util
from gretel_amd.synth import make_config
from gretel_amd.hansel import Hansel, DeviceReads
import torch
d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/e2e"
os.makedirs(d, exist_ok=True)
bam, vcf = os.path.join(d, "s.bam"), os.path.join(d, "s.vcf.gz")
t = make_config("C3", seed=0)
if not os.path.exists(bam):
    bamio.synth_to_files(t, bam, vcf)
contig, start, end = "synth", 1, 10 * t.n_snps + 10
T = time.perf_counter
from gretel_amd.hansel import table_arena
arena = None if os.environ.get("E2E_NO_ARENA") else table_arena()
for it in range(5):
    a = T()
    if not os.environ.get("E2E_NO_PREFETCH"): util.prefetch_bam(bam, contig, start, end)
    v = util.process_vcf(vcf, contig, start, end); b = T()
    rank, off, bases = util.support_table_from_bam(bam, contig, start, end, v, max_depth=(0 if os.environ.get("E2E_DEPTH0") else 8000), arena=arena); c = T()
    st = bamio.native_last_stats()
    mk = st["max_row_len"]; d0 = T()
    h = Hansel(v["N"], band=max(1, mk - 1), device=0); h._ensure(max(1, mk - 1)); torch.cuda.synchronize(); e = T()
    r = DeviceReads(h, rank, off, bases, max_k=mk); torch.cuda.synchronize(); f = T()
    h.fill_from_support(None, None, None, reads_handle=r); torch.cuda.synchronize(); g = T()
    res = h.spin(100); hh = T()
    print("vcf %.2f  decode %.2f (native %.2f)  Hansel() %.2f  DeviceReads %.2f  fill %.2f  spin %.2f   total %.2f ms"
          % ((b - a) * 1e3, (c - b) * 1e3, st["seconds"] * 1e3, (e - d0) * 1e3, (f - e) * 1e3, (g - f) * 1e3, (hh - g) * 1e3, (hh - a) * 1e3), flush=True)
    if it == 0:
        assert np.array_equal(rank, t.rank) and np.array_equal(off, t.off) and np.array_equal(bases, t.bases) and r.max_k == int(np.diff(t.off).max())
    del h, r, rank, off, bases, res
    import gc; gc.collect()
