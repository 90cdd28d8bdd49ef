"""Host-side pieces that need no GPU: the synthetic generator and the BAM/VCF decoding."""
import numpy as np

from gretel_amd.synth import make_support_table, make_config, CONFIGS


def test_generator_is_seeded_and_shaped():
    a = make_support_table(500, 20000, k=3, seed=4)
    b = make_support_table(500, 20000, k=3, seed=4)
    c = make_support_table(500, 20000, k=3, seed=5)
    assert np.array_equal(a.bases, b.bases) and np.array_equal(a.rank, b.rank)
    assert not np.array_equal(a.bases, c.bases)
    assert a.n_reads == 20000 and a.max_k == 3 and a.band == 2
    assert a.rank.min() == 0 and a.rank.max() == 500 - 3
    assert set(np.unique(a.bases).tolist()) <= set(b"ACGT")
    # every SNP has >= 2 alleles among the true haplotypes
    assert all(len(set(a.haplotypes[:, s].tolist())) >= 2 for s in range(500))


def test_tiling_bridges_every_adjacent_pair():
    t = make_support_table(300, 400, k=5, seed=0)       # almost only the tiling reads
    covered = np.zeros(300, dtype=bool)
    for r in range(t.n_reads):
        k = t.off[r + 1] - t.off[r]
        covered[t.rank[r]: t.rank[r] + k - 1] = True    # pair (s, s+1) bridged
    assert covered[:299].all()


def test_variable_k_config():
    t = make_support_table(2000, 5000, k=None, seed=1)
    ks = np.diff(t.off)
    assert ks.min() >= 2 and ks.max() <= 21
    assert (t.rank + ks <= 2000).all()
    assert set(CONFIGS) == {"C2", "C3", "C5"}


def test_result_exchange_refuses_to_drop_an_uncollected_step():
    # two staging slots: the third buffers() without a collect() in between would hand out a slot whose records nobody has
    # looked at -- the class used to finish and DROP that submission silently
    import torch
    from gretel_amd.dist import ResultExchange
    ex = ResultExchange(n_snps=10, max_paths=3, device=torch.device("cpu"), world=1, rank=0)
    for step in range(2):
        pv, rv = ex.buffers()
        pv[:] = step
        ex.submit(1, 0)
    import pytest
    with pytest.raises(RuntimeError, match="never collected"):
        ex.buffers()
    got = ex.collect()
    assert got[0]["n"] == 1 and int(got[0]["paths"][0, 0]) == 0
    pv, rv = ex.buffers()           # the collected slot is free again
    ex.submit(2, 0)
    assert ex.collect()[0]["paths"][0, 0] == 1 and ex.collect()[0]["n"] == 2 and ex.collect() is None
    ex.close()


def test_process_vcf_fast_path_is_the_loop(tmp_path, monkeypatch):
    """gretel/util.py:354-414: the vectorised scan of the POS column gives what the line-by-line loop gives -- header lines, other
    contigs, a contig whose name starts with ours, repeated positions, a last line without a newline, CRLF -- and steps aside
    (the loop then raises what int() raises) when a POS field is not a number."""
    import gzip
    import numpy as np
    import pytest
    from gretel_amd import util
    body = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT", "ctg\t5\t.\tA\tC", "ctg2\t6\t.\tA\tC", "other\t7\t.\tA\tC",
            "ctg\t12\t.\tG\tT", "ctg\t12\t.\tG\tA", "ctg\t300\t.\tG\tA", "ctg\t41"]

    def both(text, name, lo, hi, gz=False):
        f = tmp_path / ("v.vcf.gz" if gz else "v.vcf")
        (gzip.open if gz else open)(f, "wb").write(text)
        fast = util.process_vcf(str(f), name, lo, hi)
        monkeypatch.setattr(util, "_vcf_positions", lambda d, k: None)
        slow = util.process_vcf(str(f), name, lo, hi)
        monkeypatch.undo()
        assert fast["N"] == slow["N"] and fast["snp_fwd"] == slow["snp_fwd"] and fast["snp_rev"] == slow["snp_rev"]
        assert np.array_equal(fast["region"], slow["region"]) and fast["region"].dtype == slow["region"].dtype
        return fast

    v = both("\n".join(body).encode(), "ctg", 1, 100, gz=True)
    assert v["N"] == 4 and v["snp_rev"] == {0: 5, 1: 12, 2: 12, 3: 41} and v["snp_fwd"][12] == 2
    assert both("\n".join(body).encode() + b"\n", "ctg", 6, 300)["N"] == 4
    assert both("\r\n".join(body[:-1]).encode(), "ctg2", 1, 10)["N"] == 1
    assert both(b"", "ctg", 1, 10)["N"] == 0 and both(b"#only a header\n", "ctg", 1, 10)["N"] == 0
    f = tmp_path / "bad.vcf"
    f.write_bytes(b"ctg\t12x\t.\n")
    assert util._vcf_positions(f.read_bytes(), b"ctg\t") is None
    with pytest.raises(ValueError):
        util.process_vcf(str(f), "ctg", 1, 100)


def test_the_builds_hash_covers_kernel_sources_only():
    """bench.kernel_source_sha names the build a profile belongs to (profiles/*.json quote it, bench.py only quotes the counters'
    traffic when it matches): every kernel source and C-ABI header of the GPU library, not the host-side BAM decoder's header."""
    import glob
    import hashlib
    import os
    import bench
    root = os.path.dirname(os.path.abspath(bench.__file__))
    files = sorted(glob.glob(os.path.join(root, "gretel_amd", "csrc", "*.h*")) + glob.glob(os.path.join(root, "include", "*.h")) +
                   glob.glob(os.path.join(root, "include", "*.inc")))
    names = [os.path.basename(f) for f in files]
    for must in ("wpipe.hpp", "kernels.hpp", "segwalk.hpp", "segmix.hpp", "cwalk.hpp", "seg_geom.hpp", "gretel_hip.hip", "gretel_hip.h", "gh_detlog.h", "gh_logtab.inc"):
        assert must in names, must
    hh = hashlib.sha256()
    for f in files:
        if os.path.basename(f) == "gretel_io.h":
            continue
        hh.update(os.path.basename(f).encode())
        hh.update(open(f, "rb").read())
    assert bench.kernel_source_sha() == hh.hexdigest()[:16]
    # the committed counter summaries of this round belong to this build (scratch/copy_profiles.sh after profiles/collect.sh)
    import json
    import pytest
    stale = []
    for name in (bench.PROF_ROUND + "_pmc_traffic.json", bench.PROF_ROUND + "_pmc_traffic_c5.json", bench.PROF_ROUND + "_pmc_traffic_batch256.json"):
        if not os.path.exists(os.path.join(root, "profiles", name)):
            pytest.skip("profiles/%s has not been collected yet (profiles/collect.sh %s)" % (name, bench.PROF_ROUND))
        doc = json.load(open(os.path.join(root, "profiles", name)))
        assert doc["git_dirty"] is False, name
        if doc["kernel_source_sha"] != bench.kernel_source_sha():
            stale.append(name)
    if stale:       # (while kernels are being worked on this is the normal state: bench.py then leaves `traffic` null and says why)
        pytest.skip("the counter summaries %s are of another build: run profiles/collect.sh before quoting them" % ", ".join(stale))


def test_result_exchange_after_a_timeout_is_closed_and_close_returns():
    """a gather that never completes (a peer that failed never enters the collective) -- collect() raises, the slot is
    NOT handed out again while the worker may still be inside the collective, and close() in a `finally` does not hang."""
    import threading
    import time
    import pytest
    import torch
    from gretel_amd.dist import ResultExchange
    ex = ResultExchange(n_snps=10, max_paths=3, device=torch.device("cpu"), world=1, rank=0, force=False)
    # a stuck worker by hand (no process group needed): the exchange believes it is active and its worker never signals
    ex.active, ex.timeout_s = True, 0.2
    import queue as queue_mod
    ex._jobs = queue_mod.Queue()
    release = threading.Event()
    ex._thread = threading.Thread(target=lambda: release.wait(30.0), daemon=True)
    ex._thread.start()
    ex.buffers()
    ex.submit(1, 0)
    with pytest.raises(TimeoutError):
        ex.collect()
    assert ex.broken and len(ex.queue) == 1            # still tracked: its slot is not free
    for call in (ex.buffers, ex.collect, lambda: ex.submit(1, 0)):
        with pytest.raises(RuntimeError, match="timed out"):
            call()
    t0 = time.time()
    ex.close()                                          # (the worker is still blocked)
    assert time.time() - t0 < 5.0
    release.set()


def test_pinned_block_lives_as_long_as_its_views():
    """HanselBatch.spin(copy=False) hands out views of page-locked memory; the block must outlive the batch for as long
    as a view refers to it (it used to be freed with the batch or on a change of shape: use after free)."""
    import ctypes as C
    import gc
    from gretel_amd.hansel import _PinnedBlock

    class FakeLib:
        def __init__(self):
            self.live = {}
            self.libc = C.CDLL(None)
            self.libc.malloc.restype = C.c_void_p
            self.libc.free.argtypes = [C.c_void_p]

        def gh_host_alloc(self, n, out):
            p = self.libc.malloc(C.c_size_t(n))
            C.cast(out, C.POINTER(C.c_void_p))[0] = p
            self.live[p] = n
            return 0

        def gh_host_free(self, p):
            self.live.pop(p.value)
            self.libc.free(p)
            return 0

    lib = FakeLib()
    blk = _PinnedBlock(lib, 64)
    a = blk.array(np.uint8, 64).reshape(4, 16)
    view = a[1, :4]
    a[:] = 7
    del blk, a
    gc.collect()
    assert len(lib.live) == 1 and int(view.sum()) == 28    # the view keeps the block
    del view
    gc.collect()
    assert not lib.live
