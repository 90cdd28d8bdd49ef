"""Randomised parity fuzz: HIP path vs the C oracle on random windows, switches and lag counts (not part of the
default suite; run on the GPU box from the repo root: python tests/fuzz_gpu.py [seconds] [seed]).  Test infrastructure:
it is the only place outside tests/test_*.py, smoke() and bench.py's cpu_baseline that calls the oracle."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from gretel_amd.hansel import Hansel, HanselBatch
from gretel_amd.synth import make_support_table, sprinkle_deletions
from oracle.c_oracle import COracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3 and sys.argv[3] == "big"      # long windows (many LDS chunks) instead of many small ones
rng = np.random.default_rng(seed0)
t_end = time.time() + budget
n_cases = 0
n_batch = 0
n_pipe = 0
n_wide = 0
variants = {}
while time.time() < t_end:
    n = int(rng.choice([4000, 7777, 12000, 20011])) if big else int(rng.choice([2, 3, 5, 17, 64, 130, 333, 700, 1500, 3000]))
    k = int(rng.integers(2, min(n, 12) + 1)) if rng.random() < 0.7 else None
    # reads of up to 21 SNPs mostly (C5's shape), up to 33 / 64 now and then: bands beyond the 32 lanes of the wide reweight kernels
    kmx = int(rng.choice([21, 21, 21, 33, 48, 64, 72]))
    lam = 10.0 if kmx == 21 else float(rng.choice([10.0, kmx * 0.6]))
    n_haps = int(rng.integers(1, 9))
    err = float(rng.choice([0.0, 0.0, 0.01, 0.05]))
    reads = int(max(20, n * rng.integers(4, 12 if big else 40)))
    t = make_support_table(n, reads, k=k, n_haps=n_haps, err=err, seed=int(rng.integers(0, 1 << 30)), k_max=min(kmx, n), k_lambda=lam)
    if rng.random() < 0.4:
        bases = t.bases.copy()
        bases[rng.random(len(bases)) < rng.choice([0.02, 0.1, 0.3])] = ord('-')
        if rng.random() < 0.3:
            bases[rng.random(len(bases)) < 0.03] = ord('N')
        t.bases = bases
    sparse = False
    if rng.random() < 0.15 and n >= 17:
        # deletions at a few POSITIONS only (a five-candidate column here and there): the mixed-radix state space at L = 5
        sprinkle_deletions(t, float(rng.choice([0.004, 0.02, 0.06])), frac_reads=float(rng.choice([0.1, 0.3, 0.6])), seed=int(rng.integers(0, 1 << 30)))
        sparse = True
    storage = str(rng.choice(["f32", "f32", "f64"]))
    mode = str(rng.choice(["A", "A", "B", "C", "D", "E"]))
    mt = bool(rng.random() < 0.25)
    # the other switches of the unpinned arithmetic: the order candidates are offered in, zero-count candidates
    order = "".join(rng.permutation(list("ACGT-"))) if rng.random() < 0.25 else "ACGT-"
    zero = bool(rng.random() < 0.15)
    sw = dict(cand_order=order, offer_zero=zero)
    # (25..69: the packed pools' last lag counts, k_cwalk2's 33..64 with its two- and four-chunk blocks, k_cwalkg behind them)
    L = None if rng.random() < 0.4 else (int(rng.integers(1, 27)) if rng.random() < 0.85 else int(rng.integers(25, 70)))
    if sparse and rng.random() < 0.8:
        L = 5
    paths = int(rng.integers(1, 9))
    desc = dict(n=n, reads=reads, k=k, k_max=kmx, k_lambda=lam, sparse_dels=sparse, n_haps=n_haps, err=err, storage=storage, mode=mode, mt=mt, L=L, paths=paths, **sw)
    if k is not None and rng.random() < float(os.environ.get("FUZZ_BATCH_P", "0.12")):
        # batched launch over 3 windows of one shape (same N, band, switches, L)
        ts = [t] + [make_support_table(n, reads, k=k, n_haps=int(rng.integers(1, 9)), err=err, seed=int(rng.integers(0, 1 << 30)))
                    for _ in range(2)]
        # (round 6: the other windows get deletion columns of their own now and then -- the pipeline's WIDE launch, wpipe.hpp)
        for x in ts[1:]:
            if rng.random() < 0.4 and n >= 17:
                sprinkle_deletions(x, float(rng.choice([0.004, 0.02, 0.06])), frac_reads=float(rng.choice([0.1, 0.3, 0.6])), seed=int(rng.integers(0, 1 << 30)))
        if len({x.band for x in ts}) == 1:
            hs, os_ = [], []
            for x in ts:
                hh = Hansel(x.n_snps, band=x.band, storage=storage, cond_mode=mode, marginal_term=mt, **sw)
                oo = COracle(x.n_snps, x.band, storage, mode, mt, **sw)
                assert hh.fill_from_support(x.rank, x.off, x.bases) == oo.fill(x)
                hh.L = L if L is not None else 3; oo.L = hh.L
                hs.append(hh); os_.append(oo)
            # the ways of running a batch: the window pipeline (csrc/wpipe.hpp: one persistent workgroup per window; it takes what
            # it can -- row conditionals, 2..14 lags, ranked windows -- and leaves the rest to the others), windows on their own
            # streams, or kernels launched over all windows
            way = rng.random()
            os.environ.pop("GH_PIPE_NT", None)
            if way < 0.5:
                os.environ["GH_PIPE_MIN"] = "1"
                if rng.random() < 0.3:
                    os.environ["GH_PIPE_NT"] = str(rng.choice([512, 768, 1024]))
            else:
                os.environ["GH_PIPE_MIN"] = "1000000"
            if way < 0.75:
                os.environ["GH_BATCH_STREAMS_MAX"] = "-1"
            else:
                os.environ.pop("GH_BATCH_STREAMS_MAX", None)
            deep = int(rng.integers(10, 60)) if rng.random() < 0.3 else paths       # deep spins: masks move, windows are handed back
            hb = HanselBatch(hs)
            for res, oo, hh in zip(hb.spin(deep), os_, hs):
                ref = oo.spin(deep)
                if not (res["n"] == ref["n"] and res["hole_at"] == ref["hole_at"] and np.array_equal(res["paths"], ref["paths"])
                        and res["hp_current"].tolist() == ref["hp_current"].tolist() and res["hp_original"].tolist() == ref["hp_original"].tolist()
                        and res["ratio"].tolist() == ref["ratio"].tolist() and np.array_equal(hh.export_band(), oo.export_band())):
                    print("MISMATCH batch", desc, dict(way=way, deep=deep, pipe=hb.pipe_info(), nt=os.environ.get("GH_PIPE_NT")), flush=True)
                    sys.exit(1)
            n_pipe += 1 if hb.pipe_info()["windows"] else 0
            n_wide += 1 if hb.pipe_info()["windows"] and any((hh.candidate_masks()[1:] == 0x2F).any() for hh in hs) else 0
            n_batch += 1
            continue
    # the serial walkers instead of the segment-parallel / pool extension now and then (GH_WALK is read when a handle is created)
    walk = str(rng.choice(["spec", "spec1", "src"])) if rng.random() < 0.1 else None
    if walk:
        os.environ["GH_WALK"] = walk
    else:
        os.environ.pop("GH_WALK", None)
    h = Hansel(t.n_snps, band=t.band, storage=storage, cond_mode=mode, marginal_term=mt, **sw)
    os.environ.pop("GH_WALK", None)
    o = COracle(t.n_snps, t.band, storage, mode, mt, **sw)
    sent = bool(rng.random() < 0.15)                     # the end sentinels of util.py:283
    if rng.random() < 0.15:
        # reads in any order (what an unsorted BAM gives): the scattered fills instead of the LDS-counting one
        perm = rng.permutation(t.n_reads)
        ks = np.diff(t.off)
        t.bases = np.concatenate([t.bases[t.off[i]:t.off[i + 1]] for i in perm]) if t.n_reads else t.bases
        t.off = np.concatenate([[0], np.cumsum(ks[perm])]).astype(np.int64)
        t.rank = np.ascontiguousarray(t.rank[perm])
    desc.update(walk=walk, sentinels=sent)
    try:
        if rng.random() < 0.1:
            # a handle that has been used before: fill, spin, clear -- then the case proper
            h.fill_from_support(t.rank, t.off, t.bases)
            h.spin(2)
            h.clear()
        assert h.fill_from_support(t.rank, t.off, t.bases, use_end_sentinels=sent) == o.fill(t, use_end_sentinels=sent), "fill stats"
        if rng.random() < 0.1:
            h = h.copy()                                 # cmd.py:79: the spin runs on a copy of the filled tensor
        if L is not None:
            h.L = L; o.L = L
        assert h.gap_check() == o.gap_check(), "gap"
        if rng.random() < 0.15:
            # the reference's own loop (cmd.py:148-179): one generate_path, one reweight_hansel_from_path at a time
            h.snapshot_original(); o.snapshot_original()
            for it in range(paths):
                pg, po = h.generate_path(), o.generate_path()
                if po[0] is None:
                    assert pg[0] is None and pg[1] == po[1], "lone path: hole %s %s" % (pg[1], po[1])
                    break
                assert pg[0] is not None and np.array_equal(pg[0], po[0]), "lone path %d" % it
                assert np.array_equal(np.array(pg[1:]), np.array(po[1]), equal_nan=True), "lone path %d: likelihoods / minimum %s %s" % (it, pg[1:], po[1])
                ratio = max(pg[3], 0.01) if pg[3] == pg[3] else 0.01
                mg, mo = h.reweight_from_path(pg[0], ratio), o.reweight_path(po[0], ratio)
                assert abs(mg - mo) <= 1e-10 * abs(mo), "lone reweight %d" % it
            assert np.array_equal(h.export_band(), o.export_band()), "band after lone calls"
            variants["lone"] = variants.get("lone", 0) + 1
            n_cases += 1
            continue
        res, ref = h.spin(paths), o.spin(paths)
        assert res["n"] == ref["n"] and res["hole_at"] == ref["hole_at"], "n/hole %s %s" % ((res["n"], res["hole_at"]), (ref["n"], ref["hole_at"]))
        assert np.array_equal(res["paths"], ref["paths"]), "paths"
        assert res["hp_current"].tolist() == ref["hp_current"].tolist(), "hp_current"
        assert res["hp_original"].tolist() == ref["hp_original"].tolist(), "hp_original"
        assert res["ratio"].tolist() == ref["ratio"].tolist(), "ratio"
        assert np.allclose(res["magnitude"], ref["magnitude"], rtol=1e-10, atol=0), "magnitude"
        assert np.array_equal(h.export_band(), o.export_band()), "band"
        # a second spin on the reweighted tensor (tables rebuilt / reused)
        res2, ref2 = h.spin(2), o.spin(2)
        assert res2["n"] == ref2["n"] and np.array_equal(res2["paths"], ref2["paths"]), "second spin"
        clk = h.walk_clock() if res["n"] else (0, 0, 0, -1)
        v = clk[3]
        variants[v] = variants.get(v, 0) + 1
        if v == 3 and clk[1] == 6:
            variants["mixed"] = variants.get("mixed", 0) + 1
    except AssertionError as e:
        print("MISMATCH", e, desc, flush=True)
        if n <= 20:
            print(" table: rank", t.rank.tolist(), "off", t.off.tolist(), "bases", bytes(t.bases).decode(), flush=True)
            try:
                print(" gpu   ", [bytes(b"ACGTN-_"[q] for q in p).decode() for p in res["paths"]], res["hp_current"].tolist(), "variant", h.walk_clock()[3], flush=True)
                print(" oracle", [bytes(b"ACGTN-_"[q] for q in p).decode() for p in ref["paths"]], ref["hp_current"].tolist(), flush=True)
            except Exception as e2:
                print(" (no detail: %r)" % (e2,), flush=True)
        sys.exit(1)
    except Exception as e:
        print("ERROR", repr(e), desc, flush=True)
        sys.exit(1)
    n_cases += 1
print("fuzz ok: %d cases + %d batched triples (%d through the window pipeline, %d of them with five-candidate positions), walker variants %s" % (n_cases, n_batch, n_pipe, n_wide, variants))
