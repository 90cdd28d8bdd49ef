#!/usr/bin/env python3
"""
bench.py -- the hot path of Gretel on MI355X, measured the way BASELINE.json asks.

One "step" = one whole pass of the hot path over one synthetic contig window whose support
table is already resident in HBM:   clear -> BAM->Hansel fill -> 100 x {path extension,
1% clamp, reweight}   (reference gretel/util.py:226-286 + gretel/cmd.py:148-179).
Workload at every N: BASELINE.json config C3, one 10k-SNP / 1M-read / k=5 (L=5) contig PER GPU
(seed = rank): independent windows, weak scaling, results gathered to rank 0 over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--paths 100]

Prints ONE JSON line (rank 0).  `value` = haplotypes/s over all ranks.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PROF_ROUND = "r6"            # the committed profiles this build's lines quote (profiles/<round>_*: profiles/collect.sh)
sys.path.insert(0, ROOT)

PROF_STRIDE = 50       # HIP-event brackets on every 50th launch of each kernel (an event pair costs the stream ~10 us)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C5"])
    ap.add_argument("--paths", type=int, default=0, help="paths per step (default: 100; C5: 1000 = BASELINE.md's deep reweight)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocking-gather", action="store_true", help="gather a step's records inside the step (a blocking collective + copies) "
                    "instead of while the next step runs")
    ap.add_argument("--no-throughput-leg", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="throughput mode: this many independent windows per GPU in one batched launch")
    ap.add_argument("--cpu-snps", type=int, default=3000, help="SNP prefix used for the Python CPU baseline sample")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (BAM + VCF files -> haplotypes)")
    ap.add_argument("--no-spec-matrix", action="store_true", help="skip the leg that times the step under every Hansel-arithmetic switch")
    ap.add_argument("--cond", default="A", choices=list("ABCDE"), help="conditional of the Hansel arithmetic (gh_config.cond_mode); default = the frozen spec")
    ap.add_argument("--marginal-term", action="store_true", help="edge weights start with log10 marginal (gh_config.marginal_term)")
    ap.add_argument("--storage", default="f32", choices=["f32", "f64"], help="storage dtype of the tensor (gh_config.storage)")
    ap.add_argument("--cpu-full", action="store_true", help="CPU baseline as BASELINE.md section 3 plans it: the Python oracle on the "
                    "whole contig (C2: every path; C3: 3 paths; minutes), instead of the bounded sample of the default run")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and send the control records through it even with ONE rank "
                         "(a one-GPU box exercises RCCL's communicator, broadcast, gather and all-reduce that way)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="control-plane backend for N > 1: nccl = RCCL over xGMI (one GPU per rank); gloo = host sockets "
                         "(lets several ranks share one GPU: --share-gpu)")
    ap.add_argument("--dump-gathered", default="", help="rank 0 writes what it gathered in the last timed step (every rank's paths and "
                    "records) to this .npz -- the tests compare it with the oracle, rank by rank")
    ap.add_argument("--share-gpu", action="store_true", help="ranks take GPU (rank mod visible GPUs) instead of one each (gloo only)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes, one per rank, BEFORE this process
    touches the GPU (a process that has initialised HIP must never be re-exec'ed or forked), and wait for them.
    Rank 0's stdout carries the JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    sys.exit(max(abs(rc) for rc in rcs))


_JSON_FD = None


def emit_line(line):
    """The one JSON line, to the stdout this process was started with (main() points descriptor 1 at stderr meanwhile)."""
    data = (line + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
        return
    while data:
        data = data[os.write(_JSON_FD, data):]


def kernel_source_sha():
    """Identity of the build a profile was taken with: a hash over the kernel sources (the GPU box has no .git)."""
    import glob
    import hashlib
    hh = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "gretel_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "include", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "*.inc")))
    files = [f for f in files if os.path.basename(f) != "gretel_io.h"]       # (the host-side BAM decoder's ABI: no kernel includes it)
    for f in files:
        hh.update(os.path.basename(f).encode())
        hh.update(open(f, "rb").read())
    return hh.hexdigest()[:16]


def seg_geometry(n, L, R):
    """gretel_amd/csrc/seg_geom.hpp: states, entries per position, positions per segment, segments, groups."""
    ns = R ** L
    g2 = max(1, min(16, (65536 if ns > 3125 else 32768) // ns))
    smax = (25 if 2048 < ns <= 3125 else 16) * g2
    seglen = max((n + smax - 1) // smax, 8)
    S = (n + seglen - 1) // seglen
    return dict(NS=ns, NI=ns // R, seglen=seglen, S=S, G1=(S + g2 - 1) // g2)


def small_window(n, L, R):
    """seg_geom.hpp, emit_small_lds_bytes <= 64 KB: every segment map fits the emitting workgroup's LDS -- k_emit_small instead of
    k_scan + k_emit (one launch less per path)."""
    g = seg_geometry(n, L, R)
    return (g["G1"] + g["S"]) * g["NS"] * 2 + 16 <= 64 * 1024


def issue_model_seg(n, L, ranked, clock_ghz, measured_ms, rw_bytes=None):
    """k_seg is bound by vector-ALU issue (binary64 adds / compares / selects for every state of every position, then one
    table lookup per state and position), not by HBM.  Floor = the instructions of its two inner loops as compiled
    (profiles/seg_isa_count.py -> profiles/r6_seg_isa.json) x trips per SIMD x issue cost / clock, for the ONE workgroup a CU
    runs (all segments run at once: <= 256 workgroups on 256 CUs, 16 waves = 4 per SIMD)."""
    isa = json.load(open(os.path.join(ROOT, "profiles", PROF_ROUND + "_seg_isa.json")))
    sec = isa.get("rwseg", {}).get("L") if rw_bytes is not None else None      # (the same two loops as compiled inside k_rwseg)
    ent = (sec or isa["L"]).get(str(L), {}).get("R4" if ranked else "R5")
    if not ent or "next_table_loop" not in ent or "state_walk_loop" not in ent or not ranked:
        return None
    R = 4
    g = seg_geometry(n, L, R)
    nj = g["NI"] // (R if L >= 4 else 1)                     # tasks (thread-iterations of the Next loop) per position
    next_trips_per_simd = g["seglen"] * nj / 64.0 / 4.0        # wave-iterations per SIMD
    dpw = 16
    spt = max(1, (g["NS"] + 1023) // 1024)                   # states per thread of the walk loop (its body holds all of them)
    walk_trips_per_simd = 4.0 * g["seglen"] / dpw              # 16 waves = 4 per SIMD, one trip per word of 16 picks
    cyc = next_trips_per_simd * ent["next_table_loop"]["issue_cycles_per_iteration"] + \
        walk_trips_per_simd * ent["state_walk_loop"]["issue_cycles_per_iteration"]
    floor_us = cyc / (clock_ghz * 1e3)
    extra = {}
    if rw_bytes is not None:
        # k_rwseg: the reweight phase in front is HBM work (band cells and table rows, read and written once) behind three
        # dependent round trips (minimum marginal, ratio, the halo's patch); its floor is its bytes at the HBM peak
        rw_floor_us = rw_bytes / (HBM_PEAK_GBS * 1e3)
        extra = {"kernel": "k_rwseg<%d>" % L, "extension_issue_floor_us": floor_us, "reweight_hbm_floor_us": rw_floor_us,
                 "reweight_bytes": rw_bytes}
        floor_us += rw_floor_us
    return {"kernel": "k_seg<%d>" % L, "binding_resource": "vector-ALU issue (binary64)",
            "states": g["NS"], "segments": g["S"], "positions_per_segment": g["seglen"],
            "next_table_loop": {k: ent["next_table_loop"][k] for k in ("instructions", "by_class", "issue_cycles_per_iteration")},
            "state_walk_loop": {k: ent["state_walk_loop"][k] for k in ("instructions", "by_class", "issue_cycles_per_iteration")},
            "instructions_per_position_and_state": nj * ent["next_table_loop"]["instructions"] / g["NS"] + ent["state_walk_loop"]["instructions"] / dpw / spt,
            "issue_cycles_per_simd": cyc, "clock_ghz": clock_ghz, "floor_us": floor_us,
            "measured_us": measured_ms * 1e3, "frac": floor_us / (measured_ms * 1e3) if measured_ms > 0 else None,
            "isa_profile": {"file": "profiles/r6_seg_isa.json", "git_head": isa.get("git_head"), "cost_cycles": isa["cost_cycles_per_wave_instruction"]},
            "note": "floor = compute phases only (Next tables + state walk); the kernel also stages its table slice (~1.5 us) and pays "
                    "launch + teardown (~3.4 us), see DESIGN.md section 4.1", **extra}


def issue_model_pools(n, L, clock_ghz, measured_ms):
    """k_cwalk: one wavefront walks 16 pool entries through its segment, a chain of dependent steps: instructions per step
    (profiles/r6_seg_isa.json, 'cwalk') x steps per segment x cycles per instruction of a wave that shares its SIMD with at
    most one other (5 alone, 2.5 with a partner: cwalk.hpp)."""
    isa = json.load(open(os.path.join(ROOT, "profiles", PROF_ROUND + "_seg_isa.json")))
    ent = isa.get("cwalk", {}).get("L", {}).get(str(L))
    if not ent:
        return None
    smax = 512 if L <= 13 else 256
    seglen = max((n + smax - 1) // smax, 32)
    cpi = 5.0
    floor_us = seglen * ent["instructions_per_step"] * cpi / (clock_ghz * 1e3)
    return {"kernel": "k_cwalk<%d, 4>" % L, "binding_resource": "instruction issue of one wavefront per 16 pool entries",
            "instructions_per_step": ent["instructions_per_step"], "steps_per_segment": seglen, "cycles_per_instruction": cpi,
            "clock_ghz": clock_ghz, "floor_us": floor_us, "measured_us": measured_ms * 1e3,
            "frac": floor_us / (measured_ms * 1e3) if measured_ms > 0 else None,
            "isa_profile": {"file": "profiles/r6_seg_isa.json", "git_head": isa.get("git_head")},
            "note": "measured = the first k_cwalk launch of a path (every pool entry is walked; later rounds walk only what is new)"}


def edge_evals_per_path(cmask, n, L):
    """conditional lookups  sum_snp S_snp*min(L,snp)  +  reweight cells N(N+3)/2+1."""
    S = np.array([bin(int(m)).count("1") for m in cmask[1:n + 1]], dtype=np.int64)
    lag = np.minimum(L, np.arange(1, n + 1))
    return int((S * lag).sum()), n * (n + 3) // 2 + 1


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_python(table, n_prefix, n_full, n_paths=1):
    """The reference's own call structure (Python loop, one NumPy-backed call per Hansel cell:
    oracle/hansel_ref.py + oracle/gretel_ref.py), single process like gretel/cmd.py:148-179.
    n_prefix == n_full: the whole contig, `n_paths` spins (C2 in full, C3 3 paths).
    n_prefix <  n_full: a bounded sample -- the first `n_prefix` SNPs of the same contig, ONE spin, extrapolated to the
    full contig by call counts (path extension ~ N, reweight = N(N+3)/2+1 calls) -- and labelled as such."""
    from oracle import gretel_ref as G
    from oracle.hansel_ref import Hansel as PyHansel, SYMBOLS, UNSYMBOLS
    k = np.diff(table.off)
    keep = np.flatnonzero(table.rank + k <= n_prefix)
    reads = []
    b = table.bases.tobytes()
    for r in keep:
        reads.append((int(table.rank[r]), b[table.off[r]:table.off[r + 1]].decode()))
    h = PyHansel.init_matrix(SYMBOLS, UNSYMBOLS, n_prefix, band=max(1, int(k.max()) - 1))
    t0 = time.perf_counter()
    G.fill_from_support(h, reads, n_prefix)
    t_fill = time.perf_counter() - t0
    orig = h.copy()
    t_gen = t_rw = 0.0
    done = 0
    for _ in range(n_paths):
        t0 = time.perf_counter()
        path, prob, mn = G.generate_path(n_prefix, h, orig)
        t_gen += time.perf_counter() - t0
        if path is None:
            break
        t0 = time.perf_counter()
        G.reweight_hansel_from_path(h, path, max(mn, 0.01))
        t_rw += time.perf_counter() - t0
        done += 1
    done = max(done, 1)
    if n_prefix >= n_full:
        return dict(value=done / (t_gen + t_rw), unit="haplotypes/s", cores=1, kind="port", cpu=cpu_model(),
                    sample="Python oracle (reference call structure), the WHOLE contig (N=%d, %d reads), %d spins: "
                           "fill %.2fs, generate_path %.2fs, reweight %.2fs; no extrapolation" % (n_full, len(reads), done, t_fill, t_gen, t_rw))
    calls_s = n_prefix * (n_prefix + 3) // 2 + 1
    calls_f = n_full * (n_full + 3) // 2 + 1
    t_path_full = (t_gen / done) * (n_full / n_prefix) + (t_rw / done) * (calls_f / calls_s)
    return dict(value=1.0 / t_path_full, unit="haplotypes/s", cores=1, kind="port", cpu=cpu_model(),
                sample="EXTRAPOLATED: Python oracle (reference call structure), SNP prefix %d of the contig, %d reads, %d spin(s): "
                       "fill %.2fs, generate_path %.2fs, reweight %.2fs; per-path time scaled to N=%d by call "
                       "counts (x%.1f extension, x%.1f reweight); `bench.py --cpu-full` times the whole contig instead"
                       % (n_prefix, len(reads), done, t_fill, t_gen, t_rw, n_full, n_full / n_prefix, calls_f / calls_s))


def cpu_baseline_c(table, paths=3):
    """The compiled scalar port (oracle/c/gretel_oracle.c), full contig, reference pair
    enumeration (all N(N+3)/2+1 reweight calls per path), a few paths."""
    from oracle.c_oracle import COracle
    o = COracle(table.n_snps, table.band, use_libm=True)
    t0 = time.perf_counter()
    o.fill(table)
    t_fill = time.perf_counter() - t0
    o.set_full_enum(1)
    t0 = time.perf_counter()
    r = o.spin(paths)
    t_spin = time.perf_counter() - t0
    return dict(value=r["n"] / t_spin, unit="haplotypes/s", cores=1, kind="port", cpu=cpu_model(),
                sample="C oracle, whole contig, %d spins with the reference's full pair enumeration: fill %.2fs, spins %.2fs"
                       % (r["n"], t_fill, t_spin))


def end_to_end_leg(table, paths, local):
    """Secondary figure (never `value`): the same contig from FILES, as the reference's CLI takes it --
    bgzipped VCF -> SNP positions, indexed BAM -> native streaming decode (libgretel_io.so) -> support table -> PCIe
    -> GPU fill -> spins -> paths back on the host.  The files are written (untimed) from the synthetic table."""
    import shutil
    import tempfile
    from gretel_amd import bamio, util
    d = tempfile.mkdtemp(prefix="gretel_e2e_")
    try:
        t0 = time.perf_counter()
        bam, vcf = os.path.join(d, "s.bam"), os.path.join(d, "s.vcf.gz")
        contig, start, end = bamio.synth_to_files(table, bam, vcf)
        t_write = time.perf_counter() - t0
        best = None
        for _ in range(2):                                   # second pass: page cache warm, HIP warm
            t0 = time.perf_counter()
            util.prefetch_bam(bam, contig, start, end)      # (gretel_amd/cmd.py: the BAM's blocks are inflated while the VCF is parsed)
            v = util.process_vcf(vcf, contig, start, end)
            t_vcf = time.perf_counter() - t0
            # (gretel_amd.util.load_from_bam as gretel_amd/cmd.py calls it: native decode into page-locked memory, upload, GPU fill)
            t1 = time.perf_counter()
            with contextlib.redirect_stderr(io.StringIO()):              # (its two [NOTE] lines, gretel/util.py:331-334)
                h = util.load_from_bam(bam, contig, start, end, v, device=local)
            t_load = time.perf_counter() - t1
            st = bamio.native_last_stats()
            t_dec = float(st["seconds"])
            t_fill = t_load - t_dec
            t3 = time.perf_counter()
            res = h.spin(paths)
            t_spin = time.perf_counter() - t3
            wall = time.perf_counter() - t0
            cur = dict(wall_s=wall, vcf_s=t_vcf, bam_decode_s=t_dec, upload_and_fill_s=t_fill, spins_s=t_spin,
                       haplotypes=int(res["n"]), haplotypes_per_s=res["n"] / wall)
            if best is None or cur["wall_s"] < best["wall_s"]:
                best = cur
            n_reads_decoded = int(st["reads_kept"])
            del h, res
            import gc
            gc.collect()
        best.update(bam_bytes=os.path.getsize(bam), reads=n_reads_decoded, decoder=st,
                    files_written_s_untimed=t_write,
                    note="BAM (+ .bai) and bgzipped VCF of the same contig -> gretel_amd.util.process_vcf + gretel_amd.util.load_from_bam "
                         "(native BAM decode, include/gretel_io.h, + upload + GPU fill) + %d spins, as gretel_amd/cmd.py runs them; "
                         "bam_decode_s is the decoder's own clock, upload_and_fill_s the rest of load_from_bam; best of 2 passes" % paths)
        return best
    finally:
        shutil.rmtree(d, ignore_errors=True)


def batch_roofline(batch, bp, paths, n_snps, band, L, es, src_sha, pmc_file):
    """Roofline object of a batched spin.  When the window pipeline carried it (gretel_amd/csrc/wpipe.hpp: one launch, every
    window through all its paths) the kernel is k_wpipe and the bytes are SURVEY section 8(d)'s per path -- the conditional
    lookups of the extension N (1 + L) 196 + 28 N and the reweight's 8 N W -- times paths x windows; `builders_bytes` is the
    wider definition of DESIGN.md section 3 (+ marginal cells, table rows, records), what gh_batch_profile_get reports."""
    info = batch.pipe_info()
    wk, rw = bp["walk"], bp["reweight"]
    cell = 49.0 * es
    if info["windows"] > 0 and wk["launches"] > 0:
        ms = wk["ms"] / wk["launches"]
        strict = (n_snps * ((1.0 + L) * cell + 28.0) + 2.0 * es * n_snps * band) * paths * wk["windows"]
        traffic, note = None, "no PMC profile for the kernels of this build"
        try:
            pmb = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            if (n_snps, L) != (10000, 5):
                note = "profiles/%s is the C3 window's (10k SNPs, five lags): not quoted for this one" % pmc_file
            elif pmb.get("kernel_source_sha") == src_sha:
                k = next(v for kk, v in pmb["kernels"].items() if kk.startswith("k_wpipe"))
                traffic = k["hbm_bytes_per_launch_corrected"] * (paths * wk["windows"]) / float(pmb["paths"] * pmb["windows"])
                if (pmb["paths"], pmb["windows"]) == (paths, wk["windows"]):
                    note = "profiles/%s (git %s): the counters of a launch of this very shape, %d windows x %d paths" % (
                        pmc_file, pmb.get("git_head"), pmb["windows"], pmb["paths"])
                else:
                    # (k_wpipe's prologue -- G repacked into gp, pk / lmr packed: ~18 MB per window -- is paid once per launch, not
                    # per path: scaled linearly from a launch with fewer paths it is counted too often)
                    note = ("profiles/%s (git %s): %d windows x %d paths, scaled linearly to this launch -- the per-launch prologue (table "
                            "repack, ~18 MB per window) is scaled with the paths: overstated by a few per cent when the profile has fewer paths"
                            % (pmc_file, pmb.get("git_head"), pmb["windows"], pmb["paths"]))
            else:
                note = "profiles/%s was taken with kernel sources %s, this build is %s: not quoted" % (pmc_file, pmb.get("kernel_source_sha"), src_sha)
        except Exception as exc:
            note = "no usable PMC profile: %r" % (exc,)
        return {"bound": "hbm", "kernel": "k_wpipe (window pipeline: walk of path s + reweight sweep of path s-1, all paths of every window in one launch)",
                "windows_per_launch": wk["windows"], "paths_per_window": paths, "pipeline": info,
                "avg_launch_ms_hip_events": ms, "algorithmic_bytes_per_launch": strict,
                "achieved": strict / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": strict / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "builders_bytes_per_launch": wk["bytes_per_launch"], "builders_frac": wk["bytes_per_launch"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": note, "kernel_source_sha": src_sha,
                "traffic_over_algorithmic": (traffic / strict) if traffic else None,
                "traffic_rate": (traffic / (ms * 1e-3) / 1e9) if traffic else None,
                "traffic_rate_frac_of_peak": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "note": "frac prices SURVEY section 8(d)'s bytes (extension lookups + reweight cells) per path and window against 8 TB/s; "
                        "traffic_rate is what the counters saw (profiles/, scaled to this launch) over this launch's duration: with one workgroup "
                        "per CU the kernel moves 4.4-5.4 TB/s of actual traffic depending on the box, and the walker's chain of dependent "
                        "steps (150 cycles per position) is the other bound, a few per cent behind: DESIGN.md section 4.4"}
    rw_ms = rw["ms"] / max(1, rw["launches"])
    return {"bound": "hbm", "kernel": "k_marg<float, true> (batched fused reweight + marginals + table rows)", "pipeline": info,
            "windows_per_launch": rw["windows"], "avg_launch_ms_hip_events": rw_ms,
            "algorithmic_bytes_per_launch": rw["bytes_per_launch"],
            "achieved": (rw["bytes_per_launch"] / (rw_ms * 1e-3) / 1e9) if rw_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (rw["bytes_per_launch"] / (rw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rw_ms > 0 else None,
            "traffic": None, "kernel_source_sha": src_sha,
            "note": "window groups on their own streams overlap: a bracket around one group's launch also waits for the others' kernels"}


def bench_batch(args, cfg_name, paths, desc, rank, world, local, dev):
    """Throughput mode: B independent windows per GPU recovered by one batched launch per
    kernel.  Distinct synthetic contigs are expensive to generate on the host, so min(B, 8) seeds are
    generated and reused cyclically; every window still owns its tensor, tables and results."""
    import torch
    import torch.distributed as dist
    from gretel_amd.hansel import Hansel, HanselBatch, DeviceReads
    from gretel_amd.synth import make_config
    B = args.batch
    n_tab = min(B, 8)
    tables = [make_config(cfg_name, seed=rank * 1000 + q) for q in range(n_tab)]
    hs = [Hansel(tables[0].n_snps, band=tables[0].band, device=local) for _ in range(B)]
    reads = [DeviceReads(hs[q], tables[q].rank, tables[q].off, tables[q].bases) for q in range(n_tab)]
    batch = HanselBatch(hs)

    def step():
        for w, h in enumerate(hs):
            h.clear()
            h.fill_from_support(None, None, None, reads_handle=reads[w % n_tab])
        return batch.spin(paths, copy=False)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(desc["warmup"]):
        step()
    batch.profile_enable(10)
    fence()
    t0 = time.perf_counter()
    n_paths = 0
    t_fill = 0.0
    for _ in range(desc["steps"]):
        res = step()
        n_paths += sum(r["n"] for r in res)
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(n_paths)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, n_paths = float(tmax[0]), float(tsum[1])
    if rank == 0:
        t = tables[0]
        bp = batch.profile_get()
        emit_line(json.dumps({
            "roofline": batch_roofline(batch, bp, paths, t.n_snps, t.band, hs[0].L, 4.0, kernel_source_sha(), PROF_ROUND + "_pmc_traffic_batch256.json"),
            "metric": "haplotypes/sec, batched windows (throughput mode)", "value": n_paths / dt, "unit": "haplotypes/s",
            "n_gpus": world, "steps": desc["steps"], "warmup": desc["warmup"], "ms_per_step": dt / desc["steps"] * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 counts / f64 log-likelihoods",
            "data": "synthetic",
            "config": {"workload": "%d x %s windows per GPU (%d distinct seeds reused cyclically), %d-SNP / %d-read, L=%d, %d paths each; "
                                   "fill per window + one batched spin" % (B, cfg_name, n_tab, t.n_snps, t.n_reads, hs[0].L, paths),
                       "windows_per_gpu": B, "n_snps": t.n_snps, "n_reads": t.n_reads, "L": hs[0].L, "paths": paths}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                   # never returns
    args.gpus = world
    # stdout carries ONE line, the JSON: whatever a library writes to file descriptor 1 on its own (RCCL prints a version banner
    # there when its first communicator comes up, through C stdio, so it would land BEHIND the line) goes to stderr instead
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()        # does not initialise HIP
    if ndev < 1:
        sys.exit("bench.py needs a GPU (no CPU fallback for the hot path)")
    if args.share_gpu:
        if args.backend != "gloo":
            sys.exit("--share-gpu needs --backend gloo (RCCL wants one GPU per rank)")
        local = local % ndev
    elif local >= ndev:
        sys.exit("rank %d wants GPU %d but only %d are visible (use --backend gloo --share-gpu to share)" % (rank, local, ndev))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback for the hot path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")     # where the control records live
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from gretel_amd.hansel import Hansel, DeviceReads
    from gretel_amd.synth import make_config
    from gretel_amd.dist import broadcast_descriptor, gather_results, ResultExchange

    # rank 0 decides the run, everybody learns it over RCCL
    if args.paths <= 0:
        args.paths = 1000 if args.config == "C5" else 100
    desc = broadcast_descriptor(dict(paths=args.paths, steps=args.steps, warmup=args.warmup,
                                     config={"C2": 2, "C3": 3, "C5": 5}[args.config]), comm_dev, world, rank, force=use_dist)
    cfg_name = "C%d" % desc["config"]
    paths = desc["paths"]

    if args.batch > 0:
        return bench_batch(args, cfg_name, paths, desc, rank, world, local, comm_dev)

    table = make_config(cfg_name, seed=rank)               # one independent window per GPU
    spec_kw = dict(cond_mode=args.cond, marginal_term=args.marginal_term, storage=args.storage)
    h = Hansel(table.n_snps, band=table.band, device=local, **spec_kw)
    reads = DeviceReads(h, table.rank, table.off, table.bases)   # inputs resident in HBM before timing

    # The records of a step go to rank 0 while the next step runs (gretel_amd.dist.ResultExchange: spin writes into a pinned
    # slot, upload + gather + download ride on a side stream); --blocking-gather keeps the collective inside the step.
    ex = None if args.blocking_gather else ResultExchange(table.n_snps, paths, comm_dev, world, rank, force=use_dist)

    def step():
        h.clear()
        stats = h.fill_from_support(None, None, None, reads_handle=reads)
        if ex is None:
            res = h.spin(paths)
            gathered = gather_results(res, table.n_snps, paths, comm_dev, world, rank, force=use_dist, copy=False)
            return stats, res, gathered
        pv, rv = ex.buffers()
        res = h.spin(paths, out_paths=pv, out_recs=rv)
        ex.submit(res["n"], res["hole_at"])
        # what rank 0 looks at is the step before this one (its records have had a whole step to arrive)
        gathered = ex.collect() if len(ex.queue) > 1 else None
        return stats, res, gathered

    drained = {}

    def fence():
        if ex is not None:
            last = ex.drain()                           # (inside the timed region: the last step's records are on rank 0)
            if last is not None:
                drained["last"] = last
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        h.sync()

    for _ in range(desc["warmup"]):
        step()
    # HIP-event brackets on the handle's stream, on every PROF_STRIDE-th launch of each kernel: an event between
    # two kernels costs the stream a ~10 us bubble, so bracketing every launch would slow the timed region by 4 %
    h.profile_enable(PROF_STRIDE)
    h.profile_reset()
    fence()
    t0 = time.perf_counter()
    n_paths_local = 0
    for _ in range(desc["steps"]):
        stats, res, gathered = step()
        n_paths_local += res["n"]
    fence()
    dt = time.perf_counter() - t0
    if ex is not None:
        gathered = drained.get("last")                  # the last timed step's records, every rank's (rank 0)
    prof = h.profile_get()
    h.profile_enable(0)
    ev_empty_ms, ev_nop_ms = h.profile_overhead(30)      # what a bracket reads with nothing / an empty kernel inside

    if ex is not None:
        ex.close()                                      # (the worker thread: no collective of its own may run beside the ones below)
    tt = torch.tensor([dt, float(n_paths_local)], dtype=torch.float64, device=comm_dev)
    if use_dist:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, n_paths_total = float(tmax[0]), float(tsum[1])
    else:
        dt_max, n_paths_total = dt, float(n_paths_local)

    if rank == 0 and args.dump_gathered:
        np.savez(args.dump_gathered, world=world, **{"%s_%d" % (k, r): np.asarray(g[k]) for r, g in enumerate(gathered)
                                                      for k in ("n", "hole_at", "paths", "hp_current", "hp_original", "ratio", "magnitude")})
    if rank == 0:
        n, L = table.n_snps, h.L
        cyc, ticks, nsteps, variant = h.walk_clock()
        h.clear()
        h.fill_from_support(None, None, None, reads_handle=reads)
        cond_evals, rw_cells = edge_evals_per_path(h.candidate_masks(), n, L)
        hap_s = n_paths_total / dt_max
        walk = prof["walk"]
        walk_ms = walk["ms"] / max(1, walk["launches"])
        seg = prof.get("seg", {"ms": 0.0, "launches": 0, "bytes_per_launch": 0.0})
        segwalk = variant == 3                     # segment-parallel extension: k_seg + k_scan + k_emit per path
        rwseg = prof.get("rwseg", {"ms": 0.0, "launches": 0, "bytes_per_launch": 0.0})
        fused_rw = segwalk and rwseg["launches"] > 0   # k_rwseg: the reweight of path k-1 rides in the k_seg launch of path k
        if fused_rw:
            dom, dom_name = rwseg, ("k_rwseg (reweight of the path before + segment-parallel extension of this one: band cells, "
                                    "table rows, Next tables + all entry states of every segment)")
            dom_bytes_def = ("SURVEY 8(d): conditional lookups of the extension N*L*196 + the reweight of a path "
                             "(N+1)*(2*W*esz + 196*esz + 217) + the table rows it rewrites N*(7*min(W,L)*esz + L*5*6*8)")
        elif segwalk and seg["launches"]:
            dom, dom_name = seg, "k_seg (segment-parallel path extension: Next tables + all entry states of every segment)"
            dom_bytes_def = "SURVEY 8(d), conditional lookups of the extension: N*L*196 per path (the marginal cell and the original marginals, N*224, belong to k_emit)"
        elif variant == 4:
            dom, dom_name = walk, "candidate-pool path extension (per path: rounds of k_cwalk + k_clink + k_cscan, then k_cemit; bracketed as one unit)"
            dom_bytes_def = "SURVEY 8(d) path extension: N*((1+L)*196+28) per path"
        else:
            dom, dom_name = walk, "k_walk_spec (path extension: N dependent steps, one wavefront walks)"
            dom_bytes_def = "SURVEY 8(d) path extension: N*((1+L)*196+28) per path"
        dom_ms = dom["ms"] / max(1, dom["launches"])
        achieved = dom["bytes_per_launch"] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/, collected per the MI355X guide: separate
        # --pmc runs, FETCH_SIZE x2 on gfx950).  A profile is only quoted for the build it was taken with (kernel_source_sha);
        # otherwise null, with the reason.
        traffic, traffic_note = None, "no PMC profile for this config"
        src_sha = kernel_source_sha()
        try:
            pmf = PROF_ROUND + ("_pmc_traffic_c5.json" if cfg_name == "C5" else "_pmc_traffic.json")
            pmj = json.load(open(os.path.join(ROOT, "profiles", pmf)))
            if cfg_name in ("C3", "C5") and spec_kw == dict(cond_mode="A", marginal_term=False, storage="f32"):
                if pmj.get("kernel_source_sha") != src_sha:
                    traffic_note = "profiles/%s was taken with kernel sources %s (git %s), this build is %s: not quoted" % (
                        pmf, pmj.get("kernel_source_sha"), pmj.get("git_head"), src_sha)
                else:
                    want = "k_rwseg" if fused_rw else "k_seg" if segwalk else ("k_cwalk" if variant == 4 else "k_walk")
                    traffic = next(v["hbm_bytes_per_launch_corrected"] for k, v in pmj["kernels"].items() if k.startswith(want))
                    traffic_note = "profiles/%s (git %s, kernel sources %s)" % (pmf, pmj.get("git_head"), src_sha)
        except Exception as exc:
            traffic, traffic_note = None, "no usable PMC profile: %r" % (exc,)
        from gretel_amd._lib import device_clock_khz
        clock_ghz = device_clock_khz(local) / 1e6
        # what actually binds the dominant kernel: instruction issue, priced from the ISA (profiles/r6_seg_isa.json)
        issue_model = None
        try:
            if segwalk and (seg["launches"] or fused_rw):
                ranked = not bool((h.candidate_masks()[1:] == 0x2F).any())
                issue_model = issue_model_seg(n, L, ranked, clock_ghz, max(dom_ms - ev_nop_ms, 1e-6),
                                              rw_bytes=(rwseg["bytes_per_launch"] - n * L * 49.0 * (8 if spec_kw["storage"] == "f64" else 4))
                                              if fused_rw else None)
            elif variant == 4 and seg["launches"]:
                issue_model = issue_model_pools(n, L, clock_ghz, max(seg["ms"] / seg["launches"] - ev_nop_ms, 1e-6))
            elif variant == 2 and nsteps:
                isa = json.load(open(os.path.join(ROOT, "profiles", "r1_walker_isa.json")))
                ent = isa["L"].get(str(L))
                if ent:
                    floor = ent["issue_floor_cycles_per_step"]
                    issue_model = {"instructions_per_step": ent["instructions_per_step"],
                                   "cycles_per_instruction_lone_wave": isa["cycles_per_instruction_lone_wave"],
                                   "floor_cycles_per_step": floor, "measured_cycles_per_step": cyc / nsteps,
                                   "frac": floor / (cyc / nsteps)}
        except Exception as exc:
            issue_model = {"error": repr(exc)}
        # the whole step against HBM -- (B_fill + P (B_ext + B_rw)) / wall / peak
        n_adds = stats[1] + 2 * 0          # crumbs; the sentinel cases add a second observation to a handful of reads
        b_fill = 8.0 * stats[1] + float(table.n_reads) * 8.0 + float(len(table.bases))
        b_ext = n * (1 + L) * 196.0 + n * 28.0
        b_rw = 8.0 * n * min(table.band, n)
        step_bytes = b_fill + paths * (b_ext + b_rw)
        step_s = dt_max / desc["steps"]
        ranked_w = not bool((h.candidate_masks()[1:] == 0x2F).any())
        small_w = segwalk and small_window(n, L, 4 if ranked_w else 5) and os.environ.get("GH_EMIT_SMALL", "1") != "0"
        kernels_per_path = ((3 if fused_rw else 4) - (1 if small_w else 0)) if segwalk else (None if variant == 4 else 2)
        out = {
            "metric": "haplotypes/sec + SNP-edge-evals/sec on 10k-SNP synthetic contig",
            "value": hap_s,
            "unit": "haplotypes/s",
            "n_gpus": world,
            "steps": desc["steps"],
            "warmup": desc["warmup"],
            "ms_per_step": dt_max / desc["steps"] * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 counts / f64 log-likelihoods",
            "data": "synthetic",
            "config": {"workload": "%s: %d-SNP / %d-read synthetic contig per GPU, k=%s SNPs/read, L=%d, %d paths per step "
                                   "(fill + spins), seed=rank" % (cfg_name, n, table.n_reads, table.max_k, L, paths),
                       "n_snps": n, "n_reads": table.n_reads, "L": L, "band": table.band, "paths": paths,
                       "parallelism": "%d independent window(s), one per rank; %s broadcast/gather of control records, no data-path collective"
                                      % (world, "RCCL" if args.backend == "nccl" else "gloo"),
                       "backend": args.backend, "share_gpu": bool(args.share_gpu), "control_records_through_torch_distributed": bool(use_dist),
                       "gather": "blocking, inside the step (--blocking-gather)" if ex is None else
                                 "overlapped: a step's records reach rank 0 on a side stream while the next step runs (ResultExchange); the last step's inside the timed region",
                       "hansel_spec": dict(spec_kw, cand_order="ACGT-", offer_zero=False)},
            "edge_evals_per_s": hap_s * (cond_evals + rw_cells),
            "edge_evals_per_path": {"conditionals": cond_evals, "reweight_cells": rw_cells},
            "fill": {"n_slices": stats[0], "n_crumbs": stats[1],
                     "crumbs_per_s": stats[1] / (prof["fill"]["ms"] / max(1, prof["fill"]["launches"]) * 1e-3) if prof["fill"]["ms"] else None},
            "kernels_ms_per_launch": {k: (v["ms"] / v["launches"] if v["launches"] else None) for k, v in prof.items()},
            "kernels_launches": {k: v["launches"] for k, v in prof.items()},
            "roofline": {"bound": "hbm", "kernel": dom_name,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "binding_resource": (issue_model or {}).get("binding_resource", "instruction issue / launch latency (see note)"),
                         "step_frac": step_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                         "step_bytes": {"definition": "SURVEY 8(d): B_fill + P (B_ext + B_rw); B_fill = 8 x adds + sum(8 + k), "
                                                      "B_ext = N (1 + L) 196 + 28 N, B_rw = 8 N W",
                                        "fill": b_fill, "extension_per_path": b_ext, "reweight_per_path": b_rw, "step": step_bytes},
                         "launch_floor_us": {"empty_kernel_between_two_events": (ev_nop_ms - ev_empty_ms) * 1e3,
                                             "dependent_launches_per_path": kernels_per_path,
                                             "per_path": (kernels_per_path * (ev_nop_ms - ev_empty_ms) * 1e3) if kernels_per_path else None,
                                             "note": "what the dependent launches of one path cost before any of them does work (HIP events "
                                                     "around an empty kernel, minus the bracket itself); rocprofv3 reads 2.8 us for the same empty kernel"},
                         "kernel_source_sha": src_sha,
                         "algorithmic_bytes_per_launch": dom["bytes_per_launch"],
                         "algorithmic_bytes_definition": dom_bytes_def,
                         "avg_launch_ms_hip_events": dom_ms,
                         "hip_event_sampling": "every %d-th launch of each kernel inside the timed region (%d launches sampled)"
                                               % (PROF_STRIDE, dom["launches"]),
                         "hip_event_bracket_overhead_ms": {"two_events_back_to_back": ev_empty_ms, "around_an_empty_kernel": ev_nop_ms,
                                                           "note": "included in avg_launch_ms_hip_events and therefore in `achieved` (conservative): "
                                                                   "rocprofv3's average for the same kernel (profiles/) lies between the raw and the net reading"},
                         "extension_ms_per_path_hip_events": walk_ms,
                         "walker_variant": {4: "candidate-pool segments (k_cwalk/k_clink/k_cscan rounds + k_cemit)",
                                            3: ("segment-parallel (k_rwseg: reweight of the path before + k_seg; k_scan + k_emit, or k_emit_small alone in small windows)"
                                                if fused_rw else "segment-parallel (k_seg + k_scan + k_emit)"), 2: "serial, depth-2 speculation",
                                            1: "serial, depth-1 speculation, no '-' candidates", 0: "serial, depth-1 speculation"}.get(variant),
                         "walker_cycles_per_step": (cyc / nsteps) if (nsteps and variant <= 2) else None,
                         "issue_model": issue_model,
                         "note": ("k_seg evaluates Next[t][state] for all R^L states of every position (binary64 adds and compares: "
                                  "vector-ALU bound, then LDS-latency bound in the state walk), so its HBM traffic stays far below the "
                                  "bandwidth roofline by construction; a path is %s "
                                  "about 4 us of each is launch + first-touch latency; see DESIGN.md section 4"
                                  % (("2 dependent kernels (k_rwseg: the reweight of the path before + k_seg; k_emit_small) of 8-10 us,"
                                      if small_w else "3 dependent kernels (k_rwseg: the reweight of the path before + k_seg; k_scan; k_emit) of 5-21 us,")
                                     if fused_rw else "4 dependent kernels (k_seg, k_scan, k_emit, k_rw) of 5-12 us each,"))
                                 if segwalk else
                                 ("each segment of the window is walked from a pool of candidate entry states by one wavefront per 16 "
                                  "candidates (196 dependent steps per segment at C5), the chain of segments is verified exactly; "
                                  "bound by one wavefront's issue rate per step, not by HBM; see DESIGN.md section 4.2")
                                 if variant == 4 else
                                 ("not bandwidth bound: each step needs the previous step's arg-max (gretel.py:143-187), so one "
                                  "wavefront walks and its instruction issue rate (1 per 5 cycles) is the bound")},
        }
        if world == 1 and not args.no_throughput_leg:
            # secondary figure (not `value`): the same contig replicated into 32 windows and recovered by ONE batched
            # launch per kernel (gh_batch_*): what the chip does when it is given windows enough to fill it
            try:
                from gretel_amd.hansel import HanselBatch
                reps = 32
                hs = [Hansel(n, band=table.band, device=local) for _ in range(reps)]
                for hh in hs:
                    hh.fill_from_support(None, None, None, reads_handle=reads)
                hb = HanselBatch(hs)
                torch.cuda.synchronize()
                tb = time.perf_counter()
                rb = hb.spin(paths)
                tb = time.perf_counter() - tb
                out["throughput_mode"] = {"windows": reps, "value": sum(r["n"] for r in rb) / tb, "unit": "haplotypes/s",
                                          "note": "32 replicas of the benchmark contig, one batched spin of %d paths each "
                                                  "(fill not included); bench.py --batch B times distinct windows incl. fill" % paths}
                del hb, hs
            except Exception as exc:       # never let the secondary leg break the contract line
                out["throughput_mode"] = {"error": repr(exc)}
        if world == 1 and not args.no_throughput_leg and n <= 20000:
            # secondary figure: 256 replicas of the contig recovered as ONE batch (gh_batch_*: the window pipeline of csrc/wpipe.hpp
            # where it applies -- one persistent workgroup per window, one launch --, else one batched launch per kernel and path):
            # what `bench.py --batch 256` times with distinct windows.  The first call also allocates (page-locked result buffers,
            # the pipeline's packed words); the figure is the median of three calls behind it, each on freshly filled tensors.
            try:
                from gretel_amd.hansel import HanselBatch
                reps = 256
                hs = [Hansel(n, band=table.band, device=local) for _ in range(reps)]
                hb = HanselBatch(hs)
                hb.profile_enable(10)
                tbs, rb = [], None
                for it in range(4):
                    for hh in hs:
                        hh.clear()
                        hh.fill_from_support(None, None, None, reads_handle=reads)
                    torch.cuda.synchronize()
                    tb = time.perf_counter()
                    rb = hb.spin(paths, copy=False)
                    tbs.append(time.perf_counter() - tb)
                first_call, tb = tbs[0], sorted(tbs[1:])[1]
                nb = sum(r["n"] for r in rb)
                same = all(np.array_equal(r["paths"], rb[0]["paths"]) for r in rb)
                out["throughput_mode_256"] = {"windows": reps, "value": nb / tb, "unit": "haplotypes/s",
                                              "first_call_value": nb / first_call, "calls_s": tbs,
                                              "all_windows_recover_the_same_paths": bool(same),
                                              "note": "256 replicas of the benchmark contig, one batched spin of %d paths each (fill not included; "
                                                      "results copied back to the host -- page-locked buffers -- included); median of 3 calls "
                                                      "behind a first one that also allocates" % paths,
                                              "roofline": batch_roofline(hb, hb.profile_get(), paths, n, table.band, L, 4.0, src_sha, PROF_ROUND + "_pmc_traffic_batch256.json")}
                del hb, hs
                # ... and the same batch under the spec the published method describes (conditional E + marginal term, f32; see
                # value_published_spec): the pipeline's sweep then works on the to-major copy of the band as well
                hs = [Hansel(n, band=table.band, device=local, cond_mode="E", marginal_term=True) for _ in range(reps)]
                hb = HanselBatch(hs)
                tps = []
                for it in range(3):
                    for hh in hs:
                        hh.clear()
                        hh.fill_from_support(None, None, None, reads_handle=reads)
                    torch.cuda.synchronize()
                    tb = time.perf_counter()
                    rbp = hb.spin(paths, copy=False)
                    tps.append(time.perf_counter() - tb)
                out["throughput_mode_256"]["published_spec"] = {"value": sum(r["n"] for r in rbp) / min(tps[1:]), "unit": "haplotypes/s",
                                                                "spec": "cond_mode E + marginal_term, f32", "calls_s": tps, "pipeline": hb.pipe_info(),
                                                                "note": "the faster of two calls behind a first one that also allocates"}
                del hb, hs
                # ... and 256 replicas of the SPARSE-DELETION window (wide_window_sparse below: '-' at 1 % of the positions, five
                # candidates there): round 6 -- the pipeline's WIDE launch carries them (a side table for the fifth candidates, the
                # speculative walker between them, an exact stepper around them: csrc/wpipe.hpp); until round 5 one such position
                # sent a window to the batched launches
                import copy as _copy
                from gretel_amd.synth import sprinkle_deletions
                tsd = _copy.copy(table)
                tsd.bases = table.bases.copy()
                sprinkle_deletions(tsd, 0.01, seed=4321)
                hs = [Hansel(n, band=table.band, device=local) for _ in range(reps)]
                rsd = DeviceReads(hs[0], tsd.rank, tsd.off, tsd.bases)
                hb = HanselBatch(hs)
                tds, rbd = [], None
                for it in range(3):
                    for hh in hs:
                        hh.clear()
                        hh.fill_from_support(None, None, None, reads_handle=rsd)
                    torch.cuda.synchronize()
                    tb = time.perf_counter()
                    rbd = hb.spin(paths, copy=False)
                    tds.append(time.perf_counter() - tb)
                vsd = sum(r["n"] for r in rbd) / min(tds[1:])
                out["throughput_mode_256"]["sparse_deletions"] = {
                    "value": vsd, "unit": "haplotypes/s", "over_narrow": vsd / out["throughput_mode_256"]["value"], "calls_s": tds,
                    "pipeline": hb.pipe_info(), "positions_with_5_candidates": int((hs[0].candidate_masks()[1:] == 0x2F).sum()),
                    "note": "256 replicas of the sparse-deletion window ('-' on 30 % of the reads at 1 % of the positions), default spec; the "
                            "faster of two calls behind a first one that also allocates"}
                del hb, hs, rsd
            except Exception as exc:
                out["throughput_mode_256"] = dict(out.get("throughput_mode_256", {}), error=repr(exc))
        if world == 1 and not args.no_throughput_leg and L <= 5:
            # secondary figure: the same contig with 5 % of the bases read as deletions -> positions with FIVE candidates
            # (A C G T -): the window is not "narrow", the state space of the segment-parallel extension is 5^L, not 4^L
            try:
                import copy
                tw = copy.copy(table)
                bw = table.bases.copy()
                bw[np.random.default_rng(12345).random(len(bw)) < 0.05] = ord('-')
                tw.bases = bw
                hw = Hansel(n, band=table.band, device=local)
                rw = DeviceReads(hw, tw.rank, tw.off, tw.bases)
                hw.fill_from_support(None, None, None, reads_handle=rw)
                hw.spin(10)
                wide = int((hw.candidate_masks()[1:] == 0x2F).sum())
                hw.clear()
                hw.fill_from_support(None, None, None, reads_handle=rw)
                torch.cuda.synchronize()
                t0w = time.perf_counter()
                rwres = hw.spin(paths)
                dtw = time.perf_counter() - t0w
                out["wide_window"] = {"value": rwres["n"] / dtw, "unit": "haplotypes/s", "ms_per_path": dtw / max(1, rwres["n"]) * 1e3,
                                      "positions_with_5_candidates": wide, "walker_variant": hw.walk_clock()[3],
                                      "note": "same contig, 5 %% of the bases replaced by '-': %d of %d positions show A, C, G, T and '-'; "
                                              "one spin of %d paths (fill not included)" % (wide, n, paths)}
                del hw, rw
            except Exception as exc:
                out["wide_window"] = {"error": repr(exc)}
            # ... and with deletions at 1 % of the POSITIONS (a deletion column here and there: what a real pileup shows): the
            # mixed-radix state space (segmix.hpp) -- a few hundred states more behind each such column instead of 5^L everywhere
            try:
                import copy
                from gretel_amd.synth import sprinkle_deletions
                ts_ = copy.copy(table)
                ts_.bases = table.bases.copy()
                sprinkle_deletions(ts_, 0.01, seed=4321)
                hw = Hansel(n, band=table.band, device=local)
                rw = DeviceReads(hw, ts_.rank, ts_.off, ts_.bases)
                hw.fill_from_support(None, None, None, reads_handle=rw)
                hw.spin(10)
                wide = int((hw.candidate_masks()[1:] == 0x2F).sum())
                hw.clear()
                hw.fill_from_support(None, None, None, reads_handle=rw)
                torch.cuda.synchronize()
                t0w = time.perf_counter()
                rwres = hw.spin(paths)
                dtw = time.perf_counter() - t0w
                clk = hw.walk_clock()
                # ... and the same spin over again on the handle (clear + fill untimed in between): the first spin of a handle also
                # sizes and allocates what the mixed-radix flow needs (segment buffers, the halo); `value` above is that cold spin
                tws = []
                for _ in range(3):
                    hw.clear()
                    hw.fill_from_support(None, None, None, reads_handle=rw)
                    torch.cuda.synchronize()
                    t0w = time.perf_counter()
                    nw_ = hw.spin(paths)["n"]
                    tws.append((time.perf_counter() - t0w) / max(1, nw_))
                tws.sort()
                out["wide_window_sparse"] = {"value": rwres["n"] / dtw, "unit": "haplotypes/s", "ms_per_path": dtw / max(1, rwres["n"]) * 1e3,
                                             "value_repeated_spins": 1.0 / tws[1], "over_value": (1.0 / tws[1]) / out["value"],
                                             "positions_with_5_candidates": wide, "walker_variant": clk[3],
                                             "state_space": {4: "candidate ranks (4^L)", 5: "symbols (5^L)", 6: "mixed radix"}.get(clk[1], clk[1]),
                                             "most_states_per_target": clk[2],
                                             "note": "same contig, '-' on 30 %% of the reads at 1 %% of the positions: %d of %d positions show A, C, G, T "
                                                     "and '-'; one spin of %d paths (fill not included)" % (wide, n, paths)}
                del hw, rw
            except Exception as exc:
                out["wide_window_sparse"] = {"error": repr(exc)}
        if world == 1:
            # co-headline: the same step under the spec the PUBLISHED method describes (reference README.md:79-94 -> Nicholls et al.
            # 2021: naive Bayes, the marginal of the candidate times the conditionals of the earlier variants GIVEN the candidate) --
            # conditional E (or C: the "unique variants" term at the target or at the source, gretel.py:10) + marginal term, f32.
            # `value` runs the frozen default (A, no marginal term), which is a reconstruction.
            try:
                pub = {}
                for cm_ in ("E", "C"):
                    hx = Hansel(n, band=table.band, device=local, storage="f32", cond_mode=cm_, marginal_term=True)

                    def pstep():
                        hx.clear()
                        hx.fill_from_support(None, None, None, reads_handle=reads)
                        return hx.spin(paths)
                    pstep(); pstep()
                    torch.cuda.synchronize(); hx.sync()
                    ts_, nx = [], 0
                    for _ in range(7):
                        tx = time.perf_counter()
                        nx = pstep()["n"]
                        ts_.append(time.perf_counter() - tx)
                    ts_.sort()
                    pub[cm_] = nx / ts_[len(ts_) // 2]
                    del hx
                out["value_published_spec"] = {"value": min(pub.values()), "unit": "haplotypes/s", "by_conditional": pub,
                                               "over_value": min(pub.values()) / out["value"],
                                               "spec": "cond_mode E / C + marginal_term, f32 (the slower of the two is quoted)",
                                               "note": "the benchmark step (clear + fill + %d spins), 2 warm-up steps, median of 7 steps timed one by one; "
                                                       "which of the specs hanselx 0.0.92 implements is unpinned (DESIGN.md section 0)" % paths}
            except Exception as exc:
                out["value_published_spec"] = {"error": repr(exc)}
        if world == 1 and not args.no_spec_matrix:
            # the switches of the Hansel arithmetic the reference leaves to hanselx: the same step
            # (clear + fill + `paths` spins) under every conditional x marginal term x storage.  Every spec runs the same
            # kernels with the same incremental table maintenance; the figure to watch is max/min.
            try:
                import itertools
                rows = []
                for st_, cm_, mt_ in itertools.product(("f32", "f64"), "ABCDE", (False, True)):
                    hx = Hansel(n, band=table.band, device=local, storage=st_, cond_mode=cm_, marginal_term=mt_)

                    def xstep():
                        hx.clear()
                        hx.fill_from_support(None, None, None, reads_handle=reads)
                        return hx.spin(paths)
                    xstep(); xstep()
                    torch.cuda.synchronize(); hx.sync()
                    # each step ends with results on the host, so the steps are timed one by one and the median is quoted
                    # (a stall of the host in one 4 ms step would otherwise read as a slow spec)
                    ts_, nx = [], 0
                    for _ in range(7):
                        tx = time.perf_counter()
                        nx = xstep()["n"]
                        ts_.append(time.perf_counter() - tx)
                    ts_.sort()
                    rows.append({"cond_mode": cm_, "marginal_term": mt_, "storage": st_, "value": nx / ts_[len(ts_) // 2],
                                 "slowest_step_over_median": ts_[-1] / ts_[len(ts_) // 2],
                                 "walker_variant": hx.walk_clock()[3], "table_requeues": hx.walk_clock()[0]})
                    del hx
                vals = [r["value"] for r in rows]
                out["spec_matrix"] = {"unit": "haplotypes/s", "workload": "the benchmark step (clear + fill + %d spins), 2 warm-up steps, then the median of 7 steps timed one by one, per spec" % paths,
                                      "max_over_min": max(vals) / min(vals), "min": min(vals), "max": max(vals), "rows": rows}
            except Exception as exc:
                out["spec_matrix"] = {"error": repr(exc)}
        if world == 1 and not args.no_e2e:
            try:
                out["end_to_end"] = end_to_end_leg(table, paths, local)
            except Exception as exc:
                out["end_to_end"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and world == 1:
            # C2 in full, C3 three paths.  The default run stays within ~30 s of CPU work:
            # C2 = the whole contig but 5 of its paths, C3 = a SNP prefix extrapolated by call counts (labelled).
            if args.cpu_full:
                out["cpu_baseline"] = cpu_baseline_python(table, n, n, paths if n <= 2000 else 3)
            elif n <= 2000:
                out["cpu_baseline"] = cpu_baseline_python(table, n, n, min(paths, 5))
            else:
                out["cpu_baseline"] = cpu_baseline_python(table, min(args.cpu_snps, n), n)
            out["cpu_baseline_c"] = cpu_baseline_c(table, 3 if n >= 5000 else 10)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        # the figures a reader looks for first, right behind the contract's keys (the line is long: a stored tail used to lose them)
        lead = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")
        tm = out.get("throughput_mode_256") or {}
        summary = {
            "value_published_spec": (out.get("value_published_spec") or {}).get("value"),
            "throughput_mode_256": tm.get("value"),
            "throughput_mode_256_published_spec": (tm.get("published_spec") or {}).get("value"),
            "throughput_mode_256_sparse_deletions": (tm.get("sparse_deletions") or {}).get("value"),
            "wide_window_sparse": (out.get("wide_window_sparse") or {}).get("value"),
            "wide_window": (out.get("wide_window") or {}).get("value"),
            "end_to_end_wall_s": (out.get("end_to_end") or {}).get("wall_s"),
            "roofline_frac": (out.get("roofline") or {}).get("frac"),
            "unit": "haplotypes/s (end_to_end_wall_s: seconds; roofline_frac: of the HBM peak, the dominant kernel); details under the keys of the same names"}
        ordered = {k: out[k] for k in lead if k in out}
        ordered["summary"] = summary
        ordered.update({k: v for k, v in out.items() if k not in ordered})
        emit_line(json.dumps(ordered))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
