"""The N > 1 flow of bench.py on the one GPU a test box has: `python bench.py --gpus 2` starts its own two ranks
(no launcher), control records travel over gloo, both ranks compute on cuda:0 (--share-gpu).  On an 8-GPU node the
driver runs the same code with the default backend (nccl = RCCL) and one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra, paths="20"):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--steps", "2", "--warmup", "1",
                          "--paths", paths, "--no-cpu-baseline", "--no-throughput-leg", "--no-spec-matrix", "--no-e2e"] + extra,
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    # stdout carries that line and nothing else (RCCL's version banner, printed through C stdio when the first communicator
    # comes up, once landed behind it)
    assert out.stdout.strip().splitlines() == lines, out.stdout
    return json.loads(lines[0])


def test_two_ranks_share_one_gpu_over_gloo():
    one = _run([])
    two = _run(["--gpus", "2", "--backend", "gloo", "--share-gpu"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "weak" and two["config"]["backend"] == "gloo"
    assert two["steps"] == 2 and two["warmup"] == 1
    # every rank recovered its own window: 2 ranks x 2 steps x 20 paths in max-over-ranks time
    assert abs(two["value"] * two["ms_per_step"] * 1e-3 * 2 - 2 * 2 * 20) < 1e-6 * 80
    assert abs(one["value"] * one["ms_per_step"] * 1e-3 * 2 - 2 * 20) < 1e-6 * 40


def test_nccl_backend_refuses_to_share_a_gpu():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--config", "C2"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--backend gloo" in out.stderr


def _check_gathered(path, world, paths):
    """What rank 0 gathered, against the oracle: rank r owns the window of seed r (bench.py: seed = rank)."""
    import numpy as np
    from gretel_amd.synth import make_config
    from oracle.c_oracle import COracle
    z = np.load(path)
    assert int(z["world"]) == world
    for r in range(world):
        t = make_config("C2", seed=r)
        o = COracle(t.n_snps, t.band)
        o.fill(t)
        ref = o.spin(paths)
        assert int(z["n_%d" % r]) == ref["n"] == paths and int(z["hole_at_%d" % r]) == 0
        assert np.array_equal(z["paths_%d" % r], ref["paths"]), "rank %d's paths differ from the oracle's for seed %d" % (r, r)
        assert z["hp_current_%d" % r].tolist() == ref["hp_current"].tolist()
        assert z["hp_original_%d" % r].tolist() == ref["hp_original"].tolist()
        assert z["ratio_%d" % r].tolist() == ref["ratio"].tolist()
        assert np.allclose(z["magnitude_%d" % r], ref["magnitude"], rtol=1e-10, atol=0)


def test_what_rank_0_gathers_is_every_ranks_own_window(tmp_path):
    dump = str(tmp_path / "g.npz")
    two = _run(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--dump-gathered", dump])
    assert two["n_gpus"] == 2
    _check_gathered(dump, 2, 20)


def test_eight_ranks_on_one_gpu(tmp_path):
    # the world = 8 code path (ownership w mod 8, eight gathers) of BASELINE config C4, at C2 size, on the one GPU of a test box
    dump = str(tmp_path / "g8.npz")
    eight = _run(["--gpus", "8", "--backend", "gloo", "--share-gpu", "--dump-gathered", dump], paths="10")
    assert eight["n_gpus"] == 8 and eight["scaling"] == "weak"
    assert abs(eight["value"] * eight["ms_per_step"] * 1e-3 * 2 - 8 * 2 * 10) < 1e-6 * 160
    _check_gathered(dump, 8, 10)


def test_one_rank_through_rccl(tmp_path):
    # what a one-GPU box can show of the RCCL path: the process group on backend nccl (= RCCL) with ONE rank, the run descriptor
    # broadcast, the result records gathered and the timings all-reduced on cuda:0 -- the calls the driver's 8-rank launch makes
    dump = str(tmp_path / "g1.npz")
    one = _run(["--force-dist", "--backend", "nccl", "--dump-gathered", dump])
    assert one["n_gpus"] == 1 and one["config"]["backend"] == "nccl" and one["config"]["control_records_through_torch_distributed"] is True
    assert abs(one["value"] * one["ms_per_step"] * 1e-3 * 2 - 2 * 20) < 1e-6 * 40
    _check_gathered(dump, 1, 20)


def test_the_drivers_launcher_two_ranks(tmp_path):
    # the driver's own launch line for N > 1 (python -m torch.distributed.run ... bench.py --gpus N ...: RANK / LOCAL_RANK /
    # WORLD_SIZE / MASTER_* from the environment), here with gloo and both ranks on the one GPU of the test box
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dump = str(tmp_path / "gl.npz")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--config", "C2", "--paths", "20", "--backend", "gloo", "--share-gpu", "--no-cpu-baseline", "--no-throughput-leg",
                          "--no-spec-matrix", "--no-e2e", "--dump-gathered", dump], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                      # rank 0 prints the one line
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["warmup"] == 1 and two["scaling"] == "weak"
    assert abs(two["value"] * two["ms_per_step"] * 1e-3 * 2 - 2 * 2 * 20) < 1e-6 * 80
    _check_gathered(dump, 2, 20)


def test_the_bench_line_keeps_its_contract():
    # one JSON line with the keys the driver and the judge read; C2 so that the CPU baseline leg stays short
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--steps", "3", "--warmup", "1", "--no-spec-matrix",
                          "--no-throughput-leg", "--no-e2e"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("haplotypes/sec") and d["unit"] == "haplotypes/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - d["config"]["paths"]) < 1e-6 * d["config"]["paths"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "issue_model", "step_frac", "launch_floor_us", "kernel_source_sha"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["step_frac"] < 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "haplotypes/s"
    assert d["value"] > 50 * c["value"]                      # north_star: >= 50x the reference's CPU path
    # the secondary figures stand right behind the contract's keys (a stored tail of the line had lost them)
    keys = list(d)
    assert keys.index("summary") == keys.index("config") + 1 and keys.index("summary") < keys.index("roofline")
    for k in ("value_published_spec", "throughput_mode_256", "throughput_mode_256_published_spec", "throughput_mode_256_sparse_deletions",
              "wide_window_sparse", "end_to_end_wall_s", "roofline_frac"):
        assert k in d["summary"], k
    assert d["summary"]["roofline_frac"] == r["frac"]
