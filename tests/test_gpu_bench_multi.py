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


def _run(extra):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--steps", "2", "--warmup", "1",
                          "--paths", "20", "--no-cpu-baseline", "--no-throughput-leg"] + extra,
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
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
