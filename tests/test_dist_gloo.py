"""The N>1 path (descriptor broadcast, window ownership, result gather) with world_size 2
over gloo on CPU.  The records travel as tensors exactly as they do over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT
from gretel_amd import dist as gdist


def test_window_ownership_partitions():
    for world in (1, 2, 3, 8):
        owned = sorted(w for r in range(world) for w in gdist.windows_of_rank(19, world, r))
        assert owned == list(range(19))


def test_pack_roundtrip():
    rng = np.random.default_rng(0)
    res = dict(n=3, hole_at=7, paths=rng.integers(0, 7, (3, 11), dtype=np.uint8), hp_current=rng.random(3),
               hp_original=rng.random(3), ratio=rng.random(3), magnitude=rng.random(3))
    back = gdist.unpack_result(*gdist.pack_result(res, 10, 5))
    assert back["n"] == 3 and back["hole_at"] == 7
    for k in ("paths", "hp_current", "hp_original", "ratio", "magnitude"):
        assert np.array_equal(back[k], res[k])


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from gretel_amd import dist as gdist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    d = gdist.broadcast_descriptor(dict(paths=4, steps=2, warmup=1, config=3) if rank == 0 else {}, dev, world, rank)
    assert d == dict(paths=4, steps=2, warmup=1, config=3), d
    rng = np.random.default_rng(rank)
    k = 2 + rank
    res = dict(n=k, hole_at=rank * 5, paths=rng.integers(0, 7, (k, 9), dtype=np.uint8), hp_current=rng.random(k),
               hp_original=rng.random(k), ratio=rng.random(k), magnitude=rng.random(k))
    got = gdist.gather_results(res, 8, d["paths"], dev, world, rank)
    if rank == 0:
        assert len(got) == world
        for r in range(world):
            rr = np.random.default_rng(r)
            kk = 2 + r
            assert got[r]["n"] == kk and got[r]["hole_at"] == r * 5
            assert np.array_equal(got[r]["paths"], rr.integers(0, 7, (kk, 9), dtype=np.uint8))
            assert np.array_equal(got[r]["hp_current"], rr.random(kk))
        print("GATHER_OK")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()
''')


EX_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from gretel_amd import dist as gdist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    n_snps, max_paths, steps = 8, 4, 5
    ex = gdist.ResultExchange(n_snps, max_paths, dev, world, rank)

    def fake(step, r):          # what Hansel.spin would leave in the slot: k paths, their records
        rng = np.random.default_rng(1000 * step + r)
        k = 1 + (step + r) %% max_paths
        return k, rng.integers(0, 7, (k, n_snps + 1), dtype=np.uint8), rng.random((k, 5))

    def check(step, g):           # (at once: what collect() returns are views of buffers the submission after next overwrites)
        if rank != 0:
            assert g is None
            return
        assert g is not None and len(g) == world
        for r in range(world):
            k, paths, recs = fake(step, r)
            assert g[r]["n"] == k and g[r]["hole_at"] == 7 * r + step, (step, r, g[r]["n"], g[r]["hole_at"])
            assert np.array_equal(g[r]["paths"], paths) and np.array_equal(g[r]["hp_current"], recs[:, 0])
            assert np.array_equal(g[r]["magnitude"], recs[:, 3]) and np.array_equal(g[r]["min_marginal"], recs[:, 4])

    checked = 0
    for step in range(steps):
        pv, rv = ex.buffers()
        k, paths, recs = fake(step, rank)
        pv[:k] = paths; rv[:k] = recs
        ex.submit(k, 7 * rank + step)
        if len(ex.queue) > 1:
            check(step - 1, ex.collect())       # the step before this one
            checked += 1
    check(steps - 1, ex.drain())
    assert not ex.queue and checked == steps - 1
    if rank == 0:
        print("EXCHANGE_OK")
    dist.barrier()
    dist.destroy_process_group()
''')


def _run_world2(tmp_path, text, marker):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(text)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert marker in outs[0]


def test_overlapped_exchange_world2(tmp_path):
    # ResultExchange: five steps through two slots, every step's records of both ranks arrive on rank 0, in step order, one step late
    _run_world2(tmp_path, EX_WORKER % ROOT, "EXCHANGE_OK")


def test_exchange_with_one_rank_needs_no_process_group():
    import torch
    ex = gdist.ResultExchange(6, 3, torch.device("cpu"), 1, 0)
    for step in range(4):
        pv, rv = ex.buffers()
        pv[:2] = step
        rv[:2] = 0.5 * step
        ex.submit(2, 0)
        if len(ex.queue) > 1:
            g = ex.collect()
            assert len(g) == 1 and g[0]["n"] == 2 and (g[0]["paths"] == step - 1).all() and (g[0]["ratio"] == 0.5 * (step - 1)).all()
    g = ex.drain()
    assert (g[0]["paths"] == 3).all()


def test_broadcast_and_gather_world2(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]


def test_bench_launches_its_own_ranks_without_a_launcher():
    # `python bench.py --gpus 2` as the driver invokes it: the parent spawns one child per rank before touching the
    # GPU.  Without a GPU each child stops with the "needs a GPU" message -- two of them, and no request for torchrun.
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stderr.count("needs a GPU") == 2, out.stderr
    assert "torch.distributed.run" not in out.stderr
