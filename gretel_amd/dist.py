"""
Window sharding across the GPUs of one node.

Every `(contig, start, end)` run of the reference is independent (nothing in
gretel/cmd.py:69-179 is shared between runs), so rank r owns windows r, r+world, ...
and there is NO collective on the data path.  The only exchanges are control-plane:
`broadcast_descriptor` (rank 0 tells everybody what to run) and `gather_results`
(fixed-size result records back to rank 0 for the writers of gretel/cmd.py:181-240).
Both go through torch.distributed: backend "nccl" is RCCL over xGMI on the GPU box,
"gloo" in the CPU tests.
"""
from __future__ import annotations

import numpy as np

DESC_KEYS = ("paths", "steps", "warmup", "config")


def windows_of_rank(n_windows, world, rank):
    """Round-robin ownership: window w belongs to rank w % world."""
    return list(range(rank, n_windows, world))


def broadcast_descriptor(desc, device, world, rank, force=False):
    """Rank 0's run descriptor (small dict of ints) to every rank.  force: go through torch.distributed even with one rank
    (a one-GPU box can exercise RCCL's communicator and its broadcast / gather that way)."""
    if world == 1 and not force:
        return {k: int(desc[k]) for k in DESC_KEYS}
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(desc[k]) for k in DESC_KEYS] if rank == 0 else [0] * len(DESC_KEYS),
                     dtype=torch.int64, device=device)
    dist.broadcast(t, src=0)
    return {k: int(v) for k, v in zip(DESC_KEYS, t.tolist())}


def pack_result(res, n_snps, max_paths):
    """Fixed-size record of one window's recovery: paths u8[max_paths][N+1] and
    f64[max_paths][4] (hp_current, hp_original, ratio, magnitude) + [n, hole_at]."""
    paths = np.full((max_paths, n_snps + 1), 255, dtype=np.uint8)
    vals = np.zeros((max_paths + 1, 4), dtype=np.float64)
    k = int(res["n"])
    if k:
        paths[:k] = res["paths"]
        vals[:k, 0] = res["hp_current"]
        vals[:k, 1] = res["hp_original"]
        vals[:k, 2] = res["ratio"]
        vals[:k, 3] = res["magnitude"]
    vals[max_paths, 0] = k
    vals[max_paths, 1] = res["hole_at"]
    return paths, vals


def unpack_result(paths, vals):
    max_paths = paths.shape[0]
    k = int(vals[max_paths, 0])
    return dict(n=k, hole_at=int(vals[max_paths, 1]), paths=paths[:k].copy(),
                hp_current=vals[:k, 0].copy(), hp_original=vals[:k, 1].copy(),
                ratio=vals[:k, 2].copy(), magnitude=vals[:k, 3].copy())


def gather_results(res, n_snps, max_paths, device, world, rank, force=False):
    """All windows' records on rank 0 (list indexed by rank), None elsewhere."""
    if world == 1 and not force:
        return [res]
    import torch
    import torch.distributed as dist
    paths, vals = pack_result(res, n_snps, max_paths)
    # ONE collective per window and step: the path bytes and the records' doubles travel in one byte buffer
    nb_p, nb_v = paths.size, vals.size * 8
    buf = np.empty(nb_p + nb_v, dtype=np.uint8)
    buf[:nb_p] = paths.ravel()
    buf[nb_p:] = vals.view(np.uint8).ravel()
    tb = torch.from_numpy(buf).to(device)
    gb = [torch.empty_like(tb) for _ in range(world)] if rank == 0 else None
    dist.gather(tb, gb, dst=0)
    if rank != 0:
        return None
    out = []
    for g in gb:
        a = g.cpu().numpy()
        out.append(unpack_result(a[:nb_p].reshape(paths.shape), a[nb_p:].copy().view(np.float64).reshape(vals.shape)))
    return out
