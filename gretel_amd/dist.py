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


class _GatherBuffers:
    """What a gather needs, allocated once per shape: a (pinned) host staging buffer and its device twin on every rank, on
    rank 0 one [world][bytes] tensor to gather into and its (pinned) host twin.  Per step that leaves one packing pass, one
    asynchronous upload, the collective, and on rank 0 ONE download of all windows -- no per-rank copies."""

    def __init__(self, nb_p, nb_v, device, world, rank):
        import torch
        self.nb_p, self.nb_v = nb_p, nb_v
        self.off_v = (nb_p + 7) & ~7                              # the records' doubles start 8-byte aligned
        nbytes = self.off_v + nb_v
        cuda = device.type == "cuda"
        self.cuda = cuda
        self.send_host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=cuda)
        self.send_np = self.send_host.numpy()
        self.send_dev = torch.empty(nbytes, dtype=torch.uint8, device=device) if cuda else self.send_host
        self.recv_dev = self.recv_host = None
        if rank == 0:
            self.recv_dev = torch.empty((world, nbytes), dtype=torch.uint8, device=device)
            self.recv_host = torch.empty((world, nbytes), dtype=torch.uint8, pin_memory=True) if cuda else self.recv_dev


_gather_cache = {}


def gather_results(res, n_snps, max_paths, device, world, rank, force=False, copy=True):
    """All windows' records on rank 0 (list indexed by rank), None elsewhere.  ONE collective per window and step: the path
    bytes and the records' doubles travel in one byte buffer.  copy=False: the returned arrays are views of a buffer the next
    call overwrites (bench.py, which looks at one step's results at a time)."""
    if world == 1 and not force:
        return [res]
    import torch
    import torch.distributed as dist
    n1 = n_snps + 1
    key = (n_snps, max_paths, str(device), world, rank)
    b = _gather_cache.get(key)
    if b is None:
        b = _gather_cache[key] = _GatherBuffers(max_paths * n1, (max_paths + 1) * 4 * 8, device, world, rank)
    # pack (pack_result's layout) straight into the staging buffer
    k = int(res["n"])
    pv = b.send_np[:b.nb_p].reshape(max_paths, n1)
    vv = b.send_np[b.off_v:b.off_v + b.nb_v].view(np.float64).reshape(max_paths + 1, 4)
    if k:
        pv[:k] = res["paths"]
        vv[:k, 0] = res["hp_current"]
        vv[:k, 1] = res["hp_original"]
        vv[:k, 2] = res["ratio"]
        vv[:k, 3] = res["magnitude"]
    pv[k:] = 255
    vv[k:max_paths] = 0.0
    vv[max_paths] = (k, res["hole_at"], 0.0, 0.0)
    if b.cuda:
        b.send_dev.copy_(b.send_host, non_blocking=True)         # (stream-ordered: the collective is queued behind it)
    dist.gather(b.send_dev, [b.recv_dev[r] for r in range(world)] if rank == 0 else None, dst=0)
    if rank != 0:
        return None
    if b.cuda:
        b.recv_host.copy_(b.recv_dev, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    a = b.recv_host.numpy()
    out = []
    for r in range(world):
        paths = a[r, :b.nb_p].reshape(max_paths, n1)
        vals = a[r, b.off_v:b.off_v + b.nb_v].view(np.float64).reshape(max_paths + 1, 4)
        if copy:
            out.append(unpack_result(paths, vals))
        else:
            kk = int(vals[max_paths, 0])
            out.append(dict(n=kk, hole_at=int(vals[max_paths, 1]), paths=paths[:kk], hp_current=vals[:kk, 0], hp_original=vals[:kk, 1],
                            ratio=vals[:kk, 2], magnitude=vals[:kk, 3]))
    return out
