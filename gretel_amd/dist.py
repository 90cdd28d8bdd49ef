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
        # the staging buffer is packed again by the next call: the upload and the collective that read it must be through
        if b.cuda:
            torch.cuda.current_stream().synchronize()
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


class ResultExchange:
    """gather_results with the collective OFF the critical path (bench.py's default; `--blocking-gather` keeps the form above).

    The records of a step travel while the next step runs: `buffers()` hands out views of a pinned staging slot that
    Hansel.spin writes its results into directly (no packing pass), `submit()` queues upload + gather + (rank 0) download of that
    slot on a side stream -- from a worker thread, so that the caller does not even pay the host side of the collective -- and
    returns at once, `collect()` waits for the OLDEST outstanding submission and returns what rank 0
    gathered (None elsewhere).  Two slots: a slot is handed out again only when its previous submission has been collected, so
    no buffer is rewritten while a copy or the collective may still read it.  Every rank submits once per step, in step order:
    the collectives match across ranks.  Wire format per window: paths u8[max_paths][N+1], gh_path_rec f64[max_paths][5],
    f64[2] = (n, hole_at).

    Rules the class enforces or states: (1) buffers() REFUSES to hand out a slot whose previous submission has not been
    collected (a forgotten collect() used to lose a step's records silently); (2) the worker thread issues its gather on the
    default process group: the caller must not run another collective while submissions are outstanding -- drain() first, as
    bench.py does in front of its barrier -- or pass `group=` (torch.distributed.new_group) to give the exchange its own;
    (3) a submission that does not complete within `timeout_s` (a peer whose worker raised never enters the collective)
    raises instead of blocking forever -- the submission stays queued (its slot is not handed out again), the exchange refuses
    further use, and close() returns without waiting for the stuck worker."""

    NREC = 5

    def __init__(self, n_snps, max_paths, device, world, rank, force=False, slots=2, group=None, timeout_s=120.0):
        import queue as queue_mod
        import threading
        import torch
        self.n1, self.max_paths, self.device, self.world, self.rank = n_snps + 1, max_paths, device, world, rank
        self.active = world > 1 or force
        self.group, self.timeout_s = group, timeout_s
        self.cuda = device.type == "cuda"
        self.nb_p = max_paths * self.n1
        self.off_r = (self.nb_p + 7) & ~7
        self.nb_r = max(1, max_paths) * self.NREC * 8
        self.off_t = self.off_r + self.nb_r
        self.nbytes = self.off_t + 16
        self.slots = []
        for _ in range(slots):
            host = torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=self.cuda)
            sl = dict(host=host, np=host.numpy(), done=threading.Event(), error=None)
            sl["done"].set()
            if self.active:
                sl["dev"] = torch.empty(self.nbytes, dtype=torch.uint8, device=device) if self.cuda else host
                sl["recv_list"] = None
                if rank == 0:
                    sl["recv_dev"] = torch.empty((world, self.nbytes), dtype=torch.uint8, device=device)
                    sl["recv_host"] = torch.empty((world, self.nbytes), dtype=torch.uint8, pin_memory=True) if self.cuda else sl["recv_dev"]
                    sl["recv_list"] = [sl["recv_dev"][r] for r in range(world)]
                if self.cuda:
                    sl["event"] = torch.cuda.Event()
            self.slots.append(sl)
        self.side = torch.cuda.Stream(device=device) if (self.cuda and self.active) else None
        self.next_slot = 0
        self.queue = []                 # submitted, not yet collected: (slot index, n, hole_at)
        self.broken = False             # a gather timed out: a slot may be stuck inside the collective for good
        self._thread = None
        if self.active:
            self._jobs = queue_mod.Queue()
            self._thread = threading.Thread(target=self._worker, name="gretel-result-exchange", daemon=True)
            self._thread.start()

    def _views(self, a):
        paths = a[:self.nb_p].reshape(self.max_paths, self.n1)
        recs = a[self.off_r:self.off_r + self.nb_r].view(np.float64).reshape(max(1, self.max_paths), self.NREC)
        tail = a[self.off_t:self.off_t + 16].view(np.float64)
        return paths, recs, tail

    def buffers(self):
        """(paths, recs) views of the slot the next submit() sends: hand them to Hansel.spin(out_paths=, out_recs=)."""
        sl = self.slots[self.next_slot]
        if self.broken:
            raise RuntimeError("ResultExchange: an earlier gather timed out; the exchange is closed to further use")
        if any(q[0] == self.next_slot for q in self.queue):
            raise RuntimeError("ResultExchange: slot %d still holds a submission that was never collected -- call collect() (or "
                               "drain()) once per submit(); handing the slot out again would lose that step's records" % self.next_slot)
        p, r, _ = self._views(sl["np"])
        return p, r

    def submit(self, n, hole_at):
        if self.broken:
            raise RuntimeError("ResultExchange: an earlier gather timed out; the exchange is closed to further use")
        si = self.next_slot
        sl = self.slots[si]
        self.next_slot = (si + 1) % len(self.slots)
        _, _, tail = self._views(sl["np"])
        tail[0], tail[1] = float(n), float(hole_at)
        if self.active:
            # the torch calls (upload, collective, download: ~0.1 ms of host time) are made by a worker thread: the caller goes
            # straight on to its next step -- its time is spent inside the C library, which releases the interpreter lock
            sl["done"].clear()
            self._jobs.put(si)
        self.queue.append((si, int(n), int(hole_at)))

    def _worker(self):
        import torch
        import torch.distributed as dist
        while True:
            si = self._jobs.get()
            if si is None:
                return
            sl = self.slots[si]
            try:
                recv = sl["recv_list"]
                if self.cuda:
                    with torch.cuda.stream(self.side):
                        sl["dev"].copy_(sl["host"], non_blocking=True)
                        dist.gather(sl["dev"], recv, dst=0, group=self.group)
                        if self.rank == 0:
                            sl["recv_host"].copy_(sl["recv_dev"], non_blocking=True)
                        sl["event"].record(self.side)
                    sl["event"].synchronize()
                else:
                    dist.gather(sl["dev"], recv, dst=0, group=self.group)
            except BaseException as exc:           # (handed to the thread that collects)
                sl["error"] = exc
            sl["done"].set()

    def close(self):
        if self.active and self._thread is not None:
            try:
                if not self.broken:
                    self.drain()
            finally:
                self._jobs.put(None)
                # (a worker stuck inside a collective that a failed peer never entered cannot be joined: the thread is a daemon,
                # the timeout has already been reported once -- do not block the caller's cleanup on it a second time)
                self._thread.join(None if not self.broken else 1.0)
                self._thread = None

    def _finish(self, q, keep=True):
        si, n, hole = q
        sl = self.slots[si]
        if self.active:
            if self.broken:
                raise RuntimeError("ResultExchange: an earlier gather timed out; the exchange is closed to further use")
            if not sl["done"].wait(self.timeout_s):
                # the submission STAYS in the queue (its slot may still be inside the collective: buffers() must not hand it out)
                # and the exchange is closed to further use: buffers() / submit() / collect() raise, close() does not wait
                self.broken = True
                raise TimeoutError("ResultExchange: a gather did not complete within %.0f s (a peer that failed never enters the "
                                   "collective)" % self.timeout_s)
            if sl.get("error") is not None:
                exc, sl["error"] = sl["error"], None
                self.queue.remove(q)
                raise exc
        self.queue.remove(q)
        if not keep or (self.active and self.rank != 0):
            return None
        if not self.active:
            rows = [self._views(sl["np"])]
        else:
            a = sl["recv_host"].numpy()
            rows = [self._views(a[r]) for r in range(self.world)]
        out = []
        for paths, recs, tail in rows:
            k = int(tail[0])
            out.append(dict(n=k, hole_at=int(tail[1]), paths=paths[:k], hp_current=recs[:k, 0], hp_original=recs[:k, 1],
                            ratio=recs[:k, 2], magnitude=recs[:k, 3], min_marginal=recs[:k, 4]))
        return out

    def collect(self):
        """Results of the oldest outstanding submission (views of buffers that the submission after next overwrites); None on
        ranks other than 0, and when nothing is outstanding."""
        if not self.queue:
            return None
        return self._finish(self.queue[0])          # (popped by _finish once its gather is known to have completed)

    def drain(self):
        last = None
        while self.queue:
            last = self._finish(self.queue[0])
        return last
