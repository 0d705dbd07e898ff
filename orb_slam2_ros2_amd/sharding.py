"""Frame sharding of a sequence across ranks and the sequence-level gather.

Stereo pairs are independent units of work (nothing in ORBExtractor / searchByStereo crosses frames), so a
sequence of F frames is cut into contiguous blocks, one per rank, and no collective is needed on the data
path.  One exchange happens at the end: every rank sends its per-frame records to rank 0.  The records have
a fixed size per frame (padded to n_features), so a single fixed-count gather does it -- over xGMI every
sender uses its direct link to the root, ring collectives would buy nothing for ~100 KB per frame.

The functions take any initialised torch.distributed backend ("nccl" = RCCL on the GPUs, "gloo" in the CPU
tests).
"""
from __future__ import annotations

from typing import Tuple


def frame_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of rank `rank`: ceil(F/world) frames per rank, the last ranks get the rest."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = (n_frames + world - 1) // world
    b = min(n_frames, rank * per)
    e = min(n_frames, b + per)
    return b, e


def gather_frames(local, n_frames: int, rank: int, world: int, dst: int = 0, force_collective: bool = False):
    """Gather per-frame records to `dst`.

    local: tensor [n_local, ...] holding this rank's frames frame_range(n_frames, rank, world) in order.
    Returns on dst a tensor [n_frames, ...] in sequence order, elsewhere None.
    force_collective: take the collective branch with ONE rank too (an initialised process group of world size 1) -- what the
    1-GPU boxes run so that torch's RCCL and the front-end library are known to share one HIP runtime before an 8-GPU node sees them.
    """
    import torch
    import torch.distributed as dist

    b, e = frame_range(n_frames, rank, world)
    if local.shape[0] != e - b:
        raise ValueError(f"rank {rank} holds {local.shape[0]} frames, expected {e - b}")
    if world == 1 and not force_collective:
        return local
    per = (n_frames + world - 1) // world
    padded = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: e - b] = local
    outs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, outs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        rb, re = frame_range(n_frames, r, world)
        parts.append(outs[r][: re - rb])
    return torch.cat(parts, dim=0)


class WindowGather:
    """The same exchange in WINDOWS ("or per window of W frames for a streaming consumer"): every rank cuts its block into
    windows of `win` frames; as soon as a rank has packed window w it joins gather w, which runs asynchronously on the backend's own
    stream (RCCL) while the ranks compute window w + 1, and rank `dst` hands every rank's part of a finished window to `sink(first global
    frame, tensor)` -- e.g. a non-blocking copy into page-locked host memory -- so that neither the wire time nor the device-to-host
    copy of the 152 KB per frame sits at the end of the job.  All ranks take part in ceil(per / win) gathers (per = ceil(F / world)),
    a rank whose block ends early sends padding that dst ignores.

    Two send buffers and two receive sets rotate; a buffer is taken again two windows later, after the gather that read it (and, on
    dst, the sink copies that read the receive set) are known to be complete.
    """

    def __init__(self, n_frames: int, rank: int, world: int, win: int, shape_tail, dtype, device, dst: int = 0, sink=None,
                 force_collective: bool = False):
        import torch
        self.torch = torch
        self.F, self.rank, self.world, self.win, self.dst = n_frames, rank, world, max(1, int(win)), dst
        self.per = (n_frames + world - 1) // world if n_frames > 0 else 0
        self.n_windows = (self.per + self.win - 1) // self.win
        self.b, self.e = frame_range(n_frames, rank, world) if n_frames > 0 else (0, 0)
        self.collective = world > 1 or force_collective
        self.sink = sink
        self.kept = [] if (sink is None and rank == dst) else None   # no sink: dst keeps the parts and assemble() returns the sequence
        shape = (self.win,) + tuple(shape_tail)
        self.send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(2)]
        self.recv = ([[torch.empty(shape, dtype=dtype, device=device) for _ in range(world)] for _ in range(2)]
                     if (rank == dst and self.collective) else None)
        self.cuda = self.send[0].is_cuda
        # The gathers and the sink copies are issued on a stream of their own: torch's current stream is normally the LEGACY default
        # stream, and work queued there synchronises with every blocking stream of the process -- a 78 MB drain per window on it stalled
        # the front end's upload / compute pipeline (measured: the 4541-pair job 0.132 s windowed against 0.122 s with one gather at the end)
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self.fence = [None, None]     # per buffer set: an event after which the set may be rewritten
        self.inflight = None          # (window, work) of the gather not yet drained on dst
        self.next = 0

    def buffer(self, w: int):
        """The send buffer window w is packed into ([win, ...]; rows past the rank's frames are padding).  Blocks until the gather (and
        the sink copies) that last used this buffer set are complete."""
        f = self.fence[w % 2]
        if f is not None:
            f.synchronize()
            self.fence[w % 2] = None
        return self.send[w % 2]

    def local_rows(self, w: int):
        """[first, last) frames of this rank's block (block-relative) that belong to window w"""
        n = self.e - self.b
        return min(n, w * self.win), min(n, (w + 1) * self.win)

    def _drain(self):
        """dst: hand the parts of the gather in flight to the sink (after the gather, on the current stream)"""
        if self.inflight is None:
            return
        if self.side is not None:
            with self.torch.cuda.stream(self.side):
                self._drain_on_current()
        else:
            self._drain_on_current()

    def _drain_on_current(self):
        w, work = self.inflight
        self.inflight = None
        if work is not None:
            work.wait()      # RCCL: orders the current stream behind the gather; gloo: blocks until it is done
        if self.rank == self.dst:
            parts = self.recv[w % 2] if self.collective else [self.send[w % 2]]
            for r in range(self.world):
                rb, re = frame_range(self.F, r, self.world)
                lo, hi = min(re - rb, w * self.win), min(re - rb, (w + 1) * self.win)
                if hi > lo:
                    t = parts[r][: hi - lo]
                    if self.sink is not None:
                        self.sink(rb + lo, t)
                    else:
                        self.kept.append((rb + lo, t.clone()))
        if self.cuda:
            ev = self.torch.cuda.Event()
            ev.record()      # behind the gather and the sink copies of this window
            self.fence[w % 2] = ev

    def push(self, w: int):
        """Window w has been packed into buffer(w) (and the packing is complete): start its gather; drains window w - 1 first."""
        import torch.distributed as dist
        if w != self.next:
            raise ValueError(f"windows must be pushed in order (got {w}, expected {self.next})")
        self.next += 1
        self._drain()
        work = None
        if self.collective:
            if self.side is not None:
                with self.torch.cuda.stream(self.side):   # (the send buffer is complete: its pack kernel was waited for on the host)
                    work = dist.gather(self.send[w % 2], self.recv[w % 2] if self.rank == self.dst else None, dst=self.dst, async_op=True)
            else:
                work = dist.gather(self.send[w % 2], self.recv[w % 2] if self.rank == self.dst else None, dst=self.dst, async_op=True)
        self.inflight = (w, work)

    def finish(self):
        """Drain the last window; on dst without a sink return the assembled [n_frames, ...] tensor."""
        if self.next != self.n_windows:
            raise ValueError(f"{self.next} of {self.n_windows} windows were pushed")
        self._drain()
        if self.cuda:
            for k in range(2):
                if self.fence[k] is not None:
                    self.fence[k].synchronize()
                    self.fence[k] = None
        if self.kept is None:
            return None
        self.kept.sort(key=lambda it: it[0])
        if not self.kept:
            return self.send[0][:0].clone()
        return self.torch.cat([t for _, t in self.kept], dim=0)
