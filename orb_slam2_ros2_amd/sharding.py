"""Frame sharding of a sequence across ranks and the sequence-level exchange.

Stereo pairs are independent units of work (nothing in ORBExtractor / searchByStereo crosses frames), so a
sequence of F frames is cut into contiguous blocks, one per rank, and no collective is needed on the data
path.  What has to meet in one place is the sequence-level result.  Two forms:

  * gather_frames / WindowGather: every rank sends its per-frame records to rank 0 over xGMI (a fixed-count gather: records are
    padded to n_features).  Rank 0 then holds all of them in HBM -- and if the consumer lives on the host, all 690 MB of a
    KITTI-00 run cross rank 0's ONE PCIe link while the other seven links idle (12.5 ms against ~13 ms of compute per rank).
  * SharedRecordStore + WindowDrain: one node has one host memory.  The result buffer is a POSIX shared-memory segment every rank
    maps and page-locks; each rank drains its own windows into its rows over its OWN PCIe link, under the next window's compute,
    and the collective (RCCL) carries only the 16-byte per-frame summary (n, n_matches) to rank 0.

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
        if self.side is not None:
            # the send buffer may have been filled by an asynchronous copy on the caller's stream (run_sequence without collect_into)
            self.side.wait_stream(self.torch.cuda.current_stream(self.send[w % 2].device))
        if self.collective:
            if self.side is not None:
                with self.torch.cuda.stream(self.side):
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


class SharedRecordStore:
    """The sequence-level result in host memory shared by the ranks of one node: [n_frames, rec_bytes] bytes of POSIX shared memory
    (/dev/shm), created by one rank and mapped by all.  `pin()` page-locks the mapping for this process's GPU (hipHostRegister through
    torch's runtime binding), after which device -> host copies into `tensor` are asynchronous and run over this rank's own PCIe link.
    Creating / mapping the segment touches no GPU; only pin() does."""

    def __init__(self, name: str, n_frames: int, rec_bytes: int, create: bool):
        import mmap
        import os
        self.name, self.n_frames, self.rec_bytes = name, int(n_frames), int(rec_bytes)
        self.path = os.path.join("/dev/shm", name)
        self.size = max(1, self.n_frames * self.rec_bytes)
        flags = os.O_RDWR | (os.O_CREAT | os.O_EXCL if create else 0)
        fd = os.open(self.path, flags, 0o600)
        try:
            if create:
                os.ftruncate(fd, self.size)
            elif os.fstat(fd).st_size < self.size:
                raise ValueError(f"shared segment {self.path} is smaller than {self.size} bytes")
            self.map = mmap.mmap(fd, self.size)
        finally:
            os.close(fd)
        self.created, self.pinned, self._tensor = create, False, None

    @property
    def array(self):
        import numpy as np
        return np.frombuffer(self.map, dtype=np.uint8, count=self.n_frames * self.rec_bytes).reshape(self.n_frames, self.rec_bytes)

    @property
    def tensor(self):
        import torch
        if self._tensor is None:
            self._tensor = torch.from_numpy(self.array)
        return self._tensor

    def pin(self):
        """Page-lock the mapping for the current device (no-op without a GPU, or when already pinned)."""
        import torch
        if self.pinned or not torch.cuda.is_available() or self.n_frames == 0:
            return
        rc = torch.cuda.cudart().cudaHostRegister(self.tensor.data_ptr(), self.size, 0)
        if int(rc) != 0:
            raise RuntimeError(f"hipHostRegister of the shared record segment failed ({int(rc)})")
        self.pinned = True

    def close(self):
        import os
        if self.pinned:
            import torch
            torch.cuda.cudart().cudaHostUnregister(self.tensor.data_ptr())
            self.pinned = False
        self._tensor = None
        try:
            self.map.close()
        except BufferError:   # a numpy view is still alive: the mapping goes with it
            pass
        if self.created:
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass


class WindowDrain:
    """WindowGather's interface with the records going to host memory instead of rank 0's HBM: as soon as a rank has packed window w
    (buffer(w)), push(w) copies the window's rows into the rank's rows of the SharedRecordStore -- asynchronously, on a side stream,
    over the rank's own PCIe link, while window w + 1 is computed -- and starts a gather of the window's 16-byte record heads
    (n, n_matches) to `dst`.  finish() waits for the copies and the gathers, then a barrier tells dst that every rank's rows are in
    place, and returns on dst the int32 [n_frames, 4] summary table in sequence order."""

    HEAD = 16

    def __init__(self, n_frames: int, rank: int, world: int, win: int, shape_tail, dtype, device, store: SharedRecordStore, dst: int = 0,
                 force_collective: bool = False):
        import torch
        self.torch = torch
        self.F, self.rank, self.world, self.win, self.dst, self.store = n_frames, rank, world, max(1, int(win)), dst, store
        self.per = (n_frames + world - 1) // world if n_frames > 0 else 0
        self.n_windows = (self.per + self.win - 1) // self.win
        self.b, self.e = frame_range(n_frames, rank, world) if n_frames > 0 else (0, 0)
        self.collective = world > 1 or force_collective
        shape = (self.win,) + tuple(shape_tail)
        if len(shape_tail) != 1 or shape_tail[0] != store.rec_bytes or dtype != torch.uint8:
            raise ValueError("WindowDrain moves uint8 records of the store's size")
        self.send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(2)]
        # the record heads of EVERY window keep their own (tiny) buffers: nothing on the host waits for a gather before finish()
        self.heads = [torch.zeros((self.win, 4), dtype=torch.int32, device=device) for _ in range(self.n_windows)]
        self.recv = ([[torch.empty((self.win, 4), dtype=torch.int32, device=device) for _ in range(world)] for _ in range(self.n_windows)]
                     if (rank == dst and self.collective) else None)
        self.cuda = self.send[0].is_cuda
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self.fence = [None, None]
        self.works = []
        self.next = 0
        self.bytes_drained = 0

    def buffer(self, w: int):
        """The buffer window w is packed into; blocks until the copy that last read this buffer is complete."""
        f = self.fence[w % 2]
        if f is not None:
            f.synchronize()
            self.fence[w % 2] = None
        return self.send[w % 2]

    def local_rows(self, w: int):
        n = self.e - self.b
        return min(n, w * self.win), min(n, (w + 1) * self.win)

    def push(self, w: int):
        import torch.distributed as dist
        torch = self.torch
        if w != self.next:
            raise ValueError(f"windows must be pushed in order (got {w}, expected {self.next})")
        self.next += 1
        k = w % 2
        lo, hi = self.local_rows(w)

        def issue():
            if hi > lo:
                self.store.tensor[self.b + lo:self.b + hi].copy_(self.send[k][: hi - lo], non_blocking=True)
                self.bytes_drained += (hi - lo) * self.store.rec_bytes
                self.heads[w][: hi - lo] = self.send[k][: hi - lo, : self.HEAD].contiguous().view(torch.int32).reshape(hi - lo, 4)
            if self.collective:
                self.works.append(dist.gather(self.heads[w], self.recv[w] if self.rank == self.dst else None, dst=self.dst, async_op=True))
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record()
                self.fence[k] = ev

        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.send[k].device))   # whatever filled the buffer on the caller's stream
            with torch.cuda.stream(self.side):
                issue()
        else:
            issue()

    def finish(self):
        """Wait for this rank's copies and the gathers, then for every other rank's (barrier); on dst: the int32 [n_frames, 4] summary."""
        import torch.distributed as dist
        if self.next != self.n_windows:
            raise ValueError(f"{self.next} of {self.n_windows} windows were pushed")
        for wk in self.works:
            if wk is not None:
                wk.wait()
        self.works = []
        if self.side is not None:
            self.side.synchronize()
        for k in range(2):
            if self.fence[k] is not None:
                self.fence[k].synchronize()
                self.fence[k] = None
        if self.collective:
            dist.barrier()   # every rank's rows are in the shared segment
        if self.rank != self.dst:
            return None
        summary = self.torch.zeros((self.F, 4), dtype=self.torch.int32)
        for w in range(self.n_windows):
            parts = self.recv[w] if self.collective else [self.heads[w]]
            for r in range(self.world):
                rb, re = frame_range(self.F, r, self.world)
                lo, hi = min(re - rb, w * self.win), min(re - rb, (w + 1) * self.win)
                if hi > lo:
                    summary[rb + lo:rb + hi] = parts[r][: hi - lo].cpu()
        return summary
