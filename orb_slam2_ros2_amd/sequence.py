"""Sequence-level driver: the loop of example/Stereo/KittiStereo.cc:28-37 over F stereo pairs, cut into contiguous blocks per rank
(sharding.frame_range), each rank working through its block in batches, and ONE exchange at the end: the per-frame records of every
rank gathered on rank 0 (sharding.gather_frames; RCCL on the GPUs, gloo in the CPU tests).

A frame record is what Frame::createStereo leaves behind for the tracker -- left keypoints, left descriptors, right_u, depth and the
counts -- padded to n_features so that every frame has the same size:

    int32 n_keypoints | int32 n_matches | 8 bytes pad | kps [NF] x 28 B | desc [NF] x 32 B | right_u [NF] f64 | depth [NF] f64
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np

from .sharding import frame_range, gather_frames

KP_BYTES, DESC_BYTES, HEAD_BYTES = 28, 32, 16


def record_bytes(n_features: int) -> int:
    return HEAD_BYTES + n_features * (KP_BYTES + DESC_BYTES + 8 + 8)


def pack_records(kps, desc, counts, right_u, depth, n_matches):
    """Packed batch results -> uint8 records [n_pairs, record_bytes] (torch tensors in, on whatever device they live on).

    kps: uint8 [2P, NF, 28], desc: uint8 [2P, NF, 32], counts: int32 [2P], right_u / depth: float64 [P, NF], n_matches: int32 [P].
    Only the LEFT image's features go into a record (slots 2p); entries past the count are zeroed so that records are canonical."""
    import torch
    P, NF = right_u.shape[0], right_u.shape[1]
    rec = torch.zeros((P, record_bytes(NF)), dtype=torch.uint8, device=kps.device)
    if P == 0:
        return rec
    n = counts[0::2].to(torch.int32)
    head = torch.stack([n, n_matches.to(torch.int32), torch.zeros_like(n), torch.zeros_like(n)], dim=1).contiguous()
    rec[:, :HEAD_BYTES] = head.view(torch.uint8).reshape(P, HEAD_BYTES)
    live = (torch.arange(NF, device=kps.device)[None, :] < n[:, None])          # [P, NF]
    o = HEAD_BYTES
    rec[:, o:o + NF * KP_BYTES] = (kps[0::2] * live[:, :, None]).reshape(P, NF * KP_BYTES)
    o += NF * KP_BYTES
    rec[:, o:o + NF * DESC_BYTES] = (desc[0::2] * live[:, :, None]).reshape(P, NF * DESC_BYTES)
    o += NF * DESC_BYTES
    rec[:, o:o + NF * 8] = (right_u.contiguous().view(torch.uint8).reshape(P, NF, 8) * live[:, :, None]).reshape(P, NF * 8)
    o += NF * 8
    rec[:, o:o + NF * 8] = (depth.contiguous().view(torch.uint8).reshape(P, NF, 8) * live[:, :, None]).reshape(P, NF * 8)
    return rec


def unpack_record(rec: np.ndarray, n_features: int) -> dict:
    """One record (uint8 [record_bytes]) -> dict(n, n_matches, kps, desc, right_u, depth) trimmed to n keypoints."""
    from ._lib import KP_DTYPE
    rec = np.ascontiguousarray(rec, np.uint8)
    n, nm = (int(v) for v in rec[:8].view(np.int32))
    o = HEAD_BYTES
    kps = rec[o:o + n_features * KP_BYTES].view(KP_DTYPE)[:n]
    o += n_features * KP_BYTES
    desc = rec[o:o + n_features * DESC_BYTES].reshape(n_features, 32)[:n]
    o += n_features * DESC_BYTES
    ru = rec[o:o + n_features * 8].view(np.float64)[:n]
    o += n_features * 8
    dp = rec[o:o + n_features * 8].view(np.float64)[:n]
    return dict(n=n, n_matches=nm, kps=kps, desc=desc, right_u=ru, depth=dp)


def run_sequence(n_frames: int, rank: int, world: int, batch_pairs: int, submit: Callable[[Sequence[int]], object],
                 collect: Callable[[object], "object"], depth: int = 3, dst: int = 0, window: int = 0, sink=None, collect_into=None,
                 force_collective: bool = False, store=None):
    """Process this rank's block of the sequence in batches of <= batch_pairs frames, `depth` batches in flight, and gather all
    records on `dst`.

    submit(frame_ids) -> handle starts a batch (host images -> device, extraction, stereo match); collect(handle) -> uint8 tensor
    [len(frame_ids), record_bytes] on the device the collective runs on, in frame order.  Batches are collected in submission order.

    window = 0: ONE gather at the end of the sequence (sharding.gather_frames).  window = W > 0: a gather per W batches
    (sharding.WindowGather) that runs while the next window is computed; on dst every finished part goes to sink(first_frame, tensor)
    (e.g. a non-blocking copy to page-locked host memory) or, without a sink, is kept and returned assembled.  collect_into(handle,
    out) packs a batch straight into a slice of the window's send buffer (no intermediate tensor).
    store (a sharding.SharedRecordStore, window > 0): the records do not cross GPUs at all -- every rank copies its windows into its rows
    of the node's shared page-locked host segment over its own PCIe link (sharding.WindowDrain) and only the 16-byte record heads are
    gathered; the first return value is then the int32 [n_frames, 4] summary (n, n_matches, 0, 0) on dst.
    Returns (records [n_frames, record_bytes] on dst / None elsewhere or when a sink took them, number of frames this rank processed)."""
    import torch
    b, e = frame_range(n_frames, rank, world)
    B = max(1, batch_pairs)
    if store is not None and window <= 0:
        raise ValueError("run_sequence: a shared record store needs window > 0")
    if window <= 0:
        chunks, flight = [], []
        for s in range(b, e, B):
            flight.append(submit(range(s, min(e, s + B))))
            if len(flight) >= max(1, depth):
                chunks.append(collect(flight.pop(0)))
        while flight:
            chunks.append(collect(flight.pop(0)))
        if chunks:
            local = torch.cat(chunks, dim=0) if len(chunks) > 1 else chunks[0]
        else:  # an empty block (more ranks than frames): shape and device from a zero-frame batch
            local = collect(submit(range(0, 0)))[:0]
        return gather_frames(local, n_frames, rank, world, dst, force_collective), e - b

    from .sharding import WindowDrain, WindowGather
    proto = collect(submit(range(0, 0)))   # zero-frame batch: record shape, dtype and device
    win = window * B
    if store is not None:
        wg = WindowDrain(n_frames, rank, world, win, tuple(proto.shape[1:]), proto.dtype, proto.device, store, dst, force_collective)
    else:
        wg = WindowGather(n_frames, rank, world, win, tuple(proto.shape[1:]), proto.dtype, proto.device, dst, sink, force_collective)
    starts = list(range(b, e, B))
    flight, nxt = [], 0

    def top_up():
        nonlocal nxt
        while nxt < len(starts) and len(flight) < max(1, depth):
            s = starts[nxt]
            flight.append((s, min(e, s + B), submit(range(s, min(e, s + B)))))
            nxt += 1

    top_up()
    for w in range(wg.n_windows):
        lo, hi = wg.local_rows(w)           # block-relative frames of this window
        buf = wg.buffer(w)
        done = lo
        while done < hi:
            s, t, h = flight.pop(0)
            assert s - b == done and t - b <= hi, "batches do not tile the window"
            out = buf[s - b - lo:t - b - lo]
            if collect_into is not None:
                collect_into(h, out)
            else:
                out.copy_(collect(h))
            done = t - b
            top_up()
        wg.push(w)
    return wg.finish(), e - b


class DeviceSequenceProcessor:
    """submit / collect for run_sequence on one GPU: host images -> orbfe_stream_submit (upload overlapped with the compute of the
    batch before) -> records packed ON THE DEVICE from the stream's result buffers (no download), ready for the RCCL gather.

    make_pair(frame_id) -> (left, right) uint8 images; prepare() builds the page-locked batch buffers of this rank's frames before
    the clock starts (a camera driver or an image decoder would write there directly)."""

    def __init__(self, ctx, make_pair, batch_pairs, fx, bf, device, torch_pack=False, content_key=None):
        """content_key(frame_id) -> hashable identity of the frame's IMAGES (synthetic sequences repeat: batches with the same content
        share one page-locked buffer instead of 0.5 GB each)"""
        self.ctx, self.make_pair, self.batch, self.fx, self.bf, self.device = ctx, make_pair, batch_pairs, fx, bf, device
        self.pinned, self.torch_pack, self.content_key, self._by_content = {}, torch_pack, content_key, {}

    def prepare(self, frame_ids):
        from ._lib import PinnedArray
        ids = list(frame_ids)
        H, W = self.ctx.height, self.ctx.width
        cache = {}
        for s in range(0, len(ids), self.batch):
            chunk = ids[s:s + self.batch]
            ck = tuple(self.content_key(f) for f in chunk) if self.content_key else None
            if ck is not None and ck in self._by_content:
                self.pinned[(chunk[0], len(chunk))] = self._by_content[ck]
                continue
            dv = int(self.ctx.cfg.device_id)
            l, r = PinnedArray((len(chunk), H, W), np.uint8, dv), PinnedArray((len(chunk), H, W), np.uint8, dv)
            for i, f in enumerate(chunk):
                k = self.content_key(f) if self.content_key else f
                if k not in cache and len(cache) < 256:
                    cache[k] = self.make_pair(f)
                l.array[i], r.array[i] = cache[k] if k in cache else self.make_pair(f)
            self.pinned[(chunk[0], len(chunk))] = (l, r)
            if ck is not None:
                self._by_content[ck] = (l, r)

    def submit(self, frame_ids):
        ids = list(frame_ids)
        if not ids:
            return None
        key = (ids[0], len(ids))
        if key not in self.pinned:
            self.prepare(ids)
        l, r = self.pinned[key]
        return (self.ctx.stream_submit(l.array, r.array, len(ids), self.fx, self.bf, None), len(ids))

    def collect(self, handle):
        import torch
        from .torch_views import batch_result_views
        nf = self.ctx.n_features
        if handle is None:
            return torch.zeros((0, record_bytes(nf)), dtype=torch.uint8, device=self.device)
        ticket, n = handle
        if self.torch_pack:   # the same records with torch ops on views of the stream's result buffer (kept as the checker of the kernel)
            self.ctx.stream_wait(ticket)
            views = batch_result_views(self.ctx.stream_device_results(ticket, n), n, nf, self.device)
            rec = pack_records(*views)
            torch.cuda.current_stream(self.device).synchronize()   # the buffer is re-used three submits later
            return rec
        rec = torch.empty((n, self.ctx.record_bytes()), dtype=torch.uint8, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()       # (the allocation is visible to the library's stream)
        self.ctx.stream_pack_records(ticket, n, rec.data_ptr())    # the library's pack kernel, straight into the tensor
        return rec

    def collect_into(self, handle, out):
        """Pack the records of a batch straight into `out` ([n, record_bytes] uint8, contiguous, on this device; e.g. a slice of a window's
        send buffer whose previous use is known to be complete).  Returns when the records are in place."""
        if handle is None:
            return
        ticket, n = handle
        assert out.is_contiguous() and tuple(out.shape) == (n, self.ctx.record_bytes())
        self.ctx.stream_pack_records(ticket, n, out.data_ptr())
