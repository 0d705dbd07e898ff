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


def gather_frames(local, n_frames: int, rank: int, world: int, dst: int = 0):
    """Gather per-frame records to `dst`.

    local: tensor [n_local, ...] holding this rank's frames frame_range(n_frames, rank, world) in order.
    Returns on dst a tensor [n_frames, ...] in sequence order, elsewhere None.
    """
    import torch
    import torch.distributed as dist

    b, e = frame_range(n_frames, rank, world)
    if local.shape[0] != e - b:
        raise ValueError(f"rank {rank} holds {local.shape[0]} frames, expected {e - b}")
    if world == 1:
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
