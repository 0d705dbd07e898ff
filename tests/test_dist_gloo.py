"""world_size-2 test of the N>1 path on CPU (gloo): contiguous frame sharding + the sequence-level gather.
Per-frame records are produced by the oracle on small frames so the gathered sequence can be compared with a
single-process run."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records(frames):
    """fixed-size per-frame record [n_features, 8] int32: n | x,y bits | octave | first descriptor words"""
    from oracle.pyoracle import Oracle
    from orb_slam2_ros2_amd import synth
    import torch
    orc = Oracle()
    out = []
    for f in frames:
        L, _ = synth.stereo_pair(f, 320, 200, n_rect=60)
        k, d = orc.extractor(L, n_features=100, n_levels=3).extract()
        rec = np.zeros((100, 8), np.int32)
        rec[:, 0] = len(k)
        rec[:len(k), 1] = k["x"].view(np.int32)
        rec[:len(k), 2] = k["y"].view(np.int32)
        rec[:len(k), 3] = k["octave"]
        rec[:len(k), 4:8] = d[:, :16].copy().view(np.int32)
        out.append(rec)
    return torch.from_numpy(np.stack(out)) if out else torch.zeros((0, 100, 8), dtype=torch.int32)


def _worker(rank, world, port, n_frames, q):
    import torch.distributed as dist
    from orb_slam2_ros2_amd.sharding import frame_range, gather_frames
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = frame_range(n_frames, rank, world)
    local = _records(range(b, e))
    full = gather_frames(local, n_frames, rank, world, dst=0)
    if rank == 0:
        q.put(full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 4])
def test_two_rank_gather_equals_single_process_sequence(n_frames):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _records(range(n_frames)).numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)
    assert (got[:, 0, 0] > 0).all()
