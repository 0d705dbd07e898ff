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


# ---- the sequence driver bench.py --sequence uses (orb_slam2_ros2_amd/sequence.py), world size 2 over gloo ---------------------------
NF_SMALL = 100


def _small_frame_processor():
    """submit / collect for run_sequence on the CPU: the oracle produces a frame's results on small images, packed with the same
    pack_records the device path uses (CPU tensors, so gloo can gather them)."""
    import torch
    from oracle.pyoracle import Oracle
    from orb_slam2_ros2_amd import synth
    from orb_slam2_ros2_amd._lib import KP_DTYPE
    from orb_slam2_ros2_amd.sequence import pack_records
    orc = Oracle()

    def submit(frame_ids):
        return list(frame_ids)

    def collect(ids):
        P = len(ids)
        kps = np.zeros((2 * P, NF_SMALL), KP_DTYPE)
        desc = np.zeros((2 * P, NF_SMALL, 32), np.uint8)
        cnt = np.zeros(2 * P, np.int32)
        ru = np.full((P, NF_SMALL), -1.0)
        dp = np.full((P, NF_SMALL), -1.0)
        nm = np.zeros(P, np.int32)
        for i, f in enumerate(ids):
            L, R = synth.stereo_pair(f, 320, 200, n_rect=60)
            r = orc.stereo_frame(L, R, n_features=NF_SMALL, n_levels=3, fx=300.0, bf=120.0)
            nl, nr = len(r["lk"]), len(r["rk"])
            kps[2 * i, :nl], desc[2 * i, :nl], kps[2 * i + 1, :nr], desc[2 * i + 1, :nr] = r["lk"], r["ld"], r["rk"], r["rd"]
            kps[2 * i, nl:]["x"] = 123.0   # stale entries past the count, as the device arrays hold them: a record must not carry them
            cnt[2 * i], cnt[2 * i + 1], nm[i] = nl, nr, r["n_matches"]
            ru[i, :nl], dp[i, :nl] = r["right_u"], r["depth"]
        t = torch.from_numpy
        return pack_records(t(kps.view(np.uint8).reshape(2 * P, NF_SMALL, 28)), t(desc), t(cnt), t(ru), t(dp), t(nm))
    return submit, collect


def _seq_worker(rank, world, port, n_frames, batch, q):
    import torch.distributed as dist
    from orb_slam2_ros2_amd.sequence import run_sequence
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    submit, collect = _small_frame_processor()
    rec, n_local = run_sequence(n_frames, rank, world, batch, submit, collect)
    if rank == 0:
        q.put((rec.numpy(), n_local))
    else:
        assert rec is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,batch", [(7, 2), (1, 4)])
def test_sequence_driver_two_ranks_equals_one_rank(n_frames, batch):
    """run_sequence (frame_range per rank, batches in flight, the gather of the per-frame RECORDS) with two gloo ranks against the same
    function with one; n_frames = 1 leaves rank 1 with an empty block."""
    import torch.multiprocessing as mp
    from orb_slam2_ros2_amd.sequence import record_bytes, run_sequence, unpack_record
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seq_worker, args=(r, 2, port, n_frames, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, n0 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    submit, collect = _small_frame_processor()
    ref, n_all = run_sequence(n_frames, 0, 1, batch, submit, collect)
    ref = ref.numpy()
    assert n_all == n_frames and n0 == (n_frames + 1) // 2
    assert got.shape == ref.shape == (n_frames, record_bytes(NF_SMALL)) and np.array_equal(got, ref)
    u = unpack_record(got[0], NF_SMALL)
    assert u["n"] > 10 and (u["kps"]["size"] == 7).all() and u["n_matches"] == int((u["right_u"] >= 0).sum())


# ---- the windowed gather (SURVEY 8e "per window of W frames"): a gather per window while the next one is computed -------------------
def _win_worker(rank, world, port, n_frames, batch, window, use_sink, q):
    import torch
    import torch.distributed as dist
    from orb_slam2_ros2_amd.sequence import record_bytes, run_sequence
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    submit, collect = _small_frame_processor()
    host = torch.full((n_frames, record_bytes(NF_SMALL)), 0xEE, dtype=torch.uint8)
    calls = []

    def sink(first, t):
        calls.append((first, t.shape[0]))
        host[first:first + t.shape[0]].copy_(t)

    rec, n_local = run_sequence(n_frames, rank, world, batch, submit, collect, window=window, sink=sink if use_sink else None,
                                force_collective=(world == 1))
    if rank == 0:
        q.put(((host if use_sink else rec).numpy(), n_local, calls))
    else:
        assert rec is None and not calls
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,batch,window,use_sink", [(2, 7, 2, 1, True),    # per 4: two windows of 2, rank 1 ends in a ragged one
                                                                 (2, 9, 2, 2, False),   # per 5: windows of 4, rank 1 has nothing in the second
                                                                 (2, 1, 4, 1, True),    # rank 1 has an empty block
                                                                 (1, 5, 2, 1, True),    # one rank THROUGH the collective (force_collective)
                                                                 (8, 19, 2, 1, True)])  # eight ranks, `--sequence-exchange gather`: per 3, rank 6 holds 1 frame, rank 7 none
def test_windowed_gather_equals_single_gather(world, n_frames, batch, window, use_sink):
    import torch.multiprocessing as mp
    from orb_slam2_ros2_amd.sequence import record_bytes, run_sequence
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_win_worker, args=(r, world, port, n_frames, batch, window, use_sink, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, n0, calls = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    submit, collect = _small_frame_processor()
    ref, _ = run_sequence(n_frames, 0, 1, batch, submit, collect)
    assert got.shape == (n_frames, record_bytes(NF_SMALL)) and np.array_equal(got, ref.numpy())
    assert n0 == (n_frames + world - 1) // world
    if use_sink:   # every frame delivered exactly once
        seen = sorted(f for first, n in calls for f in range(first, first + n))
        assert seen == list(range(n_frames))


def test_window_gather_rejects_out_of_order_and_unfinished_windows():
    import torch
    from orb_slam2_ros2_amd.sharding import WindowGather
    wg = WindowGather(5, 0, 1, 2, (3,), torch.uint8, torch.device("cpu"))
    assert wg.n_windows == 3 and wg.local_rows(2) == (4, 5)
    with pytest.raises(ValueError):
        wg.push(1)
    wg.buffer(0)[:] = 1
    wg.push(0)
    with pytest.raises(ValueError):
        wg.finish()
    wg.buffer(1)[:] = 2
    wg.push(1)
    wg.buffer(2)[:] = 3
    wg.push(2)
    out = wg.finish()
    assert out.shape == (5, 3) and out[:, 0].tolist() == [1, 1, 2, 2, 3]


# ---- the shared host segment (one node = one host memory): every rank drains its own windows, only the record heads are gathered ----
def _drain_worker(rank, world, port, n_frames, batch, window, name, q):
    import torch.distributed as dist
    from orb_slam2_ros2_amd.sequence import record_bytes, run_sequence
    from orb_slam2_ros2_amd.sharding import SharedRecordStore
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    store = SharedRecordStore(name, n_frames, record_bytes(NF_SMALL), create=False)   # the parent created it before any rank started
    submit, collect = _small_frame_processor()
    summary, n_local = run_sequence(n_frames, rank, world, batch, submit, collect, window=window, store=store, force_collective=(world == 1))
    if rank == 0:
        q.put((summary.numpy(), n_local))
    else:
        assert summary is None
    store.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,batch,window", [(4, 11, 2, 1),   # per 3: the last rank holds 2 frames (uneven blocks, ragged windows)
                                                         (4, 2, 2, 1),    # per 1: ranks 2 and 3 hold nothing
                                                         (2, 9, 2, 2),
                                                         (1, 5, 2, 1),    # one rank through the collective
                                                         (8, 37, 2, 1),   # the node's shape: per 5, rank 7 holds 2 frames, ragged last windows
                                                         (8, 5, 2, 2)])   # per 1: ranks 5, 6, 7 hold nothing, windows wider than the blocks
def test_shared_host_segment_drain_equals_single_gather(world, n_frames, batch, window):
    """sharding.WindowDrain: the records land in a POSIX shared-memory segment (each rank writes its own rows), the collective carries
    only (n, n_matches) per frame -- against the records of a single-process run."""
    import uuid
    import torch.multiprocessing as mp
    from orb_slam2_ros2_amd.sequence import record_bytes, run_sequence, unpack_record
    from orb_slam2_ros2_amd.sharding import SharedRecordStore
    name = f"orbfe_test_{uuid.uuid4().hex[:12]}"
    store = SharedRecordStore(name, n_frames, record_bytes(NF_SMALL), create=True)
    try:
        store.array[:] = 0xEE
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_drain_worker, args=(r, world, port, n_frames, batch, window, name, q)) for r in range(world)]
        for p in procs:
            p.start()
        summary, n0 = q.get(timeout=240)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        submit, collect = _small_frame_processor()
        ref, _ = run_sequence(n_frames, 0, 1, batch, submit, collect)
        ref = ref.numpy()
        got = np.array(store.array)
        assert got.shape == ref.shape and np.array_equal(got, ref)
        assert n0 == min(n_frames, (n_frames + world - 1) // world)
        for f in range(n_frames):
            u = unpack_record(ref[f], NF_SMALL)
            assert tuple(summary[f]) == (u["n"], u["n_matches"], 0, 0)
    finally:
        store.close()
    assert not os.path.exists(os.path.join("/dev/shm", name))
