"""Child process of tests/test_gpu_rccl.py: ONE rank, backend nccl (= RCCL on ROCm), real packed frame records pushed through the
COLLECTIVE branch of the sequence-level exchange -- sharding.gather_frames(force_collective=True), the windowed gather
(sharding.WindowGather) with a ragged last window draining into page-locked host memory, and the summary gather bench.py makes --
in the same process as liborbfe_hip.so.  Prints "RCCL_ONE_RANK_OK <frames> <windows>" on success."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_frames, batch = int(sys.argv[1]), int(sys.argv[2])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s.getsockname()[1])
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    from orb_slam2_ros2_amd import synth
    from orb_slam2_ros2_amd._lib import Context
    from orb_slam2_ros2_amd.sequence import DeviceSequenceProcessor, record_bytes, run_sequence
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    ctx = Context(1241, 376, max_images=2 * batch)
    FX, BF = 718.856, 718.856 * 0.537166
    proc = DeviceSequenceProcessor(ctx, lambda f: synth.stereo_pair(f % 8), batch, FX, BF, dev, content_key=lambda f: f % 8)
    proc.prepare(range(n_frames))
    plain, n0 = run_sequence(n_frames, 0, 1, batch, proc.submit, proc.collect)                          # no collective
    one, n1 = run_sequence(n_frames, 0, 1, batch, proc.submit, proc.collect, force_collective=True)     # ONE RCCL gather at the end
    assert n0 == n1 == n_frames and one.data_ptr() != plain.data_ptr()
    plain_h, one_h = plain.cpu().numpy(), one.cpu().numpy()
    assert plain_h.shape == (n_frames, record_bytes(ctx.n_features)) and np.array_equal(one_h, plain_h), "single gather differs"
    host = torch.full((n_frames, record_bytes(ctx.n_features)), 0xEE, dtype=torch.uint8).pin_memory()
    calls = []

    def sink(first, t):
        calls.append((first, int(t.shape[0])))
        host[first:first + t.shape[0]].copy_(t, non_blocking=True)

    rec, _ = run_sequence(n_frames, 0, 1, batch, proc.submit, proc.collect, window=1, sink=sink, collect_into=proc.collect_into,
                          force_collective=True)                                                        # an RCCL gather per window
    torch.cuda.synchronize()
    assert rec is None and np.array_equal(host.numpy(), plain_h), "windowed gather differs"
    n_windows = (n_frames + batch - 1) // batch
    assert calls == [(w * batch, min(batch, n_frames - w * batch)) for w in range(n_windows)], calls
    # the same windows WITHOUT collect_into: the send buffer is filled by an asynchronous copy on torch's current stream, which the
    # gather's side stream has to wait for
    host.fill_(0xEE)
    rec, _ = run_sequence(n_frames, 0, 1, batch, proc.submit, proc.collect, window=1, sink=sink, force_collective=True)
    torch.cuda.synchronize()
    assert rec is None and np.array_equal(host.numpy(), plain_h), "windowed gather (copy_ fallback) differs"
    # one node = one host memory: the records drained into a page-locked POSIX shared-memory segment (hipHostRegister) by the rank
    # itself, RCCL carries the 16-byte record heads only (sharding.WindowDrain)
    from orb_slam2_ros2_amd.sharding import SharedRecordStore
    store = SharedRecordStore(f"orbfe_rccl1_{os.getpid()}", n_frames, record_bytes(ctx.n_features), create=True)
    try:
        store.array[:] = 0xEE
        store.pin()
        assert store.pinned and store.tensor.is_pinned()
        summary, _ = run_sequence(n_frames, 0, 1, batch, proc.submit, proc.collect, window=1, collect_into=proc.collect_into,
                                  force_collective=True, store=store)
        assert np.array_equal(np.array(store.array), plain_h), "shared-segment drain differs"
        assert np.array_equal(summary.numpy(), np.ascontiguousarray(plain_h[:, :16]).view(np.int32)), "gathered record heads differ"
    finally:
        store.close()
    # the per-pair summary gather of bench.py's step loop
    summary = torch.arange(4 * 16, dtype=torch.int32, device=dev).reshape(16, 4)
    outs = [torch.empty_like(summary)]
    dist.gather(summary, outs, dst=0)
    assert torch.equal(outs[0], summary)
    # and the library still answers after RCCL has used the device
    (lk, _), _ = ctx.extract_batch(list(synth.stereo_pair(0)))
    assert len(lk) > 1000
    ctx.close()
    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK", n_frames, n_windows)


if __name__ == "__main__":
    main()
