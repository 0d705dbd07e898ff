#!/usr/bin/env python3
"""Where does the host-image stream lose time?  Variants of the host_io leg of bench.py on one box:
   full | no download (out = NULL) | device-resident step with an independent upload / download running beside it."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from orb_slam2_ros2_amd import synth
from orb_slam2_ros2_amd._lib import Context, PinnedArray

W, H, FX, BF = 1241, 376, 718.856, 718.856 * 0.537166


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    if "dist" in sys.argv[2:]:
        import torch.distributed as dist  # noqa: F401
    if "avail" in sys.argv[2:]:
        torch.cuda.is_available()
    if "setdev" in sys.argv[2:]:
        torch.cuda.set_device(0)
    if "streams" in sys.argv[2:]:
    if "explicit" in sys.argv[2:]:
        global Context
        _C = Context
        Context = lambda w, h, max_images: _C(w, h, 2000, 8, 1.2, 20, 7, device_id=0, max_images=max_images)
    steps = 40
    fr = [synth.stereo_pair(i) for i in range(16)]
    left = np.stack([fr[i % 16][0] for i in range(B)])
    right = np.stack([fr[i % 16][1] for i in range(B)])
    if "early_dl" in sys.argv[2:]:
        e1, e2 = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    ctx = Context(W, H, max_images=2 * B)
    if "benchleg" in sys.argv[2:]:
        import bench
        import torch.distributed as dist

        def _sa():
            ctx.sync()
            torch.cuda.synchronize()
        r = bench.host_io_leg(ctx, left, right, B, 30, [None] * B, 1, _sa, dist, torch, torch.device("cuda", 0))
        print(json.dumps({"benchleg_ms": r["ms_per_step"]}))
        ctx.close()
        return
    pins = []
    for _ in range(3):
        l, r = PinnedArray(left.shape, np.uint8), PinnedArray(right.shape, np.uint8)
        l.array[...] = left
        r.array[...] = right
        pins.append((l, r))
    outs = [ctx.alloc_batch_results(B) for _ in range(3)]
    res = {"env_HSA_ENABLE_SDMA": os.environ.get("HSA_ENABLE_SDMA"), "prelude": sys.argv[2:]}
    # preludes: what bench.py does before its host_io leg (to find which of them slows the leg down)
    dl0, dr0 = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()

    def dsteps(n):
        for _ in range(n):
            ctx.stereo_batch_device(dl0.data_ptr(), dr0.data_ptr(), W, W * H, B, FX, BF)
        ctx.sync()
    if "dev" in sys.argv[2:]:
        dsteps(60)
    if "prof1" in sys.argv[2:]:
        ctx.profile_enable(1); dsteps(5); ctx.profile_read(); ctx.profile_enable(0)
    if "prof2" in sys.argv[2:]:
        ctx.profile_enable(4); dsteps(20); ctx.profile_read(); ctx.profile_enable(0)
    if "fetch" in sys.argv[2:]:
        dsteps(2); ctx.fetch_batch(0, 2 * B); ctx.fetch_stereo_batch(0, B)
    if "cand" in sys.argv[2:]:
        dsteps(2); [ctx.debug_candidates(0, l) for l in range(8)]
    if "torchops" in sys.argv[2:]:
        x = torch.empty(B, 4, dtype=torch.int32, device="cuda"); x[:, 0] = 1; torch.cuda.synchronize()

    def run(n, with_out, depth=3):
        tk = []
        for k in range(n):
            tk.append(ctx.stream_submit(pins[k % 3][0].array, pins[k % 3][1].array, B, FX, BF, outs[k % 3] if with_out else None))
            if k >= depth - 1:
                ctx.stream_wait(tk[k - depth + 1])
        for t in tk[-depth:]:
            ctx.stream_wait(t)
        ctx.sync()

    for name, wo, depth in (("full_depth3", True, 3), ("no_download_depth3", False, 3)):
        run(4, wo, depth)
        t0 = time.perf_counter()
        run(steps, wo, depth)
        res[name + "_ms"] = (time.perf_counter() - t0) / steps * 1e3

    if len(sys.argv) > 2:
        print(json.dumps(res)); ctx.close(); return
    # device-resident step alone, then with an unrelated upload (and download) looping on other streams
    dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    def dev_steps(n):
        for _ in range(n):
            ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), W, W * H, B, FX, BF)
        ctx.sync()
    dev_steps(5)
    t0 = time.perf_counter(); dev_steps(steps); res["device_step_ms"] = (time.perf_counter() - t0) / steps * 1e3
    hp = torch.empty(2 * left.nbytes, dtype=torch.uint8).pin_memory()
    dd = torch.empty(2 * left.nbytes, dtype=torch.uint8, device="cuda")
    hd = torch.empty(B * 272012, dtype=torch.uint8).pin_memory()
    d2 = torch.empty(B * 272012, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, up, down in (("device_step_with_upload_ms", True, False), ("device_step_with_updown_ms", True, True)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if up:
                with torch.cuda.stream(s1):
                    dd.copy_(hp, non_blocking=True)
            if down:
                with torch.cuda.stream(s2):
                    hd.copy_(d2, non_blocking=True)
            ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), W, W * H, B, FX, BF)
        ctx.sync()
        t_dev = time.perf_counter() - t0
        torch.cuda.synchronize()
        res[name] = t_dev / steps * 1e3
        res[name.replace("_ms", "_all_done_ms")] = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps(res, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
