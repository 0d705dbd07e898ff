#!/usr/bin/env python3
"""Host <-> device copy rates of this box for the transfer sizes of one bench step (page-locked memory, HIP events):
upload alone, download alone, both at once; plus where the GPU and this process sit (NUMA)."""
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    dev = torch.device("cuda", 0)
    up_bytes, down_bytes = 512 * 933232, 512 * 272012
    h_up = torch.empty(up_bytes, dtype=torch.uint8).pin_memory()
    h_dn = torch.empty(down_bytes, dtype=torch.uint8).pin_memory()
    d_up = torch.empty(up_bytes, dtype=torch.uint8, device=dev)
    d_dn = torch.empty(down_bytes, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def run(up, down, n=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if up:
                with torch.cuda.stream(s1):
                    d_up.copy_(h_up, non_blocking=True)
            if down:
                with torch.cuda.stream(s2):
                    h_dn.copy_(d_dn, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        return dt
    run(True, True, 3)
    out = {}
    t = run(True, False)
    out["h2d_alone_GBps"] = up_bytes / t / 1e9
    t = run(False, True)
    out["d2h_alone_GBps"] = down_bytes / t / 1e9
    t = run(True, True)
    out["both_ms"] = t * 1e3
    out["both_h2d_GBps"] = up_bytes / t / 1e9
    out["both_d2h_GBps"] = down_bytes / t / 1e9
    out["gpu_numa_nodes"] = {p: open(p).read().strip() for p in glob.glob("/sys/class/drm/card*/device/numa_node")}
    out["cpu_affinity"] = sorted(os.sched_getaffinity(0))[:4] + ["..."] + [len(os.sched_getaffinity(0))]
    try:
        out["numa_nodes_cpulist"] = {p: open(p).read().strip() for p in glob.glob("/sys/devices/system/node/node*/cpulist")}
    except Exception:
        pass
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
