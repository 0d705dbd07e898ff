"""bench.py's shared pieces: the workload constants of BASELINE.json's metric, SURVEY 8(d)'s byte model, the rank launcher and the small
timing helpers of the extra legs.  (bench.py was one 1350-line file until r6: split by leg, the emitted line unchanged.)"""
from __future__ import annotations

import json  # noqa: F401  (re-exported for the leg modules)
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, NFEAT, NLEVELS, SCALE, TH_HI, TH_LO = 1241, 376, 2000, 8, 1.2, 20, 7
FX, BF = 718.856, 718.856 * 0.537166  # config/kitti_config_00.yaml: Camera.fx, Camera.bl
HBM_PEAK_GBPS = 8000.0
PCIE_PEAK_GBPS = 63.0   # PCIe 5.0 x16, one direction (MI355X_MICROARCH.md)
BENCH_PY = os.path.join(ROOT, "bench.py")


def algorithmic_bytes(ctx, n_cand_per_image):
    """algorithmic bytes per image for each kernel, and per stereo pair in total (20 012 776 B for the KITTI shape at
    2000 features -- 8(d)'s own figure, which has no quadtree entry).  The quadtree only touches the candidate records: 4 B per
    candidate read + 4 B per selected keypoint written; that term prices the quadtree STAGE (returned separately) and is NOT part of
    the per-pair total."""
    P = sum(ctx.level_info(l).width * ctx.level_info(l).height for l in range(NLEVELS))
    S0 = W * H
    K = NFEAT
    per_image = {
        "resize": S0 + (P - S0),            # read level 0, write levels 1..7
        "blur": 2 * P,                      # read + write every plane
        "fast": P,                          # read every plane (+ candidate records, not counted)
        "orient_brief": K * (749 + 512) + K * 60,
    }
    per_pair_match = 2 * K * 32 + 2 * K * 28 + K * 12 * 121 + K * 16
    per_pair = 2 * sum(per_image.values()) + per_pair_match
    per_image["quadtree"] = 4 * n_cand_per_image + 4 * K   # stage pricing only (after the total)
    return per_image, per_pair_match, per_pair


class _StdoutToStderr:
    """RCCL prints a version banner on file descriptor 1 when its first communicator comes up; the contract is ONE JSON line on stdout.
    While this is active, everything written to fd 1 -- by Python or by a native library -- goes to stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def spawn_ranks(n: int) -> int:
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # poll: when one rank dies (a missing GPU, a failed rendezvous) the others would wait in the collective init for ever -- end them
    import threading
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            time.sleep(2.0)   # let the others fail by themselves with their own message first
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()   # exactly the children started above
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.05)
    rd.join(timeout=10)
    sys.stdout.write("".join(x or "" for x in buf))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _stats_ms(f, n, warm=5):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    a = np.sort(np.array(ts))
    return {"median_ms": float(np.median(a)), "p99_ms": float(a[min(len(a) - 1, int(0.99 * len(a)))]), "n": n}


def _kernel_us(ctx, stage, f, n=30):
    """mean device time of the kernels of one call (HIP events on the library's stream around the kernels only: inputs already
    uploaded, results not yet downloaded -- the device-resident figure)"""
    f()
    ctx.profile_enable(1)
    ctx.profile_read()
    for _ in range(n):
        f()
    ms, k = ctx.profile_read()[stage]
    ctx.profile_enable(0)
    return (ms / n) * 1e3 if k else None


def _oracle_fast():
    from oracle import pyoracle   # the CPU checker, timed beside the device on ONE host core (kind: "port")
    return pyoracle.Oracle(pyoracle.build(fast=True, out_dir=os.path.join("/tmp", f"orb_oracle_{os.getuid()}")))


def _cpu_ms(f, budget_s=1.5, max_n=5):
    f()
    ts = []
    t_end = time.perf_counter() + budget_s
    while len(ts) < max_n and (not ts or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def _gen_pair(f):
    from orb_slam2_ros2_amd import synth
    return synth.stereo_pair(f, W, H)


def generate_pairs(frames, workers=None):
    """[(left, right)] of synth.stereo_pair for every frame index of `frames`, on a pool of FORKED workers (a frame is ~0.1 s of numpy
    that mostly holds the GIL: threads are slower than one).  Must run before this process touches the GPU -- bench.py calls it before it
    imports torch."""
    import multiprocessing as mp
    frames = list(frames)
    if workers is None:
        workers = max(1, min(len(os.sched_getaffinity(0)) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))), 16))
    if workers <= 1 or len(frames) < 4:
        return [_gen_pair(f) for f in frames]
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_gen_pair, frames, chunksize=4)


def generate_pairs_child(frames):
    """{frame: (left, right)} through tools/gen_frames.py run as a CHILD process with a clean environment (no profiler preload): the child
    forks the worker pool, this process never forks -- safe whatever this process has already loaded (bench.py under rocprofv3 has the
    profiler's library, which initialises the GPU before Python starts; a GPU test session holds a context)."""
    import subprocess
    import tempfile
    frames = sorted(set(frames))
    out = {}
    if not frames:
        return out
    runs, a = [], 0   # maximal runs of consecutive indices
    for i in range(1, len(frames) + 1):
        if i == len(frames) or frames[i] != frames[i - 1] + 1:
            runs.append((frames[a], i - a))
            a = i
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    for first, count in runs:
        with tempfile.TemporaryDirectory(prefix="orbfe_frames_", dir=shm) as td:
            npy = os.path.join(td, "f.npy")
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_frames.py"), str(first), str(count), npy], check=True, env=env, timeout=1800)
            arr = np.load(npy)
        for k in range(count):
            out[first + k] = (arr[k, 0], arr[k, 1])
    return out


def golden_digests(frames):
    """the oracle's digests (prefixes) of synthetic frames: tests/golden/golden_v5.json (tools/make_golden_v5.py: every frame of the 4541-pair
    sequence); None for a frame the fixture does not hold"""
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v5.json")))
    return [g["pairs"][f] if 0 <= f < len(g["pairs"]) else None for f in frames], int(g["hex_chars"])
