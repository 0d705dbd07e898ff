"""The N > 1 path rehearsed on the ONE GPU of the box: `bench.py --gpus 2` with both ranks on cuda:0 and the collectives
over gloo (ORBFE_BENCH_ONE_DEVICE / ORBFE_BENCH_BACKEND: the bench's own test hooks) -- the same control flow as the 2-GPU RCCL run the pool
cannot host: rank launcher (children started before anything touches the GPU), per-rank frame blocks, weak-scaling step loop with its
summary gather, and the sequence job through BOTH exchanges, whose record stores must be byte-identical.  No scaling number comes out of
this (two ranks share one GPU); a real 2 / 4 / 8-GPU curve is the driver's to measure."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_two_gloo_ranks_on_one_gpu():
    env = dict(os.environ, ORBFE_BENCH_ONE_DEVICE="1", ORBFE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "32", "--steps", "4", "--warmup", "1", "--prewarm-seconds", "0.1",
           "--cpu-seconds", "0", "--host-io-steps", "3", "--sequence-leg", "301", "--sequence-unique", "24", "--content-steps", "0", "--legs", ""]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, "ONE JSON line on stdout is the contract"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 4
    assert d["verified_pairs"] == 64                      # 32 pairs per rank, every one equal to the oracle's digest (golden_v5)
    assert d["config"]["distinct_pairs_per_step"] == 32
    seq = d["sequence"]
    assert seq["frames"] == 301 and seq["exchange"]["communicator_nranks"] == 2 and seq["collective_executed"]
    assert len(seq["exchange"]["per_rank_pairs_per_s"]) == 2 and min(seq["exchange"]["per_rank_pairs_per_s"]) > 0
    a, b = seq["exchange"], seq["other_exchange"]["exchange"]
    assert {a["exchange"], b["exchange"]} == {"shared", "gather"}
    assert a["records_sha256"] == b["records_sha256"] and len(a["records_sha256"]) == 64   # both exchanges delivered the same 301 records
    assert a["host_bytes"] == b["host_bytes"] == 301 * seq["result_bytes"] // 301
    assert d["host_io"]["verified_pairs"] > 0
