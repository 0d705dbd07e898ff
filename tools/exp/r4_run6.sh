set -e
timeout -k 10 500 python3 tests/rccl_one_rank.py 40 16 2>&1 | tail -3
timeout -k 10 900 python3 bench.py > gpurun_out/r4_bench_a.json 2> gpurun_out/r4_bench_a.err || { tail -20 gpurun_out/r4_bench_a.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4_bench_a.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"])
print("sequence", json.dumps(d["sequence"])[:900])
print("ba", {k: (v.get("median_ms") if isinstance(v, dict) else v) for k, v in d["ba"].items() if k.startswith("local")})
print("latency", json.dumps(d.get("latency"))[:600])
PY
