#!/bin/bash
# Runs on the GPU box: tools/exp/sq_any.sh "<SQ counters, one pass: <= 8>" [kernel substring] -- per kernel and wave, from a short bench.py run
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
rm -rf /tmp/prof_any
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES $1 --output-format csv -d /tmp/prof_any -- python3 bench.py --steps 4 --warmup 1 --prewarm-seconds 0.2 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' --content-steps 0 > /dev/null 2> /tmp/prof_any.err || tail -5 /tmp/prof_any.err
F=$(find /tmp/prof_any -name '*counter_collection.csv' | head -1)
python3 - "$F" "${2:-k_}" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    if sys.argv[2] not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(acc.items()):
    w = max(d.get("SQ_WAVES", 1.0), 1.0)
    print(f"{k:18s} waves {w:10.0f}  per wave: " + "  ".join(f"{c} {v / w:.1f}" for c, v in sorted(d.items()) if c != "SQ_WAVES"))
PY
