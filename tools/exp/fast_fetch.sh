#!/bin/bash
# Runs on the GPU box: FETCH_SIZE of k_fast per sweep for library variants: tools/exp/fast_fetch.sh dry g8
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for v in "$@"; do
  cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so
  rm -rf /tmp/prof_f
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- python3 bench.py --steps 4 --warmup 1 --prewarm-seconds 0.2 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' > /dev/null 2> /tmp/prof_f.err || tail -3 /tmp/prof_f.err
  F=$(find /tmp/prof_f -name '*counter_collection.csv' | head -1)
  python3 - "$F" "$v" <<'PY'
import csv, sys, collections
acc = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    if r["Counter_Name"] == "FETCH_SIZE": acc[k] += float(r["Counter_Value"]); n[k] += 1
for k in ("k_fast", "k_blur", "k_resize_regions"):
    print(sys.argv[2], k, "launches", n[k], "FETCH_SIZE kB per launch", round(acc[k] / max(n[k], 1)), "-> per 8 launches x2 (guide):", round(acc[k] / max(n[k], 1) * 8 * 2 * 1024 / 1e9, 3) if k == "k_fast" else "")
PY
done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
