#!/bin/bash
# Runs on the GPU box: SQ counters of k_fast for library variants (tools/exp/ab_build.sh): tools/exp/fast_sq.sh tight pairs2
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for v in "$@"; do
  cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
    rm -rf /tmp/prof_sq
    rocprofv3 --pmc $set --output-format csv -d /tmp/prof_sq -- python3 bench.py --steps 4 --warmup 1 --prewarm-seconds 0.2 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' > /dev/null 2> /tmp/prof_sq.err || tail -3 /tmp/prof_sq.err
    F=$(find /tmp/prof_sq -name '*counter_collection.csv' | head -1)
    python3 - "$F" "$v" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1]
    if "k_fast" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, d in acc.items():
    w = d.get("SQ_WAVES", 1.0)
    print(sys.argv[2], k, "launches", n[k], {c: round(v / w, 1) for c, v in d.items() if c != "SQ_WAVES"}, "waves", int(w))
PY
  done
done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
