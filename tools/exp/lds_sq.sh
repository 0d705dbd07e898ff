#!/bin/bash
# Runs on the GPU box: LDS counters of EVERY kernel of the step for library variants (tools/exp/ab_build.sh; "cur" = the library in place):
#   tools/exp/lds_sq.sh cur fs0     ->  per kernel and wave: LDS instructions, SQ_LDS_IDX_ACTIVE (LDS-array cycles), SQ_LDS_BANK_CONFLICT
#   (extra cycles), and the LDS activity per CU and step in ms at 2.2 GHz (one LDS serves the CU's resident waves)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for v in "$@"; do
  [ "$v" = cur ] || cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so
  rm -rf /tmp/prof_sq
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d /tmp/prof_sq -- python3 bench.py --steps 4 --warmup 1 --prewarm-seconds 0.2 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' --content-steps 0 > /dev/null 2> /tmp/prof_sq.err || tail -3 /tmp/prof_sq.err
  F=$(find /tmp/prof_sq -name '*counter_collection.csv' | head -1)
  python3 - "$F" "$v" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    if not k.startswith("k_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
steps = max(n["k_quadtree"], 1)
for k, d in sorted(acc.items()):
    w = max(d.get("SQ_WAVES", 1.0), 1.0)
    idx, bc = d["SQ_LDS_IDX_ACTIVE"], d["SQ_LDS_BANK_CONFLICT"]
    print(f"{sys.argv[2]:6s} {k:18s} waves/step {w / steps:9.0f}  per wave: LDS instr {d['SQ_INSTS_LDS'] / w:7.1f}  VALU {d['SQ_INSTS_VALU'] / w:8.1f}  LDS cycles {idx / w:8.1f}  "
          f"of them bank conflicts {bc / w:7.1f} ({100 * bc / max(idx, 1):4.1f} %)  LDS activity per CU and step {idx / steps / 256 / 2.2e6:6.3f} ms")
PY
done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
