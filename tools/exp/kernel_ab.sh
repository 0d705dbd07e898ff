#!/bin/bash
# runs on the GPU box: tools/exp/kernel_ab.sh "<variants>" [pairs] -- per-kernel average durations (rocprofv3 kernel trace) of each
# variant library with every overlap switched off, so a kernel's duration is its own
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
export ORBFE_PIPELINE_STEREO=0 ORBFE_OVERLAP_BLUR=0 ORBFE_FAST_SIDE_FROM=0
for v in $1; do
  cp $R/tools/exp/libs/liborbfe_$v.so $R/orb_slam2_ros2_amd/liborbfe_hip.so
  rm -rf /tmp/kab_$v
  timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kab_$v -- python3 $R/tools/stage_times.py ${2:-512} > /tmp/kab_$v.log 2>&1
  db=$(find /tmp/kab_$v -name '*.db' | head -1)
  f=/tmp/kab_$v.csv
  python3 $R/tools/kernel_stats_from_db.py $db > $f
  echo "== $v"
  python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith('"#')))
out = {}
for r in rows:
    m = re.search(r'(k_\w+)', r['Name'])
    if not m: continue
    k = m.group(1)
    c = int(r['Calls']); t = float(r['TotalDurationNs'])
    a = out.setdefault(k, [0, 0.0]); a[0] += c; a[1] += t
print('  '.join(f"{k} {t / c / 1e3:.1f}us x{c}" for k, (c, t) in sorted(out.items())))
PY
done
cp /tmp/keep.so $R/orb_slam2_ros2_amd/liborbfe_hip.so
