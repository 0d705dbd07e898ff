#!/bin/bash
# Runs on the GPU box: bench.py's own step loop (512 DISTINCT frames, 300 steps, no legs) with library variants (tools/exp/libs/liborbfe_<v>.so;
# "cur" = the library in place), alternating:  tools/exp/bench_ab_lib.sh "old cur" [rounds]
cd ${GRAFT_REPO_ROOT:-.}
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for round in $(seq 1 ${2:-3}); do
  for v in $1; do
    if [ "$v" = cur ]; then cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so; else cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so; fi
    echo -n "$v : "
    timeout -k 10 400 python3 bench.py --steps 300 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs '' --content-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config']; print(round(d['value']), round(d['ms_per_step'],3), d['verified_pairs'], {k[6:-3]: c[k] for k in c if k.startswith('stage_')})"
  done
done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
