#!/bin/bash
# Runs on the GPU box: same-box A/B of library variants built by tools/exp/ab_build.sh: tools/exp/ab_run2.sh base n4 ...
cd ${GRAFT_REPO_ROOT:-.}
cp orb_slam2_ros2_amd/liborbfe_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp tools/exp/libs/liborbfe_$v.so orb_slam2_ros2_amd/liborbfe_hip.so
    echo -n "$v: "
    python3 bench.py --legs "" --steps 60 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['all_stages']['quadtree']['ms'], 'fast', d['roofline']['all_stages']['fast']['ms'], 'ob', d['roofline']['all_stages']['orient_brief']['ms'])"
  done
done
cp /tmp/keep.so orb_slam2_ros2_amd/liborbfe_hip.so
