set -e
bash tools/profile_round.sh r4_v1 2>&1 | tail -12
bash tools/exp/timeline.sh > gpurun_out/r4_timeline_step.txt 2>&1 || true
tail -25 gpurun_out/r4_timeline_step.txt
