#!/bin/bash
# Runs on the GPU box: half-batch streams (ORBFE_STREAMS) with and without the skew experiment (ORBFE_CHUNK_SKEW=1: a chunk's FAST waits for
# the previous chunk's, so that one chunk's vector-bound head runs beside the other's latency-bound tail).  Needs a build with the experiment.
# (needs a library built from a patched tree: the experiment was reverted after the measurement recorded in DESIGN 4.9 -- the script documents how it was run)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for kv in "ORBFE_STREAMS=1" "ORBFE_STREAMS=2" "ORBFE_STREAMS=2 ORBFE_CHUNK_SKEW=1" "ORBFE_STREAMS=3 ORBFE_CHUNK_SKEW=1" "ORBFE_STREAMS=4 ORBFE_CHUNK_SKEW=1"; do
  echo -n "$kv: "
  env $kv python3 bench.py --legs "" --steps 60 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['verified_pairs'])"
done
done
