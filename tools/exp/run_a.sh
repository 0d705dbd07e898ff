set -e
timeout -k 10 600 python3 -m pytest tests/test_gpu_dropin.py tests/test_track_chain.py -x -q -m gpu 2>&1 | tail -5
timeout -k 10 600 python3 bench.py --steps 40 --cpu-seconds 0 --host-io-steps 0 --sequence-leg 0 --legs latency > gpurun_out/r4_lat.json 2> gpurun_out/r4_lat.err || { tail -20 gpurun_out/r4_lat.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r4_lat.json')); print(json.dumps(d['latency'], indent=1)[:3000])"
