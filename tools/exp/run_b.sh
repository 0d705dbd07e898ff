set -e
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_bench_config.py tests/test_frame_glue.py tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -5
python3 tools/latency_single.py 2>&1 | grep -v amdgpu
