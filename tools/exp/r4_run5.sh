set -e
timeout -k 10 500 python3 tools/lba_sizes.py 30 43 64 100 300 2>&1 | grep -v amdgpu.ids
