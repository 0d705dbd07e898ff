#!/bin/bash
python tools/latency_single.py 2>&1 | tail -3
for b in 1 8 64 128; do python tools/step_time.py $b 200 2>/dev/null; done
