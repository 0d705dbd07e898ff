#!/usr/bin/env python3
"""gen_frames.py FIRST COUNT OUT.npy -- synthetic KITTI-shaped stereo pairs FIRST .. FIRST + COUNT - 1 (synth.stereo_pair) as one uint8
array [COUNT][2][376][1241], generated on a pool of forked workers.  For callers whose own process has already touched the GPU (the
GPU test session): they start this as a CHILD process instead of forking themselves."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from benchlib.common import generate_pairs

if __name__ == "__main__":
    first, count, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    pairs = generate_pairs(range(first, first + count))
    np.save(out, np.stack([np.stack(p) for p in pairs]))
