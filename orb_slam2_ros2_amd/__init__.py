"""orb_slam2_ros2_amd -- MI355X-native ORB front end (+ local-BA edge kernels) behind the reference's
ORBExtractor / ORBMatcher / Optimizer call shapes.  See include/orbfe.h for the C-ABI and DESIGN.md.

Importing this package does not load the HIP library; constructing any of the classes does, and raises
if the library or a device is missing (there is no CPU fallback).
"""
import os as _os

# The front end keeps up to eight HIP streams busy at once (compute, stereo match, blur, upload, download, the slot lanes, torch's and
# RCCL's own), and the HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues -- 4 by default.  Two
# streams that share a queue run one after the other: with 4 queues the upload of batch k+1 queued behind the kernels of batch k and the
# 4541-pair sequence took 0.127 s; with 16 it takes 0.087 s (profiles/r3_hw_queues.txt).  Read by the runtime at its first HIP call,
# so it is set here, before anything of this package touches the device; a value the caller exported wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .frontend import Frame, ORBExtractor, ORBMatcher, Optimizer, StereoFrontEnd  # noqa: F401

__all__ = ["Frame", "ORBExtractor", "ORBMatcher", "Optimizer", "StereoFrontEnd"]
