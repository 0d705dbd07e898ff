"""orb_slam2_ros2_amd -- MI355X-native ORB front end (+ local-BA edge kernels) behind the reference's
ORBExtractor / ORBMatcher / Optimizer call shapes.  See include/orbfe.h for the C-ABI and DESIGN.md.

Importing this package does not load the HIP library; constructing any of the classes does, and raises
if the library or a device is missing (there is no CPU fallback).
"""
from .frontend import Frame, ORBExtractor, ORBMatcher, Optimizer, StereoFrontEnd  # noqa: F401

__all__ = ["Frame", "ORBExtractor", "ORBMatcher", "Optimizer", "StereoFrontEnd"]
