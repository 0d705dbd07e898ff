"""orb_slam2_ros2_amd -- MI355X-native ORB front end (+ local-BA edge kernels) behind the reference's
ORBExtractor / ORBMatcher / Optimizer call shapes.  See include/orbfe.h for the C-ABI and DESIGN.md.

Importing this package does not load the HIP library; constructing any of the classes does, and raises
if the library or a device is missing (there is no CPU fallback).
"""


def configure_streaming() -> int:
    """Export GPU_MAX_HW_QUEUES for a process that is going to STREAM batches (orbfe_stream_submit / DeviceSequenceProcessor): the HIP
    runtime multiplexes all streams of a process onto that many hardware queues (4 by default) and reads the variable at its first HIP
    call, so this must run before anything touches the GPU (before `import torch` initialises it); a value already exported wins.
    Importing the package does not do this: one-frame-at-a-time users (ORBExtractor / StereoFrontEnd) are ~50 us per frame faster with
    the runtime's default.  Returns the value in effect."""
    import os
    from ._lib import load
    want = int(load().orbfe_recommended_hw_queues())
    return int(os.environ.setdefault("GPU_MAX_HW_QUEUES", str(want)))


from .frontend import Frame, ORBExtractor, ORBMatcher, Optimizer, StereoFrontEnd  # noqa: F401

__all__ = ["Frame", "ORBExtractor", "ORBMatcher", "Optimizer", "StereoFrontEnd", "configure_streaming"]
