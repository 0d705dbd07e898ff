import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _have_hip_device() -> bool:
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a machine without a HIP device: the gpu-marked tests are SKIPPED (with the reason), not failed -- the product
    has no CPU fallback to run them on."""
    if _have_hip_device():
        return
    skip = pytest.mark.skip(reason="needs a HIP device (liborbfe_hip.so has no CPU fallback)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/liborb_oracle.so), compiled on demand.  Checker only."""
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def kitti_pair():
    from orb_slam2_ros2_amd import synth
    return synth.stereo_pair(0)


@pytest.fixture(scope="session")
def rng():
    return np.random.default_rng(20251003)
