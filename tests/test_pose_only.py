"""Optimizer::OptimizePoseOnly (Optimizer.cc:33-203): oracle known answers (CPU) and the device optimiser against the oracle (GPU).
Tolerance: north_star asks for 1e-4 on BA pose residuals; the device differs from the oracle only by the summation order of the
6x6 normal equations, so 1e-6 on the pose is asserted."""
import numpy as np
import pytest

from orb_slam2_ros2_amd import ba_synth

ARGS = ("Xw", "meas", "info", "sigma2", "pose", "fx", "fy", "cx", "cy", "bf")


def _args(p):
    return {k: p[k] for k in ARGS}


def test_oracle_recovers_the_pose_and_flags_the_gross_outliers(orc):
    p = ba_synth.make_pose_problem()
    n_good, pose, inl = orc.pose_only_optimize(**_args(p))
    assert n_good == inl.sum()
    assert np.abs(pose - p["truth"]).max() < 5e-3           # noise-limited
    assert np.abs(p["pose"] - p["truth"]).max() > 3e-2      # the start was far away
    assert not inl[p["outlier"]].any()                      # every gross outlier is rejected
    assert inl[~p["outlier"]].mean() > 0.98
    assert abs(np.linalg.norm(pose[:4]) - 1) < 1e-12 and pose[3] > 0   # normalizeRotation


def test_oracle_mono_only_and_degenerate_inputs(orc):
    p = ba_synth.make_pose_problem(seed=11, n=300)
    a = _args(p)
    a["meas"] = a["meas"].copy()
    a["meas"][:, 2] = -1.0                                   # rightU < 0 everywhere: only EdgeSE3ProjectXYZOnlyPose
    n_good, pose, inl = orc.pose_only_optimize(**a)
    assert np.abs(pose - p["truth"]).max() < 2e-2 and not inl[p["outlier"]].any()
    e = {k: (v[:0] if isinstance(v, np.ndarray) and v.ndim and k != "pose" else v) for k, v in _args(p).items()}
    n0, pose0, _ = orc.pose_only_optimize(**e)               # no edges: g2o optimises nothing, the pose is returned unchanged
    assert n0 == 0 and np.array_equal(pose0, p["pose"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(7, 1000), (8, 2000), (9, 257), (10, 40), (12, 1024), (13, 1025), (14, 2048), (15, 2500)])   # 256- / 512-thread register kernels, the in-memory one past 2048
def test_device_optimiser_matches_oracle(orc, seed, n):
    from orb_slam2_ros2_amd._lib import Context
    from orb_slam2_ros2_amd import Optimizer
    p = ba_synth.make_pose_problem(seed=seed, n=n)
    ctx = Context(640, 480, n_features=500, max_images=1)
    n_good, pose, inl = Optimizer.OptimizePoseOnly(ctx, **_args(p))
    r_good, r_pose, r_inl = orc.pose_only_optimize(**_args(p))
    assert np.abs(pose - r_pose).max() < 1e-6
    assert (inl != r_inl).sum() <= 1 and abs(n_good - r_good) <= 1     # an edge exactly at the chi2 threshold may flip
    assert not inl[p["outlier"]].any()
    again = ctx.pose_only_optimize(**_args(p))
    assert np.array_equal(again[1], pose) and np.array_equal(again[2], inl)   # deterministic reductions
    mono = _args(p)
    mono["meas"] = mono["meas"].copy()
    mono["meas"][:, 2] = -1.0
    g = ctx.pose_only_optimize(**mono)
    o = orc.pose_only_optimize(**mono)
    assert np.abs(g[1] - o[1]).max() < 1e-6 and (g[2] != o[2]).sum() <= 1
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 2, 5, 63, 64, 65, 255, 256, 257])
def test_device_optimiser_small_and_wave_boundary_sizes(orc, n):
    """Edge counts around the register kernel's thread / wave boundaries (and none at all): pose, inlier flags and count against the oracle."""
    from orb_slam2_ros2_amd._lib import Context
    p = ba_synth.make_pose_problem(seed=3 + n, n=max(n, 1))
    a = _args(p)
    if n == 0:
        for k in ("Xw", "meas", "info", "sigma2"):
            a[k] = a[k][:0]
    ctx = Context(640, 480, n_features=500, max_images=1)
    g = ctx.pose_only_optimize(**a)
    o = orc.pose_only_optimize(**a)
    ctx.close()
    assert np.abs(g[1] - o[1]).max() < 1e-6 and abs(g[0] - o[0]) <= 1 and (g[2] != o[2]).sum() <= 1
