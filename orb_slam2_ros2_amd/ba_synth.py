"""Synthetic local-BA problem of BASELINE config 5: 60 keyframes on a 3 m arc, 3000 points in a
6x3x4 m box, each point observed by 3..8 keyframes that see it, 80 % stereo / 20 % mono edges, integer-only
Irwin-Hall measurement noise scaled by the octave sigma, 5 % gross outliers.  Intrinsics from
config/tum_config_f2.yaml (fx 520.9 fy 521.0 cx 325.1 cy 249.7, bf = fx*0.0767889)."""
from __future__ import annotations

import numpy as np

from .synth import hash_u64

FX, FY, CX, CY = 520.908620, 521.007327, 325.141442, 249.701764
BF = FX * 0.0767889


def _u(seed, stream, idx):
    """uniform [0,1) doubles with 53 random bits"""
    return (hash_u64(seed, stream, idx) >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def _irwin_hall(seed, stream, idx):
    """~N(0,1): sum of 12 uniform 16-bit draws, integer-only generator"""
    idx = np.asarray(idx, np.uint64)
    s = np.zeros(idx.shape, np.float64)
    for k in range(12):
        s += (hash_u64(seed, stream + k, idx) & np.uint64(0xFFFF)).astype(np.float64)
    return s / 65536.0 - 6.0


def _quat_from_yaw(yaw):
    return np.stack([np.zeros_like(yaw), np.sin(yaw / 2), np.zeros_like(yaw), np.cos(yaw / 2)], 1)  # rotation about y


def make_problem(seed=42, n_kf=60, n_pt=3000, scale_factor=1.2, n_levels=8, with_truth=False):
    # camera centres on an arc of radius 3 m looking roughly along +z
    ang = np.linspace(-0.5, 0.5, n_kf)
    centres = np.stack([3.0 * np.sin(ang), 0.05 * np.cos(7 * ang), 3.0 * (1 - np.cos(ang))], 1)
    yaw = -0.6 * ang
    q_cw = _quat_from_yaw(yaw)  # world -> camera rotation (about y)

    def rot(q, v):
        qv, qw = q[..., :3], q[..., 3:4]
        uv = 2 * np.cross(qv, v)
        return v + qw * uv + np.cross(qv, uv)

    t_cw = -rot(q_cw, centres)
    poses = np.concatenate([q_cw, t_cw], 1)
    pid = np.arange(n_pt)
    points = np.stack([(_u(seed, 1, pid) - 0.5) * 6.0, (_u(seed, 2, pid) - 0.5) * 3.0, 2.0 + _u(seed, 3, pid) * 4.0], 1)

    sigma = np.float32(scale_factor) ** np.arange(n_levels, dtype=np.float32)
    e_pose, e_pt, meas, stereo, info, delta = [], [], [], [], [], []
    d_mono, d_stereo = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815)))
    eid = 0
    for p in range(n_pt):
        want = 3 + int(hash_u64(seed, 4, p) % np.uint64(6))
        start = int(hash_u64(seed, 5, p) % np.uint64(n_kf))
        got = 0
        for j in range(n_kf):
            k = (start + j * 7) % n_kf
            pc = rot(poses[k, :4], points[p]) + poses[k, 4:]
            if pc[2] < 0.3:
                continue
            u, v = FX * pc[0] / pc[2] + CX, FY * pc[1] / pc[2] + CY
            if not (0 <= u < 640 and 0 <= v < 480):
                continue
            octave = int(hash_u64(seed, 6, eid) % np.uint64(n_levels))
            s = float(sigma[octave])
            nz = _irwin_hall(seed, 10, np.array([3 * eid, 3 * eid + 1, 3 * eid + 2])) * s
            if hash_u64(seed, 7, eid) % np.uint64(20) == 0:  # 5 % gross outliers
                nz = nz + np.array([35.0, -28.0, 31.0])
            is_st = (hash_u64(seed, 8, eid) % np.uint64(5)) != 0  # 80 % stereo
            ur = u - BF / pc[2]
            # measurements are float32 in the reference (kp.pt.x, rightU stored from float math)
            meas.append([np.float32(u + nz[0]), np.float32(v + nz[1]), np.float32(ur + nz[2]) if is_st else -1.0])
            inv_s = np.float32(1.0) / sigma[octave]
            # quirk Q9: stereo edges use invSigma^2, mono edges use invSigma (Optimizer.cc:301 vs :319)
            info.append(float(np.float32(inv_s) ** 2) if is_st else float(inv_s))
            delta.append(d_stereo if is_st else d_mono)
            stereo.append(1 if is_st else 0)
            e_pose.append(k)
            e_pt.append(p)
            eid += 1
            got += 1
            if got >= want:
                break
    # perturb the estimates so errors/Jacobians are not trivially zero
    kid = np.arange(n_kf)
    poses_est = poses.copy()
    poses_est[:, 4:] += 0.02 * np.stack([_irwin_hall(seed, 30, kid), _irwin_hall(seed, 50, kid), _irwin_hall(seed, 70, kid)], 1)
    points_est = points + 0.03 * np.stack([_irwin_hall(seed, 90, pid), _irwin_hall(seed, 110, pid), _irwin_hall(seed, 130, pid)], 1)
    out = dict(poses=poses_est, points=points_est, edge_pose=np.asarray(e_pose, np.int32), edge_point=np.asarray(e_pt, np.int32),
               meas=np.asarray(meas, np.float64), is_stereo=np.asarray(stereo, np.uint8), info=np.asarray(info, np.float64),
               huber_delta=np.asarray(delta, np.float64), fx=FX, fy=FY, cx=CX, cy=CY, bf=BF)
    if with_truth:  # the noise-free vertices the measurements were generated from
        out.update(poses_true=poses, points_true=points)
    return out


def make_pose_problem(seed=7, n=1000, scale_factor=1.2, n_levels=8, outlier_every=12):
    """One frame for Optimizer::OptimizePoseOnly (Optimizer.cc:33-203): n map points seen by a camera whose initial pose is a
    perturbed version of the truth; 70 % stereo / 30 % mono observations with octave-scaled noise, every `outlier_every`-th
    observation a gross outlier.  Returns the arrays of the C-ABI call."""
    idx = np.arange(n)
    X = np.stack([(_u(seed, 1, idx) - 0.5) * 8.0, (_u(seed, 2, idx) - 0.5) * 4.0, 3.0 + _u(seed, 3, idx) * 9.0], 1)
    q_true = np.array([0.02, -0.03, 0.01, 0.0])
    q_true[3] = np.sqrt(1 - (q_true[:3] ** 2).sum())
    t_true = np.array([0.10, -0.05, 0.20])

    def rot(q, v):
        qv, qw = q[:3], q[3]
        uv = 2 * np.cross(qv, v)
        return v + qw * uv + np.cross(qv, uv)

    Pc = rot(q_true, X) + t_true
    u = FX * Pc[:, 0] / Pc[:, 2] + CX
    v = FY * Pc[:, 1] / Pc[:, 2] + CY
    ur = u - BF / Pc[:, 2]
    octave = (hash_u64(seed, 4, idx) % np.uint64(n_levels)).astype(np.int64)
    sig = (np.float32(scale_factor) ** octave.astype(np.float32)).astype(np.float32)
    nz = np.stack([_irwin_hall(seed, 10, idx), _irwin_hall(seed, 30, idx), _irwin_hall(seed, 50, idx)], 1) * sig[:, None]
    out = (idx % outlier_every) == 0
    nz[out] += np.array([40.0, -33.0, 36.0])
    stereo = (hash_u64(seed, 5, idx) % np.uint64(10)) < np.uint64(7)
    meas = np.stack([np.float32(u + nz[:, 0]), np.float32(v + nz[:, 1]), np.where(stereo, np.float32(ur + nz[:, 2]), -1.0)], 1).astype(np.float64)
    inv_s = (np.float32(1.0) / sig).astype(np.float32)
    info = (inv_s.astype(np.float32) ** 2).astype(np.float64)   # getScaledFactorInv2 (float pow), both edge kinds (Optimizer.cc:85,106)
    sigma2 = (sig ** 2).astype(np.float32)                        # getScaledFactor2
    q0 = q_true + np.array([0.004, -0.003, 0.002, 0.0])
    q0 /= np.linalg.norm(q0)
    pose0 = np.concatenate([q0, t_true + np.array([0.05, -0.04, 0.06])])
    return dict(Xw=X, meas=meas, info=info, sigma2=sigma2, pose=pose0, fx=FX, fy=FY, cx=CX, cy=CY, bf=BF,
                truth=np.concatenate([q_true, t_true]), outlier=out)
