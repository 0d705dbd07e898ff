// se3_dev.h -- device-side SE3 helpers shared by the pose-only and local-BA kernels (g2o types/slam3d/se3quat.h restated:
// Eigen quaternion-vector product, SE3Quat::exp, operator*, normalizeRotation).  fp64, no FMA contraction.
#pragma once
#include <hip/hip_runtime.h>

namespace orbfe {

struct PoseDev {
  double q[4], t[3];
};

__device__ __forceinline__ void quat_rotate(const double* q, const double* v, double* out) {
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  double ux = qy * v[2] - qz * v[1], uy = qz * v[0] - qx * v[2], uz = qx * v[1] - qy * v[0];
  ux += ux;
  uy += uy;
  uz += uz;
  out[0] = v[0] + qw * ux + (qy * uz - qz * uy);
  out[1] = v[1] + qw * uy + (qz * ux - qx * uz);
  out[2] = v[2] + qw * uz + (qx * uy - qy * ux);
}

// SE3Quat::exp(update) * T, normalizeRotation (g2o se3quat.h); update = (omega, upsilon)
__device__ inline void pose_oplus(const PoseDev& T, const double* upd, PoseDev& out) {
  const double wx = upd[0], wy = upd[1], wz = upd[2];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  const double Om[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
  double Om2[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += Om[i][k] * Om[k][j];
      Om2[i][j] = a;
    }
  double R[3][3], V[3][3];
  const double st = sin(theta), ct = cos(theta);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double I = (i == j) ? 1.0 : 0.0;
      if (theta < 0.00001) {
        R[i][j] = I + Om[i][j] + 0.5 * Om2[i][j];
        V[i][j] = I + 0.5 * Om[i][j] + (1. / 6.) * Om2[i][j];
      } else {
        R[i][j] = I + st / theta * Om[i][j] + (1 - ct) / (theta * theta) * Om2[i][j];
        V[i][j] = I + (1 - ct) / (theta * theta) * Om[i][j] + (theta - st) / (theta * theta * theta) * Om2[i][j];
      }
    }
  double q[4];
  const double tr = R[0][0] + R[1][1] + R[2][2];
  if (tr > 0) {
    double t = sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[2][1] - R[1][2]) * t;
    q[1] = (R[0][2] - R[2][0]) * t;
    q[2] = (R[1][0] - R[0][1]) * t;
  } else {
    // (the three cases of Eigen's largest-diagonal branch spelled out: indexing R and q by a run-time i would move both out of registers)
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (i == 0 ? R[2][2] > R[0][0] : R[2][2] > R[1][1]) i = 2;
    if (i == 0) {
      double t = sqrt(R[0][0] - R[1][1] - R[2][2] + 1.0);
      q[0] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (R[2][1] - R[1][2]) * t;
      q[1] = (R[1][0] + R[0][1]) * t;
      q[2] = (R[2][0] + R[0][2]) * t;
    } else if (i == 1) {
      double t = sqrt(R[1][1] - R[2][2] - R[0][0] + 1.0);
      q[1] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (R[0][2] - R[2][0]) * t;
      q[2] = (R[2][1] + R[1][2]) * t;
      q[0] = (R[0][1] + R[1][0]) * t;
    } else {
      double t = sqrt(R[2][2] - R[0][0] - R[1][1] + 1.0);
      q[2] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (R[1][0] - R[0][1]) * t;
      q[0] = (R[0][2] + R[2][0]) * t;
      q[1] = (R[1][2] + R[2][1]) * t;
    }
  }
  double te[3];
  for (int i = 0; i < 3; ++i) te[i] = V[i][0] * upd[3] + V[i][1] * upd[4] + V[i][2] * upd[5];
  const double ax = q[0], ay = q[1], az = q[2], aw = q[3];
  const double bx = T.q[0], by = T.q[1], bz = T.q[2], bw = T.q[3];
  out.q[3] = aw * bw - ax * bx - ay * by - az * bz;
  out.q[0] = aw * bx + ax * bw + ay * bz - az * by;
  out.q[1] = aw * by + ay * bw + az * bx - ax * bz;
  out.q[2] = aw * bz + az * bw + ax * by - ay * bx;
  double rt[3];
  quat_rotate(q, T.t, rt);
  for (int i = 0; i < 3; ++i) out.t[i] = te[i] + rt[i];
  if (out.q[3] < 0)
    for (int i = 0; i < 4; ++i) out.q[i] = -out.q[i];
  const double n = sqrt(out.q[0] * out.q[0] + out.q[1] * out.q[1] + out.q[2] * out.q[2] + out.q[3] * out.q[3]);
  for (int i = 0; i < 4; ++i) out.q[i] /= n;
}


}  // namespace orbfe
