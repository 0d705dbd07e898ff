// ba_oracle.cpp -- CPU ORACLE for the local-BA edge evaluation (test infrastructure, NOT product code).
//
// PARITY UNPINNED (see orb_oracle.cpp): g2o (tag 20241228) is not vendored in the reference and not
// installed here; this restates the published arithmetic of
//   g2o/types/sba/edge_project_xyz.cpp          EdgeSE3ProjectXYZ::computeError / linearizeOplus / cam_project
//   g2o/types/sba/edge_project_stereo_xyz.cpp   EdgeStereoSE3ProjectXYZ (same, 3-D error with bf)
//   g2o/types/slam3d/se3quat.h                  SE3Quat::map  (Eigen quaternion * vector)
//   g2o/core/robust_kernel_impl.cpp             RobustKernelHuber::robustify
//   g2o/core/base_binary_edge.hpp               constructQuadraticForm (H blocks, b)
// as configured by Optimizer::OptimizeLocalMap (src/ORB_SLAM2/src/Optimizer.cc:296-330).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Mat3 {
  double m[3][3];
};

// Eigen::Quaternion::toRotationMatrix
Mat3 quat_to_rot(double qx, double qy, double qz, double qw) {
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  Mat3 R;
  R.m[0][0] = 1 - (tyy + tzz);
  R.m[0][1] = txy - twz;
  R.m[0][2] = txz + twy;
  R.m[1][0] = txy + twz;
  R.m[1][1] = 1 - (txx + tzz);
  R.m[1][2] = tyz - twx;
  R.m[2][0] = txz - twy;
  R.m[2][1] = tyz + twx;
  R.m[2][2] = 1 - (txx + tyy);
  return R;
}

// SE3Quat::map: _r * xyz + _t with Eigen's quaternion-vector product
void se3_map(const double* T, const double* X, double out[3]) {
  const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
  double uv[3] = {qy * X[2] - qz * X[1], qz * X[0] - qx * X[2], qx * X[1] - qy * X[0]};
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  out[0] = X[0] + qw * uv[0] + (qy * uv[2] - qz * uv[1]) + T[4];
  out[1] = X[1] + qw * uv[1] + (qz * uv[0] - qx * uv[2]) + T[5];
  out[2] = X[2] + qw * uv[2] + (qx * uv[1] - qy * uv[0]) + T[6];
}

}  // namespace

extern "C" {

// Same SoA problem layout as orbfe_ba_problem (include/orbfe.h).  Any output pointer may be NULL.
void orc_ba_eval_edges(int n_edges, const double* poses, const double* points, const int32_t* edge_pose, const int32_t* edge_point,
                       const double* meas, const uint8_t* is_stereo, const double* info, const double* delta, double fx, double fy,
                       double cx, double cy, double bf, double* error, double* chi2, double* rho, double* j_point, double* j_pose,
                       uint8_t* depth_pos) {
  for (int e = 0; e < n_edges; ++e) {
    const double* T = poses + (size_t)edge_pose[e] * 7;
    const double* X = points + (size_t)edge_point[e] * 3;
    double P[3];
    se3_map(T, X, P);
    const double x = P[0], y = P[1], z = P[2];
    const bool st = is_stereo[e] != 0;
    const double* m = meas + (size_t)e * 3;
    // cam_project
    const double u = x / z * fx + cx, v = y / z * fy + cy;
    double err[3] = {m[0] - u, m[1] - v, 0.0};
    if (st) err[2] = m[2] - (u - bf / z);
    if (error) std::memcpy(error + (size_t)e * 3, err, sizeof err);
    // chi2 = e^T * Omega * e, Omega = info * I
    const double w = info[e];
    double c2 = err[0] * (w * err[0]) + err[1] * (w * err[1]);
    if (st) c2 += err[2] * (w * err[2]);
    if (chi2) chi2[e] = c2;
    double r0 = c2, r1 = 1.0;
    if (delta[e] > 0) {  // RobustKernelHuber
      const double dsqr = delta[e] * delta[e];
      if (c2 > dsqr) {
        const double sq = std::sqrt(c2);
        r0 = 2 * sq * delta[e] - dsqr;
        r1 = delta[e] / sq;
      }
    }
    if (rho) {
      rho[(size_t)e * 2] = r0;
      rho[(size_t)e * 2 + 1] = r1;
    }
    if (depth_pos) depth_pos[e] = z > 0.0;
    const double z_2 = z * z;
    if (j_point) {
      const Mat3 R = quat_to_rot(T[0], T[1], T[2], T[3]);
      double* J = j_point + (size_t)e * 9;
      if (st) {
        for (int k = 0; k < 3; ++k) {
          J[0 + k] = -fx * R.m[0][k] / z + fx * x * R.m[2][k] / z_2;
          J[3 + k] = -fy * R.m[1][k] / z + fy * y * R.m[2][k] / z_2;
          J[6 + k] = J[0 + k] - bf * R.m[2][k] / z_2;
        }
      } else {
        // _jacobianOplusXi = -1./z * tmp * R with tmp = [[fx,0,-x/z*fx],[0,fy,-y/z*fy]]
        const double tmp[2][3] = {{fx, 0, -x / z * fx}, {0, fy, -y / z * fy}};
        const double s = -1. / z;
        for (int r = 0; r < 2; ++r)
          for (int k = 0; k < 3; ++k) {
            double acc = 0;
            for (int i = 0; i < 3; ++i) acc += (s * tmp[r][i]) * R.m[i][k];
            J[3 * r + k] = acc;
          }
        J[6] = J[7] = J[8] = 0.0;
      }
    }
    if (j_pose) {
      double* J = j_pose + (size_t)e * 18;
      J[0] = x * y / z_2 * fx;
      J[1] = -(1 + (x * x / z_2)) * fx;
      J[2] = y / z * fx;
      J[3] = -1. / z * fx;
      J[4] = 0;
      J[5] = x / z_2 * fx;
      J[6] = (1 + y * y / z_2) * fy;
      J[7] = -x * y / z_2 * fy;
      J[8] = -x / z * fy;
      J[9] = 0;
      J[10] = -1. / z * fy;
      J[11] = y / z_2 * fy;
      for (int k = 12; k < 18; ++k) J[k] = 0.0;
      if (st) {
        J[12] = J[0] - bf * y / z_2;
        J[13] = J[1] + bf * x / z_2;
        J[14] = J[2];
        J[15] = J[3];
        J[16] = 0;
        J[17] = J[5] - bf / z_2;
      }
    }
  }
}

// g2o BaseBinaryEdge::constructQuadraticForm for every active edge, accumulated the way BlockSolver_6_3 lays the system out:
// vertex 0 = point (Xi, Jacobian A), vertex 1 = pose (Xj, Jacobian B).  With Omega = info*I and w = rho'(chi2) (Huber) or 1:
//   b_point -= A^T (w Omega) e        H_ll += A^T (w Omega) A
//   b_pose  -= B^T (w Omega) e        H_pp += B^T (w Omega) B        H_pl(e) = B^T (w Omega) A      (pose not fixed)
// Points are never fixed in the local BA (they are the marginalised set, Optimizer.cc:265); fixed poses get no blocks.
void orc_ba_build_system(int n_poses, int n_points, int n_edges, const double* poses, const double* points, const int32_t* edge_pose,
                         const int32_t* edge_point, const double* meas, const uint8_t* is_stereo, const double* info, const double* delta,
                         double fx, double fy, double cx, double cy, double bf, const uint8_t* pose_fixed, double* Hpp, double* bp,
                         double* Hll, double* bl, double* Hpl, double* chi2_robust_sum) {
  std::vector<double> err((size_t)n_edges * 3), chi2(n_edges), rho((size_t)n_edges * 2), jp((size_t)n_edges * 9), jx((size_t)n_edges * 18);
  orc_ba_eval_edges(n_edges, poses, points, edge_pose, edge_point, meas, is_stereo, info, delta, fx, fy, cx, cy, bf, err.data(),
                    chi2.data(), rho.data(), jp.data(), jx.data(), nullptr);
  std::memset(Hpp, 0, sizeof(double) * 36 * n_poses);
  std::memset(bp, 0, sizeof(double) * 6 * n_poses);
  std::memset(Hll, 0, sizeof(double) * 9 * n_points);
  std::memset(bl, 0, sizeof(double) * 3 * n_points);
  double total = 0;
  for (int e = 0; e < n_edges; ++e) {
    const int rows = is_stereo[e] ? 3 : 2;
    const double w = rho[(size_t)e * 2 + 1] * info[e];
    const double* A = &jp[(size_t)e * 9];
    const double* B = &jx[(size_t)e * 18];
    const double* er = &err[(size_t)e * 3];
    const int pi = edge_point[e], ki = edge_pose[e];
    total += rho[(size_t)e * 2];
    for (int a = 0; a < 3; ++a) {
      double s = 0;
      for (int r = 0; r < rows; ++r) s += A[3 * r + a] * (w * er[r]);
      bl[(size_t)pi * 3 + a] -= s;
      for (int c = 0; c < 3; ++c) {
        double h = 0;
        for (int r = 0; r < rows; ++r) h += A[3 * r + a] * w * A[3 * r + c];
        Hll[(size_t)pi * 9 + 3 * a + c] += h;
      }
    }
    if (Hpl) std::memset(Hpl + (size_t)e * 18, 0, sizeof(double) * 18);
    if (pose_fixed && pose_fixed[ki]) continue;
    for (int a = 0; a < 6; ++a) {
      double s = 0;
      for (int r = 0; r < rows; ++r) s += B[6 * r + a] * (w * er[r]);
      bp[(size_t)ki * 6 + a] -= s;
      for (int c = 0; c < 6; ++c) {
        double h = 0;
        for (int r = 0; r < rows; ++r) h += B[6 * r + a] * w * B[6 * r + c];
        Hpp[(size_t)ki * 36 + 6 * a + c] += h;
      }
      if (Hpl)
        for (int c = 0; c < 3; ++c) {
          double h = 0;
          for (int r = 0; r < rows; ++r) h += B[6 * r + a] * w * A[3 * r + c];
          Hpl[(size_t)e * 18 + 3 * a + c] = h;
        }
    }
  }
  if (chi2_robust_sum) *chi2_robust_sum = total;
}

// SE3Quat::exp(update) * T  -- VertexSE3Expmap::oplusImpl; update = (omega, upsilon).  Used by the
// finite-difference Jacobian check in tests.
void orc_se3_oplus(const double* T, const double* upd, double* out) {
  const double wx = upd[0], wy = upd[1], wz = upd[2];
  const double theta = std::sqrt(wx * wx + wy * wy + wz * wz);
  double Om[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
  double Om2[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += Om[i][k] * Om[k][j];
      Om2[i][j] = a;
    }
  double R[3][3], V[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double I = (i == j) ? 1.0 : 0.0;
      if (theta < 0.00001) {
        R[i][j] = I + Om[i][j] + 0.5 * Om2[i][j];
        V[i][j] = I + 0.5 * Om[i][j] + (1. / 6.) * Om2[i][j];
      } else {
        R[i][j] = I + std::sin(theta) / theta * Om[i][j] + (1 - std::cos(theta)) / (theta * theta) * Om2[i][j];
        V[i][j] = I + (1 - std::cos(theta)) / (theta * theta) * Om[i][j] + (theta - std::sin(theta)) / (theta * theta * theta) * Om2[i][j];
      }
    }
  // quaternion of R (Eigen's algorithm)
  double q[4];
  const double tr = R[0][0] + R[1][1] + R[2][2];
  if (tr > 0) {
    double t = std::sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[2][1] - R[1][2]) * t;
    q[1] = (R[0][2] - R[2][0]) * t;
    q[2] = (R[1][0] - R[0][1]) * t;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k][j] - R[j][k]) * t;
    q[j] = (R[j][i] + R[i][j]) * t;
    q[k] = (R[k][i] + R[i][k]) * t;
  }
  double te[3];
  for (int i = 0; i < 3; ++i) te[i] = V[i][0] * upd[3] + V[i][1] * upd[4] + V[i][2] * upd[5];
  // (q, te) * (T.q, T.t):  q_out = q * Tq ; t_out = q * Tt + te
  const double ax = q[0], ay = q[1], az = q[2], aw = q[3];
  const double bx = T[0], by = T[1], bz = T[2], bw = T[3];
  out[3] = aw * bw - ax * bx - ay * by - az * bz;
  out[0] = aw * bx + ax * bw + ay * bz - az * by;
  out[1] = aw * by + ay * bw + az * bx - ax * bz;
  out[2] = aw * bz + az * bw + ax * by - ay * bx;
  const double n = std::sqrt(out[0] * out[0] + out[1] * out[1] + out[2] * out[2] + out[3] * out[3]);
  for (int i = 0; i < 4; ++i) out[i] /= n;
  const double Tq[7] = {ax, ay, az, aw, te[0], te[1], te[2]};
  double p[3];
  se3_map(Tq, T + 4, p);
  out[4] = p[0];
  out[5] = p[1];
  out[6] = p[2];
}

}  // extern "C"
