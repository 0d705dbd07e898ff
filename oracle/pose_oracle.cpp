// pose_oracle.cpp -- CPU ORACLE for Optimizer::OptimizePoseOnly (test infrastructure, NOT product code).
//
// PARITY UNPINNED (see orb_oracle.cpp).  Restates, for ONE 6-DoF pose vertex and unary edges,
//   src/ORB_SLAM2/src/Optimizer.cc:33-203   the reference's driver: mono edge when rightU < 0 else stereo edge, information
//                                           = invSigma2(octave), Huber deltas sqrt(5.991)/sqrt(7.815) (as float), four rounds of
//                                           optimize(10) each restarting from the INITIAL pose, re-classification of every edge with
//                                           chi2 > 5.991*sigma2 (mono) / 7.815*sigma2 (stereo), robust kernels dropped in round i==2
//   g2o (20241228) types/sba/edge_project_xyz_onlypose.*, edge_project_stereo_xyz_onlypose.*  (error, Jacobian with 1/z products)
//   g2o core/base_unary_edge.hpp constructQuadraticForm, core/robust_kernel_impl.cpp Huber,
//   g2o core/optimization_algorithm_levenberg.cpp (tau = 1e-5, gain ratio, lambda *= max(1/3, min(2/3, 1-(2rho-1)^3)), ni doubling,
//       at most 10 trials per iteration, lambda re-initialised at every optimize() call),
//   g2o types/slam3d/se3quat.h (exp, operator*, normalizeRotation), LinearSolverDense (dense symmetric solve of the 6x6 system).
// The final image-bounds / positive-depth check of Optimizer.cc:180-190 uses Frame::project2UV with the PRE-optimisation float pose
// and stays on the caller's side of the boundary.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Pose {
  double q[4];  // x y z w
  double t[3];
};

void quat_rotate(const double* q, const double* v, double* out) {  // Eigen quaternion * vector
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  double uv[3] = {qy * v[2] - qz * v[1], qz * v[0] - qx * v[2], qx * v[1] - qy * v[0]};
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  out[0] = v[0] + qw * uv[0] + (qy * uv[2] - qz * uv[1]);
  out[1] = v[1] + qw * uv[1] + (qz * uv[0] - qx * uv[2]);
  out[2] = v[2] + qw * uv[2] + (qx * uv[1] - qy * uv[0]);
}

// SE3Quat::exp(update) * T with normalizeRotation (update = omega, upsilon)
Pose oplus(const Pose& T, const double* upd) {
  const double wx = upd[0], wy = upd[1], wz = upd[2];
  const double theta = std::sqrt(wx * wx + wy * wy + wz * wz);
  const double Om[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
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
  double q[4];  // Eigen: quaternion from rotation matrix
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
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k][j] - R[j][k]) * t;
    q[j] = (R[j][i] + R[i][j]) * t;
    q[k] = (R[k][i] + R[i][k]) * t;
  }
  double te[3];
  for (int i = 0; i < 3; ++i) te[i] = V[i][0] * upd[3] + V[i][1] * upd[4] + V[i][2] * upd[5];
  Pose out;
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
  const double n = std::sqrt(out.q[0] * out.q[0] + out.q[1] * out.q[1] + out.q[2] * out.q[2] + out.q[3] * out.q[3]);
  for (int i = 0; i < 4; ++i) out.q[i] /= n;
  return out;
}

struct Problem {
  int n;
  const double* Xw;    // [n][3]
  const double* meas;  // [n][3]  u, v, u_right (mono edge when u_right < 0, Optimizer.cc:77)
  const double* info;  // [n]     invSigma2(octave)
  double fx, fy, cx, cy, bf;
  double d_mono, d_stereo;
};

struct EdgeState {
  std::vector<double> err;   // [n][3] error at the last computeError of each edge
  std::vector<uint8_t> level;  // 0 active, 1 outlier
  std::vector<uint8_t> robust; // kernel present
};

void edge_error(const Problem& P, const Pose& T, int i, double* e) {
  double p[3];
  quat_rotate(T.q, P.Xw + 3 * i, p);
  const double x = p[0] + T.t[0], y = p[1] + T.t[1], z = p[2] + T.t[2];
  const double* m = P.meas + 3 * i;
  const double u = x / z * P.fx + P.cx, v = y / z * P.fy + P.cy;  // cam_project
  e[0] = m[0] - u;
  e[1] = m[1] - v;
  e[2] = (m[2] < 0) ? 0.0 : m[2] - (u - P.bf / z);
}

double edge_chi2(const Problem& P, int i, const double* e) {
  const double w = P.info[i];
  double c = e[0] * (w * e[0]) + e[1] * (w * e[1]);
  if (!(P.meas[3 * i + 2] < 0)) c += e[2] * (w * e[2]);
  return c;
}

void huber(double e, double delta, double* rho) {
  const double dsqr = delta * delta;
  if (e <= dsqr) {
    rho[0] = e;
    rho[1] = 1.;
  } else {
    const double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
  }
}

void compute_active_errors(const Problem& P, const Pose& T, EdgeState& S) {
  for (int i = 0; i < P.n; ++i)
    if (S.level[i] == 0) edge_error(P, T, i, &S.err[3 * i]);
}

double active_robust_chi2(const Problem& P, const EdgeState& S) {
  double chi = 0;
  for (int i = 0; i < P.n; ++i) {
    if (S.level[i] != 0) continue;
    const double c = edge_chi2(P, i, &S.err[3 * i]);
    if (S.robust[i]) {
      double rho[2];
      huber(c, (P.meas[3 * i + 2] < 0) ? P.d_mono : P.d_stereo, rho);
      chi += rho[0];
    } else
      chi += c;
  }
  return chi;
}

// linearizeOplus of every active edge at T + constructQuadraticForm: H (6x6), b (6)
void build_system(const Problem& P, const Pose& T, const EdgeState& S, double H[6][6], double b[6]) {
  std::memset(H, 0, sizeof(double) * 36);
  std::memset(b, 0, sizeof(double) * 6);
  for (int i = 0; i < P.n; ++i) {
    if (S.level[i] != 0) continue;
    double p[3];
    quat_rotate(T.q, P.Xw + 3 * i, p);
    const double x = p[0] + T.t[0], y = p[1] + T.t[1], z = p[2] + T.t[2];
    const double invz = 1.0 / z, invz_2 = invz * invz;
    const bool st = !(P.meas[3 * i + 2] < 0);
    const int rows = st ? 3 : 2;
    double J[3][6];
    J[0][0] = x * y * invz_2 * P.fx;
    J[0][1] = -(1 + (x * x * invz_2)) * P.fx;
    J[0][2] = y * invz * P.fx;
    J[0][3] = -invz * P.fx;
    J[0][4] = 0;
    J[0][5] = x * invz_2 * P.fx;
    J[1][0] = (1 + y * y * invz_2) * P.fy;
    J[1][1] = -x * y * invz_2 * P.fy;
    J[1][2] = -x * invz * P.fy;
    J[1][3] = 0;
    J[1][4] = -invz * P.fy;
    J[1][5] = y * invz_2 * P.fy;
    if (st) {
      J[2][0] = J[0][0] - P.bf * y * invz_2;
      J[2][1] = J[0][1] + P.bf * x * invz_2;
      J[2][2] = J[0][2];
      J[2][3] = J[0][3];
      J[2][4] = 0;
      J[2][5] = J[0][5] - P.bf * invz_2;
    }
    const double* e = &S.err[3 * i];
    double w = P.info[i];
    double r1 = 1.0;
    if (S.robust[i]) {
      double rho[2];
      huber(edge_chi2(P, i, e), st ? P.d_stereo : P.d_mono, rho);
      r1 = rho[1];
    }
    for (int a = 0; a < 6; ++a) {
      double s = 0;
      for (int r = 0; r < rows; ++r) s += J[r][a] * (w * e[r]);
      b[a] -= r1 * s;
      for (int c = 0; c < 6; ++c) {
        double h = 0;
        for (int r = 0; r < rows; ++r) h += J[r][a] * (r1 * w) * J[r][c];
        H[a][c] += h;
      }
    }
  }
}

// dense symmetric positive-definite solve (Cholesky); returns false if not PD
bool solve6(const double A[6][6], const double* b, double* x) {
  double L[6][6] = {};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0)) return false;
        L[i][i] = std::sqrt(s);
      } else
        L[i][j] = s / L[j][j];
    }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
  return true;
}

// SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg on the single pose vertex
void optimize(const Problem& P, Pose& T, EdgeState& S, int iterations) {
  bool any = false;
  for (int i = 0; i < P.n; ++i) any = any || S.level[i] == 0;
  if (!any) return;  // g2o: "0 vertices to optimize"
  double lambda = 0, ni = 2;
  for (int it = 0; it < iterations; ++it) {
    compute_active_errors(P, T, S);
    double current_chi = active_robust_chi2(P, S);
    double temp_chi = current_chi;
    double H[6][6], b[6];
    build_system(P, T, S, H, b);
    if (it == 0) {
      double max_diag = 0;
      for (int j = 0; j < 6; ++j) max_diag = std::max(std::fabs(H[j][j]), max_diag);
      lambda = 1e-5 * max_diag;
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const Pose backup = T;  // push()
      double Hl[6][6];
      std::memcpy(Hl, H, sizeof Hl);
      for (int j = 0; j < 6; ++j) Hl[j][j] += lambda;
      double x[6] = {0, 0, 0, 0, 0, 0};
      const bool ok2 = solve6(Hl, b, x);
      T = oplus(T, x);
      compute_active_errors(P, T, S);
      temp_chi = active_robust_chi2(P, S);
      if (!ok2) temp_chi = std::numeric_limits<double>::max();
      rho = current_chi - temp_chi;
      double scale = 0;
      for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(temp_chi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        const double scale_factor = std::max(1. / 3., alpha);
        lambda *= scale_factor;
        ni = 2;
        current_chi = temp_chi;
      } else {
        lambda *= ni;
        ni *= 2;
        T = backup;  // pop()
        if (!std::isfinite(lambda)) break;
      }
      ++qmax;
    } while (rho < 0 && qmax < 10);
    if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) break;  // Terminate
  }
}

}  // namespace

extern "C" {

// Returns edges - nBad of the last round (the value OptimizePoseOnly would return before its projection check).
// sigma2[i] = getScaledFactor2(octave) as float.  inlier_out[i] in {0,1}.
int orc_pose_only_optimize(int n, const double* Xw, const double* meas, const double* info, const float* sigma2, const double* pose_in,
                           double fx, double fy, double cx, double cy, double bf, double* pose_out, uint8_t* inlier_out) {
  Problem P{n, Xw, meas, info, fx, fy, cx, cy, bf, (double)(float)std::sqrt(5.991), (double)(float)std::sqrt(7.815)};
  EdgeState S;
  S.err.assign((size_t)n * 3, 0.0);
  S.level.assign(n, 0);
  S.robust.assign(n, 1);
  Pose T0;
  std::memcpy(T0.q, pose_in, 4 * sizeof(double));
  std::memcpy(T0.t, pose_in + 4, 3 * sizeof(double));
  Pose T = T0;
  for (int i = 0; i < n; ++i) edge_error(P, T, i, &S.err[3 * i]);  // edge->computeError() at construction (Optimizer.cc:90,111)
  std::vector<uint8_t> inlier(n, 1);
  int n_bad = 0;
  for (int round = 0; round < 4; ++round) {
    n_bad = 0;
    T = T0;  // poseVertex->setEstimate(se3): every round restarts from the initial pose (Optimizer.cc:127)
    optimize(P, T, S, 10);
    for (int pass = 0; pass < 2; ++pass) {  // mono edges first, then stereo (two std::map loops, Optimizer.cc:132-177)
      for (int i = 0; i < n; ++i) {
        const bool st = !(meas[3 * i + 2] < 0);
        if ((int)st != pass) continue;
        if (!inlier[i]) edge_error(P, T, i, &S.err[3 * i]);
        const double c = edge_chi2(P, i, &S.err[3 * i]);
        const double th = (st ? 7.815 : 5.991) * sigma2[i];
        if (c > th) {
          inlier[i] = 0;
          S.level[i] = 1;
          ++n_bad;
        } else {
          inlier[i] = 1;
          S.level[i] = 0;
        }
        if (round == 2) S.robust[i] = 0;
      }
    }
  }
  std::memcpy(pose_out, T.q, 4 * sizeof(double));
  std::memcpy(pose_out + 4, T.t, 3 * sizeof(double));
  if (inlier_out) std::memcpy(inlier_out, inlier.data(), n);
  return n - n_bad;
}

}  // extern "C"
