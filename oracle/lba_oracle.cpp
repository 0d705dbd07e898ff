// lba_oracle.cpp -- CPU ORACLE for the g2o part of Optimizer::OptimizeLocalMap (test infrastructure, NOT product code).
//
// PARITY UNPINNED (see orb_oracle.cpp): g2o (20241228) is not vendored in the reference and not installed here.  Restates
//   src/ORB_SLAM2/src/Optimizer.cc:336-362   the reference's driver: optimize(5) with Huber kernels on every edge, then every
//                                            edge with chi2 > 5.991 (mono) / 7.815 (stereo) or non-positive depth goes to
//                                            level 1, ALL robust kernels are dropped, optimize(10) on level 0;
//   src/ORB_SLAM2/src/Optimizer.cc:364-391   final computeError() + the same test on every edge (reported, the map
//                                            bookkeeping that follows stays with the caller);
//   g2o core/optimization_algorithm_levenberg.cpp   tau = 1e-5, gain ratio with scale = dx.(lambda dx + b) + 1e-3,
//                                            lambda *= max(1/3, min(2/3, 1-(2rho-1)^3)) / lambda *= ni, ni *= 2, at most 10
//                                            trials per iteration, lambda re-initialised by every optimize() call;
//   g2o core/block_solver.hpp                BlockSolver_6_3 with marginalised points: lambda on both diagonals, Schur
//                                            complement Hpp - Hpl Hll^-1 Hpl^T, back-substitution for the points;
//   g2o solvers/eigen/linear_solver_eigen.h  sparse LLT of the reduced system -- restated as a DENSE Cholesky (same
//                                            factorisation up to the elimination order, i.e. up to rounding).
// edge->chi2() after optimize() is the chi2 of the LAST EVALUATED trial (accepted or not), exactly as g2o leaves _error;
// isDepthPositive() always reads the current estimates.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

extern "C" {
void orc_ba_eval_edges(int n_edges, const double* poses, const double* points, const int32_t* edge_pose, const int32_t* edge_point,
                       const double* meas, const uint8_t* is_stereo, const double* info, const double* delta, double fx, double fy,
                       double cx, double cy, double bf, double* error, double* chi2, double* rho, double* j_point, double* j_pose,
                       uint8_t* depth_positive);
void orc_ba_build_system(int n_poses, int n_points, int n_edges, const double* poses, const double* points, const int32_t* edge_pose,
                         const int32_t* edge_point, const double* meas, const uint8_t* is_stereo, const double* info, const double* delta,
                         double fx, double fy, double cx, double cy, double bf, const uint8_t* pose_fixed, double* Hpp, double* bp,
                         double* Hll, double* bl, double* Hpl, double* chi2_robust_sum);
void orc_se3_oplus(const double* T, const double* upd, double* out);
}

namespace {

struct Lba {
  int NK, NP, E;
  const int32_t *ek, *ep;
  const double* meas;
  const uint8_t* st;
  const double* info;  // as given
  double fx, fy, cx, cy, bf;
  const uint8_t* fixed;
  std::vector<double> poses, points;      // current estimates
  std::vector<double> info_eff, delta;    // information with level-1 edges zeroed, Huber delta (<= 0: none)
  std::vector<uint8_t> level;
  std::vector<double> chi2_last;          // chi2 of every edge at its last evaluation (g2o's _error)
};

// computeActiveErrors + activeRobustChi2: sum of rho(chi2) over the active edges; refreshes chi2_last of the active edges
double active_chi2(Lba& S) {
  std::vector<double> chi2(S.E), rho((size_t)S.E * 2);
  orc_ba_eval_edges(S.E, S.poses.data(), S.points.data(), S.ek, S.ep, S.meas, S.st, S.info_eff.data(), S.delta.data(), S.fx, S.fy, S.cx,
                    S.cy, S.bf, nullptr, chi2.data(), rho.data(), nullptr, nullptr, nullptr);
  double total = 0;
  std::vector<double> plain(S.E);
  orc_ba_eval_edges(S.E, S.poses.data(), S.points.data(), S.ek, S.ep, S.meas, S.st, S.info, S.delta.data(), S.fx, S.fy, S.cx, S.cy, S.bf,
                    nullptr, plain.data(), nullptr, nullptr, nullptr, nullptr);
  for (int e = 0; e < S.E; ++e)
    if (S.level[e] == 0) {
      total += rho[(size_t)e * 2];
      S.chi2_last[e] = plain[e];
    }
  return total;
}

// dense Cholesky solve of the n x n symmetric system (row-major A is overwritten); false if not positive definite
bool chol_solve(std::vector<double>& A, int n, const double* b, double* x) {
  for (int j = 0; j < n; ++j) {
    double s = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(s > 0) || !std::isfinite(s)) return false;
    const double d = std::sqrt(s);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double t = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) t -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = t / d;
    }
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * y[k];
    y[i] = s / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * x[k];
    x[i] = s / A[(size_t)i * n + i];
  }
  return true;
}

bool inv3(const double* M, double* R) {  // Eigen 3x3 inverse (cofactors / determinant)
  const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  if (det == 0 || !std::isfinite(det)) return false;
  const double id = 1.0 / det;
  R[0] = c00 * id;
  R[1] = (M[2] * M[7] - M[1] * M[8]) * id;
  R[2] = (M[1] * M[5] - M[2] * M[4]) * id;
  R[3] = c01 * id;
  R[4] = (M[0] * M[8] - M[2] * M[6]) * id;
  R[5] = (M[2] * M[3] - M[0] * M[5]) * id;
  R[6] = c02 * id;
  R[7] = (M[1] * M[6] - M[0] * M[7]) * id;
  R[8] = (M[0] * M[4] - M[1] * M[3]) * id;
  return true;
}

// BlockSolver::solve with marginalised points: (H + lambda I) dx = b by Schur complement.  dxp [NK][6] (0 for fixed), dxl [NP][3].
bool schur_solve(const Lba& S, const double* Hpp, const double* bp, const double* Hll, const double* bl, const double* Hpl, double lambda,
                 double* dxp, double* dxl) {
  std::vector<int> slot(S.NK, -1);
  int nf = 0;
  for (int k = 0; k < S.NK; ++k)
    if (!(S.fixed && S.fixed[k])) slot[k] = nf++;
  const int n = 6 * nf;
  std::vector<double> A((size_t)n * n, 0.0), rhs(n, 0.0), Dinv((size_t)S.NP * 9);
  for (int k = 0; k < S.NK; ++k) {
    if (slot[k] < 0) continue;
    for (int a = 0; a < 6; ++a) {
      for (int c = 0; c < 6; ++c) A[(size_t)(6 * slot[k] + a) * n + 6 * slot[k] + c] = Hpp[(size_t)k * 36 + 6 * a + c];
      A[(size_t)(6 * slot[k] + a) * n + 6 * slot[k] + a] += lambda;
      rhs[6 * slot[k] + a] = bp[(size_t)k * 6 + a];
    }
  }
  for (int p = 0; p < S.NP; ++p) {
    double D[9];
    std::memcpy(D, Hll + (size_t)p * 9, sizeof D);
    D[0] += lambda, D[4] += lambda, D[8] += lambda;
    if (!inv3(D, &Dinv[(size_t)p * 9])) return false;
  }
  // point -> edges lists
  std::vector<std::vector<int>> obs(S.NP);
  for (int e = 0; e < S.E; ++e)
    if (slot[S.ek[e]] >= 0) obs[S.ep[e]].push_back(e);
  for (int p = 0; p < S.NP; ++p) {
    const double* Di = &Dinv[(size_t)p * 9];
    for (int e1 : obs[p]) {
      double W[18];  // Hpl(e1) * Dinv  (6x3)
      for (int a = 0; a < 6; ++a)
        for (int c = 0; c < 3; ++c) {
          double s = 0;
          for (int k = 0; k < 3; ++k) s += Hpl[(size_t)e1 * 18 + 3 * a + k] * Di[3 * k + c];
          W[3 * a + c] = s;
        }
      const int i = slot[S.ek[e1]];
      for (int a = 0; a < 6; ++a) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += W[3 * a + k] * bl[(size_t)p * 3 + k];
        rhs[6 * i + a] -= s;
      }
      for (int e2 : obs[p]) {
        const int j = slot[S.ek[e2]];
        for (int a = 0; a < 6; ++a)
          for (int c = 0; c < 6; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += W[3 * a + k] * Hpl[(size_t)e2 * 18 + 3 * c + k];
            A[(size_t)(6 * i + a) * n + 6 * j + c] -= s;
          }
      }
    }
  }
  std::vector<double> x(n, 0.0);
  if (n > 0 && !chol_solve(A, n, rhs.data(), x.data())) return false;
  std::fill(dxp, dxp + (size_t)S.NK * 6, 0.0);
  for (int k = 0; k < S.NK; ++k)
    if (slot[k] >= 0) std::memcpy(dxp + (size_t)k * 6, &x[6 * slot[k]], 6 * sizeof(double));
  for (int p = 0; p < S.NP; ++p) {
    double r[3] = {bl[(size_t)p * 3], bl[(size_t)p * 3 + 1], bl[(size_t)p * 3 + 2]};
    for (int e : obs[p]) {
      const double* dk = dxp + (size_t)S.ek[e] * 6;
      for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int a = 0; a < 6; ++a) s += Hpl[(size_t)e * 18 + 3 * a + c] * dk[a];
        r[c] -= s;
      }
    }
    const double* Di = &Dinv[(size_t)p * 9];
    for (int a = 0; a < 3; ++a) dxl[(size_t)p * 3 + a] = Di[3 * a] * r[0] + Di[3 * a + 1] * r[1] + Di[3 * a + 2] * r[2];
  }
  return true;
}

// SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg; returns the number of solve() calls made
int optimize(Lba& S, int iterations, const volatile int* stop, double* lambda_out) {
  bool any = false;
  for (int e = 0; e < S.E; ++e) any = any || S.level[e] == 0;
  if (!any) return 0;
  std::vector<double> Hpp((size_t)S.NK * 36), bp((size_t)S.NK * 6), Hll((size_t)S.NP * 9), bl((size_t)S.NP * 3), Hpl((size_t)S.E * 18);
  std::vector<double> dxp((size_t)S.NK * 6), dxl((size_t)S.NP * 3);
  double lambda = 0, ni = 2;
  int done = 0;
  for (int it = 0; it < iterations; ++it) {
    if (stop && *stop) break;
    ++done;
    double current_chi = active_chi2(S);
    double temp_chi = current_chi;
    orc_ba_build_system(S.NK, S.NP, S.E, S.poses.data(), S.points.data(), S.ek, S.ep, S.meas, S.st, S.info_eff.data(), S.delta.data(), S.fx,
                        S.fy, S.cx, S.cy, S.bf, S.fixed, Hpp.data(), bp.data(), Hll.data(), bl.data(), Hpl.data(), nullptr);
    if (it == 0) {
      double max_diag = 0;
      for (int k = 0; k < S.NK; ++k)
        for (int a = 0; a < 6; ++a) max_diag = std::max(std::fabs(Hpp[(size_t)k * 36 + 7 * a]), max_diag);
      for (int p = 0; p < S.NP; ++p)
        for (int a = 0; a < 3; ++a) max_diag = std::max(std::fabs(Hll[(size_t)p * 9 + 4 * a]), max_diag);
      lambda = 1e-5 * max_diag;
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const std::vector<double> poses_backup = S.poses, points_backup = S.points;  // push()
      const bool ok2 = schur_solve(S, Hpp.data(), bp.data(), Hll.data(), bl.data(), Hpl.data(), lambda, dxp.data(), dxl.data());
      if (ok2) {
        for (int k = 0; k < S.NK; ++k) {
          if (S.fixed && S.fixed[k]) continue;
          double out[7];
          orc_se3_oplus(&S.poses[(size_t)k * 7], &dxp[(size_t)k * 6], out);
          if (out[3] < 0)  // SE3Quat::normalizeRotation: w >= 0
            for (int a = 0; a < 4; ++a) out[a] = -out[a];
          std::memcpy(&S.poses[(size_t)k * 7], out, sizeof out);
        }
        for (size_t i = 0; i < S.points.size(); ++i) S.points[i] += dxl[i];
      } else {
        std::fill(dxp.begin(), dxp.end(), 0.0);
        std::fill(dxl.begin(), dxl.end(), 0.0);
      }
      temp_chi = active_chi2(S);
      if (!ok2) temp_chi = std::numeric_limits<double>::max();
      rho = current_chi - temp_chi;
      double scale = 0;
      for (int k = 0; k < S.NK; ++k)
        if (!(S.fixed && S.fixed[k]))
          for (int a = 0; a < 6; ++a) scale += dxp[(size_t)k * 6 + a] * (lambda * dxp[(size_t)k * 6 + a] + bp[(size_t)k * 6 + a]);
      for (size_t i = 0; i < dxl.size(); ++i) scale += dxl[i] * (lambda * dxl[i] + bl[i]);
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
        S.poses = poses_backup;  // pop()
        S.points = points_backup;
        if (!std::isfinite(lambda)) break;
      }
      ++qmax;
    } while (rho < 0 && qmax < 10 && !(stop && *stop));
    if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) break;  // Terminate
  }
  if (lambda_out) *lambda_out = lambda;
  return done;
}

}  // namespace

extern "C" {

// Problem layout as orbfe_ba_problem.  Outputs (any may be NULL): poses_out [n_poses][7], points_out [n_points][3],
// level_out [n_edges] (1: excluded from the second round), chi2_out [n_edges] / bad_out [n_edges] (final computeError() and the
// chi2 / depth test of Optimizer.cc:364-391), iters_out[2] (solve() calls of the two rounds).
void orc_ba_local_optimize(int n_poses, int n_points, int n_edges, const double* poses, const double* points, const int32_t* edge_pose,
                           const int32_t* edge_point, const double* meas, const uint8_t* is_stereo, const double* info,
                           const double* huber_delta, double fx, double fy, double cx, double cy, double bf, const uint8_t* pose_fixed,
                           int iters1, int iters2, double* poses_out, double* points_out, uint8_t* level_out, double* chi2_out,
                           uint8_t* bad_out, int32_t* iters_out) {
  Lba S;
  S.NK = n_poses, S.NP = n_points, S.E = n_edges;
  S.ek = edge_pose, S.ep = edge_point, S.meas = meas, S.st = is_stereo, S.info = info;
  S.fx = fx, S.fy = fy, S.cx = cx, S.cy = cy, S.bf = bf, S.fixed = pose_fixed;
  S.poses.assign(poses, poses + (size_t)n_poses * 7);
  S.points.assign(points, points + (size_t)n_points * 3);
  S.info_eff.assign(info, info + n_edges);
  S.delta.assign(huber_delta, huber_delta + n_edges);
  S.level.assign(n_edges, 0);
  S.chi2_last.assign(n_edges, 0.0);
  int it1 = optimize(S, iters1, nullptr, nullptr);
  {  // Optimizer.cc:338-359
    std::vector<uint8_t> dp(n_edges);
    orc_ba_eval_edges(n_edges, S.poses.data(), S.points.data(), edge_pose, edge_point, meas, is_stereo, info, S.delta.data(), fx, fy, cx, cy,
                      bf, nullptr, nullptr, nullptr, nullptr, nullptr, dp.data());
    for (int e = 0; e < n_edges; ++e) {
      const double th = is_stereo[e] ? 7.815 : 5.991;
      if (S.chi2_last[e] > th || !dp[e]) {
        S.level[e] = 1;
        S.info_eff[e] = 0.0;
      }
      S.delta[e] = -1.0;  // setRobustKernel(nullptr)
    }
  }
  int it2 = optimize(S, iters2, nullptr, nullptr);
  if (poses_out) std::memcpy(poses_out, S.poses.data(), sizeof(double) * S.poses.size());
  if (points_out) std::memcpy(points_out, S.points.data(), sizeof(double) * S.points.size());
  if (level_out) std::memcpy(level_out, S.level.data(), n_edges);
  if (chi2_out || bad_out) {
    std::vector<double> c2(n_edges);
    std::vector<uint8_t> dp(n_edges);
    orc_ba_eval_edges(n_edges, S.poses.data(), S.points.data(), edge_pose, edge_point, meas, is_stereo, info, S.delta.data(), fx, fy, cx, cy,
                      bf, nullptr, c2.data(), nullptr, nullptr, nullptr, dp.data());
    for (int e = 0; e < n_edges; ++e) {
      if (chi2_out) chi2_out[e] = c2[e];
      if (bad_out) bad_out[e] = (c2[e] > (is_stereo[e] ? 7.815 : 5.991) || !dp[e]) ? 1 : 0;
    }
  }
  if (iters_out) iters_out[0] = it1, iters_out[1] = it2;
}

}  // extern "C"
