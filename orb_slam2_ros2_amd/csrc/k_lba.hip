// k_lba.hip -- the linear algebra of one Levenberg-Marquardt trial of the local bundle adjustment, on the device.
//
// Replaces what g2o runs inside optimizer.optimize() under Optimizer::OptimizeLocalMap (src/ORB_SLAM2/src/Optimizer.cc:336,361)
// once the normal-equation blocks exist (k_ba.hip): BlockSolver_6_3::setLambda / solve with marginalised points
// (g2o core/block_solver.hpp: lambda on both diagonals, Hll^-1 per point, Schur complement Hpp - Hpl Hll^-1 Hpl^T,
// reduced right-hand side, back-substitution), the LLT of the reduced system (solvers/eigen/linear_solver_eigen.h, here a
// dense Cholesky by one workgroup), SparseOptimizer::update (VertexSE3Expmap / VertexPointXYZ oplus) and the scalars the
// Levenberg control needs (activeRobustChi2, computeLambdaInit, computeScale).  SURVEY 8f, row f4.
// All fp64, no FMA contraction, every sum in a fixed order: results are run-to-run identical.  Block scatter-adds keyed by
// vertex ids and a <= 1024-row triangular factorisation are not dense contractions worth MFMA tiles.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"
#include "se3_dev.h"

namespace orbfe {

// fixed-order workgroup sum / max of one double per thread (blockDim.x == 1024)
__device__ __forceinline__ double block_reduce_1024(double v, double* sh, bool take_max) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) sh[t] = take_max ? fmax(sh[t], sh[t + o]) : sh[t] + sh[t + o];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// activeRobustChi2 (sum of rho(chi2) over level-0 edges) + g2o's per-edge _error bookkeeping: chi2_last is refreshed for the
// edges that were just evaluated, i.e. the active ones
__global__ __launch_bounds__(1024) void k_lba_chi2_sum(int n_edges, const double* __restrict__ chi2, const double* __restrict__ rho,
                                                       const uint8_t* __restrict__ level, double* __restrict__ chi2_last,
                                                       double* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double sh[1024];
  double acc = 0;
  for (int e = threadIdx.x; e < n_edges; e += 1024)
    if (level[e] == 0) {
      acc += rho[(size_t)e * 2];
      chi2_last[e] = chi2[e];
    }
  const double s = block_reduce_1024(acc, sh, false);
  if (threadIdx.x == 0) out[0] = s;
}

// computeLambdaInit: max |H_jj| over the active vertices (fixed poses have no block)
__global__ __launch_bounds__(1024) void k_lba_maxdiag(int n_poses, int n_points, const double* __restrict__ Hpp,
                                                      const double* __restrict__ Hll, const uint8_t* __restrict__ fixed,
                                                      double* __restrict__ out) {
  __shared__ double sh[1024];
  double m = 0;
  for (int i = threadIdx.x; i < n_poses * 6; i += 1024) {
    const int k = i / 6, a = i - 6 * k;
    if (!(fixed && fixed[k])) m = fmax(m, fabs(Hpp[(size_t)k * 36 + 7 * a]));
  }
  for (int i = threadIdx.x; i < n_points * 3; i += 1024) {
    const int p = i / 3, a = i - 3 * p;
    m = fmax(m, fabs(Hll[(size_t)p * 9 + 4 * a]));
  }
  const double r = block_reduce_1024(m, sh, true);
  if (threadIdx.x == 0) out[0] = r;
}

// (Hll + lambda I)^-1 per point (Eigen's 3x3 inverse: cofactors / determinant)
__global__ __launch_bounds__(256) void k_lba_point_inv(int n_points, const double* __restrict__ Hll, const double* __restrict__ lambda_p,
                                                       double* __restrict__ Dinv, int* __restrict__ ok) {
#pragma clang fp contract(off)
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_points) return;
  const double lambda = lambda_p[0];
  double M[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) M[i] = Hll[(size_t)p * 9 + i];
  M[0] += lambda, M[4] += lambda, M[8] += lambda;
  const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  if (det == 0 || !isfinite(det)) {
    *ok = 0;
    return;
  }
  const double id = 1.0 / det;
  double* R = Dinv + (size_t)p * 9;
  R[0] = c00 * id;
  R[1] = (M[2] * M[7] - M[1] * M[8]) * id;
  R[2] = (M[1] * M[5] - M[2] * M[4]) * id;
  R[3] = c01 * id;
  R[4] = (M[0] * M[8] - M[2] * M[6]) * id;
  R[5] = (M[2] * M[3] - M[0] * M[5]) * id;
  R[6] = c02 * id;
  R[7] = (M[1] * M[6] - M[0] * M[7]) * id;
  R[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}

// W(e) = Hpl(e) * Dinv(point(e))   (6x3 per edge; zero for edges of fixed poses because their Hpl is zero)
__global__ __launch_bounds__(256) void k_lba_edge_w(int n_edges, const int32_t* __restrict__ edge_point, const double* __restrict__ Hpl,
                                                    const double* __restrict__ Dinv, double* __restrict__ W) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  const double* H = Hpl + (size_t)e * 18;
  const double* D = Dinv + (size_t)edge_point[e] * 9;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += H[3 * a + k] * D[3 * k + c];
      W[(size_t)e * 18 + 3 * a + c] = s;
    }
}

// Reduced system, one wave per block (i, j) of free poses: lanes 0..35 own S entries, lanes 36..41 (diagonal blocks only)
// the right-hand side.  pairs = (e1, e2) with pose(e1) = i, pose(e2) = j observing the same point, in host order.
//   S_ij = [i == j] (Hpp_i + lambda I) - sum_pairs W(e1) Hpl(e2)^T          rhs_i = bp_i - sum_{e1 of pose i} W(e1) bl(point(e1))
// S is stored column-major with leading dimension n = 6 nf (symmetric, so both triangles are filled).
__global__ __launch_bounds__(64) void k_lba_schur(int nf, const int32_t* __restrict__ free_pose, const int32_t* __restrict__ pair_off,
                                                  const int2* __restrict__ pairs, const int32_t* __restrict__ ps_off,
                                                  const int32_t* __restrict__ ps_edges, const int32_t* __restrict__ edge_point,
                                                  const double* __restrict__ Hpp, const double* __restrict__ bp,
                                                  const double* __restrict__ bl, const double* __restrict__ Hpl,
                                                  const double* __restrict__ W, const double* __restrict__ lambda_p,
                                                  double* __restrict__ S, double* __restrict__ rhs) {
#pragma clang fp contract(off)
  const int i = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
  const int n = 6 * nf;
  const int ki = free_pose[i];
  if (lane < 36) {
    const int a = lane / 6, c = lane - 6 * a;
    double acc = 0;
    if (i == j) {
      acc = Hpp[(size_t)ki * 36 + lane];
      if (a == c) acc += lambda_p[0];
    }
    for (int q = pair_off[i * nf + j]; q < pair_off[i * nf + j + 1]; ++q) {
      const int2 pr = pairs[q];
      const double* w = W + (size_t)pr.x * 18 + 3 * a;
      const double* h = Hpl + (size_t)pr.y * 18 + 3 * c;
      acc -= w[0] * h[0] + w[1] * h[1] + w[2] * h[2];
    }
    S[(size_t)(6 * i + a) + (size_t)(6 * j + c) * n] = acc;
  } else if (lane < 42 && i == j) {
    const int a = lane - 36;
    double acc = bp[(size_t)ki * 6 + a];
    for (int q = ps_off[ki]; q < ps_off[ki + 1]; ++q) {
      const int e = ps_edges[q];
      const double* w = W + (size_t)e * 18 + 3 * a;
      const double* b = bl + (size_t)edge_point[e] * 3;
      acc -= w[0] * b[0] + w[1] * b[1] + w[2] * b[2];
    }
    rhs[6 * i + a] = acc;
  }
}

// Dense Cholesky of the reduced system and both triangular solves by ONE workgroup, in 6x6 blocks (n = 6 nb, nb <= LBA_MAX_FREE).
// Right-looking: per block column (a) the 6x6 diagonal block is factorised by one lane in LDS, (b) every row below it -- and the
// right-hand side, carried along as an extra row, which makes the forward substitution part of the factorisation -- is solved
// against that block, written back to S and parked in an LDS panel, (c) the trailing matrix takes its rank-6 update from the
// panel.  L is also mirrored into the upper triangle of S so that the backward substitution reads coalesced columns, and the
// diagonal blocks stay in LDS.  The dependent chain is nb block steps (a few us each) instead of n column steps through global
// memory (a first version with one barrier pair per scalar column took 3.5 ms at n = 240; this one takes ~0.2 ms).
// ok = 0 if a pivot is not positive (g2o: the linear solver fails, the trial is rejected).
__global__ __launch_bounds__(1024) void k_lba_chol_solve(int n, double* __restrict__ S, const double* __restrict__ rhs,
                                                         double* __restrict__ x, int* __restrict__ ok) {
#pragma clang fp contract(off)
  __shared__ double P[6 * LBA_MAX_FREE + 1][6];  // panel rows of the current block column; row n = the right-hand-side row
  __shared__ double yv[6 * LBA_MAX_FREE];         // right-hand side -> y -> x
  __shared__ double Ld[LBA_MAX_FREE][36];         // factorised diagonal blocks (row-major, lower)
  __shared__ int s_ok;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nb = n / 6;
  if (t < n) yv[t] = rhs[t];
  if (t == 0) s_ok = 1;
  __syncthreads();
  for (int kb = 0; kb < nb; ++kb) {
    const int c0 = 6 * kb;
    double* Lk = Ld[kb];
    if (t < 36) Lk[t] = S[(size_t)(c0 + t / 6) + (size_t)(c0 + t % 6) * n];
    __syncthreads();
    if (t == 0) {  // 6x6 Cholesky, lower triangle in place
      for (int j = 0; j < 6; ++j) {
        double d = Lk[7 * j];
        for (int k = 0; k < j; ++k) d -= Lk[6 * j + k] * Lk[6 * j + k];
        if (!(d > 0) || !isfinite(d)) {
          s_ok = 0;
          break;
        }
        d = sqrt(d);
        Lk[7 * j] = d;
        for (int i = j + 1; i < 6; ++i) {
          double v = Lk[6 * i + j];
          for (int k = 0; k < j; ++k) v -= Lk[6 * i + k] * Lk[6 * j + k];
          Lk[6 * i + j] = v / d;
        }
      }
    }
    __syncthreads();
    if (!s_ok) break;  // uniform
    // (b) rows r in [c0 + 6, n) and the right-hand-side row r == n:  P[r] = S[r, c0..c0+5] * Lk^-T
    for (int r = c0 + 6 + t; r <= n; r += 1024) {
      double v[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double sv = (r < n) ? S[(size_t)r + (size_t)(c0 + c) * n] : yv[c0 + c];
        for (int m = 0; m < c; ++m) sv -= v[m] * Lk[6 * c + m];
        v[c] = sv / Lk[7 * c];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        P[r][c] = v[c];
        if (r < n) {
          S[(size_t)r + (size_t)(c0 + c) * n] = v[c];  // L
          S[(size_t)(c0 + c) + (size_t)r * n] = v[c];  // L^T, for the backward substitution
        } else {
          yv[c0 + c] = v[c];  // y of this block
        }
      }
    }
    __syncthreads();
    // (c) trailing update of the lower triangle (and of the right-hand-side row): one wave per column, lanes over its rows
    for (int c = c0 + 6 + wave; c < n; c += 16) {
      const double p0 = P[c][0], p1 = P[c][1], p2 = P[c][2], p3 = P[c][3], p4 = P[c][4], p5 = P[c][5];
      for (int r = c + lane; r <= n; r += 64) {
        const double sv = P[r][0] * p0 + P[r][1] * p1 + P[r][2] * p2 + P[r][3] * p3 + P[r][4] * p4 + P[r][5] * p5;
        if (r < n)
          S[(size_t)r + (size_t)c * n] -= sv;
        else
          yv[c] -= sv;
      }
    }
    __syncthreads();
  }
  if (!s_ok) {
    if (t == 0) *ok = 0;
    if (t < n) x[t] = 0.0;
    return;
  }
  for (int kb = nb - 1; kb >= 0; --kb) {  // L^T x = y, block by block from the bottom
    const int c0 = 6 * kb;
    const double* Lk = Ld[kb];
    if (t == 0) {
      for (int a = 5; a >= 0; --a) {
        double v = yv[c0 + a];
        for (int m = a + 1; m < 6; ++m) v -= Lk[6 * m + a] * yv[c0 + m];
        yv[c0 + a] = v / Lk[7 * a];
      }
    }
    __syncthreads();
    if (t < c0) {  // rows above: y[t] -= sum_a L[c0+a][t] x[c0+a], L^T read from the upper triangle (coalesced in t)
      double v = yv[t];
#pragma unroll
      for (int a = 0; a < 6; ++a) v -= S[(size_t)t + (size_t)(c0 + a) * n] * yv[c0 + a];
      yv[t] = v;
    }
    __syncthreads();
  }
  if (t < n) x[t] = yv[t];
}

// ---- the same factorisation for reduced systems past the LDS-resident solver (more than LBA_MAX_FREE free keyframes) --------------
// Optimizer::OptimizeLocalMap takes ALL keyframes covisible with the current one (getConnectedKfs(0), src/Optimizer.cc:232) -- there is
// no bound on their number, so the reduced system must not have one either.  Same right-looking 6x6-blocked algorithm, with the
// panel, the diagonal blocks and the right-hand side in global memory and one block column per pair of launches: the panel step by
// one workgroup (the dependent part: a 6x6 factorisation and n - c0 independent row solves), the trailing update by as many
// workgroups as it has columns / 16 -- that update is where the n^3 / 3 flops are, and a single CU would take ~40 ms for it at
// 300 keyframes.  A failed pivot clears *ok; every later launch of the factorisation then returns at once.
__global__ __launch_bounds__(1024) void k_lba_chol_panel(int n, int kb, double* __restrict__ S, double* __restrict__ Pg,
                                                         double* __restrict__ Ldg, double* __restrict__ yg, int* __restrict__ ok) {
#pragma clang fp contract(off)
  __shared__ double Lk[36];
  __shared__ int s_ok;
  const int t = threadIdx.x, c0 = 6 * kb;
  if (*ok == 0) return;  // uniform
  if (t < 36) Lk[t] = S[(size_t)(c0 + t / 6) + (size_t)(c0 + t % 6) * n];
  if (t == 0) s_ok = 1;
  __syncthreads();
  if (t == 0) {
    for (int j = 0; j < 6; ++j) {
      double d = Lk[7 * j];
      for (int k = 0; k < j; ++k) d -= Lk[6 * j + k] * Lk[6 * j + k];
      if (!(d > 0) || !isfinite(d)) {
        s_ok = 0;
        break;
      }
      d = sqrt(d);
      Lk[7 * j] = d;
      for (int i = j + 1; i < 6; ++i) {
        double v = Lk[6 * i + j];
        for (int k = 0; k < j; ++k) v -= Lk[6 * i + k] * Lk[6 * j + k];
        Lk[6 * i + j] = v / d;
      }
    }
  }
  __syncthreads();
  if (!s_ok) {
    if (t == 0) *ok = 0;
    return;
  }
  if (t < 36) Ldg[(size_t)kb * 36 + t] = Lk[t];
  for (int r = c0 + 6 + t; r <= n; r += 1024) {  // rows below the diagonal block, and the right-hand-side row r == n
    double v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double sv = (r < n) ? S[(size_t)r + (size_t)(c0 + c) * n] : yg[c0 + c];
      for (int m = 0; m < c; ++m) sv -= v[m] * Lk[6 * c + m];
      v[c] = sv / Lk[7 * c];
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      Pg[(size_t)r * 6 + c] = v[c];
      if (r < n) {
        S[(size_t)r + (size_t)(c0 + c) * n] = v[c];  // L
        S[(size_t)(c0 + c) + (size_t)r * n] = v[c];  // L^T, for the backward substitution
      } else {
        yg[c0 + c] = v[c];  // y of this block
      }
    }
  }
}

__global__ __launch_bounds__(1024) void k_lba_chol_trail(int n, int kb, double* __restrict__ S, const double* __restrict__ Pg,
                                                         double* __restrict__ yg, const int* __restrict__ ok) {
#pragma clang fp contract(off)
  if (*ok == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = 6 * kb + 6 + (int)blockIdx.x * 16 + wave;  // one wave per column of the trailing matrix, lanes over its rows
  if (c >= n) return;
  const double* pc = Pg + (size_t)c * 6;
  const double p0 = pc[0], p1 = pc[1], p2 = pc[2], p3 = pc[3], p4 = pc[4], p5 = pc[5];
  for (int r = c + lane; r <= n; r += 64) {
    const double* pr = Pg + (size_t)r * 6;
    const double sv = pr[0] * p0 + pr[1] * p1 + pr[2] * p2 + pr[3] * p3 + pr[4] * p4 + pr[5] * p5;
    if (r < n)
      S[(size_t)r + (size_t)c * n] -= sv;
    else
      yg[c] -= sv;
  }
}

__global__ __launch_bounds__(1024) void k_lba_chol_back(int n, const double* __restrict__ S, const double* __restrict__ Ldg,
                                                        double* __restrict__ yg, double* __restrict__ x, const int* __restrict__ ok) {
#pragma clang fp contract(off)
  const int t = threadIdx.x;
  if (*ok == 0) {
    for (int i = t; i < n; i += 1024) x[i] = 0.0;
    return;
  }
  const int nb = n / 6;
  for (int kb = nb - 1; kb >= 0; --kb) {  // L^T x = y, block by block from the bottom
    const int c0 = 6 * kb;
    const double* Lk = Ldg + (size_t)kb * 36;
    if (t == 0) {
      for (int a = 5; a >= 0; --a) {
        double v = yg[c0 + a];
        for (int m = a + 1; m < 6; ++m) v -= Lk[6 * m + a] * yg[c0 + m];
        yg[c0 + a] = v / Lk[7 * a];
      }
    }
    __syncthreads();  // (one workgroup: its own global writes are visible to it after the barrier)
    for (int r = t; r < c0; r += 1024) {
      double v = yg[r];
#pragma unroll
      for (int a = 0; a < 6; ++a) v -= S[(size_t)r + (size_t)(c0 + a) * n] * yg[c0 + a];
      yg[r] = v;
    }
    __syncthreads();
  }
  for (int i = t; i < n; i += 1024) x[i] = yg[i];
}

// SparseOptimizer::update: poses <- exp(dx) * pose (free poses), points <- point + Dinv (bl - sum Hpl^T dxp)
__global__ __launch_bounds__(256) void k_lba_update(int n_poses, int n_points, const int32_t* __restrict__ pose_slot,
                                                    const double* __restrict__ x, const int32_t* __restrict__ pt_off,
                                                    const int32_t* __restrict__ pt_edges, const int32_t* __restrict__ edge_pose,
                                                    const double* __restrict__ Hpl, const double* __restrict__ bl,
                                                    const double* __restrict__ Dinv, double* __restrict__ poses, double* __restrict__ points,
                                                    double* __restrict__ dxp, double* __restrict__ dxl) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n_points) {
    const int p = t;
    double r0 = bl[(size_t)p * 3], r1 = bl[(size_t)p * 3 + 1], r2 = bl[(size_t)p * 3 + 2];
    for (int q = pt_off[p]; q < pt_off[p + 1]; ++q) {
      const int e = pt_edges[q];
      const int s = pose_slot[edge_pose[e]];
      if (s < 0) continue;
      const double* h = Hpl + (size_t)e * 18;
      const double* d = x + 6 * s;
      double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        s0 += h[3 * a] * d[a];
        s1 += h[3 * a + 1] * d[a];
        s2 += h[3 * a + 2] * d[a];
      }
      r0 -= s0, r1 -= s1, r2 -= s2;
    }
    const double* D = Dinv + (size_t)p * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double dv = D[3 * a] * r0 + D[3 * a + 1] * r1 + D[3 * a + 2] * r2;
      dxl[(size_t)p * 3 + a] = dv;
      points[(size_t)p * 3 + a] += dv;
    }
  } else if (t < n_points + n_poses) {
    const int k = t - n_points;
    const int s = pose_slot[k];
    double upd[6] = {0, 0, 0, 0, 0, 0};
    if (s >= 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a) upd[a] = x[6 * s + a];
      PoseDev T, R;
      for (int a = 0; a < 4; ++a) T.q[a] = poses[(size_t)k * 7 + a];
      for (int a = 0; a < 3; ++a) T.t[a] = poses[(size_t)k * 7 + 4 + a];
      pose_oplus(T, upd, R);
      for (int a = 0; a < 4; ++a) poses[(size_t)k * 7 + a] = R.q[a];
      for (int a = 0; a < 3; ++a) poses[(size_t)k * 7 + 4 + a] = R.t[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) dxp[(size_t)k * 6 + a] = upd[a];
  }
}

// computeScale: sum_i dx_i (lambda dx_i + b_i) over the free poses and the points
__global__ __launch_bounds__(1024) void k_lba_scale(int n_poses, int n_points, const uint8_t* __restrict__ fixed,
                                                    const double* __restrict__ dxp, const double* __restrict__ bp,
                                                    const double* __restrict__ dxl, const double* __restrict__ bl,
                                                    const double* __restrict__ lambda_p, double* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double sh[1024];
  const double lambda = lambda_p[0];
  double acc = 0;
  for (int i = threadIdx.x; i < n_poses * 6; i += 1024)
    if (!(fixed && fixed[i / 6])) acc += dxp[i] * (lambda * dxp[i] + bp[i]);
  for (int i = threadIdx.x; i < n_points * 3; i += 1024) acc += dxl[i] * (lambda * dxl[i] + bl[i]);
  const double s = block_reduce_1024(acc, sh, false);
  if (threadIdx.x == 0) out[0] = s;
}

// Optimizer.cc:338-359 between the two rounds: level 1 for chi2 > 5.991 / 7.815 or non-positive depth, kernels dropped
__global__ __launch_bounds__(256) void k_lba_classify(int n_edges, const double* __restrict__ chi2_last, const uint8_t* __restrict__ depth_pos,
                                                      const uint8_t* __restrict__ is_stereo, uint8_t* __restrict__ level,
                                                      double* __restrict__ info_eff, double* __restrict__ delta_eff) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  const double th = is_stereo[e] ? 7.815 : 5.991;
  if (chi2_last[e] > th || !depth_pos[e]) {
    level[e] = 1;
    info_eff[e] = 0.0;
  }
  delta_eff[e] = -1.0;
}

// final report (Optimizer.cc:364-391): chi2 at the final estimates and the same test
__global__ __launch_bounds__(256) void k_lba_final(int n_edges, const double* __restrict__ chi2, const uint8_t* __restrict__ depth_pos,
                                                   const uint8_t* __restrict__ is_stereo, uint8_t* __restrict__ bad) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  bad[e] = (chi2[e] > (is_stereo[e] ? 7.815 : 5.991) || !depth_pos[e]) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
void launch_lba_chi2_sum(hipStream_t s, int n_edges, const double* chi2, const double* rho, const uint8_t* level, double* chi2_last,
                         double* out) {
  hipLaunchKernelGGL(k_lba_chi2_sum, dim3(1), dim3(1024), 0, s, n_edges, chi2, rho, level, chi2_last, out);
}
void launch_lba_maxdiag(hipStream_t s, int n_poses, int n_points, const double* Hpp, const double* Hll, const uint8_t* fixed, double* out) {
  hipLaunchKernelGGL(k_lba_maxdiag, dim3(1), dim3(1024), 0, s, n_poses, n_points, Hpp, Hll, fixed, out);
}
void launch_lba_solve(hipStream_t s, int n_poses, int n_points, int n_edges, int nf, const int32_t* free_pose, const int32_t* pose_slot,
                      const int32_t* pair_off, const int2* pairs, const int32_t* ps_off, const int32_t* ps_edges, const int32_t* pt_off,
                      const int32_t* pt_edges, const int32_t* edge_pose, const int32_t* edge_point, const uint8_t* fixed, const double* Hpp,
                      const double* bp, const double* Hll, const double* bl, const double* Hpl, const double* lambda_p, double* Dinv, double* W,
                      double* S, double* rhs, double* x, int* ok, double* poses, double* points, double* dxp, double* dxl, double* scale_out,
                      double* big_scratch) {
  if (n_points > 0) hipLaunchKernelGGL(k_lba_point_inv, dim3((n_points + 255) / 256), dim3(256), 0, s, n_points, Hll, lambda_p, Dinv, ok);
  if (n_edges > 0) hipLaunchKernelGGL(k_lba_edge_w, dim3((n_edges + 255) / 256), dim3(256), 0, s, n_edges, edge_point, Hpl, Dinv, W);
  if (nf > 0) {
    hipLaunchKernelGGL(k_lba_schur, dim3(nf, nf), dim3(64), 0, s, nf, free_pose, pair_off, pairs, ps_off, ps_edges, edge_point, Hpp, bp, bl,
                       Hpl, W, lambda_p, S, rhs);
    if (nf <= LBA_MAX_FREE) {
      hipLaunchKernelGGL(k_lba_chol_solve, dim3(1), dim3(1024), 0, s, 6 * nf, S, rhs, x, ok);
    } else {  // big_scratch: panel [(n + 1) * 6] | diagonal blocks [nf * 36] | right-hand side [n]
      const int n = 6 * nf;
      double* Pg = big_scratch;
      double* Ldg = Pg + (size_t)(n + 1) * 6;
      double* yg = Ldg + (size_t)nf * 36;
      (void)hipMemcpyAsync(yg, rhs, sizeof(double) * n, hipMemcpyDeviceToDevice, s);
      for (int kb = 0; kb < nf; ++kb) {
        hipLaunchKernelGGL(k_lba_chol_panel, dim3(1), dim3(1024), 0, s, n, kb, S, Pg, Ldg, yg, ok);
        const int cols = n - 6 * kb - 6;
        if (cols > 0) hipLaunchKernelGGL(k_lba_chol_trail, dim3((cols + 15) / 16), dim3(1024), 0, s, n, kb, S, Pg, yg, ok);
      }
      hipLaunchKernelGGL(k_lba_chol_back, dim3(1), dim3(1024), 0, s, n, S, Ldg, yg, x, ok);
    }
  }
  const int nt = n_points + n_poses;
  if (nt > 0)
    hipLaunchKernelGGL(k_lba_update, dim3((nt + 255) / 256), dim3(256), 0, s, n_poses, n_points, pose_slot, x, pt_off, pt_edges, edge_pose, Hpl,
                       bl, Dinv, poses, points, dxp, dxl);
  hipLaunchKernelGGL(k_lba_scale, dim3(1), dim3(1024), 0, s, n_poses, n_points, fixed, dxp, bp, dxl, bl, lambda_p, scale_out);
}
void launch_lba_classify(hipStream_t s, int n_edges, const double* chi2_last, const uint8_t* depth_pos, const uint8_t* is_stereo,
                         uint8_t* level, double* info_eff, double* delta_eff) {
  if (n_edges > 0)
    hipLaunchKernelGGL(k_lba_classify, dim3((n_edges + 255) / 256), dim3(256), 0, s, n_edges, chi2_last, depth_pos, is_stereo, level, info_eff,
                       delta_eff);
}
void launch_lba_final(hipStream_t s, int n_edges, const double* chi2, const uint8_t* depth_pos, const uint8_t* is_stereo, uint8_t* bad) {
  if (n_edges > 0) hipLaunchKernelGGL(k_lba_final, dim3((n_edges + 255) / 256), dim3(256), 0, s, n_edges, chi2, depth_pos, is_stereo, bad);
}

}  // namespace orbfe
