// k_lm.hip -- Optimizer::OptimizeLocalMap's two optimizer.optimize() calls (src/ORB_SLAM2/src/Optimizer.cc:336-361) with g2o's
// Levenberg-Marquardt CONTROL on the device: lambda, the gain ratio, accept / restore, the trial and iteration counters, the round
// switch (outlier classification, :338-359) and the stop-flag poll all live in one small state record that a one-lane control
// kernel advances between the trials; the host only enqueues a fixed stream of (gated) kernels and synchronises ONCE per call
// (twice when a round needed more rejected trials than were provisioned).  Round 2 ran this loop on the host with two blocking
// synchronisations per trial: 12.2 ms for BASELINE config 5, most of it round trips and three kernels written for convenience.
//
// One TRIAL (= one step of the enqueued stream), every kernel gated by state.run_step (six launches; every launch of a dependent
// stream costs 4.6 us before it does anything, so the pose side of the system rides in k_lm_prep):
//   k_lm_ctrl        (lm_ctrl_body) decide the outstanding trial (rho = (chi_cur - chi_trial) / (scale + 1e-3); accept: lambda *= max(1/3, 1 - (2 rho - 1)^3),
//                    swap the estimate / system buffers; reject: lambda *= ni, ni *= 2), start the next iteration or trial, or finish
//   k_lm_prep        per point: (Hll + lambda I)^-1, W(e) = Hpl(e) Dinv for its edges             (BlockSolver_6_3::solve, marginalised points)
//                    + Hpp / bp per pose of the CURRENT system from its stored edge terms (blocks after the point blocks)
//   k_lm_schur       one wave per block (i >= j) of free poses: S_ij = [i == j](Hpp_i + lambda I) - sum W(e1) Hpl(e2)^T, lanes over the pairs
//   k_lm_chol        dense Cholesky + both substitutions of the reduced system by ONE workgroup with the matrix in REGISTERS:
//                    one 6x6 block per thread, the panel of a block column exchanged through LDS (up to 42 free keyframes)
//   k_lm_update      oplus into the OTHER estimate buffer (no push / pop copies), computeScale partial sums
//   k_lm_linpoints   eight lanes per point, one edge each, at the trial estimate: error, chi2, Huber weight, both Jacobians ONCE, Hpl,
//                    Hll / bl of the point, partial sums of the robust chi2 -- kept as the system of the next iteration if the trial is
//                    accepted (g2o rebuilds the same numbers); vertex -> edge lists are host-built CSR
// All fp64, contraction off, every sum in a fixed order: run-to-run identical.  Scatter-adds of 6x6 / 6x3 / 3x3 blocks keyed by vertex
// ids and a <= 258-row triangular factorisation are no dense contractions worth MFMA tiles; past LM_CHOL_MAX_NB free
// keyframes the reduced system IS one (6 nf rows, dense): k_lmbig.hip factorises it in 48 x 48 tiles with the trailing updates on the
// fp64 matrix cores, gated by the same state record.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"
#include "se3_dev.h"
#include "wave_ops.h"

namespace orbfe {

// ---------------------------------------------------------------------------------------------------------------------------------
// per-edge terms at one estimate (what linearizeOplus + robustify leave behind): 32 doubles
//   [0..8] A = d e / d point (3x3, row-major; row 2 zero for mono)   [9..26] B = d e / d pose (3x6)   [27..29] w * e   [30] w   [31] rows
// ---------------------------------------------------------------------------------------------------------------------------------
#define LM_TERM 32

__device__ __forceinline__ double block_sum_256(double v, double* sh) {  // fixed tree, blockDim.x == 256
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) sh[t] = sh[t] + sh[t + o];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// The estimates, the edge terms and the normal-equation blocks exist twice: [cur] = the current estimate and its system,
// [cur ^ 1] = the trial's.  `which` = 0: work on cur, 1: on cur ^ 1.
struct LmBuffers {
  double* poses[2];
  double* points[2];
  double* terms[2];   // [E][LM_TERM]
  double* Hpl[2];     // [E][18]
  double* Hpp[2];     // [NK][36]
  double* bp[2];      // [NK][6]
  double* Hll[2];     // [NP][9]
  double* bl[2];      // [NP][3]
  double* chi_part[2];  // per linearize block: sum of rho(chi2) over its active edges
};

__device__ __forceinline__ bool lm_gate(const LmState* st, int gate) {
  // gate 0: always | 1: run_step | 2: run_switch | 3: run_final
  if (gate == 0) return true;
  const volatile int32_t* p = gate == 1 ? &st->run_step : (gate == 2 ? &st->run_switch : &st->run_final);
  return *p != 0;
}

// (fixed tree over the 64 lanes by data-parallel-primitive moves: wave_ops.h)
__device__ __forceinline__ double wave_sum_fixed(double v) { return wave_sum_f64(v); }

__device__ void lm_ctrl_body(LmState* __restrict__ st, const LmBuffers& B, int mode, int chi_blocks, int scale_blocks,
                             const double* __restrict__ scale_part, const volatile uint8_t* __restrict__ abort_flag);

// One edge at the estimate of buffer `buf`: error, chi2, Huber weight, both Jacobians, Hpl -> terms / Hpl / chi2_last; returns the
// robust chi2 (0 for an inactive edge) and leaves A (3x3), w e (3), w and the row count in registers for the point's blocks.
__device__ __forceinline__ double lm_linearize_edge(int e, int buf, const LmBuffers& B, const int32_t* __restrict__ edge_pose, int pt,
                                                    const double* __restrict__ meas, const uint8_t* __restrict__ is_stereo,
                                                    const double* __restrict__ info, const double* __restrict__ delta,
                                                    const uint8_t* __restrict__ pose_fixed, const uint8_t* __restrict__ level,
                                                    const BaParamsDev& prm, double* __restrict__ chi2_last, int write_last, double (&A)[9],
                                                    double (&we)[3], double& w_out, int& rows_out) {
#pragma clang fp contract(off)
  double r0 = 0.0;
  const int kp = edge_pose[e];
  const double* T = B.poses[buf] + (size_t)kp * 7;
  const double* X = B.points[buf] + (size_t)pt * 3;
  const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
  const double X0 = X[0], X1 = X[1], X2 = X[2];
  double uvx = qy * X2 - qz * X1, uvy = qz * X0 - qx * X2, uvz = qx * X1 - qy * X0;
  uvx += uvx;
  uvy += uvy;
  uvz += uvz;
  const double x = X0 + qw * uvx + (qy * uvz - qz * uvy) + T[4];
  const double y = X1 + qw * uvy + (qz * uvx - qx * uvz) + T[5];
  const double z = X2 + qw * uvz + (qx * uvy - qy * uvx) + T[6];
  const bool stq = is_stereo[e] != 0;
  const double fx = prm.fx, fy = prm.fy, cx = prm.cx, cy = prm.cy, bf = prm.bf;
  const double* m = meas + (size_t)e * 3;
  const double u = x / z * fx + cx, v = y / z * fy + cy;
  const double e0 = m[0] - u, e1 = m[1] - v;
  const double e2 = stq ? (m[2] - (u - bf / z)) : 0.0;
  const double wi = info[e];
  const double c2 = stq ? (e0 * (wi * e0) + e1 * (wi * e1) + e2 * (wi * e2)) : (e0 * (wi * e0) + e1 * (wi * e1));
  // RobustKernelHuber::robustify (delta <= 0: no kernel)
  const double dl = delta[e];
  double rr0 = c2, r1 = 1.0;
  if (dl > 0.0) {
    const double dsqr = dl * dl;
    if (c2 > dsqr) {
      const double sq = sqrt(c2);
      rr0 = 2 * sq * dl - dsqr;
      r1 = dl / sq;
    }
  }
  if (level[e] == 0) {  // activeRobustChi2 + the per-edge _error bookkeeping of the ACTIVE edges (g2o evaluates only those)
    r0 = rr0;
    if (write_last) chi2_last[e] = c2;
  }
  const double w = r1 * wi;
  double* t = B.terms[buf] + (size_t)e * LM_TERM;
  const double z_2 = z * z;
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  double Bm[18];
  if (stq) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      A[k] = -fx * R[k] / z + fx * x * R[6 + k] / z_2;
      A[3 + k] = -fy * R[3 + k] / z + fy * y * R[6 + k] / z_2;
      A[6 + k] = A[k] - bf * R[6 + k] / z_2;
    }
  } else {
    const double t02 = -x / z * fx, t12 = -y / z * fy, s = -1. / z;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      A[k] = (s * fx) * R[k] + (s * t02) * R[6 + k];
      A[3 + k] = (s * fy) * R[3 + k] + (s * t12) * R[6 + k];
      A[6 + k] = 0.0;
    }
  }
  Bm[0] = x * y / z_2 * fx;
  Bm[1] = -(1 + (x * x / z_2)) * fx;
  Bm[2] = y / z * fx;
  Bm[3] = -1. / z * fx;
  Bm[4] = 0;
  Bm[5] = x / z_2 * fx;
  Bm[6] = (1 + y * y / z_2) * fy;
  Bm[7] = -x * y / z_2 * fy;
  Bm[8] = -x / z * fy;
  Bm[9] = 0;
  Bm[10] = -1. / z * fy;
  Bm[11] = y / z_2 * fy;
  if (stq) {
    Bm[12] = Bm[0] - bf * y / z_2;
    Bm[13] = Bm[1] + bf * x / z_2;
    Bm[14] = Bm[2];
    Bm[15] = Bm[3];
    Bm[16] = 0;
    Bm[17] = Bm[5] - bf / z_2;
  } else {
#pragma unroll
    for (int k = 12; k < 18; ++k) Bm[k] = 0.0;
  }
  const int rows = stq ? 3 : 2;
#pragma unroll
  for (int k = 0; k < 9; ++k) t[k] = A[k];
#pragma unroll
  for (int k = 0; k < 18; ++k) t[9 + k] = Bm[k];
  we[0] = w * e0, we[1] = w * e1, we[2] = w * e2;
  t[27] = we[0];
  t[28] = we[1];
  t[29] = we[2];
  t[30] = w;
  t[31] = (double)rows;
  w_out = w;
  rows_out = rows;
  // Hpl(e) = B^T W A (zero for the edges of fixed poses: they have no block)
  double* out = B.Hpl[buf] + (size_t)e * 18;
  if (pose_fixed[kp]) {
#pragma unroll
    for (int i = 0; i < 18; ++i) out[i] = 0.0;
  } else {
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double h = 0;
        _Pragma("unroll") for (int r = 0; r < 3; ++r) if (r < rows) h += Bm[6 * r + a] * w * A[3 * r + c];
        out[3 * a + c] = h;
      }
  }
  return r0;
}

// The system at one estimate, point side: 32 points per block, EIGHT lanes per point, one edge each (a point's edges in list order):
// the lane linearises its edge (terms, Hpl, chi2) and the eight lanes sum Hll / bl of their point in a fixed butterfly -- the edge terms
// never come back from memory for this.  (Until late r3 this was two launches, one lane per edge and then eight lanes per point reading
// the terms back: 10 + 17 us, and every launch of this stream costs 4.6 us before it does anything.)  The robust chi2 is summed per block
// in a fixed tree -> chi_part[buf][block]; the control step adds the blocks.
// (Measured and dropped in r3: the control step of the next trial in the tail of this kernel -- a fence, a ticket and a barrier in each
//  of its 94 blocks, then ~4 us of serial loads in the last one: 11.7 -> 23.2 us against 6.2 us for the launch it saves.)
__global__ __launch_bounds__(256) void k_lm_linpoints(int n_points, LmBuffers B, const LmState* __restrict__ st, int gate, int which,
                                                      const int32_t* __restrict__ pt_off, const int32_t* __restrict__ pt_edges,
                                                      const int32_t* __restrict__ edge_pose, const double* __restrict__ meas,
                                                      const uint8_t* __restrict__ is_stereo, const double* __restrict__ info,
                                                      const double* __restrict__ delta, const uint8_t* __restrict__ pose_fixed,
                                                      const uint8_t* __restrict__ level, BaParamsDev prm, double* __restrict__ chi2_last,
                                                      int write_last) {
#pragma clang fp contract(off)
  __shared__ double sh[256];
  if (!lm_gate(st, gate)) return;
  const int buf = st->cur ^ which;
  const int p = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.0;
  double r0 = 0.0;
  if (p < n_points) {
    for (int i = pt_off[p] + sub; i < pt_off[p + 1]; i += 8) {
      const int e = pt_edges[i];
      double A[9], we[3], w;
      int rows;
      r0 += lm_linearize_edge(e, buf, B, edge_pose, p, meas, is_stereo, info, delta, pose_fixed, level, prm, chi2_last, write_last, A, we, w, rows);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double s = 0;
        _Pragma("unroll") for (int r = 0; r < 3; ++r) if (r < rows) s += A[3 * r + a] * we[r];
        acc[9 + a] -= s;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double h = 0;
          _Pragma("unroll") for (int r = 0; r < 3; ++r) if (r < rows) h += A[3 * r + a] * w * A[3 * r + c];
          acc[3 * a + c] += h;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    double v = acc[k];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    acc[k] = v;
  }
  if (p < n_points && sub == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) B.Hll[buf][(size_t)p * 9 + k] = acc[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) B.bl[buf][(size_t)p * 3 + k] = acc[9 + k];
  }
  const double s = block_sum_256(r0, sh);
  if (threadIdx.x == 0) B.chi_part[buf][blockIdx.x] = s;
}

// Hpp / bp per pose (one 256-thread block per pose: one edge per thread) from the stored terms of buffer buf.  Partial sums are
// combined in a fixed butterfly + a fixed order over the four waves.
__device__ __forceinline__ void lm_pose_block(int k, int n_poses, int buf, const LmBuffers& B, const uint8_t* __restrict__ pose_fixed,
                                              const int32_t* __restrict__ ps_off, const int32_t* __restrict__ ps_edges, double (*part)[42]) {
#pragma clang fp contract(off)
  if (k >= n_poses) return;
  const double* terms = B.terms[buf];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc[42];
#pragma unroll
  for (int i = 0; i < 42; ++i) acc[i] = 0.0;
  if (!pose_fixed[k]) {
    for (int q = ps_off[k] + (int)threadIdx.x; q < ps_off[k + 1]; q += 256) {
      const double* t = terms + (size_t)ps_edges[q] * LM_TERM;
      const int rows = (int)t[31];
      const double w = t[30];
      double Bv[18], we[3];
#pragma unroll
      for (int i = 0; i < 18; ++i) Bv[i] = t[9 + i];
#pragma unroll
      for (int i = 0; i < 3; ++i) we[i] = t[27 + i];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double s = 0;
        _Pragma("unroll") for (int r = 0; r < 3; ++r) if (r < rows) s += Bv[6 * r + a] * we[r];
        acc[36 + a] -= s;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double h = 0;
          _Pragma("unroll") for (int r = 0; r < 3; ++r) if (r < rows) h += Bv[6 * r + a] * w * Bv[6 * r + c];
          acc[6 * a + c] += h;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 42; ++i) {
    const double v = wave_sum_fixed(acc[i]);
    if (lane == 0) part[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 42) {
    const double v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
    if (threadIdx.x < 36)
      B.Hpp[buf][(size_t)k * 36 + threadIdx.x] = v;
    else
      B.bp[buf][(size_t)k * 6 + threadIdx.x - 36] = v;
  }
}
__global__ __launch_bounds__(256) void k_lm_poseblocks(int n_poses, LmBuffers B, const LmState* __restrict__ st, int gate, int which,
                                                       const uint8_t* __restrict__ pose_fixed, const int32_t* __restrict__ ps_off,
                                                       const int32_t* __restrict__ ps_edges) {
  __shared__ double part[4][42];
  if (!lm_gate(st, gate)) return;
  lm_pose_block((int)blockIdx.x, n_poses, st->cur ^ which, B, pose_fixed, ps_off, ps_edges, part);
}

// computeLambdaInit: max |H_jj| over the active vertices of the CURRENT system -> state.maxdiag
__global__ __launch_bounds__(1024) void k_lm_maxdiag(int n_poses, int n_points, LmBuffers B, LmState* __restrict__ st, int gate,
                                                     const uint8_t* __restrict__ fixed) {
  __shared__ double sh[1024];
  if (!lm_gate(st, gate)) return;
  const int buf = st->cur;
  const double* Hpp = B.Hpp[buf];
  const double* Hll = B.Hll[buf];
  double m = 0;
  for (int i = threadIdx.x; i < n_poses * 6; i += 1024) {
    const int k = i / 6, a = i - 6 * k;
    if (!fixed[k]) m = fmax(m, fabs(Hpp[(size_t)k * 36 + 7 * a]));
  }
  for (int i = threadIdx.x; i < n_points * 3; i += 1024) {
    const int p = i / 3, a = i - 3 * p;
    m = fmax(m, fabs(Hll[(size_t)p * 9 + 4 * a]));
  }
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) st->maxdiag = sh[0];
}

// per point: Dinv = (Hll + lambda I)^-1 (Eigen's 3x3 inverse: cofactors / determinant), then W(e) = Hpl(e) Dinv for the point's edges.
// EIGHT lanes per point (each inverts the same block and takes one edge): one dependent chain per lane instead of one per edge.
// Blocks [pt_blocks, pt_blocks + n_poses): Hpp / bp of the CURRENT system (lm_pose_block).  The pose side of a system is only ever used
// once its estimate is the current one (k_lm_schur, k_lm_update's scale), so it is not built with the trial's system but here, beside the
// point inverses of the trial that follows -- for an accepted trial that is its first computation, after a rejected one a recomputation
// of the same numbers.  One launch less per trial, and this work runs beside the point blocks instead of after them.
__global__ __launch_bounds__(256) void k_lm_prep(int n_points, int n_poses, int pt_blocks, LmBuffers B, LmState* __restrict__ st,
                                                 const int32_t* __restrict__ pt_off, const int32_t* __restrict__ pt_edges,
                                                 double* __restrict__ Dinv, double* __restrict__ W, const uint8_t* __restrict__ pose_fixed,
                                                 const int32_t* __restrict__ ps_off, const int32_t* __restrict__ ps_edges) {
#pragma clang fp contract(off)
  __shared__ double part[4][42];
  if (!lm_gate(st, 1)) return;
  if ((int)blockIdx.x >= pt_blocks) {
    lm_pose_block((int)blockIdx.x - pt_blocks, n_poses, st->cur, B, pose_fixed, ps_off, ps_edges, part);
    return;
  }
  const int p = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
  if (p >= n_points) return;
  const int buf = st->cur;
  const double lambda = st->lambda;
  double M[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) M[i] = B.Hll[buf][(size_t)p * 9 + i];
  M[0] += lambda, M[4] += lambda, M[8] += lambda;
  const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  if (det == 0 || !isfinite(det)) {
    if (sub == 0) st->ok = 0;
    return;
  }
  const double id = 1.0 / det;
  double D[9];
  D[0] = c00 * id;
  D[1] = (M[2] * M[7] - M[1] * M[8]) * id;
  D[2] = (M[1] * M[5] - M[2] * M[4]) * id;
  D[3] = c01 * id;
  D[4] = (M[0] * M[8] - M[2] * M[6]) * id;
  D[5] = (M[2] * M[3] - M[0] * M[5]) * id;
  D[6] = c02 * id;
  D[7] = (M[1] * M[6] - M[0] * M[7]) * id;
  D[8] = (M[0] * M[4] - M[1] * M[3]) * id;
  if (sub == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) Dinv[(size_t)p * 9 + i] = D[i];
  }
  for (int q = pt_off[p] + sub; q < pt_off[p + 1]; q += 8) {
    const int e = pt_edges[q];
    const double* H = B.Hpl[buf] + (size_t)e * 18;
    double hv[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) hv[k] = H[k];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double h0 = hv[3 * a], h1 = hv[3 * a + 1], h2 = hv[3 * a + 2];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double s = 0;
        s += h0 * D[c];
        s += h1 * D[3 + c];
        s += h2 * D[6 + c];
        W[(size_t)e * 18 + 3 * a + c] = s;
      }
    }
  }
}

// The pair lists of the reduced system, built on the device once per call: block (i >= j) of free poses gets the pairs (e1, e2) of edges
// with pose(e1) = i, pose(e2) = j that observe the same point, in the order of pose i's edge list, at pairs[b * cap ...], pair_cnt[b] of
// them.  k_lm_pair_table: table[slot j][point] = the edge of pose j that observes the point (or -1; a pose observes a point at most once --
// the host checks that and takes the host-driven path otherwise).  k_lm_pairs: one wave per block walks pose i's list, looks each point up
// in row j and compacts the hits in list order (ballot + prefix count: deterministic).  The host built these lists until late r3:
// two passes over sum(observations^2) ~ 100 k combinations, 0.37 ms of a 3.1 ms call before the first kernel could start.
__global__ __launch_bounds__(256) void k_lm_pair_table(int n_edges, int n_points, const int32_t* __restrict__ edge_pose,
                                                       const int32_t* __restrict__ edge_point, const int32_t* __restrict__ pose_slot,
                                                       int32_t* __restrict__ table) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  const int s = pose_slot[edge_pose[e]];
  if (s >= 0) table[(size_t)s * n_points + edge_point[e]] = e;
}
__global__ __launch_bounds__(64) void k_lm_pairs(int nf, int n_points, int cap, const int32_t* __restrict__ free_pose,
                                                 const int32_t* __restrict__ ps_off, const int32_t* __restrict__ ps_edges,
                                                 const int32_t* __restrict__ edge_point, const int32_t* __restrict__ table,
                                                 int2* __restrict__ pairs, int32_t* __restrict__ pair_cnt) {
  const int b = blockIdx.x;
  int i = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= b) ++i;
  while (i * (i + 1) / 2 > b) --i;
  const int j = b - i * (i + 1) / 2;
  const int lane = threadIdx.x;
  const int ki = free_pose[i];
  const int32_t* row = table + (size_t)j * n_points;
  int2* out = pairs + (size_t)b * cap;
  int n = 0;
  const int q_end = ps_off[ki + 1];
  for (int q0 = ps_off[ki]; q0 < q_end; q0 += 64) {
    const int q = q0 + lane;
    int e1 = -1, e2 = -1;
    if (q < q_end) {
      e1 = ps_edges[q];
      e2 = row[edge_point[e1]];
    }
    const bool hit = e2 >= 0;
    const unsigned long long m = __ballot(hit);
    if (hit) out[n + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(e1, e2);
    n += __popcll(m);
  }
  if (lane == 0) pair_cnt[b] = n;
}

// Reduced system, one wave per block (i >= j) of free poses, the lanes over the block's pairs (e1, e2): pose(e1) = i, pose(e2) = j, same
// point.  Sblk: lower-triangular blocks, block (i, j) at (i (i + 1) / 2 + j) * 36, row-major inside.  nf more waves form the right-hand
// side rhs_i = bp_i - sum_{e of pose i} W(e) bl(point(e)).
__global__ __launch_bounds__(64) void k_lm_schur(int nf, LmBuffers B, const LmState* __restrict__ st, const int32_t* __restrict__ free_pose,
                                                 const int32_t* __restrict__ pair_cnt, int pair_cap, const int2* __restrict__ pairs,
                                                 const int32_t* __restrict__ ps_off, const int32_t* __restrict__ ps_edges,
                                                 const int32_t* __restrict__ edge_point, const double* __restrict__ W,
                                                 double* __restrict__ Sblk, double* __restrict__ rhs, double* __restrict__ M, int ld) {
  // M != nullptr: the dense layout of the blocked solver (k_lmbig.hip) -- block (i, j) at rows 6 i, columns 6 j of a row-major matrix of
  // pitch ld, the right-hand side in row ld
#pragma clang fp contract(off)
  if (!lm_gate(st, 1)) return;
  const int lane = threadIdx.x;
  const int buf = st->cur;
  const int n_blk = nf * (nf + 1) / 2;
  if ((int)blockIdx.x >= n_blk) {
    // the right-hand side of block row i in a wave of its own: behind the diagonal block's sum in the same wave it doubled the longest
    // chain of dependent round trips of this launch (26 us; the other 780 waves were done after 12)
    const int i = (int)blockIdx.x - n_blk;
    const int ki = free_pose[i];
    double r[6] = {0, 0, 0, 0, 0, 0};
    for (int q = ps_off[ki] + lane; q < ps_off[ki + 1]; q += 64) {
      const int e = ps_edges[q];
      const double* w = W + (size_t)e * 18;
      const double* bb = B.bl[buf] + (size_t)edge_point[e] * 3;
      const double b0 = bb[0], b1 = bb[1], b2 = bb[2];
#pragma unroll
      for (int a = 0; a < 6; ++a) r[a] += w[3 * a] * b0 + w[3 * a + 1] * b1 + w[3 * a + 2] * b2;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double s = wave_sum_fixed(r[a]);
      if (lane == a) (M ? M[(size_t)ld * ld + 6 * i + a] : rhs[6 * i + a]) = B.bp[buf][(size_t)ki * 6 + a] - s;
    }
    return;
  }
  // blockIdx.x -> (i, j), i >= j
  const int b = blockIdx.x;
  int i = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= b) ++i;
  while (i * (i + 1) / 2 > b) --i;
  const int j = b - i * (i + 1) / 2;
  const double* Hpl = B.Hpl[buf];
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  const int q0 = b * pair_cap, q1 = q0 + pair_cnt[b];
  // (measured and dropped: four trips of the pair list in flight at once for the ~260-pair diagonal blocks -- the clamped loads of the
  //  one-trip blocks, 780 of 820, cost more than the diagonal ones gain: 17.5 -> 21 us)
  for (int q = q0 + lane; q < q1; q += 64) {
    const int2 pr = pairs[q];
    const double* w = W + (size_t)pr.x * 18;
    const double* h = Hpl + (size_t)pr.y * 18;
    double wv[18], hv[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) wv[k] = w[k], hv[k] = h[k];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[6 * a + c] += wv[3 * a] * hv[3 * c] + wv[3 * a + 1] * hv[3 * c + 1] + wv[3 * a + 2] * hv[3 * c + 2];
  }
  const int ki = free_pose[i];
  double* out = Sblk + (size_t)b * 36;
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    const double s = wave_sum_fixed(acc[k]);
    if (lane == k) {
      double v = -s;
      if (i == j) {
        double d = B.Hpp[buf][(size_t)ki * 36 + k];
        if (k / 6 == k % 6) d += st->lambda;
        v = d - s;
      }
      if (M)
        M[(size_t)(6 * i + k / 6) * ld + 6 * j + k % 6] = v;
      else
        out[k] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Cholesky of the reduced system with the matrix in registers: ONE workgroup of 512 threads (two waves per SIMD: 256 registers per
// lane, nothing spills), nb <= LM_CHOL_MAX_NB block rows.  Thread t < 448 owns up to TWO 6x6 blocks of the strictly lower triangle
// (linear indices t and t + 448: 896 >= 42 * 41 / 2); the threads of the LAST wave own one diagonal block each (thread 448 + I) and the
// matching block of the right-hand side, which is carried as an extra block row so that the forward substitution is part of the
// factorisation.  The last wave has nothing else to do: a diagonal block is factorised as soon as its last update has landed, while the
// other waves are still updating the rest of the trailing matrix (look-ahead).
// Per block column kb: (1) L_kk (lower Cholesky of the diagonal block, by its owner) and the reciprocals of its diagonal -> LDS;
// barrier; (2) the owners of the blocks (I, kb), I > kb, and of y_kb solve X L_kk^T = A in registers and publish X (the panel) in LDS;
// barrier; (3) the owners of (I, J), J > kb: A -= P_I P_J^T from LDS.  Two barriers per block column, no global traffic.
// Backward substitution: x_kb from the diagonal owner, then the owners of (kb, J < kb) subtract L_{kb,J}^T x_kb from y_J in LDS (one
// writer per J).  ok = 0 if a pivot is not positive (g2o: the linear solver fails, the trial is rejected), x = 0 then.
// ---------------------------------------------------------------------------------------------------------------------------------
#define LM_PSTRIDE 38  // doubles per panel block in LDS: 304 bytes = 76 dwords, 76 mod 64 = 12 -> sixteen lanes' 16-byte reads hit distinct banks
#define LM_CHOL_THREADS 512
#define LM_CHOL_OFF 448  // threads that own off-diagonal blocks
#ifndef LM_DIAG_WAVE
#define LM_DIAG_WAVE 3
#endif

// t -> (I, J), I > J: the strictly-lower blocks in COLUMN-major order (column 0 first, rows ascending inside a column).  At block
// column kb the blocks still being updated (J > kb) are then a contiguous TAIL of the order and the panel (J == kb) a contiguous run
// just before it: whole waves drop out of the update as the factorisation proceeds (a wave with one active lane issues as many
// instructions as a full one -- with the row-major order every wave kept a few active lanes to the end and a block column cost the same
// 7.6 us at kb = 39 as at kb = 0).
__device__ __forceinline__ void lm_tri_index(int t, int nb, int& I, int& J) {
  int j = 0, base = 0;
  while (j < nb - 1 && base + (nb - 1 - j) <= t) {
    base += nb - 1 - j;
    ++j;
  }
  J = j;
  I = j + 1 + (t - base);
}

// X L^T = A for the rows of one block, in place (L, inv: LDS)
__device__ __forceinline__ void lm_panel_solve(double (&A)[36], const double* L, const double* inv) {
#pragma clang fp contract(fast)
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double sv = A[6 * a + c];
#pragma unroll
      for (int m = 0; m < c; ++m) sv = fma(-v[m], L[6 * c + m], sv);  // (LDS broadcast reads; copies in registers would spill)
      v[c] = sv * inv[c];
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) A[6 * a + c] = v[c];
  }
}
// A -= P_I P_J^T (panel blocks in LDS)
__device__ __forceinline__ void lm_block_update(double (&A)[36], const double* PI, const double* PJ) {
#pragma clang fp contract(fast)  // fused multiply-adds here: the trailing update is two thirds of the kernel's instructions, and the
                                 // factorisation's rounding is not part of any bit-exact contract (its order already differs from g2o's LLT)
  // The block is updated in three parts of two COLUMNS: a third of P_J (12 doubles) is held in registers and every row of P_I is read once
  // per part -- 6 + 18 reads of 16 bytes per part, 72 per block, where re-reading P_J for every row took 126.  (All of P_J, or half of it,
  // in registers beside two blocks per thread spills.)  Measured: the same 118 us either way -- the phase is bound by fp64 issue on the
  // SIMD that hosts the diagonal wave, not by the LDS port (see k_lm_chol).
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    double pj[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) pj[k] = PJ[12 * h + k];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double pi[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) pi[k] = PI[6 * a + k];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const double* q = pj + 6 * cc;
        double acc = A[6 * a + 2 * h + cc];
        acc = fma(-pi[0], q[0], acc);
        acc = fma(-pi[1], q[1], acc);
        acc = fma(-pi[2], q[2], acc);
        acc = fma(-pi[3], q[3], acc);
        acc = fma(-pi[4], q[4], acc);
        acc = fma(-pi[5], q[5], acc);
        A[6 * a + 2 * h + cc] = acc;
      }
    }
  }
}

// the same for a DIAGONAL block: only the lower triangle is ever read (21 of the 36 entries)
__device__ __forceinline__ void lm_diag_update(double (&A)[36], const double* PJ) {
#pragma clang fp contract(fast)
  double pj[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) pj[k] = PJ[k];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c <= a; ++c) {
      const double* pa = pj + 6 * a;
      const double* q = pj + 6 * c;
      double acc = A[6 * a + c];
      acc = fma(-pa[0], q[0], acc);
      acc = fma(-pa[1], q[1], acc);
      acc = fma(-pa[2], q[2], acc);
      acc = fma(-pa[3], q[3], acc);
      acc = fma(-pa[4], q[4], acc);
      acc = fma(-pa[5], q[5], acc);
      A[6 * a + c] = acc;
    }
}

#ifdef LM_CHOL_STAMPS  // diagnostic build only (tools/exp/chol_bench.hip): where a block column's time goes, per wave
__device__ long long g_lm_stamps[8][8];
__device__ unsigned int g_lm_hwid[8];
#define LM_ST(k) \
  if ((t & 63) == 0) { const long long now_ = __builtin_amdgcn_s_memtime(); g_lm_stamps[t >> 6][k] += now_ - last_; last_ = now_; }
#define LM_ST_INIT long long last_ = __builtin_amdgcn_s_memtime();
#else
#define LM_ST(k)
#define LM_ST_INIT
#endif

struct LmCholShared {
  double P[(LM_CHOL_MAX_NB + 1) * LM_PSTRIDE];  // panel blocks of the current column; slot nb: the rhs row (6)
  double Lk[36 + 6];                            // L_kk (row-major, lower) + 1 / diagonal
  double yv[6 * LM_CHOL_MAX_NB];                // y, then x
  int ok;
};

// The two roles run SEPARATE copies of the block-column loop (same barriers at the same points): the waves of off-diagonal owners keep
// two blocks (144 registers) live, the diagonal wave one block + the right-hand side + the reciprocals -- written as one loop with
// role branches the register allocator spilled ~100 bytes per lane into scratch in every phase (a block column then cost 5 us).
__device__ __forceinline__ void lm_chol_offdiag(int nb, int t, LmCholShared& sh, const double* __restrict__ Sblk) {
#pragma clang fp contract(fast)
  const int n_off = nb * (nb - 1) / 2;
  int I0 = -1, J0 = -1, I1 = -1, J1 = -1;
  double A0[36], A1[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) A0[k] = 0.0, A1[k] = 0.0;
  if (t < n_off) {
    lm_tri_index(t, nb, I0, J0);
    const double* src = Sblk + ((size_t)I0 * (I0 + 1) / 2 + J0) * 36;
#pragma unroll
    for (int k = 0; k < 36; ++k) A0[k] = src[k];
  }
  if (t + LM_CHOL_OFF < n_off) {
    lm_tri_index(t + LM_CHOL_OFF, nb, I1, J1);
    const double* src = Sblk + ((size_t)I1 * (I1 + 1) / 2 + J1) * 36;
#pragma unroll
    for (int k = 0; k < 36; ++k) A1[k] = src[k];
  }
  __syncthreads();  // (A) s_ok initialised
  LM_ST_INIT
  __syncthreads();  // (B) first diagonal block factorised
  LM_ST(0)
  for (int kb = 0; kb < nb; ++kb) {
    if (!sh.ok) break;  // uniform (read after a barrier)
    if (J0 == kb) {
      lm_panel_solve(A0, sh.Lk, sh.Lk + 36);
      double* dst = sh.P + (size_t)I0 * LM_PSTRIDE;
#pragma unroll
      for (int k = 0; k < 36; ++k) dst[k] = A0[k];
    }
    if (J1 == kb) {
      lm_panel_solve(A1, sh.Lk, sh.Lk + 36);
      double* dst = sh.P + (size_t)I1 * LM_PSTRIDE;
#pragma unroll
      for (int k = 0; k < 36; ++k) dst[k] = A1[k];
    }
    LM_ST(1)
    __syncthreads();
    LM_ST(2)
    if (J0 > kb) lm_block_update(A0, sh.P + (size_t)I0 * LM_PSTRIDE, sh.P + (size_t)J0 * LM_PSTRIDE);
    if (J1 > kb) lm_block_update(A1, sh.P + (size_t)I1 * LM_PSTRIDE, sh.P + (size_t)J1 * LM_PSTRIDE);
    LM_ST(3)
    __syncthreads();
    LM_ST(4)
  }
  if (!sh.ok) return;
  // Backward substitution (the chain lives in the diagonal wave, see lm_chol_diag): the SUB-diagonal blocks L_{I,I-1} go to it through
  // the panel area, which nobody reads any more; the other blocks of row kb are applied here, one step behind the chain.
  if (I0 == J0 + 1) {
    double* dst = sh.P + (size_t)I0 * LM_PSTRIDE;
#pragma unroll
    for (int k = 0; k < 36; ++k) dst[k] = A0[k];
  }
  if (I1 == J1 + 1 && I1 >= 0) {
    double* dst = sh.P + (size_t)I1 * LM_PSTRIDE;
#pragma unroll
    for (int k = 0; k < 36; ++k) dst[k] = A1[k];
  }
  __syncthreads();  // (C)
  for (int kb = nb - 1; kb >= 0; --kb) {
    __syncthreads();  // x_kb published by the diagonal owner
    // y_J -= L_{kb,J}^T x_kb for J <= kb - 2: the only writer of y_J in this step; y_J is read two or more steps later
    // both blocks of this thread at once (their LDS round trips and chains overlap), two partial sums per entry: this update sits
    // between two barriers of every step -- one block after the other with six-deep chains it was 0.6 us per step, longer than the chain
    // of the diagonal wave it runs beside
    const bool u0 = I0 == kb && J0 < kb - 1, u1 = I1 == kb && J1 < kb - 1;
    if (u0 || u1) {
      double xk[6], v0[6], v1[6];
      const int j0 = u0 ? J0 : 0, j1 = u1 ? J1 : 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) xk[a] = sh.yv[6 * kb + a], v0[a] = sh.yv[6 * j0 + a], v1[a] = sh.yv[6 * j1 + a];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double p0 = fma(-A0[c], xk[0], v0[c]), q0 = -A0[18 + c] * xk[3];
        double p1 = fma(-A1[c], xk[0], v1[c]), q1 = -A1[18 + c] * xk[3];
        p0 = fma(-A0[6 + c], xk[1], p0), q0 = fma(-A0[24 + c], xk[4], q0);
        p1 = fma(-A1[6 + c], xk[1], p1), q1 = fma(-A1[24 + c], xk[4], q1);
        p0 = fma(-A0[12 + c], xk[2], p0), q0 = fma(-A0[30 + c], xk[5], q0);
        p1 = fma(-A1[12 + c], xk[2], p1), q1 = fma(-A1[30 + c], xk[5], q1);
        v0[c] = p0 + q0, v1[c] = p1 + q1;
      }
      if (u0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) sh.yv[6 * J0 + c] = v0[c];
      }
      if (u1) {
#pragma unroll
        for (int c = 0; c < 6; ++c) sh.yv[6 * J1 + c] = v1[c];
      }
    }
  }
  LM_ST(6)
}

__device__ __forceinline__ void lm_chol_diag(int nb, int t, LmCholShared& sh, const double* __restrict__ Sblk, const double* __restrict__ rhs) {
#pragma clang fp contract(fast)
  const int I0 = t - LM_CHOL_OFF;
  const bool diag = I0 < nb;
  double A0[36], Y[6], invd[6];  // the diagonal block, its right-hand side, 1 / L_jj (substitutions multiply, they do not divide)
#pragma unroll
  for (int k = 0; k < 36; ++k) A0[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) Y[k] = 0.0, invd[k] = 0.0;
  if (diag) {
    const double* src = Sblk + ((size_t)I0 * (I0 + 1) / 2 + I0) * 36;
#pragma unroll
    for (int k = 0; k < 36; ++k) A0[k] = src[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) Y[k] = rhs[6 * I0 + k];
  }
  auto factor_diag = [&]() {  // lower Cholesky of A0 in place, right-looking; publish L and 1 / diag
    bool good = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double d = A0[7 * j];
      if (!(d > 0) || !isfinite(d)) good = false;
      // y ~ 1 / sqrt(d): the hardware estimate y0 (~23 bits) and ONE third-order step y0 (1 + e / 2 + 3 e^2 / 8), e = 1 - d y0^2 (error
      // ~ e^3: below double rounding), four dependent operations deep -- two Newton steps were six; L_jj = d y.  The six pivots of a
      // diagonal block are the serial spine of the factorisation (a dependent fp64 operation of a lone wave takes ~30 cycles): sqrt()
      // followed by a division was ~60 dependent operations per pivot
      const double y0 = __builtin_amdgcn_rsq(d);
      const double e = fma(-(d * y0), y0, 1.0);
      const double y = fma(y0 * e, fma(0.375, e, 0.5), y0);
      A0[7 * j] = d * y;
      invd[j] = y;
#pragma unroll
      for (int i2 = j + 1; i2 < 6; ++i2) A0[6 * i2 + j] *= y;
#pragma unroll
      for (int i2 = j + 1; i2 < 6; ++i2)
#pragma unroll
        for (int k = j + 1; k <= i2; ++k) A0[6 * i2 + k] = fma(-A0[6 * i2 + j], A0[6 * k + j], A0[6 * i2 + k]);
    }
    if (!good) sh.ok = 0;
#pragma unroll
    for (int k = 0; k < 36; ++k) sh.Lk[k] = A0[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) sh.Lk[36 + k] = invd[k];
  };
  __syncthreads();  // (A)
  LM_ST_INIT
  if (diag && I0 == 0) factor_diag();
  __syncthreads();  // (B)
  LM_ST(0)
  for (int kb = 0; kb < nb; ++kb) {
    if (!sh.ok) break;
    if (diag && I0 == kb) {  // y_kb L_kk^T = Y (the factor is still in this thread's registers)
      double v[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double sv = Y[c];
#pragma unroll
        for (int m = 0; m < c; ++m) sv = fma(-v[m], A0[6 * c + m], sv);
        v[c] = sv * invd[c];
      }
      double* dst = sh.P + (size_t)nb * LM_PSTRIDE;
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[k] = v[k], sh.yv[6 * kb + k] = v[k];
    }
    LM_ST(1)
    __syncthreads();
    LM_ST(2)
    if (diag && I0 > kb) {
      const double* PJ = sh.P + (size_t)I0 * LM_PSTRIDE;
      lm_diag_update(A0, PJ);
      const double* PY = sh.P + (size_t)nb * LM_PSTRIDE;
      double py[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) py[k] = PY[k];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double* q = PJ + 6 * c;
        double acc = Y[c];
#pragma unroll
        for (int m = 0; m < 6; ++m) acc = fma(-py[m], q[m], acc);
        Y[c] = acc;
      }
      LM_ST(5)
      if (I0 == kb + 1) factor_diag();  // look-ahead: the next diagonal block is complete now
    }
    LM_ST(3)
    __syncthreads();
    LM_ST(4)
  }
  if (!sh.ok) return;
  // Backward substitution L^T x = y.  Step kb needs y_kb complete, and the last contribution to it comes from the step before
  // (L_{kb+1,kb}^T x_{kb+1}): with every row's blocks applied by their owners that is two workgroup barriers per block row on the
  // critical path (0.9 us each: a quarter of the kernel).  The diagonal lane kb therefore keeps the sub-diagonal block L_{kb+1,kb} itself and
  // applies it in its own step; the owners of the other blocks of row kb + 1 work one step behind (their targets y_J, J <= kb - 1, are
  // read at step J at the earliest, a barrier later): ONE barrier per block row, and the chain stays in this wave.
  __syncthreads();  // (C) sub-diagonal blocks published
  double Sub[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) Sub[k] = 0.0;
  if (diag && I0 + 1 < nb) {
    const double* src = sh.P + (size_t)(I0 + 1) * LM_PSTRIDE;
#pragma unroll
    for (int k = 0; k < 36; ++k) Sub[k] = src[k];
  }
  for (int kb = nb - 1; kb >= 0; --kb) {
    if (diag && I0 == kb) {
      double xn[6], yk[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        xn[a] = (kb + 1 < nb) ? sh.yv[6 * (kb + 1) + a] : 0.0;
        yk[a] = sh.yv[6 * kb + a];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) {  // two partial sums: the chain is three fused multiply-adds and an add deep, not six
        double s0 = fma(-Sub[c], xn[0], yk[c]), s1 = -Sub[18 + c] * xn[3];
        s0 = fma(-Sub[6 + c], xn[1], s0), s1 = fma(-Sub[24 + c], xn[4], s1);
        s0 = fma(-Sub[12 + c], xn[2], s0), s1 = fma(-Sub[30 + c], xn[5], s1);
        yk[c] = s0 + s1;
      }
      double xv[6];
#pragma unroll
      for (int a = 5; a >= 0; --a) {
        double v = yk[a];
#pragma unroll
        for (int m = a + 1; m < 6; ++m) v = fma(-A0[6 * m + a], xv[m], v);
        xv[a] = v * invd[a];
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) sh.yv[6 * kb + a] = xv[a];
    }
    __syncthreads();
  }
  LM_ST(6)
}

__global__ __launch_bounds__(LM_CHOL_THREADS) void k_lm_chol(int nb, LmState* __restrict__ st, const double* __restrict__ Sblk,
                                                             const double* __restrict__ rhs, double* __restrict__ x) {
  __shared__ __attribute__((aligned(16))) LmCholShared sh;
  if (!lm_gate(st, 1)) return;
  const int t = threadIdx.x;
  if (t == 0) sh.ok = 1;
  // Role by wave: the diagonal wave is wave LM_DIAG_WAVE of the workgroup, the owners of off-diagonal blocks are the other seven in
  // order.  Waves w and w + 4 share a SIMD, and the diagonal wave (250 fp64 instructions per block column, the serial spine) should share
  // its SIMD with the LIGHTEST owner wave: in column-major order that is the last one (threads 384..447 of the order: blocks of columns
  // 11-13 only, no second block), which is wave 7 when the diagonal wave is wave 3: 118 -> 112 us at 40 block rows (tools/exp/chol_bench.hip;
  // s_setprio(3) in the diagonal wave on top of that: no gain).
  const int wv = t >> 6;
#ifdef LM_CHOL_STAMPS
  if ((t & 63) == 0) g_lm_hwid[wv] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID, all 32 bits
#endif
  if (wv != LM_DIAG_WAVE)
    lm_chol_offdiag(nb, wv < LM_DIAG_WAVE ? t : t - 64, sh, Sblk);
  else
    lm_chol_diag(nb, LM_CHOL_OFF + (t & 63), sh, Sblk, rhs);
  __syncthreads();
  if (!sh.ok) {
    if (t == 0) st->ok = 0;
    for (int k = t; k < 6 * nb; k += LM_CHOL_THREADS) x[k] = 0.0;
    return;
  }
  for (int k = t; k < 6 * nb; k += LM_CHOL_THREADS) x[k] = sh.yv[k];
}

// SparseOptimizer::update into the OTHER buffer: poses <- exp(dx) * pose (free poses; fixed ones are copied), points <- point + Dinv (bl
// - sum Hpl^T dxp); computeScale partial sums sum_i dx_i (lambda dx_i + b_i) per block -> scale_part[blockIdx.x].
// Blocks [0, pt_blocks): 32 points each, eight lanes per point (one edge each, fixed butterfly); the blocks after them: 256 poses each.
__global__ __launch_bounds__(256) void k_lm_update(int n_poses, int n_points, int pt_blocks, LmBuffers B, const LmState* __restrict__ st,
                                                   const int32_t* __restrict__ pose_slot, const double* __restrict__ x,
                                                   const int32_t* __restrict__ pt_off, const int32_t* __restrict__ pt_edges,
                                                   const int32_t* __restrict__ edge_pose, const double* __restrict__ Dinv,
                                                   double* __restrict__ scale_part) {
#pragma clang fp contract(off)
  __shared__ double sh[256];
  if (!lm_gate(st, 1)) return;
  const int cur = st->cur, nxt = cur ^ 1;
  const double lambda = st->lambda;
  double acc = 0.0;
  if ((int)blockIdx.x < pt_blocks) {
    const int p = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
    double s0 = 0, s1 = 0, s2 = 0;
    if (p < n_points) {
      for (int q = pt_off[p] + sub; q < pt_off[p + 1]; q += 8) {
        const int e = pt_edges[q];
        const int s = pose_slot[edge_pose[e]];
        if (s < 0) continue;
        const double* h = B.Hpl[cur] + (size_t)e * 18;
        const double* d = x + 6 * s;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          s0 += h[3 * a] * d[a];
          s1 += h[3 * a + 1] * d[a];
          s2 += h[3 * a + 2] * d[a];
        }
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      s0 += __shfl_xor(s0, o);
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    if (p < n_points && sub == 0) {
      const double* bl = B.bl[cur] + (size_t)p * 3;
      const double r0 = bl[0] - s0, r1 = bl[1] - s1, r2 = bl[2] - s2;
      const double* D = Dinv + (size_t)p * 9;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double dv = D[3 * a] * r0 + D[3 * a + 1] * r1 + D[3 * a + 2] * r2;
        B.points[nxt][(size_t)p * 3 + a] = B.points[cur][(size_t)p * 3 + a] + dv;
        acc += dv * (lambda * dv + bl[a]);
      }
    }
  } else {
    const int k = ((int)blockIdx.x - pt_blocks) * 256 + threadIdx.x;
    if (k < n_poses) {
      const int s = pose_slot[k];
      const double* src = B.poses[cur] + (size_t)k * 7;
      double* dst = B.poses[nxt] + (size_t)k * 7;
      if (s >= 0) {
        double upd[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) upd[a] = x[6 * s + a];
        // (every loop over these small arrays unrolled: left as loops the arrays stay private memory, the compiler moves them to LDS
        //  addressed by the flat thread number and reads the workgroup sizes for that from the dispatch packet -- in HOST memory: a scalar
        //  load across PCIe at the start of every wave; see k_brief.hip, r6)
        PoseDev T, R;
#pragma unroll
        for (int a = 0; a < 4; ++a) T.q[a] = src[a];
#pragma unroll
        for (int a = 0; a < 3; ++a) T.t[a] = src[4 + a];
        pose_oplus(T, upd, R);
#pragma unroll
        for (int a = 0; a < 4; ++a) dst[a] = R.q[a];
#pragma unroll
        for (int a = 0; a < 3; ++a) dst[4 + a] = R.t[a];
        const double* bp = B.bp[cur] + (size_t)k * 6;
#pragma unroll
        for (int a = 0; a < 6; ++a) acc += upd[a] * (lambda * upd[a] + bp[a]);
      } else {
        for (int a = 0; a < 7; ++a) dst[a] = src[a];
      }
    }
  }
  const double s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) scale_part[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The control kernel: one lane.  mode 0: between two trials | 1: the round-switch point (after the trials provisioned for round 0)
// | 2: after the switch group | 3: the final point | 4: after the final group.
// g2o: SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve (tau = 1e-5, <= 10 trials per iteration).
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lm_sum(const double* p, int n) {  // all 64 lanes: lane-strided partial sums, then a fixed butterfly
  const volatile double* vp = p;  // (in the tail of k_lm_linpoints these were written by other blocks of the SAME launch: no cached copies)
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 64) s += vp[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  return s;
}

__device__ void lm_ctrl_body(LmState* __restrict__ st, const LmBuffers& B, int mode, int chi_blocks, int scale_blocks,
                             const double* __restrict__ scale_part, const volatile uint8_t* __restrict__ abort_flag) {
#pragma clang fp contract(off)
  LmState s = *st;  // (every lane runs the same scalar program; lane 0 stores the result)
  if (abort_flag && *abort_flag) s.stopped = 1;
  if (mode == 2 && s.run_switch) {  // the switch group has run: round 1 starts on the system it rebuilt (new information / no kernels)
    s.run_switch = 0;
    s.switched = 1;
    s.round = 1;
    s.it = 0;
    s.phase = 0;
    s.need_chi = 1;
  }
  const int cur0 = s.cur;
  const double sum_cur = lm_sum(B.chi_part[cur0], chi_blocks), sum_trial = lm_sum(B.chi_part[cur0 ^ 1], chi_blocks);
  const double sum_scale = lm_sum(scale_part, scale_blocks);
  if (mode == 4 && s.run_final) {
    s.run_final = 0;
    s.finalized = 1;
  }
  if (s.need_chi) {
    if (s.cur) s.chi_of[1] = sum_cur; else s.chi_of[0] = sum_cur;  // (selects, not a variable index: the state record stays in registers -- as an indexed array it lived in scratch memory)
    s.need_chi = 0;
  }
  // (a) a trial ran since the last control point: decide it
  if (s.phase == 1) {
    const double chi_trial = sum_trial;
    const double scale = sum_scale;
    const bool ok2 = s.ok != 0;
    const double temp_chi = ok2 ? chi_trial : 1.7976931348623157e308;
    double rho = (s.current_chi - temp_chi) / (scale + 1e-3);
    if (!ok2) rho = -1.0;  // the linear solver failed: the trial is rejected whatever its step looked like
    bool lambda_dead = false;
    if (rho > 0 && isfinite(temp_chi)) {
      double alpha = 1. - pow((2 * rho - 1), 3);
      alpha = fmin(alpha, 2. / 3.);
      s.lambda *= fmax(1. / 3., alpha);
      s.ni = 2;
      s.current_chi = temp_chi;
      if (s.cur) s.chi_of[0] = chi_trial; else s.chi_of[1] = chi_trial;
      s.cur ^= 1;  // accept: the trial's estimate AND its system become current (no pop)
    } else {
      s.lambda *= s.ni;
      s.ni *= 2;
      if (!isfinite(s.lambda)) lambda_dead = true;  // (g2o breaks out of the trial loop before ++qmax)
    }
    if (!lambda_dead) ++s.qmax;
    s.rho = rho;
    if (!lambda_dead && rho < 0 && s.qmax < 10 && !s.stopped) {
      s.phase = 3;  // another trial of the same iteration: same system, new lambda
    } else if (lambda_dead || s.qmax == 10 || rho == 0 || !isfinite(s.lambda)) {
      s.phase = 2;  // OptimizationAlgorithm::Terminate
    } else {
      ++s.it;
      s.phase = 0;
    }
  }
  s.run_step = 0;
  if (mode == 0) {  // (b) schedule: the trial kernels of this step follow
    if (s.phase == 0) {  // `for (it ...) { if (stopped) break; ++done; computeActiveErrors; buildSystem; ...` -- both already in the CURRENT buffers
      if (s.it >= (s.round ? s.iters[1] : s.iters[0]) || s.stopped) {
        s.phase = 2;
      } else {
        if (s.round) ++s.done[1]; else ++s.done[0];
        s.current_chi = s.cur ? s.chi_of[1] : s.chi_of[0];
        if (s.it == 0) {
          s.lambda = 1e-5 * s.maxdiag;  // computeLambdaInit, tau = 1e-5
          s.ni = 2;
        }
        s.qmax = 0;
        s.phase = 3;
      }
    }
    if (s.phase == 3 && s.round < 2) {
      s.phase = 1;
      s.run_step = 1;
      s.ok = 1;
    }
  } else if (mode == 1) {  // round 0 -> classification + round 1, unless stopped (Optimizer.cc:338: `if (!isStop)`)
    s.run_switch = 0;
    if (s.round == 0 && !s.switched) {
      if (s.phase == 0 && (s.it >= s.iters[0] || s.stopped)) s.phase = 2;  // the budget ended exactly with the last decision
      if (s.phase == 2) {
        if (s.stopped)
          s.round = 2;  // both rounds are over
        else
          s.run_switch = 1;
      }
    }
  } else if (mode == 3) {
    s.run_final = 0;
    if (s.round == 1) {
      if (s.phase == 0 && (s.it >= s.iters[1] || s.stopped)) s.phase = 2;
      if (s.phase == 2) s.round = 2;
    }
    if (s.round == 2 && !s.finalized) s.run_final = 1;
  }
  if (threadIdx.x == 0) *st = s;
}
__global__ __launch_bounds__(64) void k_lm_ctrl(LmState* st, LmBuffers B, int mode, int chi_blocks, int scale_blocks,
                                                const double* __restrict__ scale_part, const volatile uint8_t* __restrict__ abort_flag,
                                                LmState* state_out) {
  lm_ctrl_body(st, B, mode, chi_blocks, scale_blocks, scale_part, abort_flag);
  if (state_out && threadIdx.x == 0) *state_out = *st;  // (lane 0 wrote *st itself) -> the result block, for the host
}

// Optimizer.cc:338-359 between the two rounds: level 1 for chi2 > 5.991 / 7.815 or non-positive depth at the CURRENT estimate, kernels dropped
__global__ __launch_bounds__(256) void k_lm_classify(int n_edges, LmBuffers B, const LmState* __restrict__ st, const int32_t* __restrict__ edge_pose,
                                                     const int32_t* __restrict__ edge_point, const double* __restrict__ chi2_last,
                                                     const uint8_t* __restrict__ is_stereo, uint8_t* __restrict__ level,
                                                     double* __restrict__ info_eff, double* __restrict__ delta_eff) {
#pragma clang fp contract(off)
  if (!lm_gate(st, 2)) return;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  const int buf = st->cur;
  const double* T = B.poses[buf] + (size_t)edge_pose[e] * 7;
  const double* X = B.points[buf] + (size_t)edge_point[e] * 3;
  const double qx = T[0], qy = T[1], qw = T[3];
  const double X0 = X[0], X1 = X[1], X2 = X[2];
  double uvx = qy * X2 - T[2] * X1, uvy = T[2] * X0 - qx * X2;
  uvx += uvx;
  uvy += uvy;
  double uvz = qx * X1 - qy * X0;
  uvz += uvz;
  const double z = X2 + qw * uvz + (qx * uvy - qy * uvx) + T[6];  // isDepthPositive: (T * X).z > 0
  const double th = is_stereo[e] ? 7.815 : 5.991;
  if (chi2_last[e] > th || !(z > 0.0)) {
    level[e] = 1;
    info_eff[e] = 0.0;
  }
  delta_eff[e] = -1.0;
}

// final report (Optimizer.cc:364-391): chi2 at the final estimates with the ORIGINAL information, the same test; and the estimates
// themselves into the output staging buffers
__global__ __launch_bounds__(256) void k_lm_final(int n_edges, int n_poses, int n_points, LmBuffers B, const LmState* __restrict__ st,
                                                  const int32_t* __restrict__ edge_pose, const int32_t* __restrict__ edge_point,
                                                  const double* __restrict__ meas, const uint8_t* __restrict__ is_stereo,
                                                  const double* __restrict__ info, BaParamsDev prm, const uint8_t* __restrict__ level,
                                                  double* __restrict__ chi2_out, uint8_t* __restrict__ bad, uint8_t* __restrict__ level_out,
                                                  double* __restrict__ poses_out, double* __restrict__ points_out) {
#pragma clang fp contract(off)
  if (!lm_gate(st, 3)) return;
  const int buf = st->cur;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n_edges) {
    const double* T = B.poses[buf] + (size_t)edge_pose[e] * 7;
    const double* X = B.points[buf] + (size_t)edge_point[e] * 3;
    const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
    const double X0 = X[0], X1 = X[1], X2 = X[2];
    double uvx = qy * X2 - qz * X1, uvy = qz * X0 - qx * X2, uvz = qx * X1 - qy * X0;
    uvx += uvx;
    uvy += uvy;
    uvz += uvz;
    const double x = X0 + qw * uvx + (qy * uvz - qz * uvy) + T[4];
    const double y = X1 + qw * uvy + (qz * uvx - qx * uvz) + T[5];
    const double z = X2 + qw * uvz + (qx * uvy - qy * uvx) + T[6];
    const bool stq = is_stereo[e] != 0;
    const double* m = meas + (size_t)e * 3;
    const double u = x / z * prm.fx + prm.cx, v = y / z * prm.fy + prm.cy;
    const double e0 = m[0] - u, e1 = m[1] - v;
    const double e2 = stq ? (m[2] - (u - prm.bf / z)) : 0.0;
    const double w = info[e];
    const double c2 = stq ? (e0 * (w * e0) + e1 * (w * e1) + e2 * (w * e2)) : (e0 * (w * e0) + e1 * (w * e1));
    chi2_out[e] = c2;
    bad[e] = (c2 > (stq ? 7.815 : 5.991) || !(z > 0.0)) ? 1 : 0;
    level_out[e] = level[e];
  }
  for (int i = e; i < n_poses * 7; i += gridDim.x * 256) poses_out[i] = B.poses[buf][i];
  for (int i = e; i < n_points * 3; i += gridDim.x * 256) points_out[i] = B.points[buf][i];
}

// k_lmbig.hip: the blocked multi-workgroup solver of windows past LM_CHOL_MAX_NB free keyframes
void launch_lm_chol_big(hipStream_t s, const LmLaunch& L);

// ---------------------------------------------------------------------------------------------------------------------------------
static LmBuffers lm_buffers(const LmLaunch& L) {
  LmBuffers B;
  for (int k = 0; k < 2; ++k) {
    B.poses[k] = L.poses[k], B.points[k] = L.points[k], B.terms[k] = L.terms[k], B.Hpl[k] = L.Hpl[k], B.Hpp[k] = L.Hpp[k], B.bp[k] = L.bp[k];
    B.Hll[k] = L.Hll[k], B.bl[k] = L.bl[k], B.chi_part[k] = L.chi_part[k];
  }
  return B;
}

// the system of buffer cur ^ which, gated: the point side and, with_poses, the pose side as a launch of its own (inside a trial it
// rides with k_lm_prep of the NEXT trial instead)
void launch_lm_build(hipStream_t s, const LmLaunch& L, int gate, int which, int write_last, bool with_poses) {
  const LmBuffers B = lm_buffers(L);
  const int pb = (L.NP + 31) / 32;
  if (pb > 0)
    hipLaunchKernelGGL(k_lm_linpoints, dim3(pb), dim3(256), 0, s, L.NP, B, L.state, gate, which, L.pt_off, L.pt_edges, L.edge_pose, L.meas,
                       L.is_stereo, L.info_eff, L.delta_eff, L.fixed, L.level, L.prm, L.chi2_last, write_last);
  if (with_poses && L.NK > 0)
    hipLaunchKernelGGL(k_lm_poseblocks, dim3(L.NK), dim3(256), 0, s, L.NK, B, L.state, gate, which, L.fixed, L.ps_off, L.ps_edges);
}
// the pair lists of the reduced system (k_lm_pair_table / k_lm_pairs); L.pair_table holds 0xFF bytes (-1) when this is called
void launch_lm_pairs(hipStream_t s, const LmLaunch& L) {
  if (L.nf <= 0 || L.E <= 0) return;
  hipLaunchKernelGGL(k_lm_pair_table, dim3((L.E + 255) / 256), dim3(256), 0, s, L.E, L.NP, L.edge_pose, L.edge_point, L.pose_slot, L.pair_table);
  hipLaunchKernelGGL(k_lm_pairs, dim3(L.nf * (L.nf + 1) / 2), dim3(64), 0, s, L.nf, L.NP, L.pair_cap, L.free_pose, L.ps_off, L.ps_edges, L.edge_point,
                     L.pair_table, L.pairs, L.pair_cnt);
}
void launch_lm_maxdiag(hipStream_t s, const LmLaunch& L, int gate) {
  hipLaunchKernelGGL(k_lm_maxdiag, dim3(1), dim3(1024), 0, s, L.NK, L.NP, lm_buffers(L), L.state, gate, L.fixed);
}
void launch_lm_ctrl(hipStream_t s, const LmLaunch& L, int mode) {
  hipLaunchKernelGGL(k_lm_ctrl, dim3(1), dim3(64), 0, s, L.state, lm_buffers(L), mode, (L.NP + 31) / 32, (L.NP + 31) / 32 + (L.NK + 255) / 256, L.scale_part, L.abort_flag,
                     mode == 4 ? L.state_out : (LmState*)nullptr);
}
// one trial: solve + update + the system at the trial estimate (the control step that decides it is the caller's next launch)
void launch_lm_step(hipStream_t s, const LmLaunch& L) {
  const LmBuffers B = lm_buffers(L);
  const int pb = (L.NP + 31) / 32;
  if (pb + L.NK > 0)
    hipLaunchKernelGGL(k_lm_prep, dim3(pb + L.NK), dim3(256), 0, s, L.NP, L.NK, pb, B, L.state, L.pt_off, L.pt_edges, L.Dinv, L.W, L.fixed, L.ps_off,
                       L.ps_edges);
  if (L.nf > 0) {
    hipLaunchKernelGGL(k_lm_schur, dim3(L.nf * (L.nf + 1) / 2 + L.nf), dim3(64), 0, s, L.nf, B, L.state, L.free_pose, L.pair_cnt, L.pair_cap, L.pairs, L.ps_off,
                       L.ps_edges, L.edge_point, L.W, L.Sblk, L.rhs, L.M, L.ld);
    if (L.M)
      launch_lm_chol_big(s, L);
    else
      hipLaunchKernelGGL(k_lm_chol, dim3(1), dim3(LM_CHOL_THREADS), 0, s, L.nf, L.state, L.Sblk, L.rhs, L.x);
  }
  const int ub = pb + (L.NK + 255) / 256;
  if (ub > 0)
    hipLaunchKernelGGL(k_lm_update, dim3(ub), dim3(256), 0, s, L.NK, L.NP, pb, B, L.state, L.pose_slot, L.x, L.pt_off, L.pt_edges,
                       L.edge_pose, L.Dinv, L.scale_part);
  launch_lm_build(s, L, 1, 1, 1, false);
}
// `n` trials as one group: a control step and a trial each, then the caller's control point (launch_lm_switch / launch_lm_final)
void launch_lm_steps(hipStream_t s, const LmLaunch& L, int n) {
  for (int k = 0; k < n; ++k) {
    launch_lm_ctrl(s, L, 0);
    launch_lm_step(s, L);
  }
}
void launch_lm_switch(hipStream_t s, const LmLaunch& L) {
  const LmBuffers B = lm_buffers(L);
  launch_lm_ctrl(s, L, 1);
  if (L.E > 0)
    hipLaunchKernelGGL(k_lm_classify, dim3((L.E + 255) / 256), dim3(256), 0, s, L.E, B, L.state, L.edge_pose, L.edge_point, L.chi2_last, L.is_stereo,
                       L.level, L.info_eff, L.delta_eff);
  launch_lm_build(s, L, 2, 0, 1, true);
  launch_lm_maxdiag(s, L, 2);
  launch_lm_ctrl(s, L, 2);
}
void launch_lm_final(hipStream_t s, const LmLaunch& L) {
  const LmBuffers B = lm_buffers(L);
  launch_lm_ctrl(s, L, 3);
  const int n = std::max(L.E, 1);
  hipLaunchKernelGGL(k_lm_final, dim3((n + 255) / 256), dim3(256), 0, s, L.E, L.NK, L.NP, B, L.state, L.edge_pose, L.edge_point, L.meas, L.is_stereo,
                     L.info, L.prm, L.level, L.chi2_out, L.bad, L.level_out, L.poses_out, L.points_out);
  launch_lm_ctrl(s, L, 4);
}

}  // namespace orbfe
