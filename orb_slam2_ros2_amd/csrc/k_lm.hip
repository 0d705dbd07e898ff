// k_lm.hip -- Optimizer::OptimizeLocalMap's two optimizer.optimize() calls (src/ORB_SLAM2/src/Optimizer.cc:336-361) with g2o's
// Levenberg-Marquardt CONTROL on the device: lambda, the gain ratio, accept / restore, the trial and iteration counters, the round
// switch (outlier classification, :338-359) and the stop-flag poll all live in one small state record that a one-lane control
// kernel advances between the trials; the host only enqueues a fixed stream of (gated) kernels and synchronises ONCE per call
// (twice when a round needed more rejected trials than were provisioned).  Round 2 ran this loop on the host with two blocking
// synchronisations per trial: 12.2 ms for BASELINE config 5, most of it round trips and three kernels written for convenience.
//
// One TRIAL (= one step of the enqueued stream), every kernel gated by state.run_step:
//   k_lm_ctrl        decide the outstanding trial (rho = (chi_cur - chi_trial) / (scale + 1e-3); accept: lambda *= max(1/3, 1 - (2 rho - 1)^3),
//                    swap the estimate / system buffers; reject: lambda *= ni, ni *= 2), start the next iteration or trial, or finish
//   k_lm_prep        per point: (Hll + lambda I)^-1, W(e) = Hpl(e) Dinv for its edges             (BlockSolver_6_3::solve, marginalised points)
//   k_lm_schur       one wave per block (i >= j) of free poses: S_ij = [i == j](Hpp_i + lambda I) - sum W(e1) Hpl(e2)^T, lanes over the pairs
//   k_lm_chol        dense Cholesky + both substitutions of the reduced system by ONE workgroup with the matrix in REGISTERS:
//                    one 6x6 block per thread, the panel of a block column exchanged through LDS (up to 42 free keyframes)
//   k_lm_update      oplus into the OTHER estimate buffer (no push / pop copies), computeScale partial sums
//   k_lm_linearize   one lane per edge at the trial estimate: error, chi2, Huber weight, both Jacobians ONCE, Hpl, partial sums of
//                    the robust chi2 -- kept as the system of the next iteration if the trial is accepted (g2o rebuilds the same numbers)
//   k_lm_blocks      Hll / bl per point and Hpp / bp per pose from the stored edge terms (segmented sums over host-built CSR lists)
// All fp64, contraction off, every sum in a fixed order: run-to-run identical.  Scatter-adds of 6x6 / 6x3 / 3x3 blocks keyed by vertex
// ids and a <= 258-row triangular factorisation are no dense contractions worth MFMA tiles.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"
#include "se3_dev.h"

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

__global__ __launch_bounds__(256) void k_lm_linearize(int n_edges, LmBuffers B, const LmState* __restrict__ st, int gate, int which,
                                                      const int32_t* __restrict__ edge_pose, const int32_t* __restrict__ edge_point,
                                                      const double* __restrict__ meas, const uint8_t* __restrict__ is_stereo,
                                                      const double* __restrict__ info, const double* __restrict__ delta,
                                                      const uint8_t* __restrict__ pose_fixed, const uint8_t* __restrict__ level, BaParamsDev prm,
                                                      double* __restrict__ chi2_last, int write_last) {
#pragma clang fp contract(off)
  __shared__ double sh[256];
  if (!lm_gate(st, gate)) return;
  const int buf = st->cur ^ which;
  const int e = blockIdx.x * 256 + threadIdx.x;
  double r0 = 0.0;
  if (e < n_edges) {
    const int kp = edge_pose[e];
    const double* T = B.poses[buf] + (size_t)kp * 7;
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
    double A[9], Bm[18];
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
    t[27] = w * e0;
    t[28] = w * e1;
    t[29] = w * e2;
    t[30] = w;
    t[31] = (double)rows;
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
          for (int r = 0; r < rows; ++r) h += Bm[6 * r + a] * w * A[3 * r + c];
          out[3 * a + c] = h;
        }
    }
  }
  const double s = block_sum_256(r0, sh);
  if (threadIdx.x == 0) B.chi_part[buf][blockIdx.x] = s;
}

// Hll / bl per point (blocks [0, pt_blocks)) and Hpp / bp per pose (one 256-thread block per pose after them), from the stored terms
__global__ __launch_bounds__(256) void k_lm_blocks(int n_points, int n_poses, int pt_blocks, LmBuffers B, const LmState* __restrict__ st, int gate,
                                                   int which, const uint8_t* __restrict__ pose_fixed, const int32_t* __restrict__ pt_off,
                                                   const int32_t* __restrict__ pt_edges, const int32_t* __restrict__ ps_off,
                                                   const int32_t* __restrict__ ps_edges) {
#pragma clang fp contract(off)
  __shared__ double part[4][42];
  if (!lm_gate(st, gate)) return;
  const int buf = st->cur ^ which;
  const double* terms = B.terms[buf];
  if ((int)blockIdx.x < pt_blocks) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_points) return;
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int i = pt_off[p]; i < pt_off[p + 1]; ++i) {
      const double* t = terms + (size_t)pt_edges[i] * LM_TERM;
      const int rows = (int)t[31];
      const double w = t[30];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double s = 0;
        for (int r = 0; r < rows; ++r) s += t[3 * r + a] * t[27 + r];
        b[a] -= s;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double h = 0;
          for (int r = 0; r < rows; ++r) h += t[3 * r + a] * w * t[3 * r + c];
          H[3 * a + c] += h;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) B.Hll[buf][(size_t)p * 9 + k] = H[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) B.bl[buf][(size_t)p * 3 + k] = b[k];
    return;
  }
  const int k = (int)blockIdx.x - pt_blocks;
  if (k >= n_poses) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc = 0.0;
  if (!pose_fixed[k] && lane < 42) {  // lane = entry: (a, c) of Hpp for lane < 36, else entry of bp; wave wv takes the edges q = wv (mod 4)
    const int a = lane < 36 ? lane / 6 : lane - 36, c = lane < 36 ? lane - 6 * (lane / 6) : 0;
    for (int q = ps_off[k] + wv; q < ps_off[k + 1]; q += 4) {
      const double* t = terms + (size_t)ps_edges[q] * LM_TERM;
      const int rows = (int)t[31];
      if (lane < 36) {
        const double w = t[30];
        double h = 0;
        for (int r = 0; r < rows; ++r) h += t[9 + 6 * r + a] * w * t[9 + 6 * r + c];
        acc += h;
      } else {
        double s = 0;
        for (int r = 0; r < rows; ++r) s += t[9 + 6 * r + a] * t[27 + r];
        acc -= s;
      }
    }
  }
  if (lane < 42) part[wv][lane] = acc;
  __syncthreads();
  if (threadIdx.x < 42) {
    const double v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
    if (threadIdx.x < 36)
      B.Hpp[buf][(size_t)k * 36 + threadIdx.x] = v;
    else
      B.bp[buf][(size_t)k * 6 + threadIdx.x - 36] = v;
  }
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

// per point: Dinv = (Hll + lambda I)^-1 (Eigen's 3x3 inverse: cofactors / determinant), then W(e) = Hpl(e) Dinv for the point's edges
__global__ __launch_bounds__(256) void k_lm_prep(int n_points, LmBuffers B, LmState* __restrict__ st, const int32_t* __restrict__ pt_off,
                                                 const int32_t* __restrict__ pt_edges, double* __restrict__ Dinv, double* __restrict__ W) {
#pragma clang fp contract(off)
  if (!lm_gate(st, 1)) return;
  const int p = blockIdx.x * 256 + threadIdx.x;
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
    st->ok = 0;
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
#pragma unroll
  for (int i = 0; i < 9; ++i) Dinv[(size_t)p * 9 + i] = D[i];
  for (int q = pt_off[p]; q < pt_off[p + 1]; ++q) {
    const int e = pt_edges[q];
    const double* H = B.Hpl[buf] + (size_t)e * 18;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double h0 = H[3 * a], h1 = H[3 * a + 1], h2 = H[3 * a + 2];
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

// Reduced system, one wave per block (i >= j) of free poses, the lanes over the block's pairs (e1, e2): pose(e1) = i, pose(e2) = j, same
// point.  Sblk: lower-triangular blocks, block (i, j) at (i (i + 1) / 2 + j) * 36, row-major inside.  The diagonal blocks also form
// the right-hand side rhs_i = bp_i - sum_{e of pose i} W(e) bl(point(e)).
__device__ __forceinline__ double wave_sum_fixed(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);  // fixed butterfly: deterministic
  return v;
}

__global__ __launch_bounds__(64) void k_lm_schur(int nf, LmBuffers B, const LmState* __restrict__ st, const int32_t* __restrict__ free_pose,
                                                 const int32_t* __restrict__ pair_off, const int2* __restrict__ pairs,
                                                 const int32_t* __restrict__ ps_off, const int32_t* __restrict__ ps_edges,
                                                 const int32_t* __restrict__ edge_point, const double* __restrict__ W,
                                                 double* __restrict__ Sblk, double* __restrict__ rhs) {
#pragma clang fp contract(off)
  if (!lm_gate(st, 1)) return;
  // blockIdx.x -> (i, j), i >= j
  const int b = blockIdx.x;
  int i = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= b) ++i;
  while (i * (i + 1) / 2 > b) --i;
  const int j = b - i * (i + 1) / 2;
  const int lane = threadIdx.x;
  const int buf = st->cur;
  const double* Hpl = B.Hpl[buf];
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  const int q0 = pair_off[i * nf + j], q1 = pair_off[i * nf + j + 1];
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
      out[k] = v;
    }
  }
  if (i == j) {
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
      if (lane == a) rhs[6 * i + a] = B.bp[buf][(size_t)ki * 6 + a] - s;
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

__device__ __forceinline__ void lm_tri_index(int t, int& I, int& J) {  // t -> (I, J), I > J, strictly lower: t = I (I - 1) / 2 + J
  I = (int)((sqrt(8.0 * (double)t + 1.0) + 1.0) * 0.5);
  while (I * (I - 1) / 2 > t) --I;
  while ((I + 1) * I / 2 <= t) ++I;
  J = t - I * (I - 1) / 2;
}

// X L^T = A for the rows of one block, in place (L, inv: LDS)
__device__ __forceinline__ void lm_panel_solve(double (&A)[36], const double* L, const double* inv) {
#pragma clang fp contract(off)
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double sv = A[6 * a + c];
#pragma unroll
      for (int m = 0; m < c; ++m) sv -= v[m] * L[6 * c + m];
      v[c] = sv * inv[c];
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) A[6 * a + c] = v[c];
  }
}
// A -= P_I P_J^T (panel blocks in LDS)
__device__ __forceinline__ void lm_block_update(double (&A)[36], const double* PI, const double* PJ) {
#pragma clang fp contract(off)
  double pj[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) pj[k] = PJ[k];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double pi[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) pi[k] = PI[6 * a + k];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double* q = pj + 6 * c;
      A[6 * a + c] -= pi[0] * q[0] + pi[1] * q[1] + pi[2] * q[2] + pi[3] * q[3] + pi[4] * q[4] + pi[5] * q[5];
    }
  }
}

__global__ __launch_bounds__(LM_CHOL_THREADS) void k_lm_chol(int nb, LmState* __restrict__ st, const double* __restrict__ Sblk,
                                                             const double* __restrict__ rhs, double* __restrict__ x) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) double P[(LM_CHOL_MAX_NB + 1) * LM_PSTRIDE];  // panel blocks of the current column; slot nb: the rhs row (6)
  __shared__ __attribute__((aligned(16))) double Lk[36 + 6];                            // L_kk (row-major, lower) + 1 / diagonal
  __shared__ double yv[6 * LM_CHOL_MAX_NB];                                             // y, then x
  __shared__ int s_ok;
  if (!lm_gate(st, 1)) return;
  const int t = threadIdx.x;
  const int n_off = nb * (nb - 1) / 2;  // strictly-lower blocks
  const bool diag = t >= LM_CHOL_OFF && t - LM_CHOL_OFF < nb;
  // off-diagonal owners: blocks (I0, J0) and (I1, J1); a diagonal owner keeps its block in A0 (I0 == J0) and its right-hand side in Y
  int I0 = -1, J0 = -1, I1 = -1, J1 = -1;
  double A0[36], A1[36], Y[6];
#pragma unroll
  for (int k = 0; k < 36; ++k) A0[k] = 0.0, A1[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) Y[k] = 0.0;
  if (t < LM_CHOL_OFF) {
    if (t < n_off) {
      lm_tri_index(t, I0, J0);
      const double* src = Sblk + ((size_t)I0 * (I0 + 1) / 2 + J0) * 36;
#pragma unroll
      for (int k = 0; k < 36; ++k) A0[k] = src[k];
    }
    if (t + LM_CHOL_OFF < n_off) {
      lm_tri_index(t + LM_CHOL_OFF, I1, J1);
      const double* src = Sblk + ((size_t)I1 * (I1 + 1) / 2 + J1) * 36;
#pragma unroll
      for (int k = 0; k < 36; ++k) A1[k] = src[k];
    }
  } else if (diag) {
    I0 = J0 = t - LM_CHOL_OFF;
    const double* src = Sblk + ((size_t)I0 * (I0 + 1) / 2 + I0) * 36;
#pragma unroll
    for (int k = 0; k < 36; ++k) A0[k] = src[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) Y[k] = rhs[6 * I0 + k];
  }
  if (t == 0) s_ok = 1;
  __syncthreads();

  auto factor_diag = [&]() {  // lower Cholesky of A0 in place (this thread owns a diagonal block), publish L and 1 / diag
    bool good = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double d = A0[7 * j];
#pragma unroll
      for (int k = 0; k < j; ++k) d -= A0[6 * j + k] * A0[6 * j + k];
      if (!(d > 0) || !isfinite(d)) good = false;
      d = sqrt(d);
      A0[7 * j] = d;
      const double inv = 1.0 / d;
      Lk[36 + j] = inv;
#pragma unroll
      for (int i2 = j + 1; i2 < 6; ++i2) {
        double v = A0[6 * i2 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) v -= A0[6 * i2 + k] * A0[6 * j + k];
        A0[6 * i2 + j] = v * inv;
      }
    }
    if (!good) s_ok = 0;
#pragma unroll
    for (int k = 0; k < 36; ++k) Lk[k] = A0[k];
  };
  if (diag && I0 == 0) factor_diag();
  __syncthreads();

  for (int kb = 0; kb < nb; ++kb) {
    if (!s_ok) break;  // uniform (read after a barrier)
    // (2) panel: X L_kk^T = A for the blocks (I, kb), I > kb, and the right-hand-side block kb
    if (!diag) {
      if (J0 == kb) {
        lm_panel_solve(A0, Lk, Lk + 36);
        double* dst = P + (size_t)I0 * LM_PSTRIDE;
#pragma unroll
        for (int k = 0; k < 36; ++k) dst[k] = A0[k];
      }
      if (J1 == kb) {
        lm_panel_solve(A1, Lk, Lk + 36);
        double* dst = P + (size_t)I1 * LM_PSTRIDE;
#pragma unroll
        for (int k = 0; k < 36; ++k) dst[k] = A1[k];
      }
    } else if (I0 == kb) {  // y_kb L_kk^T = Y (the factor is still in this thread's registers)
      double v[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double sv = Y[c];
#pragma unroll
        for (int m = 0; m < c; ++m) sv -= v[m] * A0[6 * c + m];
        v[c] = sv * Lk[36 + c];
      }
      double* dst = P + (size_t)nb * LM_PSTRIDE;
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[k] = v[k], yv[6 * kb + k] = v[k];
    }
    __syncthreads();
    // (3) trailing update: blocks (I, J) with J > kb (diagonal ones included), the right-hand-side blocks J > kb
    if (!diag) {
      if (J0 > kb) lm_block_update(A0, P + (size_t)I0 * LM_PSTRIDE, P + (size_t)J0 * LM_PSTRIDE);
      if (J1 > kb) lm_block_update(A1, P + (size_t)I1 * LM_PSTRIDE, P + (size_t)J1 * LM_PSTRIDE);
    } else if (I0 > kb) {
      const double* PJ = P + (size_t)I0 * LM_PSTRIDE;
      lm_block_update(A0, PJ, PJ);
      const double* PY = P + (size_t)nb * LM_PSTRIDE;
      double py[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) py[k] = PY[k];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double* q = PJ + 6 * c;
        Y[c] -= py[0] * q[0] + py[1] * q[1] + py[2] * q[2] + py[3] * q[3] + py[4] * q[4] + py[5] * q[5];
      }
      if (I0 == kb + 1) factor_diag();  // look-ahead: the next diagonal block is complete now
    }
    __syncthreads();
  }
  if (!s_ok) {
    if (t == 0) st->ok = 0;
    for (int k = t; k < 6 * nb; k += LM_CHOL_THREADS) x[k] = 0.0;
    return;
  }
  // backward substitution L^T x = y, block by block from the bottom
  for (int kb = nb - 1; kb >= 0; --kb) {
    if (diag && I0 == kb) {
      double xv[6];
#pragma unroll
      for (int a = 5; a >= 0; --a) {
        double v = yv[6 * kb + a];
#pragma unroll
        for (int m = a + 1; m < 6; ++m) v -= A0[6 * m + a] * xv[m];
        xv[a] = v / A0[7 * a];
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) yv[6 * kb + a] = xv[a];
    }
    __syncthreads();
    if (!diag) {  // y_J -= L_{kb,J}^T x_kb : the only writer of y_J in this step (a thread's two blocks have different (I, J))
      if (I0 == kb) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double v = yv[6 * J0 + c];
#pragma unroll
          for (int a = 0; a < 6; ++a) v -= A0[6 * a + c] * yv[6 * kb + a];
          yv[6 * J0 + c] = v;
        }
      }
      if (I1 == kb) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double v = yv[6 * J1 + c];
#pragma unroll
          for (int a = 0; a < 6; ++a) v -= A1[6 * a + c] * yv[6 * kb + a];
          yv[6 * J1 + c] = v;
        }
      }
    }
    __syncthreads();
  }
  for (int k = t; k < 6 * nb; k += LM_CHOL_THREADS) x[k] = yv[k];
}

// SparseOptimizer::update into the OTHER buffer: poses <- exp(dx) * pose (free poses; fixed ones are copied), points <- point + Dinv (bl
// - sum Hpl^T dxp); computeScale partial sums sum_i dx_i (lambda dx_i + b_i) per block -> scale_part[blockIdx.x]
__global__ __launch_bounds__(256) void k_lm_update(int n_poses, int n_points, LmBuffers B, const LmState* __restrict__ st,
                                                   const int32_t* __restrict__ pose_slot, const double* __restrict__ x,
                                                   const int32_t* __restrict__ pt_off, const int32_t* __restrict__ pt_edges,
                                                   const int32_t* __restrict__ edge_pose, const double* __restrict__ Dinv,
                                                   double* __restrict__ scale_part) {
#pragma clang fp contract(off)
  __shared__ double sh[256];
  if (!lm_gate(st, 1)) return;
  const int cur = st->cur, nxt = cur ^ 1;
  const double lambda = st->lambda;
  const int t = blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  if (t < n_points) {
    const int p = t;
    const double* bl = B.bl[cur] + (size_t)p * 3;
    double r0 = bl[0], r1 = bl[1], r2 = bl[2];
    for (int q = pt_off[p]; q < pt_off[p + 1]; ++q) {
      const int e = pt_edges[q];
      const int s = pose_slot[edge_pose[e]];
      if (s < 0) continue;
      const double* h = B.Hpl[cur] + (size_t)e * 18;
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
      B.points[nxt][(size_t)p * 3 + a] = B.points[cur][(size_t)p * 3 + a] + dv;
      acc += dv * (lambda * dv + bl[a]);
    }
  } else if (t < n_points + n_poses) {
    const int k = t - n_points;
    const int s = pose_slot[k];
    const double* src = B.poses[cur] + (size_t)k * 7;
    double* dst = B.poses[nxt] + (size_t)k * 7;
    if (s >= 0) {
      double upd[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) upd[a] = x[6 * s + a];
      PoseDev T, R;
      for (int a = 0; a < 4; ++a) T.q[a] = src[a];
      for (int a = 0; a < 3; ++a) T.t[a] = src[4 + a];
      pose_oplus(T, upd, R);
      for (int a = 0; a < 4; ++a) dst[a] = R.q[a];
      for (int a = 0; a < 3; ++a) dst[4 + a] = R.t[a];
      const double* bp = B.bp[cur] + (size_t)k * 6;
#pragma unroll
      for (int a = 0; a < 6; ++a) acc += upd[a] * (lambda * upd[a] + bp[a]);
    } else {
      for (int a = 0; a < 7; ++a) dst[a] = src[a];
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
__device__ __forceinline__ double lm_sum(const double* p, int n) {
  double s = 0;
  for (int i = 0; i < n; ++i) s += p[i];
  return s;
}

__global__ __launch_bounds__(64) void k_lm_ctrl(LmState* __restrict__ st, LmBuffers B, int mode, int chi_blocks, int scale_blocks,
                                                const double* __restrict__ scale_part, const volatile uint8_t* __restrict__ abort_flag) {
#pragma clang fp contract(off)
  if (threadIdx.x != 0) return;
  LmState s = *st;
  if (abort_flag && *abort_flag) s.stopped = 1;
  if (mode == 2 && s.run_switch) {  // the switch group has run: round 1 starts on the system it rebuilt (new information / no kernels)
    s.run_switch = 0;
    s.switched = 1;
    s.round = 1;
    s.it = 0;
    s.phase = 0;
    s.need_chi = 1;
  }
  if (mode == 4 && s.run_final) {
    s.run_final = 0;
    s.finalized = 1;
  }
  if (s.need_chi) {
    s.chi_of[s.cur] = lm_sum(B.chi_part[s.cur], chi_blocks);
    s.need_chi = 0;
  }
  // (a) a trial ran since the last control point: decide it
  if (s.phase == 1) {
    const double chi_trial = lm_sum(B.chi_part[s.cur ^ 1], chi_blocks);
    const double scale = lm_sum(scale_part, scale_blocks);
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
      s.chi_of[s.cur ^ 1] = chi_trial;
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
      if (s.it >= s.iters[s.round] || s.stopped) {
        s.phase = 2;
      } else {
        ++s.done[s.round];
        s.current_chi = s.chi_of[s.cur];
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
  *st = s;
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
                                                  const double* __restrict__ info, BaParamsDev prm, double* __restrict__ chi2_out,
                                                  uint8_t* __restrict__ bad, double* __restrict__ poses_out, double* __restrict__ points_out) {
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
  }
  for (int i = e; i < n_poses * 7; i += gridDim.x * 256) poses_out[i] = B.poses[buf][i];
  for (int i = e; i < n_points * 3; i += gridDim.x * 256) points_out[i] = B.points[buf][i];
}

// ---------------------------------------------------------------------------------------------------------------------------------
static LmBuffers lm_buffers(const LmLaunch& L) {
  LmBuffers B;
  for (int k = 0; k < 2; ++k) {
    B.poses[k] = L.poses[k], B.points[k] = L.points[k], B.terms[k] = L.terms[k], B.Hpl[k] = L.Hpl[k], B.Hpp[k] = L.Hpp[k], B.bp[k] = L.bp[k];
    B.Hll[k] = L.Hll[k], B.bl[k] = L.bl[k], B.chi_part[k] = L.chi_part[k];
  }
  return B;
}

// linearize + blocks of buffer cur ^ which, gated
void launch_lm_build(hipStream_t s, const LmLaunch& L, int gate, int which, int write_last) {
  const LmBuffers B = lm_buffers(L);
  const int eb = (L.E + 255) / 256, pb = (L.NP + 255) / 256;
  if (L.E > 0)
    hipLaunchKernelGGL(k_lm_linearize, dim3(eb), dim3(256), 0, s, L.E, B, L.state, gate, which, L.edge_pose, L.edge_point, L.meas, L.is_stereo,
                       L.info_eff, L.delta_eff, L.fixed, L.level, L.prm, L.chi2_last, write_last);
  if (pb + L.NK > 0)
    hipLaunchKernelGGL(k_lm_blocks, dim3(pb + L.NK), dim3(256), 0, s, L.NP, L.NK, pb, B, L.state, gate, which, L.fixed, L.pt_off, L.pt_edges, L.ps_off,
                       L.ps_edges);
}
void launch_lm_maxdiag(hipStream_t s, const LmLaunch& L, int gate) {
  hipLaunchKernelGGL(k_lm_maxdiag, dim3(1), dim3(1024), 0, s, L.NK, L.NP, lm_buffers(L), L.state, gate, L.fixed);
}
void launch_lm_ctrl(hipStream_t s, const LmLaunch& L, int mode) {
  hipLaunchKernelGGL(k_lm_ctrl, dim3(1), dim3(64), 0, s, L.state, lm_buffers(L), mode, (L.E + 255) / 256, (L.NP + L.NK + 255) / 256, L.scale_part,
                     L.abort_flag);
}
// one trial: ctrl + solve + update + the system at the trial estimate
void launch_lm_step(hipStream_t s, const LmLaunch& L) {
  const LmBuffers B = lm_buffers(L);
  launch_lm_ctrl(s, L, 0);
  if (L.NP > 0) hipLaunchKernelGGL(k_lm_prep, dim3((L.NP + 255) / 256), dim3(256), 0, s, L.NP, B, L.state, L.pt_off, L.pt_edges, L.Dinv, L.W);
  if (L.nf > 0) {
    hipLaunchKernelGGL(k_lm_schur, dim3(L.nf * (L.nf + 1) / 2), dim3(64), 0, s, L.nf, B, L.state, L.free_pose, L.pair_off, L.pairs, L.ps_off,
                       L.ps_edges, L.edge_point, L.W, L.Sblk, L.rhs);
    hipLaunchKernelGGL(k_lm_chol, dim3(1), dim3(LM_CHOL_THREADS), 0, s, L.nf, L.state, L.Sblk, L.rhs, L.x);
  }
  const int nt = L.NP + L.NK;
  if (nt > 0)
    hipLaunchKernelGGL(k_lm_update, dim3((nt + 255) / 256), dim3(256), 0, s, L.NK, L.NP, B, L.state, L.pose_slot, L.x, L.pt_off, L.pt_edges,
                       L.edge_pose, L.Dinv, L.scale_part);
  launch_lm_build(s, L, 1, 1, 1);
}
void launch_lm_switch(hipStream_t s, const LmLaunch& L) {
  const LmBuffers B = lm_buffers(L);
  launch_lm_ctrl(s, L, 1);
  if (L.E > 0)
    hipLaunchKernelGGL(k_lm_classify, dim3((L.E + 255) / 256), dim3(256), 0, s, L.E, B, L.state, L.edge_pose, L.edge_point, L.chi2_last, L.is_stereo,
                       L.level, L.info_eff, L.delta_eff);
  launch_lm_build(s, L, 2, 0, 1);
  launch_lm_maxdiag(s, L, 2);
  launch_lm_ctrl(s, L, 2);
}
void launch_lm_final(hipStream_t s, const LmLaunch& L) {
  const LmBuffers B = lm_buffers(L);
  launch_lm_ctrl(s, L, 3);
  const int n = std::max(L.E, 1);
  hipLaunchKernelGGL(k_lm_final, dim3((n + 255) / 256), dim3(256), 0, s, L.E, L.NK, L.NP, B, L.state, L.edge_pose, L.edge_point, L.meas, L.is_stereo,
                     L.info, L.prm, L.chi2_out, L.bad, L.poses_out, L.points_out);
  launch_lm_ctrl(s, L, 4);
}

}  // namespace orbfe
