// k_pose.hip -- Optimizer::OptimizePoseOnly on the device: one workgroup runs the whole optimisation of one frame.
//
// Replaces src/ORB_SLAM2/src/Optimizer.cc:33-203 (the g2o part): one VertexSE3Expmap, EdgeSE3ProjectXYZOnlyPose /
// EdgeStereoSE3ProjectXYZOnlyPose per observed map point (mono when rightU < 0), information invSigma2(octave), Huber
// sqrt(5.991)/sqrt(7.815); four rounds of optimize(10), each restarting from the initial pose, edges re-classified with
// chi2 > 5.991*sigma2 / 7.815*sigma2 after every round, robust kernels dropped in the third round; g2o's Levenberg-Marquardt
// control (tau 1e-5, gain ratio, lambda *= max(1/3, min(2/3, 1-(2rho-1)^3)), ni doubling, <= 10 trials per iteration).
// A single 6-DoF vertex makes every iteration: evaluate <= 2000 edges in parallel, reduce 21+6+1 doubles over the workgroup
// (fixed tree => deterministic), solve a 6x6 system on one lane.  No host round trip per iteration.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"
#include "se3_dev.h"
#include "wave_ops.h"

namespace orbfe {

#define POSE_THREADS 256

__device__ bool solve6(const double* A /*6x6 row-major*/, const double* b, double* x) {
  double L[36];
  for (int i = 0; i < 36; ++i) L[i] = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[6 * i + j];
      for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
      if (i == j) {
        if (!(s > 0)) return false;
        L[6 * i + i] = sqrt(s);
      } else
        L[6 * i + j] = s / L[6 * j + j];
    }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
    y[i] = s / L[6 * i + i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
    x[i] = s / L[6 * i + i];
  }
  return true;
}

// (H + lambda I) x = b with H and b where they are (LDS), for the register-resident kernel, where this lane-0 solve and the pose update
// behind it are the longest serial piece of a pass (9 k of its ~20 k cycles: ~33 fp64 divisions and square roots of ~100 dependent cycles
// each): one reciprocal square root per pivot -- v_rsq_f64 and a third-order correction y0 (1 + e / 2 + 3 e^2 / 8), e = 1 - s y0^2, error
// below double rounding -- L_ii = s y, and every division by L_ii a multiplication by y.
__device__ __forceinline__ bool solve6_lambda(const double* H, double lambda, const double* b, double* x) {
#pragma clang fp contract(off)
  double L[36], inv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = (i == j) ? H[6 * i + j] + lambda : H[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
      if (i == j) {
        if (!(s > 0) || !isfinite(s)) return false;
        const double y0 = __builtin_amdgcn_rsq(s);
        const double e = fma(-(s * y0), y0, 1.0);
        const double y = fma(y0 * e, fma(0.375, e, 0.5), y0);
        L[6 * i + i] = s * y;
        inv[i] = y;
      } else
        L[6 * i + j] = s * inv[j];
    }
  }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
    y[i] = s * inv[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
    x[i] = s * inv[i];
  }
  return true;
}

struct PoseShared {
  PoseDev T, T0, backup;
  double H[36], b[6], x[6];
  double red[28][POSE_THREADS / 64];
  double lambda, ni, current_chi, temp_chi, rho;
  int qmax, cont_inner, stop_outer, any_active, n_bad;
};

// sum of v over the workgroup, result in every thread (fixed order: wave butterfly, then waves 0..3)
__device__ __forceinline__ double block_sum(double v, double (*slot)[POSE_THREADS / 64], int row) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) slot[row][threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int w = 0; w < POSE_THREADS / 64; ++w) s += slot[row][w];
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(POSE_THREADS) void k_pose_only(int n, const double* __restrict__ Xw, const double* __restrict__ meas,
                                                            const double* __restrict__ info, const float* __restrict__ sigma2,
                                                            const double* __restrict__ pose_in, BaParamsDev prm, double d_mono,
                                                            double d_stereo, double* __restrict__ err, uint8_t* __restrict__ level,
                                                            uint8_t* __restrict__ robust, uint8_t* __restrict__ inlier,
                                                            double* __restrict__ pose_out, int32_t* __restrict__ n_good) {
#pragma clang fp contract(off)
  __shared__ PoseShared S;
  const int tid = threadIdx.x;
  if (tid == 0) {
    for (int i = 0; i < 4; ++i) S.T0.q[i] = pose_in[i];
    for (int i = 0; i < 3; ++i) S.T0.t[i] = pose_in[4 + i];
    S.T = S.T0;
  }
  for (int i = tid; i < n; i += POSE_THREADS) {
    level[i] = 0;
    robust[i] = 1;
    inlier[i] = 1;
  }
  __syncthreads();

  auto edge_error = [&](int i, const PoseDev& T, double* e) {
    double p[3];
    quat_rotate(T.q, Xw + 3 * i, p);
    const double x = p[0] + T.t[0], y = p[1] + T.t[1], z = p[2] + T.t[2];
    const double* m = meas + 3 * i;
    const double u = x / z * prm.fx + prm.cx, v = y / z * prm.fy + prm.cy;
    e[0] = m[0] - u;
    e[1] = m[1] - v;
    e[2] = (m[2] < 0) ? 0.0 : m[2] - (u - prm.bf / z);
  };
  auto edge_chi2 = [&](int i, const double* e) {
    const double w = info[i];
    double c = e[0] * (w * e[0]) + e[1] * (w * e[1]);
    if (!(meas[3 * i + 2] < 0)) c += e[2] * (w * e[2]);
    return c;
  };
  // computeActiveErrors + activeRobustChi2 at the current S.T
  auto eval_chi = [&]() {
    const PoseDev T = S.T;
    double part = 0;
    for (int i = tid; i < n; i += POSE_THREADS) {
      if (level[i] != 0) continue;
      double e[3];
      edge_error(i, T, e);
      err[3 * i] = e[0];
      err[3 * i + 1] = e[1];
      err[3 * i + 2] = e[2];
      const double c = edge_chi2(i, e);
      if (robust[i]) {
        const double dl = (meas[3 * i + 2] < 0) ? d_mono : d_stereo;
        const double dsqr = dl * dl;
        part += (c <= dsqr) ? c : (2 * sqrt(c) * dl - dsqr);
      } else
        part += c;
    }
    return block_sum(part, S.red, 27);
  };
  // linearizeOplus + constructQuadraticForm of every active edge at S.T (errors of the last evaluation)
  auto build_system = [&]() {
    const PoseDev T = S.T;
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0;
    for (int i = tid; i < n; i += POSE_THREADS) {
      if (level[i] != 0) continue;
      double p[3];
      quat_rotate(T.q, Xw + 3 * i, p);
      const double x = p[0] + T.t[0], y = p[1] + T.t[1], z = p[2] + T.t[2];
      const double invz = 1.0 / z, invz_2 = invz * invz;
      const bool st = !(meas[3 * i + 2] < 0);
      double J[3][6];
      J[0][0] = x * y * invz_2 * prm.fx;
      J[0][1] = -(1 + (x * x * invz_2)) * prm.fx;
      J[0][2] = y * invz * prm.fx;
      J[0][3] = -invz * prm.fx;
      J[0][4] = 0;
      J[0][5] = x * invz_2 * prm.fx;
      J[1][0] = (1 + y * y * invz_2) * prm.fy;
      J[1][1] = -x * y * invz_2 * prm.fy;
      J[1][2] = -x * invz * prm.fy;
      J[1][3] = 0;
      J[1][4] = -invz * prm.fy;
      J[1][5] = y * invz_2 * prm.fy;
      J[2][0] = st ? J[0][0] - prm.bf * y * invz_2 : 0.0;
      J[2][1] = st ? J[0][1] + prm.bf * x * invz_2 : 0.0;
      J[2][2] = st ? J[0][2] : 0.0;
      J[2][3] = st ? J[0][3] : 0.0;
      J[2][4] = 0;
      J[2][5] = st ? J[0][5] - prm.bf * invz_2 : 0.0;
      const double e[3] = {err[3 * i], err[3 * i + 1], st ? err[3 * i + 2] : 0.0};
      const double w = info[i];
      double r1 = 1.0;
      if (robust[i]) {
        const double c = edge_chi2(i, e);
        const double dl = st ? d_stereo : d_mono;
        if (c > dl * dl) r1 = dl / sqrt(c);
      }
      int k = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        acc[21 + a] -= r1 * (J[0][a] * (w * e[0]) + J[1][a] * (w * e[1]) + J[2][a] * (w * e[2]));
#pragma unroll
        for (int c2 = a; c2 < 6; ++c2) acc[k++] += J[0][a] * (r1 * w) * J[0][c2] + J[1][a] * (r1 * w) * J[1][c2] + J[2][a] * (r1 * w) * J[2][c2];
      }
    }
    double tot[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) tot[k] = block_sum(acc[k], S.red, k);
    if (tid == 0) {
      int k = 0;
      for (int a = 0; a < 6; ++a)
        for (int c2 = a; c2 < 6; ++c2) {
          S.H[6 * a + c2] = tot[k];
          S.H[6 * c2 + a] = tot[k];
          ++k;
        }
      for (int a = 0; a < 6; ++a) S.b[a] = tot[21 + a];
    }
    __syncthreads();
  };

  // edge->computeError() at construction (Optimizer.cc:90,111)
  for (int i = tid; i < n; i += POSE_THREADS) {
    double e[3];
    edge_error(i, S.T0, e);
    err[3 * i] = e[0];
    err[3 * i + 1] = e[1];
    err[3 * i + 2] = e[2];
  }
  __syncthreads();

  for (int round = 0; round < 4; ++round) {
    if (tid == 0) {
      S.T = S.T0;  // every round restarts from the initial pose (Optimizer.cc:127)
      S.n_bad = 0;
      S.any_active = 0;
    }
    __syncthreads();
    for (int i = tid; i < n; i += POSE_THREADS)
      if (level[i] == 0) S.any_active = 1;  // benign race: every writer stores 1
    __syncthreads();
    if (S.any_active) {
      // ---- SparseOptimizer::optimize(10) with OptimizationAlgorithmLevenberg ----
      for (int it = 0; it < 10; ++it) {
        const double chi = eval_chi();
        build_system();
        if (tid == 0) {
          S.current_chi = chi;
          if (it == 0) {
            double md = 0;
            for (int j = 0; j < 6; ++j) md = fmax(fabs(S.H[7 * j]), md);
            S.lambda = 1e-5 * md;
            S.ni = 2;
          }
          S.rho = 0;
          S.qmax = 0;
          S.stop_outer = 0;
        }
        __syncthreads();
        for (int trial = 0; trial < 10; ++trial) {
          if (tid == 0) {
            S.backup = S.T;
            double Hl[36];
            for (int j = 0; j < 36; ++j) Hl[j] = S.H[j];
            for (int j = 0; j < 6; ++j) Hl[7 * j] += S.lambda;
            for (int j = 0; j < 6; ++j) S.x[j] = 0;
            S.cont_inner = solve6(Hl, S.b, S.x) ? 1 : 0;  // ok2
            PoseDev Tn;
            pose_oplus(S.T, S.x, Tn);
            S.T = Tn;
          }
          __syncthreads();
          const double tchi = eval_chi();
          if (tid == 0) {
            double temp_chi = S.cont_inner ? tchi : 1.7976931348623157e308;
            double rho = S.current_chi - temp_chi;
            double scale = 0;
            for (int j = 0; j < 6; ++j) scale += S.x[j] * (S.lambda * S.x[j] + S.b[j]);
            scale += 1e-3;
            rho /= scale;
            bool finite_lambda = true;
            if (rho > 0 && isfinite(temp_chi)) {
              double alpha = 1. - pow((2 * rho - 1), 3);
              alpha = fmin(alpha, 2. / 3.);
              S.lambda *= fmax(1. / 3., alpha);
              S.ni = 2;
              S.current_chi = temp_chi;
            } else {
              S.lambda *= S.ni;
              S.ni *= 2;
              S.T = S.backup;
              finite_lambda = isfinite(S.lambda);
            }
            S.rho = rho;
            S.qmax += 1;
            S.cont_inner = (finite_lambda && rho < 0 && S.qmax < 10) ? 1 : 0;
            if (!S.cont_inner) S.stop_outer = (S.qmax == 10 || rho == 0 || !isfinite(S.lambda)) ? 1 : 0;
          }
          __syncthreads();
          if (!S.cont_inner) break;
        }
        if (S.stop_outer) break;
      }
    }
    // ---- re-classification (Optimizer.cc:132-177): mono edges, then stereo edges; counts only ----
    {
      const PoseDev T = S.T;
      int bad = 0;
      for (int i = tid; i < n; i += POSE_THREADS) {
        const bool st = !(meas[3 * i + 2] < 0);
        double e[3] = {err[3 * i], err[3 * i + 1], err[3 * i + 2]};
        if (!inlier[i]) {
          edge_error(i, T, e);
          err[3 * i] = e[0];
          err[3 * i + 1] = e[1];
          err[3 * i + 2] = e[2];
        }
        const double c = edge_chi2(i, e);
        const double th = (st ? 7.815 : 5.991) * (double)sigma2[i];
        if (c > th) {
          inlier[i] = 0;
          level[i] = 1;
          ++bad;
        } else {
          inlier[i] = 1;
          level[i] = 0;
        }
        if (round == 2) robust[i] = 0;
      }
      const double tb = block_sum((double)bad, S.red, 27);
      if (tid == 0) S.n_bad = (int)tb;
      __syncthreads();
    }
  }
  if (tid == 0) {
    for (int i = 0; i < 4; ++i) pose_out[i] = S.T.q[i];
    for (int i = 0; i < 3; ++i) pose_out[4 + i] = S.T.t[i];
    *n_good = n - S.n_bad;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same optimisation with every edge in REGISTERS (r3): 512 threads, a thread owns up to POSE_EPT = 4 edges (n <= 2048: a frame has
// at most nFeatures matched map points) -- map point, measurement, information, the last error, its flags -- for the whole call, so the
// forty-odd passes over the edges touch no memory; the 27 sums of the normal equations are reduced together (a fixed tree per wave by
// data-parallel-primitive moves, ONE barrier, the wave partials added in wave order) where the version above runs 27 block reductions of
// two barriers each; and an iteration that follows an ACCEPTED trial does not evaluate the errors again (g2o does, and gets the numbers
// the trial just produced: same pose, same arithmetic).
// The loop is written as ONE state machine with a single evaluation site, a single build site and a single solve site: unrolled over
// the register-resident edges and with fp64 divisions expanded, the straightforward nesting (evaluation inlined at three places) came to
// 12 000 instructions = 96 KB of code, more than the 64 KB instruction cache two compute units share, and the kernel ran at the speed of
// instruction fetch (10 us per iteration whatever the arithmetic did).  Thresholds, decisions and their order are those of the kernel above.
// ---------------------------------------------------------------------------------------------------------------------------------
#define POSE_EPT 4
#define POSE_RT 512
#ifdef POSE_STAMPS  // diagnostic build only (tools/exp/pose_bench.hip): cycles per section of the state machine, wave 0
__device__ long long g_pose_stamps[8];
#define PS_BEGIN long long ps_t_ = __builtin_amdgcn_s_memtime();
#define PS(k)                                                  \
  {                                                            \
    const long long now_ = __builtin_amdgcn_s_memtime();       \
    if (tid == 0) g_pose_stamps[k] += now_ - ps_t_;            \
    ps_t_ = now_;                                              \
  }
#define PS_COUNT(k) \
  if (tid == 0) g_pose_stamps[k] += 1;
#else
#define PS_BEGIN
#define PS(k)
#define PS_COUNT(k)
#endif
#define POSE_TRIALS 10  // g2o's Levenberg-Marquardt gives an iteration at most 10 trial steps (qmax)
#define POSE_STAGE_LD(NT) ((NT) + 8)  // doubles per row of the staged sums: rows 8 double-banks apart => the readers of a wave 2 per bank pair
template <int NT>
struct PoseSharedR {
  PoseDev T0;
  double H[36], b[6];
  double red[28][NT / 64];
  // every trial step an iteration can take, solved at once on POSE_TRIALS lanes when its system is built (see the build site)
  PoseDev trial_T[POSE_TRIALS];
  double trial_x[POSE_TRIALS][6];
  int trial_ok[POSE_TRIALS];
  // the 27 sums of the normal equations staged per thread (59 KB with 256 threads, 112 KB with 512: static LDS past 64 KB is fine on gfx950)
  double stage[27 * POSE_STAGE_LD(NT)];
};
template <int NT>
__global__ __launch_bounds__(NT) void k_pose_only_reg(int n, const double* __restrict__ Xw, const double* __restrict__ meas,
                                                           const double* __restrict__ info, const float* __restrict__ sigma2,
                                                           const double* __restrict__ pose_in, BaParamsDev prm, double d_mono,
                                                           double d_stereo, uint8_t* __restrict__ inlier_out,
                                                           double* __restrict__ pose_out, int32_t* __restrict__ n_good,
                                                           const int32_t* __restrict__ n_dev) {
#pragma clang fp contract(off)
  __shared__ PoseSharedR<NT> S;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (n_dev) {  // the tracking chain: the edge list was built on the device, `n` is only its upper bound; < 0: no optimisation at all
    const int nd = *n_dev;
    // (both register kernels are launched: the one whose capacity class the count falls into runs, the other returns)
    if (NT == 256 ? nd > 256 * POSE_EPT : (nd >= 0 && nd <= 256 * POSE_EPT)) return;
    if (nd < 0) {
      if (tid < 7) pose_out[tid] = pose_in[tid];
      if (tid == 0) *n_good = 0;
      return;
    }
    n = min(n, nd);
  }
  double X[POSE_EPT][3], M[POSE_EPT][3], W[POSE_EPT], E[POSE_EPT][3], TH[POSE_EPT];
  bool have[POSE_EPT], st[POSE_EPT], lvl0[POSE_EPT], rob[POSE_EPT], inl[POSE_EPT];
#pragma unroll
  for (int u = 0; u < POSE_EPT; ++u) {
    const int i = tid + u * NT;
    have[u] = i < n;
    const int ii = have[u] ? i : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) X[u][k] = (n > 0) ? Xw[3 * ii + k] : 0.0, M[u][k] = (n > 0) ? meas[3 * ii + k] : 0.0;
    W[u] = (n > 0) ? info[ii] : 0.0;
    st[u] = !(M[u][2] < 0);
    TH[u] = (st[u] ? 7.815 : 5.991) * (double)((n > 0) ? sigma2[ii] : 0.f);
    lvl0[u] = have[u], rob[u] = true, inl[u] = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) E[u][k] = 0.0;
  }
  if (tid == 0) {
    for (int i = 0; i < 4; ++i) S.T0.q[i] = pose_in[i];
    for (int i = 0; i < 3; ++i) S.T0.t[i] = pose_in[4 + i];
  }
  __syncthreads();

  auto edge_error = [&](int u, const PoseDev& T, double* e) {
    double p[3];
    quat_rotate(T.q, X[u], p);
    const double x = p[0] + T.t[0], y = p[1] + T.t[1], z = p[2] + T.t[2];
    const double uu = x / z * prm.fx + prm.cx, v = y / z * prm.fy + prm.cy;
    e[0] = M[u][0] - uu;
    e[1] = M[u][1] - v;
    e[2] = (M[u][2] < 0) ? 0.0 : M[u][2] - (uu - prm.bf / z);
  };
  auto edge_chi2 = [&](int u, const double* e) {
    double c = e[0] * (W[u] * e[0]) + e[1] * (W[u] * e[1]);
    if (st[u]) c += e[2] * (W[u] * e[2]);
    return c;
  };

  // edge->computeError() at construction (Optimizer.cc:90,111) -- (with the state machine below this is the only other use of edge_error
  // besides the evaluation site and the re-classification)
  {
    const PoseDev T = S.T0;
#pragma unroll
    for (int u = 0; u < POSE_EPT; ++u)
      if (have[u]) edge_error(u, T, E[u]);
  }

  // The Levenberg-Marquardt control runs in EVERY thread (late r4): damping, gain ratio, trial counter and the iteration's base pose are
  // registers with the same value everywhere -- every thread reads the same sums and the same trial table and executes the same
  // instructions -- so a trial is: read the trial's pose (LDS broadcast), evaluate, ONE barrier for the sum, decide.  With the control on
  // lane 0 and its state in LDS a trial took four barriers and two store-barrier-load round trips (~1.2 k of a pass's ~5 k cycles).
  PoseDev Tb = S.T0;  // the current estimate
  int n_bad = 0;
  int sum_slot = 0;  // the evaluation's per-wave partial sums alternate between two rows: one barrier per sum
  auto block_sum_alt = [&](double v) {
    const double w = wave_sum_f64(v);
    double* row = S.red[27 - sum_slot];  // (rows 26 and 27: the 512-thread build's 27 sums use rows 0 .. 26 between two barriers of their own)
    if (lane == 0) row[wv] = w;
    __syncthreads();
    double sm = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) sm += row[k];
    sum_slot ^= 1;
    return sm;
  };
  for (int round = 0; round < 4; ++round) {
    Tb = S.T0;  // every round restarts from the initial pose (Optimizer.cc:127)
    n_bad = 0;
    bool mine = false;
#pragma unroll
    for (int u = 0; u < POSE_EPT; ++u) mine = mine || lvl0[u];
    const int any_active = __syncthreads_or(mine ? 1 : 0);
    if (any_active) {
      // ---- SparseOptimizer::optimize(10) with OptimizationAlgorithmLevenberg, as a state machine ----
      int it = 0;            // iterations completed
      bool do_solve = false; // the pass starts with a trial step and its evaluation decides that trial
      double chi_cur = 0, current_chi = 0, lambda = 0, ni = 2;
      int qmax = 0;
      PoseDev Te = Tb;  // the pose the pass evaluates at
      bool trial_ok = true;
      PS_BEGIN
      for (;;) {
        PS(0)
        if (do_solve) {  // trial qmax of this iteration: solved when the system was built
          Te = S.trial_T[qmax];
          trial_ok = S.trial_ok[qmax] != 0;  // ok2
        }
        PS(1)  // trial step taken
        PS_COUNT(6)
        // computeActiveErrors + activeRobustChi2 at Te -- the ONE evaluation site
        double chi;
        {
          double part = 0;
#pragma unroll
          for (int u = 0; u < POSE_EPT; ++u) {
            if (!lvl0[u]) continue;
            edge_error(u, Te, E[u]);
            const double c = edge_chi2(u, E[u]);
            if (rob[u]) {
              const double dl = st[u] ? d_stereo : d_mono;
              const double dsqr = dl * dl;
              part += (c <= dsqr) ? c : (2 * sqrt(c) * dl - dsqr);
            } else
              part += c;
          }
          chi = block_sum_alt(part);
        }
        PS(2)  // evaluation
        if (!do_solve) {
          chi_cur = chi;  // the errors of the current estimate: an iteration starts
        } else {
          // decide the trial (every thread, the same numbers)
          double xs[6], bs[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) xs[j] = S.trial_x[qmax][j], bs[j] = S.b[j];
          const double temp_chi = trial_ok ? chi : 1.7976931348623157e308;
          double rho = current_chi - temp_chi;
          double scale = 0;
#pragma unroll
          for (int j = 0; j < 6; ++j) scale += xs[j] * (lambda * xs[j] + bs[j]);
          scale += 1e-3;
          rho /= scale;
          bool finite_lambda = true, accepted = false;
          if (rho > 0 && isfinite(temp_chi)) {
            double alpha = 1. - pow((2 * rho - 1), 3);
            alpha = fmin(alpha, 2. / 3.);
            lambda *= fmax(1. / 3., alpha);
            ni = 2;
            current_chi = temp_chi;
            accepted = true;
            Tb = Te;
          } else {
            lambda *= ni;
            ni *= 2;
            finite_lambda = isfinite(lambda);  // (the estimate stays Tb: g2o restores its backup)
          }
          qmax += 1;
          const bool cont_inner = finite_lambda && rho < 0 && qmax < 10;
          if (cont_inner) continue;  // another trial of the same iteration (new lambda, the same system)
          const bool stop_outer = qmax == 10 || rho == 0 || !isfinite(lambda);
          if (stop_outer) break;
          if (++it == 10) break;
          if (!accepted) {  // (a trial neither accepted nor retried, e.g. a NaN gain ratio: the next iteration re-evaluates at the restored pose)
            do_solve = false;
            Te = Tb;
            continue;
          }
          chi_cur = chi;  // accepted: these ARE the errors and the chi2 of the new current estimate
        }
        PS(3)  // decision
        PS_COUNT(7)
        // linearizeOplus + constructQuadraticForm of every active edge at the current estimate (errors of the last evaluation) -- the ONE build site
        {
          const PoseDev T = Tb;
          double acc[27];
#pragma unroll
          for (int k = 0; k < 27; ++k) acc[k] = 0;
#pragma unroll
          for (int u = 0; u < POSE_EPT; ++u) {
            if (!lvl0[u]) continue;
            double p[3];
            quat_rotate(T.q, X[u], p);
            const double x = p[0] + T.t[0], y = p[1] + T.t[1], z = p[2] + T.t[2];
            const double invz = 1.0 / z, invz_2 = invz * invz;
            const bool s3 = st[u];
            double J[3][6];
            J[0][0] = x * y * invz_2 * prm.fx;
            J[0][1] = -(1 + (x * x * invz_2)) * prm.fx;
            J[0][2] = y * invz * prm.fx;
            J[0][3] = -invz * prm.fx;
            J[0][4] = 0;
            J[0][5] = x * invz_2 * prm.fx;
            J[1][0] = (1 + y * y * invz_2) * prm.fy;
            J[1][1] = -x * y * invz_2 * prm.fy;
            J[1][2] = -x * invz * prm.fy;
            J[1][3] = 0;
            J[1][4] = -invz * prm.fy;
            J[1][5] = y * invz_2 * prm.fy;
            J[2][0] = s3 ? J[0][0] - prm.bf * y * invz_2 : 0.0;
            J[2][1] = s3 ? J[0][1] + prm.bf * x * invz_2 : 0.0;
            J[2][2] = s3 ? J[0][2] : 0.0;
            J[2][3] = s3 ? J[0][3] : 0.0;
            J[2][4] = 0;
            J[2][5] = s3 ? J[0][5] - prm.bf * invz_2 : 0.0;
            const double e[3] = {E[u][0], E[u][1], s3 ? E[u][2] : 0.0};
            const double w = W[u];
            double r1 = 1.0;
            if (rob[u]) {
              const double c = edge_chi2(u, e);
              const double dl = s3 ? d_stereo : d_mono;
              if (c > dl * dl) r1 = dl / sqrt(c);
            }
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              acc[21 + a] -= r1 * (J[0][a] * (w * e[0]) + J[1][a] * (w * e[1]) + J[2][a] * (w * e[2]));
#pragma unroll
              for (int c2 = a; c2 < 6; ++c2) acc[k++] += J[0][a] * (r1 * w) * J[0][c2] + J[1][a] * (r1 * w) * J[1][c2] + J[2][a] * (r1 * w) * J[2][c2];
            }
          }
          PS(5)  // build: per-edge arithmetic
          // (the previous system's b and trial table are rewritten only behind the barrier below: every thread has read them by then)
          {
            // every thread stages its 27 partial sums; NT / 32 threads per sum add 32 staged values each (ascending), a fixed tree joins
            // them: 27 + 32 LDS accesses and ~36 additions per thread, where 27 wave trees of data-parallel-primitive moves took as long as
            // the edges' arithmetic (6.7 k of a build's 13.5 k cycles with 256 threads)
            constexpr int PARTS = NT / 32, LD = POSE_STAGE_LD(NT);
#pragma unroll
            for (int k = 0; k < 27; ++k) S.stage[k * LD + tid] = acc[k];
            __syncthreads();
            if (tid < 27 * PARTS) {
              const int k = tid / PARTS, part = tid % PARTS;
              const double* row = S.stage + k * LD + part;
              double sm = 0;
#pragma unroll
              for (int j = 0; j < 32; ++j) sm += row[PARTS * j];
              sm += dpp_f64<ORBFE_DPP_ROW_SHR(1), 0xf>(sm);  // the last lane of each group of PARTS (8: half a row of 16, 16: a row) ends up with the group's sum
              sm += dpp_f64<ORBFE_DPP_ROW_SHR(2), 0xf>(sm);
              sm += dpp_f64<ORBFE_DPP_ROW_SHR(4), 0xf>(sm);
              if (PARTS == 16) sm += dpp_f64<ORBFE_DPP_ROW_SHR(8), 0xf>(sm);
              if (part == PARTS - 1) {
                if (k >= 21) {
                  S.b[k - 21] = sm;
                } else {  // k -> (a, c2), a <= c2, in the order the sums were laid out
                  int a = 0, rem = k;
                  while (rem >= 6 - a) rem -= 6 - a, ++a;
                  const int c2 = a + rem;
                  S.H[6 * a + c2] = sm;
                  S.H[6 * c2 + a] = sm;
                }
              }
            }
            __syncthreads();
          }
        }
        PS(4)  // build
        current_chi = chi_cur;
        if (it == 0) {
          double md = 0;
#pragma unroll
          for (int j = 0; j < 6; ++j) md = fmax(fabs(S.H[7 * j]), md);
          lambda = 1e-5 * md;
          ni = 2;
        }
        qmax = 0;
        // The trial steps of this iteration, all at once: a rejected trial multiplies lambda by ni and doubles ni, the system and the pose
        // it starts from stay -- so trial q's damping is known now, and lane q solves (H + lambda_q I) x = b and applies x to the pose
        // beside the others (one instruction stream; a call of 28 iterations takes ~59 trials, most of the rejected ones in the runs of ten
        // that end a converged round).  Same recurrence, same solve, same update as taking them one by one.
        if (tid < POSE_TRIALS) {
          double lam = lambda, nu = ni;
          for (int q = 0; q < tid; ++q) lam *= nu, nu *= 2;
          double xs[6] = {0, 0, 0, 0, 0, 0};
          const bool ok = solve6_lambda(S.H, lam, S.b, xs);
          PoseDev Tn;
          pose_oplus(Tb, xs, Tn);
          S.trial_T[tid] = Tn;
          for (int j = 0; j < 6; ++j) S.trial_x[tid][j] = xs[j];
          S.trial_ok[tid] = ok ? 1 : 0;
        }
        __syncthreads();
        PS(1)  // solve + oplus of the iteration's trials
        do_solve = true;
      }
    }
    // ---- re-classification (Optimizer.cc:132-177): mono edges, then stereo edges; counts only ----
    {
      int bad = 0;
#pragma unroll
      for (int u = 0; u < POSE_EPT; ++u) {
        if (!have[u]) continue;
        if (!inl[u]) edge_error(u, Tb, E[u]);
        const double c = edge_chi2(u, E[u]);
        if (c > TH[u]) {
          inl[u] = false;
          lvl0[u] = false;
          ++bad;
        } else {
          inl[u] = true;
          lvl0[u] = true;
        }
        if (round == 2) rob[u] = false;
      }
      n_bad = (int)block_sum_alt((double)bad);
    }
  }
#pragma unroll
  for (int u = 0; u < POSE_EPT; ++u)
    if (have[u]) inlier_out[tid + u * NT] = inl[u] ? 1 : 0;
  if (tid == 0) {
    for (int i = 0; i < 4; ++i) pose_out[i] = Tb.q[i];
    for (int i = 0; i < 3; ++i) pose_out[4 + i] = Tb.t[i];
    *n_good = n - n_bad;
  }
}

void launch_pose_only(hipStream_t s, int n, const double* Xw, const double* meas, const double* info, const float* sigma2,
                      const double* pose_in, BaParamsDev prm, double d_mono, double d_stereo, double* err, uint8_t* level,
                      uint8_t* robust, uint8_t* inlier, double* pose_out, int32_t* n_good, const int32_t* n_dev) {
  // n_dev (nullable, the tracking chain): the edge count in device memory, n its upper bound (register kernels only)
  // (past POSE_RT * POSE_EPT edges: round 2's kernel, which re-reads the edges from memory every pass)
  // 256 threads up to 1024 edges (one wave per SIMD: the 27 wave reductions of a build are issued once per SIMD, not twice), 512 up to 2048
  if (n_dev && n > 256 * POSE_EPT && n <= POSE_RT * POSE_EPT)  // the count is on the device: the 256-thread kernel takes it if it fits (20 % faster there), else the next launch does
    hipLaunchKernelGGL(k_pose_only_reg<256>, dim3(1), dim3(256), 0, s, 256 * POSE_EPT, Xw, meas, info, sigma2, pose_in, prm, d_mono, d_stereo, inlier,
                       pose_out, n_good, n_dev);
  if (n <= 256 * POSE_EPT)
    hipLaunchKernelGGL(k_pose_only_reg<256>, dim3(1), dim3(256), 0, s, n, Xw, meas, info, sigma2, pose_in, prm, d_mono, d_stereo, inlier, pose_out,
                       n_good, n_dev);
  else if (n <= POSE_RT * POSE_EPT)
    hipLaunchKernelGGL(k_pose_only_reg<POSE_RT>, dim3(1), dim3(POSE_RT), 0, s, n, Xw, meas, info, sigma2, pose_in, prm, d_mono, d_stereo, inlier, pose_out,
                       n_good, n_dev);
  else
    hipLaunchKernelGGL(k_pose_only, dim3(1), dim3(POSE_THREADS), 0, s, n, Xw, meas, info, sigma2, pose_in, prm, d_mono, d_stereo, err,
                       level, robust, inlier, pose_out, n_good);
}

}  // namespace orbfe
