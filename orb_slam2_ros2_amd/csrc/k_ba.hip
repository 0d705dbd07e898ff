// k_ba.hip -- local-BA reprojection error / Jacobian / Huber evaluation, one lane per edge, fp64.
//
// Replaces the per-edge arithmetic g2o runs for the graph Optimizer::OptimizeLocalMap builds
// (src/ORB_SLAM2/src/Optimizer.cc:296-330): EdgeStereoSE3ProjectXYZ / EdgeSE3ProjectXYZ computeError,
// linearizeOplus, chi2, isDepthPositive, and RobustKernelHuber::robustify (g2o 20241228 types_sba,
// core/robust_kernel_impl).  Pose vertex = SE3Quat (qx,qy,qz,qw,tx,ty,tz), map(X) = q*X + t.
// HBM-bound gather/scatter: 304 B per edge + the vertices; nothing here is a dense
// contraction, so no MFMA.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"

namespace orbfe {

__global__ __launch_bounds__(256) void k_ba_edges(int n_edges, const double* __restrict__ poses, const double* __restrict__ points,
                                                  const int32_t* __restrict__ edge_pose, const int32_t* __restrict__ edge_point,
                                                  const double* __restrict__ meas, const uint8_t* __restrict__ is_stereo,
                                                  const double* __restrict__ info, const double* __restrict__ delta, BaParamsDev prm,
                                                  double* __restrict__ error, double* __restrict__ chi2, double* __restrict__ rho,
                                                  double* __restrict__ jpoint, double* __restrict__ jpose,
                                                  uint8_t* __restrict__ depth_pos) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  const double* T = poses + (size_t)edge_pose[e] * 7;
  const double* X = points + (size_t)edge_point[e] * 3;
  const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
  const double X0 = X[0], X1 = X[1], X2 = X[2];
  // Eigen quaternion * vector: uv = 2 * (q.vec x v);  v + w*uv + q.vec x uv
  double uvx = qy * X2 - qz * X1, uvy = qz * X0 - qx * X2, uvz = qx * X1 - qy * X0;
  uvx += uvx;
  uvy += uvy;
  uvz += uvz;
  const double x = X0 + qw * uvx + (qy * uvz - qz * uvy) + T[4];
  const double y = X1 + qw * uvy + (qz * uvx - qx * uvz) + T[5];
  const double z = X2 + qw * uvz + (qx * uvy - qy * uvx) + T[6];
  const bool st = is_stereo[e] != 0;
  const double fx = prm.fx, fy = prm.fy, cx = prm.cx, cy = prm.cy, bf = prm.bf;
  const double* m = meas + (size_t)e * 3;
  const double u = x / z * fx + cx;
  const double v = y / z * fy + cy;
  const double e0 = m[0] - u, e1 = m[1] - v;
  const double e2 = st ? (m[2] - (u - bf / z)) : 0.0;
  error[(size_t)e * 3 + 0] = e0;
  error[(size_t)e * 3 + 1] = e1;
  error[(size_t)e * 3 + 2] = e2;
  const double w = info[e];
  // e^T (w I) e the way Eigen evaluates it: dot(e, (w*I)*e)
  const double c2 = st ? (e0 * (w * e0) + e1 * (w * e1) + e2 * (w * e2)) : (e0 * (w * e0) + e1 * (w * e1));
  chi2[e] = c2;
  // RobustKernelHuber::robustify (delta <= 0: no kernel => rho = (chi2, 1))
  const double dl = delta[e];
  double r0 = c2, r1 = 1.0;
  if (dl > 0.0) {
    const double dsqr = dl * dl;
    if (c2 > dsqr) {
      const double sq = sqrt(c2);
      r0 = 2 * sq * dl - dsqr;
      r1 = dl / sq;
    }
  }
  rho[(size_t)e * 2 + 0] = r0;
  rho[(size_t)e * 2 + 1] = r1;
  if (depth_pos) depth_pos[e] = z > 0.0;
  if (!jpoint && !jpose) return;
  const double z_2 = z * z;
  if (jpoint) {
    // rotation matrix of the unit quaternion (Eigen toRotationMatrix)
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    const double R00 = 1 - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
    const double R10 = txy + twz, R11 = 1 - (txx + tzz), R12 = tyz - twx;
    const double R20 = txz - twy, R21 = tyz + twx, R22 = 1 - (txx + tyy);
    double* J = jpoint + (size_t)e * 9;
    if (st) {  // EdgeStereoSE3ProjectXYZ::linearizeOplus
      J[0] = -fx * R00 / z + fx * x * R20 / z_2;
      J[1] = -fx * R01 / z + fx * x * R21 / z_2;
      J[2] = -fx * R02 / z + fx * x * R22 / z_2;
      J[3] = -fy * R10 / z + fy * y * R20 / z_2;
      J[4] = -fy * R11 / z + fy * y * R21 / z_2;
      J[5] = -fy * R12 / z + fy * y * R22 / z_2;
      J[6] = J[0] - bf * R20 / z_2;
      J[7] = J[1] - bf * R21 / z_2;
      J[8] = J[2] - bf * R22 / z_2;
    } else {  // EdgeSE3ProjectXYZ: -1/z * tmp * R, tmp = [[fx,0,-x/z*fx],[0,fy,-y/z*fy]]
      const double t02 = -x / z * fx, t12 = -y / z * fy, s = -1. / z;
      J[0] = s * (fx * R00 + t02 * R20);
      J[1] = s * (fx * R01 + t02 * R21);
      J[2] = s * (fx * R02 + t02 * R22);
      J[3] = s * (fy * R10 + t12 * R20);
      J[4] = s * (fy * R11 + t12 * R21);
      J[5] = s * (fy * R12 + t12 * R22);
      J[6] = J[7] = J[8] = 0.0;
    }
  }
  if (jpose) {
    double* J = jpose + (size_t)e * 18;
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
    if (st) {
      J[12] = J[0] - bf * y / z_2;
      J[13] = J[1] + bf * x / z_2;
      J[14] = J[2];
      J[15] = J[3];
      J[16] = 0;
      J[17] = J[5] - bf / z_2;
    } else {
      J[12] = J[13] = J[14] = J[15] = J[16] = J[17] = 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Normal-equation build (g2o BaseBinaryEdge::constructQuadraticForm + BlockSolver_6_3 layout):
//   H_ll(point) += A^T W A,  b_l -= A^T W e      one lane per point, its edges in ascending edge order
//   H_pp(pose)  += B^T W B,  b_p -= B^T W e      one wave per free pose, lanes stride over its edges, fixed-order reduce
//   H_pl(edge)   = B^T W A                       one lane per edge
// with W = rho'(chi2) * info * I.  Segmented sums over host-built CSR lists: no atomics, run-to-run deterministic.
// A scatter-add of 6x6 / 6x3 / 3x3 blocks keyed by vertex id is not a dense contraction, hence no MFMA.
// ---------------------------------------------------------------------------------------------
struct EdgeTerms {
  double e[3], w, A[9], B[18];
  int rows;
};

__device__ __forceinline__ void edge_terms(int e, const double* __restrict__ poses, const double* __restrict__ points,
                                           const int32_t* __restrict__ edge_pose, const int32_t* __restrict__ edge_point,
                                           const double* __restrict__ meas, const uint8_t* __restrict__ is_stereo,
                                           const double* __restrict__ info, const double* __restrict__ delta, const BaParamsDev& prm,
                                           EdgeTerms& t) {
#pragma clang fp contract(off)
  const double* T = poses + (size_t)edge_pose[e] * 7;
  const double* X = points + (size_t)edge_point[e] * 3;
  const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
  const double X0 = X[0], X1 = X[1], X2 = X[2];
  double uvx = qy * X2 - qz * X1, uvy = qz * X0 - qx * X2, uvz = qx * X1 - qy * X0;
  uvx += uvx;
  uvy += uvy;
  uvz += uvz;
  const double x = X0 + qw * uvx + (qy * uvz - qz * uvy) + T[4];
  const double y = X1 + qw * uvy + (qz * uvx - qx * uvz) + T[5];
  const double z = X2 + qw * uvz + (qx * uvy - qy * uvx) + T[6];
  const bool st = is_stereo[e] != 0;
  t.rows = st ? 3 : 2;
  const double fx = prm.fx, fy = prm.fy, cx = prm.cx, cy = prm.cy, bf = prm.bf;
  const double* m = meas + (size_t)e * 3;
  const double u = x / z * fx + cx, v = y / z * fy + cy;
  t.e[0] = m[0] - u;
  t.e[1] = m[1] - v;
  t.e[2] = st ? (m[2] - (u - bf / z)) : 0.0;
  const double wi = info[e];
  const double c2 = st ? (t.e[0] * (wi * t.e[0]) + t.e[1] * (wi * t.e[1]) + t.e[2] * (wi * t.e[2])) : (t.e[0] * (wi * t.e[0]) + t.e[1] * (wi * t.e[1]));
  double r1 = 1.0;
  const double dl = delta[e];
  if (dl > 0.0 && c2 > dl * dl) r1 = dl / sqrt(c2);
  t.w = r1 * wi;
  const double z_2 = z * z;
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  double* J = t.A;
  if (st) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      J[k] = -fx * R[k] / z + fx * x * R[6 + k] / z_2;
      J[3 + k] = -fy * R[3 + k] / z + fy * y * R[6 + k] / z_2;
      J[6 + k] = J[k] - bf * R[6 + k] / z_2;
    }
  } else {
    const double t02 = -x / z * fx, t12 = -y / z * fy, s = -1. / z;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      J[k] = (s * fx) * R[k] + (s * t02) * R[6 + k];
      J[3 + k] = (s * fy) * R[3 + k] + (s * t12) * R[6 + k];
      J[6 + k] = 0.0;
    }
  }
  double* Bm = t.B;
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
  if (st) {
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
}

#define BA_ARGS                                                                                                                   \
  const double *__restrict__ poses, const double *__restrict__ points, const int32_t *__restrict__ edge_pose,                      \
      const int32_t *__restrict__ edge_point, const double *__restrict__ meas, const uint8_t *__restrict__ is_stereo,              \
      const double *__restrict__ info, const double *__restrict__ delta, BaParamsDev prm
#define BA_PASS poses, points, edge_pose, edge_point, meas, is_stereo, info, delta, prm

__global__ __launch_bounds__(256) void k_ba_point_blocks(int n_points, BA_ARGS, const int32_t* __restrict__ pt_off,
                                                         const int32_t* __restrict__ pt_edges, double* __restrict__ Hll,
                                                         double* __restrict__ bl) {
#pragma clang fp contract(off)
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_points) return;
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  for (int i = pt_off[p]; i < pt_off[p + 1]; ++i) {
    EdgeTerms t;
    edge_terms(pt_edges[i], BA_PASS, t);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double s = 0;
      for (int r = 0; r < t.rows; ++r) s += t.A[3 * r + a] * (t.w * t.e[r]);
      b[a] -= s;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double h = 0;
        for (int r = 0; r < t.rows; ++r) h += t.A[3 * r + a] * t.w * t.A[3 * r + c];
        H[3 * a + c] += h;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) Hll[(size_t)p * 9 + k] = H[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) bl[(size_t)p * 3 + k] = b[k];
}

__global__ __launch_bounds__(64) void k_ba_pose_blocks(int n_poses, BA_ARGS, const uint8_t* __restrict__ pose_fixed,
                                                       const int32_t* __restrict__ ps_off, const int32_t* __restrict__ ps_edges,
                                                       double* __restrict__ Hpp, double* __restrict__ bp) {
#pragma clang fp contract(off)
  const int k = blockIdx.x;
  const int lane = threadIdx.x;
  double acc[42];
#pragma unroll
  for (int i = 0; i < 42; ++i) acc[i] = 0.0;
  if (!(pose_fixed && pose_fixed[k])) {
    for (int i = ps_off[k] + lane; i < ps_off[k + 1]; i += 64) {
      EdgeTerms t;
      edge_terms(ps_edges[i], BA_PASS, t);
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double s = 0;
        for (int r = 0; r < t.rows; ++r) s += t.B[6 * r + a] * (t.w * t.e[r]);
        acc[36 + a] -= s;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double h = 0;
          for (int r = 0; r < t.rows; ++r) h += t.B[6 * r + a] * t.w * t.B[6 * r + c];
          acc[6 * a + c] += h;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 42; ++i) {
    double v = acc[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);  // fixed butterfly order: deterministic
    acc[i] = v;
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 36; ++i) Hpp[(size_t)k * 36 + i] = acc[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) bp[(size_t)k * 6 + i] = acc[36 + i];
  }
}

__global__ __launch_bounds__(256) void k_ba_edge_blocks(int n_edges, BA_ARGS, const uint8_t* __restrict__ pose_fixed,
                                                        double* __restrict__ Hpl) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  double* out = Hpl + (size_t)e * 18;
  if (pose_fixed && pose_fixed[edge_pose[e]]) {
#pragma unroll
    for (int i = 0; i < 18; ++i) out[i] = 0.0;
    return;
  }
  EdgeTerms t;
  edge_terms(e, BA_PASS, t);
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double h = 0;
      for (int r = 0; r < t.rows; ++r) h += t.B[6 * r + a] * t.w * t.A[3 * r + c];
      out[3 * a + c] = h;
    }
}

void launch_ba_system(hipStream_t s, int n_poses, int n_points, int n_edges, const double* poses, const double* points,
                      const int32_t* edge_pose, const int32_t* edge_point, const double* meas, const uint8_t* is_stereo,
                      const double* info, const double* delta, BaParamsDev prm, const uint8_t* pose_fixed, const int32_t* pt_off,
                      const int32_t* pt_edges, const int32_t* ps_off, const int32_t* ps_edges, double* Hpp, double* bp, double* Hll,
                      double* bl, double* Hpl) {
  if (n_points > 0)
    hipLaunchKernelGGL(k_ba_point_blocks, dim3((n_points + 255) / 256), dim3(256), 0, s, n_points, BA_PASS, pt_off, pt_edges, Hll, bl);
  if (n_poses > 0)
    hipLaunchKernelGGL(k_ba_pose_blocks, dim3(n_poses), dim3(64), 0, s, n_poses, BA_PASS, pose_fixed, ps_off, ps_edges, Hpp, bp);
  if (n_edges > 0 && Hpl)
    hipLaunchKernelGGL(k_ba_edge_blocks, dim3((n_edges + 255) / 256), dim3(256), 0, s, n_edges, BA_PASS, pose_fixed, Hpl);
}

void launch_ba_edges(hipStream_t s, int n_edges, const double* d_poses, const double* d_points, const int32_t* d_edge_pose,
                     const int32_t* d_edge_point, const double* d_meas, const uint8_t* d_is_stereo, const double* d_info,
                     const double* d_delta, BaParamsDev prm, double* d_error, double* d_chi2, double* d_rho, double* d_jpoint,
                     double* d_jpose, uint8_t* d_depth_pos) {
  if (n_edges <= 0) return;
  hipLaunchKernelGGL(k_ba_edges, dim3((n_edges + 255) / 256), dim3(256), 0, s, n_edges, d_poses, d_points, d_edge_pose,
                     d_edge_point, d_meas, d_is_stereo, d_info, d_delta, prm, d_error, d_chi2, d_rho, d_jpoint, d_jpose, d_depth_pos);
}

}  // namespace orbfe
