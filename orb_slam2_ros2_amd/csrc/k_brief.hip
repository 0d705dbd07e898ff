// k_brief.hip -- intensity-centroid orientation + rotated BRIEF-256 + keypoint assembly, one wavefront
// per keypoint.
//
// Replaces ORBExtractor::computeBRIEF (src/ORB_SLAM2/src/ORBExtractor.cc:397-456), getGrayCentroid
// (:465-487), rotateTemplate (:534-540) and the level concatenation of ORBExtractor::extract (:499-508).
//
//  * IC moments: the 749-pixel disc (umax table of initMaxU, :217-236) is summed in int32 across the 64
//    lanes (two patch rows per step) on the UN-blurred plane, then wave-reduced.
//  * theta = atan2(m01, m10), cos/sin: fp64, evaluated with the shared deterministic routines of
//    orb_math.h (bit-identical on host and device; <= 1 ulp from libm).
//  * 256 tests: lane l evaluates pairs l, l+64, l+128, l+192; the rotated offsets are rounded exactly as
//    the reference does (double product -> float, float add, round-half-even); one __ballot per group of
//    64 tests yields 8 descriptor bytes already in the reference's LSB-first order.
#include <hip/hip_runtime.h>

#include "orb_math.h"
#include "orbfe_internal.h"

namespace orbfe {

struct UmaxTab {
  int8_t u[16];
};

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void k_orient_brief(const LevelDev* __restrict__ lv, int n_levels,
                                                      const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                      size_t img_pitch, const uint32_t* __restrict__ sel,
                                                      const int32_t* __restrict__ sel_count, int n_features,
                                                      const int8_t* __restrict__ pattern, UmaxTab umax,
                                                      orbfe_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                      KpAux* __restrict__ aux, int32_t* __restrict__ n_kp, double* __restrict__ theta_out,
                                                      int rows0) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);  // output keypoint index inside the image
  const int img = blockIdx.y;
  // level-major concatenation (ORBExtractor.cc:501-506): find the level that owns output index k
  const int32_t* sc = sel_count + (size_t)img * n_levels;
  int level = -1, j = 0, acc = 0;
  for (int l = 0; l < n_levels; ++l) {
    const int c = sc[l];
    if (level < 0 && k < acc + c) {
      level = l;
      j = k - acc;
    }
    acc += c;
  }
  if (k == 0 && lane == 0) n_kp[img] = acc;
  if (level < 0) return;  // wave-uniform
  const LevelDev& L = lv[level];
  const uint32_t rec = sel[(size_t)img * n_features + L.quota_off + j];
  const int x = (int)ORBFE_REC_X(rec) + ORBFE_EDGE, y = (int)ORBFE_REC_Y(rec) + ORBFE_EDGE;  // level coordinates
  const int resp = (int)ORBFE_REC_R(rec);
  const uint8_t* I = pyr + (size_t)img * img_pitch + L.plane_off;
  const uint8_t* W = blur + (size_t)img * img_pitch + L.plane_off;
  const int stride = L.stride;

  // ---- intensity centroid (ORBExtractor.cc:465-487): m10 = sum dx*I, m01 = sum dy*I over the disc ----
  int m10 = 0, m01 = 0;
  {
    const int half = lane >> 5;      // two rows per step
    const int dx = (lane & 31) - 15;  // -15..16 (16 unused)
    for (int r = 0; r < 16; ++r) {
      const int dy = -15 + 2 * r + half;  // -15 .. 16
      if (dy <= 15) {
        const int ady = dy < 0 ? -dy : dy;
        const int d = umax.u[ady];
        if (dx >= -d && dx <= d) {
          const int v = I[(size_t)(y + dy) * stride + (x + dx)];
          m10 += dx * v;
          m01 += dy * v;
        }
      }
    }
  }
  m10 = wave_sum_i(m10);
  m01 = wave_sum_i(m01);
  const double theta = orbmath::det_atan2((double)m01, (double)m10);
  double sn, cs;
  orbmath::det_sincos(theta, &sn, &cs);

  // ---- rotated BRIEF on the blurred plane (ORBExtractor.cc:439-454) ----
  const float px = (float)x, py = (float)y;
  unsigned long long bits[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int pair = g * 64 + lane;
    const int8_t* t = pattern + pair * 4;
    const float x1 = (float)t[0], y1 = (float)t[1], x2 = (float)t[2], y2 = (float)t[3];
    // float * double -> double, one rounding to float (rotateTemplate, :537-538)
    const float p1x = (float)((double)x1 * cs - (double)y1 * sn);
    const float p1y = (float)((double)x1 * sn + (double)y1 * cs);
    const float p2x = (float)((double)x2 * cs - (double)y2 * sn);
    const float p2y = (float)((double)x2 * sn + (double)y2 * cs);
    const int r1 = __float2int_rn(py + p1y), c1 = __float2int_rn(px + p1x);
    const int r2 = __float2int_rn(py + p2y), c2 = __float2int_rn(px + p2x);
    const int v1 = W[(size_t)r1 * stride + c1];
    const int v2 = W[(size_t)r2 * stride + c2];
    bits[g] = __ballot(v1 < v2);
  }
  if (lane < 4) {
    unsigned long long* d64 = (unsigned long long*)(desc + ((size_t)img * n_features + k) * 32);
    unsigned long long b = bits[0];
    if (lane == 1) b = bits[1];
    if (lane == 2) b = bits[2];
    if (lane == 3) b = bits[3];
    d64[lane] = b;
  }
  if (lane == 0) {
    orbfe_keypoint kp;
    kp.x = px * L.sf;  // keypoint.pt *= scale[octave] (ORBExtractor.cc:408-409)
    kp.y = py * L.sf;
    kp.size = 7.0f;
    kp.angle = (float)(theta / 3.14159265358979323846 * 180);  // ORBExtractor.cc:407
    kp.response = (float)resp;
    kp.octave = level;
    kp.class_id = -1;
    kps[(size_t)img * n_features + k] = kp;
    // createRowIndexDB band (ORBMatcher.cc:924-927), stored with the keypoint for the stereo matcher
    const float r = (float)(2.0 * (double)L.sf);
    const unsigned row = (unsigned)__float2int_rn(kp.y);
    const int max_row = min(rows0, __float2int_rn((float)row + r + 1.0f));
    const int min_row = max(0, __float2int_rn((float)row - r));
    KpAux a;
    a.row_min = (int16_t)min_row;
    a.row_max = (int16_t)max_row;
    aux[(size_t)img * n_features + k] = a;
    if (theta_out) theta_out[(size_t)img * n_features + k] = theta;
  }
}

void launch_orient_brief(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint8_t* d_pyr, const uint8_t* d_blur,
                         size_t img_pitch, const uint32_t* d_sel, const int32_t* d_sel_count, int n_features,
                         const int8_t* d_pattern, const int umax[16], orbfe_keypoint* d_kps, uint8_t* d_desc, KpAux* d_aux,
                         int32_t* d_n_kp, double* d_theta, int rows0, int n_img) {
  if (n_img <= 0 || n_features <= 0) return;
  UmaxTab u;
  for (int i = 0; i < 16; ++i) u.u[i] = (int8_t)umax[i];
  hipLaunchKernelGGL(k_orient_brief, dim3((n_features + 3) / 4, n_img), dim3(256), 0, s, d_lv, n_levels, d_pyr, d_blur,
                     img_pitch, d_sel, d_sel_count, n_features, d_pattern, u, d_kps, d_desc, d_aux, d_n_kp, d_theta, rows0);
}

}  // namespace orbfe
