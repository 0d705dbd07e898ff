// k_glue.hip -- the frame-level steps either side of the ORB path.
//
// Replaces cv::cvtColor(COLOR_RGB2GRAY / COLOR_BGR2GRAY) of Tracking::grabFrame (src/ORB_SLAM2/src/Tracking.cc:55-68),
// Camera::undistortPoints (src/Camera.cc:29-39, cv::undistortPoints with P = K: 5 fixed iterations in fp64) and the RGB-D tail
// of Frame::Frame (src/Frame.cc:136-158: depth lookup at the distorted keypoint with truncated indices, rightU = x_u - bf / d).
// Byte arithmetic and fp64 without FMA contraction: bit-identical to an un-contracted x86-64 build of the same formulas.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"

namespace orbfe {

// 4 pixels (12 source bytes as 3 aligned words) -> one destination word.  order 1: RGB, 2: BGR.
__global__ __launch_bounds__(256) void k_cvt_gray(const uint8_t* __restrict__ src, size_t src_stride, uint8_t* __restrict__ dst,
                                                  int dst_stride, int w, int order, int variant) {
  const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int y = blockIdx.y;
  if (x4 >= w) return;
  const uint32_t* s = (const uint32_t*)(src + (size_t)y * src_stride + 3 * (size_t)x4);  // rows and 12-byte groups are 4-aligned
  const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];
  const uint32_t c[12] = {w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24, w1 & 255u, (w1 >> 8) & 255u,
                          (w1 >> 16) & 255u, w1 >> 24, w2 & 255u, (w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24};
  // variant 0: 14-bit coefficients 4899 / 9617 / 1868, variant 1: 15-bit 9798 / 19235 / 3735 (un-vendored OpenCV: a selectable decision)
  const uint32_t cr = variant ? 9798u : 4899u, cg = variant ? 19235u : 9617u, cb = variant ? 3735u : 1868u;
  const uint32_t shift = variant ? 15u : 14u, half = 1u << (shift - 1);
  const uint32_t c0 = order == 1 ? cr : cb, c2 = order == 1 ? cb : cr;  // R2Y / B2Y on the first / third channel
  uint32_t out = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t g = (c[3 * j] * c0 + c[3 * j + 1] * cg + c[3 * j + 2] * c2 + half) >> shift;
    out |= g << (8 * j);
  }
  *(uint32_t*)(dst + (size_t)y * dst_stride + x4) = out;  // the row padding absorbs the tail
}

// one lane per keypoint of the slot: depth lookup at the distorted position, undistortion in place, rightU
__global__ __launch_bounds__(256) void k_frame_rgbd(orbfe_keypoint* __restrict__ kps, const int32_t* __restrict__ n_kp_ptr, int n_features,
                                                    orbfe_camera cam, const uint8_t* __restrict__ depth, int depth_type, size_t depth_stride,
                                                    float depth_scale, double* __restrict__ depth_out, double* __restrict__ right_u_out,
                                                    orbfe_keypoint* __restrict__ kps_host) {
  // kps_host (nullable, orbfe_frame_rgbd_image): the undistorted keypoint records written to page-locked host memory as well; depth /
  // depth_out / right_u_out may then be page-locked host pointers too (a thousand 2-byte reads and two 8 KB arrays over PCIe)
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_features) return;
  const int n = *n_kp_ptr;
  if (i >= n) {
    depth_out[i] = -1.0;
    right_u_out[i] = -1.0;
    return;
  }
  const float px = kps[i].x, py = kps[i].y;
  float xu = px, yu = py;
  if (cam.k1 != 0.0f) {  // Camera.cc:31
    const double fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k0 = cam.k1, k1 = cam.k2, k2 = cam.p1, k3 = cam.p2, k4 = cam.k3;
    double x = px, y = py;
    const double u = x, v = y;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
      if (icdist < 0) {
        x = (u - cx) * ifx;
        y = (v - cy) * ify;
        break;
      }
      const double dx = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
      const double dy = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0 * r2 + 0 * r2 * r2;
      x = (x0 - dx) * icdist;
      y = (y0 - dy) * icdist;
    }
    const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
    xu = (float)(xx * ww);
    yu = (float)(yy * ww);
    kps[i].x = xu;
    kps[i].y = yu;
  }
  double d_out = -1.0, ru_out = -1.0;
  if (depth) {
    const int ix = (int)px, iy = (int)py;  // Mat::at<float>(float, float): truncated indices (quirk Q10)
    const uint8_t* row = depth + (size_t)iy * depth_stride;
    const float raw = depth_type == 0 ? (float)((const uint16_t*)row)[ix] : ((const float*)row)[ix];
    const float d = raw / depth_scale;
    if (d > 0) {
      d_out = (double)d;
      ru_out = (double)(xu - cam.bf / d);
    }
  }
  depth_out[i] = d_out;
  right_u_out[i] = ru_out;
  if (kps_host) {
    orbfe_keypoint k = kps[i];
    k.x = xu, k.y = yu;
    kps_host[i] = k;
  }
}

// Frame records for the sequence-level gather: what Frame::createStereo leaves behind for the tracker, one fixed-size
// record per stereo pair, packed from the stream's result buffer:
//   int32 n_keypoints | int32 n_matches | 8 bytes pad | left keypoints [NF x 28 B] | left descriptors [NF x 32 B] | right_u [NF] f64 | depth [NF] f64
// Entries past the keypoint count are zeroed, so a record depends on nothing but its frame.  One workgroup per pair, 4-byte units
// (every section starts on a multiple of 4: 16, 16 + 28 NF, ...); a pure copy, HBM-bound.
__global__ __launch_bounds__(256) void k_pack_records(const uint8_t* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                      const int32_t* __restrict__ counts, const uint8_t* __restrict__ right_u,
                                                      const uint8_t* __restrict__ depth, const int32_t* __restrict__ n_match, int nf,
                                                      uint32_t* __restrict__ out, size_t rec_words) {
  const int p = blockIdx.x;
  const int n = min(max(counts[2 * p], 0), nf);  // the LEFT image of pair p sits in slot 2p
  uint32_t* rec = out + (size_t)p * rec_words;
  if (threadIdx.x < 4) rec[threadIdx.x] = threadIdx.x == 0 ? (uint32_t)n : (threadIdx.x == 1 ? (uint32_t)n_match[p] : 0u);
  const uint32_t* src[4] = {(const uint32_t*)(kps + (size_t)(2 * p) * nf * 28), (const uint32_t*)(desc + (size_t)(2 * p) * nf * 32),
                            (const uint32_t*)(right_u + (size_t)p * nf * 8), (const uint32_t*)(depth + (size_t)p * nf * 8)};
  const int wpe[4] = {7, 8, 2, 2};  // words per entry
  size_t o = 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int live = n * wpe[s], total = nf * wpe[s];
    for (int i = threadIdx.x; i < total; i += 256) rec[o + i] = i < live ? src[s][i] : 0u;
    o += (size_t)total;
  }
}

void launch_pack_records(hipStream_t s, const uint8_t* d_kps, const uint8_t* d_desc, const int32_t* d_counts, const uint8_t* d_ru,
                         const uint8_t* d_dp, const int32_t* d_nm, int nf, int n_pairs, void* d_out) {
  if (n_pairs <= 0) return;
  const size_t rec_words = 4 + (size_t)nf * (7 + 8 + 2 + 2);
  hipLaunchKernelGGL(k_pack_records, dim3(n_pairs), dim3(256), 0, s, d_kps, d_desc, d_counts, d_ru, d_dp, d_nm, nf, (uint32_t*)d_out, rec_words);
}

void launch_cvt_gray(hipStream_t s, const uint8_t* d_src, size_t src_stride, uint8_t* d_dst, int dst_stride, int w, int h, int order,
                     int variant) {
  hipLaunchKernelGGL(k_cvt_gray, dim3((w + 1023) / 1024, h), dim3(256), 0, s, d_src, src_stride, d_dst, dst_stride, w, order, variant);
}
void launch_frame_rgbd(hipStream_t s, orbfe_keypoint* d_kps, const int32_t* d_n_kp, int n_features, const orbfe_camera& cam,
                       const uint8_t* d_depth, int depth_type, size_t depth_stride, float depth_scale, double* d_depth_out, double* d_right_u,
                       orbfe_keypoint* h_kps) {
  hipLaunchKernelGGL(k_frame_rgbd, dim3((n_features + 255) / 256), dim3(256), 0, s, d_kps, d_n_kp, n_features, cam, d_depth, depth_type,
                     depth_stride, depth_scale, d_depth_out, d_right_u, h_kps);
}

}  // namespace orbfe
