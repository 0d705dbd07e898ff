// glue_oracle.cpp -- CPU ORACLE for the frame-level steps either side of the ORB path (test infrastructure, NOT product code).
//
// PARITY UNPINNED (see orb_oracle.cpp): OpenCV is not vendored in the reference and not installed here.  Restates
//   cv::cvtColor(COLOR_RGB2GRAY / COLOR_BGR2GRAY) on CV_8UC3 as called by Tracking::grabFrame (src/ORB_SLAM2/src/Tracking.cc:55-68):
//       OpenCV 4.x color_rgb.simd.hpp RGB2Gray<uchar>: 14-bit fixed point, R2Y = 4899, G2Y = 9617, B2Y = 1868,
//       gray = (R*4899 + G*9617 + B*1868 + (1 << 13)) >> 14;
//   cv::undistortPoints(pts, pts, K, D, noArray(), K) as called by Camera::undistortPoints (src/Camera.cc:29-39):
//       OpenCV 4.5 undistort.dispatch.cpp cvUndistortPointsInternal with the default criteria (5 fixed iterations, no epsilon
//       test), float points / float K and D widened to double, 5 distortion coefficients (k1 k2 p1 p2 k3), result rounded to float;
//       the early-outs of Camera.cc:31 (no coefficients, k1 == 0, no keypoints) are part of the restatement;
//   the RGB-D tail of Frame::Frame (src/Frame.cc:136-158): depth image converted to float and divided by the depth scale,
//       d = depth.at<float>(kp.pt.y, kp.pt.x) at the DISTORTED keypoint with float->int truncation (quirk Q10),
//       depth = d and rightU = kpU.pt.x - bf / d (float arithmetic) where d > 0, else -1.
#include <cmath>
#include <cstdint>
#include <cstring>

extern "C" {

// order: 1 = RGB, 2 = BGR (the reference's Camera.Color values, Tracking.cc:56,63); src rows stride bytes apart, 3 bytes per pixel
// variant 0: the 14-bit coefficients (R 4899, G 9617, B 1868, >> 14), variant 1: the 15-bit ones of newer OpenCV 4.x builds (9798, 19235,
// 3735, >> 15).  cv::cvtColor is un-vendored third-party arithmetic: which one the reference's OpenCV uses could not be checked here.
void orc_cvt_gray_v(const uint8_t* src, int w, int h, int stride, int order, int variant, uint8_t* dst, int dst_stride) {
  const int cr = variant ? 9798 : 4899, cg = variant ? 19235 : 9617, cb = variant ? 3735 : 1868, sh = variant ? 15 : 14;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const uint8_t* p = src + (size_t)y * stride + 3 * x;
      const int r = order == 1 ? p[0] : p[2], g = p[1], b = order == 1 ? p[2] : p[0];
      dst[(size_t)y * dst_stride + x] = (uint8_t)((r * cr + g * cg + b * cb + (1 << (sh - 1))) >> sh);
    }
}
void orc_cvt_gray(const uint8_t* src, int w, int h, int stride, int order, uint8_t* dst, int dst_stride) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const uint8_t* p = src + (size_t)y * stride + 3 * x;
      const int r = order == 1 ? p[0] : p[2], g = p[1], b = order == 1 ? p[2] : p[0];
      dst[(size_t)y * dst_stride + x] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
    }
}

// xy: n points (x, y) float, undistorted in place.  K = (fx fy cx cy), D = (k1 k2 p1 p2 k3), all float as the reference stores them.
void orc_undistort_points(int n, float* xy, const float* K, const float* D) {
  if (!D || D[0] == 0.0f || n == 0) return;  // Camera.cc:31
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const double ifx = 1. / fx, ify = 1. / fy;
  const double k0 = D[0], k1 = D[1], k2 = D[2], k3 = D[3], k4 = D[4];
  for (int i = 0; i < n; ++i) {
    double x = xy[2 * i], y = xy[2 * i + 1];
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
    xy[2 * i] = (float)(xx * ww);
    xy[2 * i + 1] = (float)(yy * ww);
  }
}

// depth_type 0: uint16 image, 1: float image (rows stride bytes apart).  xy: distorted keypoints, xy_u: undistorted.
void orc_rgbd_lookup(int n, const float* xy, const float* xy_u, const void* depth, int depth_type, int stride, float depth_scale, float bf,
                     double* depth_out, double* right_u_out) {
  for (int i = 0; i < n; ++i) {
    const int px = (int)xy[2 * i], py = (int)xy[2 * i + 1];  // Mat::at<float>(float, float): the indices are truncated (Q10)
    float raw;
    if (depth_type == 0)
      raw = (float)*(const uint16_t*)((const uint8_t*)depth + (size_t)py * stride + 2 * px);
    else
      raw = *(const float*)((const uint8_t*)depth + (size_t)py * stride + 4 * px);
    const float d = raw / depth_scale;  // convertTo(CV_32F); depthImg /= dScale
    depth_out[i] = -1.0;
    right_u_out[i] = -1.0;
    if (d > 0) {
      depth_out[i] = (double)d;
      right_u_out[i] = (double)(xy_u[2 * i] - bf / d);
    }
  }
}

}  // extern "C"

// ---- MapPoint::isInVision + MapPoint::predictLevel (src/MapPoint.cc:141-201) ----------------------------------------------------
// cv::Mat arithmetic restated (OpenCV 4.x core, not in the reference tree -- a documented decision, parity unpinned):
//   Rcw * X + tcw : MatExpr GEMM; the 3x3 * 3x1 CV_32F fast path of gemmImpl forms float s = a0*b0 + a1*b1 + a2*b2 and stores
//                   (float)(s*alpha + c*beta) with alpha = beta = 1.0 (double)
//   cv::norm(NORM_L2), Mat::dot on CV_32F: double accumulation of float elements
// R: row-major 3x3 (Rcw), t: tcw, cam: fx fy cx cy, bounds: minU maxU minV maxV.  Outputs as documented in include/orbfe.h.
extern "C" void orc_project_map_points(int n, const float* pos, const float* vdir, const float* max_dist, const float* min_dist, const float* R,
                                       const float* t, const float* cam, const float* bounds, float scale_factor, float* uv, float* dist_out,
                                       float* cos_out, int8_t* level_out, uint8_t* visible) {
  const float log_sf = std::log(scale_factor);  // std::log(float) (MapPoint.cc:195)
  for (int i = 0; i < n; ++i) {
    float pc[3];
    for (int r = 0; r < 3; ++r) {
      const float s = R[3 * r] * pos[3 * i] + R[3 * r + 1] * pos[3 * i + 1] + R[3 * r + 2] * pos[3 * i + 2];
      pc[r] = (float)((double)s * 1.0 + (double)t[r] * 1.0);
    }
    float u = 0.f, v = 0.f, distance = 0.f, cos_theta = 0.f;
    int level = 0;
    uint8_t vis = 0;
    do {
      if (pc[2] < 0) break;                                                                        // :151
      const float x = pc[0], y = pc[1], z = pc[2];
      distance = std::sqrt(x * x + y * y + z * z);                                                 // :155
      if (!(distance < max_dist[i] && distance > min_dist[i])) break;                              // MapPoint.h:150-156
      u = x / z * cam[0] + cam[2];                                                                 // :159-160
      v = y / z * cam[1] + cam[3];
      if (!(u < bounds[1] && v < bounds[3] && u > bounds[0] && v > bounds[2])) break;              // Frame.h:259-264
      float vd[3];
      for (int r = 0; r < 3; ++r) vd[r] = R[3 * r] * vdir[3 * i] + R[3 * r + 1] * vdir[3 * i + 1] + R[3 * r + 2] * vdir[3 * i + 2];
      double nn = 0.0, dot = 0.0;
      for (int r = 0; r < 3; ++r) nn += (double)vd[r] * (double)vd[r];
      const float vabs = (float)std::sqrt(nn);                                                     // :166
      for (int r = 0; r < 3; ++r) dot += (double)vd[r] * (double)pc[r];
      cos_theta = (float)(dot / (double)(distance * vabs));                                        // :167
      if (cos_theta < 0.5) break;
      vis = 1;
      level = (int)std::nearbyint(std::log(max_dist[i] / distance) / log_sf);                      // cvRound (:195)
      level = level < 0 ? 0 : (level > 7 ? 7 : level);
    } while (false);
    uv[2 * i] = u, uv[2 * i + 1] = v, dist_out[i] = distance, cos_out[i] = cos_theta, level_out[i] = (int8_t)level, visible[i] = vis;
  }
}
