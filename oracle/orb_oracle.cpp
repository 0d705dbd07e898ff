// orb_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// PARITY UNPINNED: the reference (sunshanlu/ORB_SLAM2_ROS2) ships no tests, golden vectors or fixtures
// for this path (its only test, src/ORB_SLAM2/test/TxtVsProto.cc, times map I/O), and it cannot be built
// here (needs OpenCV/g2o/rclcpp, none installed).  This file is therefore a *restatement*: control flow
// follows the reference file:line cited at every function; the arithmetic that lives in un-vendored
// third-party code (OpenCV 4.x cv::resize / cv::GaussianBlur / cv::FAST / cvRound, g2o 20241228 SE3 edges)
// is restated from those libraries' published algorithms and documented where a choice had to be made.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// The product (orb_slam2_ros2_amd/csrc) never links, includes or calls anything in oracle/.
//
// Plain C++17 + libm + pthreads; exported surface is extern "C" for ctypes.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <thread>
#include <vector>

#include "../orb_slam2_ros2_amd/csrc/orb_math.h"  // shared deterministic atan2/sincos (math_mode 1)

namespace {

// ----------------------------------------------------------------------------------------------
// OpenCV rounding helpers (cvRound = round-half-to-even via the SSE cvt instructions; the default
// MXCSR/fenv rounding mode makes lrint identical).  SURVEY Appendix A.1.
// ----------------------------------------------------------------------------------------------
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_floor(float v) {
  int i = (int)v;
  return i - (i > v);
}
inline int cv_floor(double v) {
  int i = (int)v;
  return i - (i > v);
}
inline int cv_ceil(float v) {
  int i = (int)v;
  return i + (i < v);
}
inline short sat_short(float v) {
  int i = cv_round(v);
  return (short)std::min(32767, std::max(-32768, i));
}

struct Plane {
  int w = 0, h = 0;
  std::vector<uint8_t> d;  // tight, stride == w
  uint8_t at(int y, int x) const { return d[(size_t)y * w + x]; }
};

// cv::KeyPoint memory layout: 28 bytes, no padding.
struct KeyPoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// ----------------------------------------------------------------------------------------------
// cv::resize(src, dst, Size(dw,dh), 0, 0, INTER_LINEAR) for CV_8UC1 -- call site
// ORBExtractor.cc:316.  OpenCV imgproc/resize.cpp fixed-point bilinear path: 11-bit coefficients,
// horizontal pass into int32, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2.
// ----------------------------------------------------------------------------------------------
void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  const int ONE = 2048;
  double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  double scale_x = 1.0 / inv_scale_x, scale_y = 1.0 / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
  int xmax = dw;
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) {
      fx = 0;
      sx = 0;
    }
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) {
        fx = 0;
        sx = sw - 1;
      }
    }
    xofs[dx] = sx;
    ialpha[2 * dx] = sat_short((1.f - fx) * ONE);
    ialpha[2 * dx + 1] = sat_short(fx * ONE);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[2 * dy] = sat_short((1.f - fy) * ONE);
    ibeta[2 * dy + 1] = sat_short(fy * ONE);
  }
  std::vector<int> row0(dw), row1(dw);
  auto hline = [&](int sy, std::vector<int>& out) {
    sy = std::min(std::max(sy, 0), sh - 1);  // rows are clipped, weights are not
    const uint8_t* S = src + (size_t)sy * sstride;
    int dx = 0;
    for (; dx < xmax; ++dx) out[dx] = S[xofs[dx]] * ialpha[2 * dx] + S[xofs[dx] + 1] * ialpha[2 * dx + 1];
    for (; dx < dw; ++dx) out[dx] = S[xofs[dx]] * ONE;
  };
  for (int dy = 0; dy < dh; ++dy) {
    hline(yofs[dy], row0);
    hline(yofs[dy] + 1, row1);
    int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int dx = 0; dx < dw; ++dx) {
      int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
      D[dx] = (uint8_t)std::min(255, std::max(0, v));
    }
  }
}

// ----------------------------------------------------------------------------------------------
// cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8UC1 -- call site
// ORBExtractor.cc:319.  OpenCV's 8-bit path is separable 8.8 fixed point: row pass u8*tap -> u16
// (saturating), column pass u16*tap -> 16.16, result (v + 0x8000) >> 16.  taps[] is a parameter:
// variant 0 = {18,34,48,56,48,34,18} (OpenCV >= 4.3 error-diffused rounding, sum 256, DEFAULT),
// variant 1 = {18,34,49,55,49,34,18} (older per-tap rounding, sum 257).  SURVEY Appendix A.3.
// ----------------------------------------------------------------------------------------------
const int kGaussTaps[2][7] = {{18, 34, 48, 56, 48, 34, 18}, {18, 34, 49, 55, 49, 34, 18}};

inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0) p = -p;
    else p = 2 * (n - 1) - p;
  }
  return p;
}

void gauss7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, const int taps[7]) {
  std::vector<uint16_t> tmp((size_t)w * h);
  for (int y = 0; y < h; ++y) {
    const uint8_t* S = src + (size_t)y * sstride;
    for (int x = 0; x < w; ++x) {
      uint32_t acc = 0;
      for (int k = 0; k < 7; ++k) {
        acc += (uint32_t)taps[k] * S[reflect101(x + k - 3, w)];
        if (acc > 65535u) acc = 65535u;  // ufixedpoint16 saturating add
      }
      tmp[(size_t)y * w + x] = (uint16_t)acc;
    }
  }
  for (int y = 0; y < h; ++y) {
    uint8_t* D = dst + (size_t)y * dstride;
    for (int x = 0; x < w; ++x) {
      uint64_t acc = 0;
      for (int k = 0; k < 7; ++k) acc += (uint64_t)taps[k] * tmp[(size_t)reflect101(y + k - 3, h) * w + x];
      if (acc > 0xFFFFFFFFull) acc = 0xFFFFFFFFull;
      uint64_t r = (acc + 0x8000ull) >> 16;
      D[x] = (uint8_t)std::min<uint64_t>(255, r);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// cv::FAST(patch, kps, threshold, nonmaxSuppression=true), TYPE_9_16 -- call sites
// ORBExtractor.cc:365,367.  Restated from OpenCV features2d/src/fast.cpp (FAST_t<16>) and
// fast_score.cpp (cornerScore<16>): row-rolling buffers, threshold table, arc count > 8, score with
// the same pruning, 3x3 strict NMS against scores of the same patch.  SURVEY Appendix A.4.
// ----------------------------------------------------------------------------------------------
struct FastKp {
  int x, y, score;
};
const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int v = ptr[0];
  short d[N];
  for (int k = 0; k < N; ++k) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

void fast9_16(const uint8_t* img, int stride, int cols, int rows, int threshold, bool nonmax, std::vector<FastKp>& out) {
  out.clear();
  const int K = 8, N = 25;
  if (cols < 7 || rows < 7) return;
  int pixel[25];
  for (int k = 0; k < 16; ++k) pixel[k] = kRingDx[k] + kRingDy[k] * stride;
  for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t threshold_tab[512];
  for (int i = -255; i <= 255; ++i) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

  std::vector<uint8_t> sbuf((size_t)cols * 3, 0);
  std::vector<int> cbuf((size_t)(cols + 1) * 3, 0);
  uint8_t* buf[3] = {sbuf.data(), sbuf.data() + cols, sbuf.data() + 2 * cols};
  int* cpbuf[3] = {cbuf.data() + 1, cbuf.data() + (cols + 1) + 1, cbuf.data() + 2 * (cols + 1) + 1};

  for (int i = 3; i < rows - 2; ++i) {
    const uint8_t* ptr = img + (size_t)i * stride + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3];
    std::memset(curr, 0, cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; ++j, ++ptr) {
        int v = ptr[0];
        const uint8_t* tab = &threshold_tab[0] - v + 255;
        int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
        d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
        d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
        d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
        d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
        d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
        bool is_corner = false;
        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (int k = 0; k < N; ++k) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                is_corner = true;
                break;
              }
            } else
              count = 0;
          }
        }
        if (!is_corner && (d & 2)) {
          int vt = v + threshold, count = 0;
          for (int k = 0; k < N; ++k) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                is_corner = true;
                break;
              }
            } else
              count = 0;
          }
        }
        if (is_corner) {
          cornerpos[ncorners++] = j;
          if (nonmax) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; ++k) {
      int j = cornerpos[k];
      int score = prev[j];
      if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                      score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
        out.push_back({j, i - 1, score});
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Quadtree keypoint selection -- ORBExtractor.h:18-93, ORBExtractor.cc:19-192.
// Same containers as the reference (std::multimap<count, node, greater>, std::set<size_t>) so the
// tie-breaking (equal counts pop in insertion order) is the library's, not a re-derivation.
// ----------------------------------------------------------------------------------------------
struct QCand {
  float x, y, response;
};
struct QNode {
  double row_begin, row_end, col_begin, col_end;
  std::vector<size_t> kps;
  std::vector<int> children;
};

struct Quadtree {
  const std::vector<QCand>& all;
  std::vector<QNode> nodes;
  unsigned need;
  long n_splits = 0;
  Quadtree(int x, int y, const std::vector<QCand>& cands, unsigned need_nodes) : all(cands), need(need_nodes) {
    QNode root;  // ORBExtractor.cc:19-28
    root.row_begin = 0;
    root.row_end = y;
    root.col_begin = 0;
    root.col_end = x;
    for (size_t i = 0; i < cands.size(); ++i) root.kps.push_back(i);
    nodes.push_back(std::move(root));
  }
  bool is_in(const QNode& n, const QCand& kp) const {  // ORBExtractor.h:55-62 (strict on all sides)
    return kp.x > n.col_begin && kp.x < n.col_end && kp.y > n.row_begin && kp.y < n.row_end;
  }
  int make_child(int parent, double rb, double re, double cb, double ce) {  // ORBExtractor.cc:39-53
    QNode c;
    c.row_begin = rb;
    c.row_end = re;
    c.col_begin = cb;
    c.col_end = ce;
    for (size_t idx : nodes[parent].kps)
      if (is_in(c, all[idx])) c.kps.push_back(idx);
    nodes.push_back(std::move(c));
    return (int)nodes.size() - 1;
  }
  void split_node(int id) {  // ORBExtractor.cc:60-72: rows outer, cols inner
    double rb = nodes[id].row_begin, re = nodes[id].row_end, cb = nodes[id].col_begin, ce = nodes[id].col_end;
    double rows[3] = {rb, (rb + re) / 2, re};
    double cols[3] = {cb, (cb + ce) / 2, ce};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) {
        int c = make_child(id, rows[i], rows[i + 1], cols[j], cols[j + 1]);
        nodes[id].children.push_back(c);
      }
  }
  void init_split(int id) {  // ORBExtractor.cc:81-96
    const double w = nodes[id].col_end, h = nodes[id].row_end;
    const int nIni = (int)std::round(w / h);
    const float hX = (float)((double)w / nIni);
    std::vector<double> cols = {nodes[id].col_begin};
    for (std::size_t idx = 1; idx < (std::size_t)nIni; ++idx) cols.push_back((double)((float)idx * hX));
    cols.push_back(nodes[id].col_end);
    for (int i = 0; i < nIni; ++i) {
      int c = make_child(id, nodes[id].row_begin, nodes[id].row_end, cols[i], cols[i + 1]);
      nodes[id].children.push_back(c);
    }
  }
  size_t get_feature(int id) const {  // ORBExtractor.cc:103-117 (first maximum wins, strict >)
    size_t best = 0;
    float best_r = 0.0f;
    for (size_t idx : nodes[id].kps)
      if (all[idx].response > best_r) {
        best = idx;
        best_r = all[idx].response;
      }
    return best;
  }
  std::set<size_t> run() {  // ORBExtractor.cc:144-192
    std::multimap<size_t, int, std::greater<size_t>> to_split;
    init_split(0);
    to_split.insert({nodes[0].kps.size(), 0});
    unsigned n_nodes = 1;
    while (n_nodes < need && !to_split.empty()) {
      auto it = to_split.begin();
      int id = it->second;
      if (id != 0) split_node(id);
      to_split.erase(it);
      n_nodes -= 1;
      ++n_splits;
      std::vector<int> ch = nodes[id].children;  // copy: nodes may reallocate
      for (int c : ch) {
        if (nodes[c].kps.empty()) continue;
        to_split.insert({nodes[c].kps.size(), c});
        n_nodes += 1;
      }
    }
    std::set<size_t> ret;
    size_t n_take = std::min((size_t)need, to_split.size());
    auto it = to_split.begin();
    for (size_t i = 0; i < n_take; ++i, ++it) {
      // The reference calls getFeature() on a node; an empty root (no candidates, need==1) would make it
      // index levelKps[0] out of bounds.  The restatement yields nothing in that undefined case.
      if (nodes[it->second].kps.empty()) continue;
      ret.insert(get_feature(it->second));
    }
    return ret;
  }
};

// ----------------------------------------------------------------------------------------------
// ORBExtractor -- ORBExtractor.cc:205-508.
// ----------------------------------------------------------------------------------------------
struct BriefPair {
  float x1, y1, x2, y2;
};
const int8_t kEmbeddedPattern[256][4] = {
#include "../orb_slam2_ros2_amd/csrc/brief_pattern.inc"
};

struct Extractor {
  int n_features, n_levels, th_hi, th_lo, math_mode;
  float scale_factor;
  std::vector<float> sf;      // mvfScaledFactors
  std::vector<int> quota;     // mvnFeatures
  std::vector<Plane> pyr;     // mvPyramids  (un-blurred)
  std::vector<Plane> blur;    // mvBriefMat
  std::vector<BriefPair> tem; // mvBriefTem
  std::vector<unsigned> umax; // mvMaxColIdx
  static const int kBorder = 19;    // mnBorderSize (ORBExtractor.cc:523)
  static const int kCentroidR = 15; // mnCentroidR (ORBExtractor.cc:518)
  std::vector<std::vector<QCand>> last_cands;  // debugging aid for the parity tests
  std::vector<long> last_splits;
  std::vector<double> last_theta;

  void init_max_u() {  // ORBExtractor.cc:217-236
    umax.assign(kCentroidR + 1, 0);
    int v, v0, vmax = cv_floor(kCentroidR * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(kCentroidR * std::sqrt(2.f) / 2);
    const double hp2 = kCentroidR * kCentroidR;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = kCentroidR, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  // returns false on ImageSizeError (ORBExtractor.cc:310-314)
  bool init_pyramid(const uint8_t* img, int w, int h, int stride, const int taps[7]) {  // :278-320
    pyr.resize(n_levels);
    blur.resize(n_levels);
    for (int l = 0; l < n_levels; ++l) sf.push_back((float)std::pow((double)scale_factor, (double)l));
    quota.assign(n_levels, 0);
    float scale = 1.0f / scale_factor;
    int sum = 0;
    int nfeats = cv_round((double)(n_features * (1 - scale)) / (1 - std::pow((double)scale, (double)n_levels)));
    for (int l = 0; l < n_levels - 1; ++l) {
      quota[l] = nfeats;
      sum += nfeats;
      nfeats = cv_round(nfeats * scale);
    }
    quota[n_levels - 1] = std::max(0, n_features - sum);

    pyr[0].w = w;
    pyr[0].h = h;
    pyr[0].d.resize((size_t)w * h);
    for (int y = 0; y < h; ++y) std::memcpy(&pyr[0].d[(size_t)y * w], img + (size_t)y * stride, w);
    for (int i = 1; i < n_levels; ++i) {
      int lw = cv_round(w / sf[i]);
      int lh = cv_round(h / sf[i]);
      if (lw < 2 * kBorder || lh < 2 * kBorder) return false;
      pyr[i].w = lw;
      pyr[i].h = lh;
      pyr[i].d.resize((size_t)lw * lh);
      resize_linear_u8(pyr[0].d.data(), w, h, w, pyr[i].d.data(), lw, lh, lw);  // always from level 0 (Q1)
    }
    for (int i = 0; i < n_levels; ++i) {
      blur[i].w = pyr[i].w;
      blur[i].h = pyr[i].h;
      blur[i].d.resize(pyr[i].d.size());
      gauss7_u8(pyr[i].d.data(), pyr[i].w, pyr[i].h, pyr[i].w, blur[i].d.data(), pyr[i].w, taps);
    }
    return true;
  }

  // ORBExtractor.cc:331-387.  Returns keypoints of one level (level coordinates, angle -1).
  void extract_fast(int level, std::vector<KeyPoint>& out) {
    const Plane& image = pyr[level];
    int max_bx = image.w - kBorder + 3, max_by = image.h - kBorder + 3;
    int min_bx = kBorder - 3, min_by = kBorder - 3;
    int w = max_bx - min_bx, h = max_by - min_by;
    const int n_cols = w / 30, n_rows = h / 30;
    std::vector<QCand> level_kps;
    if (n_cols > 0 && n_rows > 0) {
      const int w_cell = (int)std::ceil(w / n_cols);  // integer division first (Q2)
      const int h_cell = (int)std::ceil(h / n_rows);
      std::vector<FastKp> cell;
      for (int idx = 0; idx < n_rows; ++idx) {
        int ini_y = min_by + idx * h_cell, max_y = ini_y + h_cell + 6;
        if (ini_y >= max_by - 6) continue;
        if (max_y > max_by) max_y = max_by;
        for (int jdx = 0; jdx < n_cols; ++jdx) {
          int ini_x = min_bx + jdx * w_cell, max_x = ini_x + w_cell + 6;
          if (ini_x >= max_bx - 6) continue;
          if (max_x > max_bx) max_x = max_bx;
          const uint8_t* patch = image.d.data() + (size_t)ini_y * image.w + ini_x;
          fast9_16(patch, image.w, max_x - ini_x, max_y - ini_y, th_hi, true, cell);
          if (cell.empty()) fast9_16(patch, image.w, max_x - ini_x, max_y - ini_y, th_lo, true, cell);
          for (const FastKp& k : cell)
            level_kps.push_back({(float)k.x + (float)(jdx * w_cell), (float)k.y + (float)(idx * h_cell), (float)k.score});
        }
      }
    }
    Quadtree qt(w, h, level_kps, (unsigned)quota[level]);
    std::set<size_t> idxs = qt.run();
    for (size_t id : idxs) {
      KeyPoint kp;
      kp.x = level_kps[id].x + (float)min_bx;
      kp.y = level_kps[id].y + (float)min_by;
      kp.size = 7.f;
      kp.angle = -1.f;
      kp.response = level_kps[id].response;
      kp.octave = level;
      kp.class_id = -1;
      out.push_back(kp);
    }
    last_splits[level] = qt.n_splits;
    last_cands[level] = std::move(level_kps);
  }

  // ORBExtractor.cc:465-487
  void gray_centroid_moments(const Plane& image, float px, float py, int& m10_out, int& m01_out) const {
    int x = cv_round(px), y = cv_round(py);
    int m10 = 0, m01 = 0;
    for (int dx = -kCentroidR; dx <= kCentroidR; ++dx) m10 += dx * image.at(y, x + dx);
    for (int dy = 1; dy <= kCentroidR; ++dy) {
      int v_sum = 0;
      int d = (int)umax[dy];
      for (int dx = -d; dx <= d; ++dx) {
        int up = image.at(y + dy, x + dx), down = image.at(y - dy, x + dx);
        m10 += dx * (up + down);
        v_sum += (up - down);
      }
      m01 += v_sum * dy;
    }
    m10_out = m10;
    m01_out = m01;
  }
  double angle_of(int m01, int m10) const {
    return math_mode ? orbmath::det_atan2((double)m01, (double)m10) : std::atan2((double)m01, (double)m10);
  }

  // ORBExtractor.cc:427-456 + rotateTemplate :534-540
  double compute_brief_one(int level, float px, float py, uint8_t* desc) const {
    const Plane& image = pyr[level];
    const Plane& work = blur[level];
    int m10, m01;
    gray_centroid_moments(image, px, py, m10, m01);
    double theta = angle_of(m01, m10);
    double c, s;
    if (math_mode) orbmath::det_sincos(theta, &s, &c);
    else {
      c = std::cos(theta);
      s = std::sin(theta);
    }
    uint8_t value = 0;
    int bias = 0, nb = 0;
    for (const BriefPair& pp : tem) {
      float p1x = (float)(pp.x1 * c - pp.y1 * s), p1y = (float)(pp.x1 * s + pp.y1 * c);
      float p2x = (float)(pp.x2 * c - pp.y2 * s), p2y = (float)(pp.x2 * s + pp.y2 * c);
      uint8_t v1 = work.at(cv_round(py + p1y), cv_round(px + p1x));
      uint8_t v2 = work.at(cv_round(py + p2y), cv_round(px + p2x));
      value |= (uint8_t)((v1 < v2) << bias);
      if (bias == 7) {
        desc[nb++] = value;
        bias = 0;
        value = 0;
        continue;
      }
      ++bias;
    }
    return theta;
  }

  // ORBExtractor.cc:499-508 + :397-415
  int extract(KeyPoint* kps_out, uint8_t* desc_out, int cap) {
    std::vector<KeyPoint> kps;
    last_cands.assign(n_levels, {});
    last_splits.assign(n_levels, 0);
    for (int level = 0; level < n_levels; ++level) {
      std::vector<KeyPoint> lk;
      extract_fast(level, lk);
      kps.insert(kps.end(), lk.begin(), lk.end());
    }
    last_theta.assign(kps.size(), 0.0);
    int n = (int)std::min<size_t>(kps.size(), (size_t)cap);
    for (int i = 0; i < n; ++i) {
      KeyPoint& kp = kps[i];
      double angle = compute_brief_one(kp.octave, kp.x, kp.y, desc_out + (size_t)i * 32);
      last_theta[i] = angle;
      kp.angle = (float)(angle / M_PI * 180);
      kp.x *= sf[kp.octave];
      kp.y *= sf[kp.octave];
      kps_out[i] = kp;
    }
    return n;
  }
};

// ----------------------------------------------------------------------------------------------
// ORBMatcher pieces -- ORBMatcher.cc:18-81, 841-1011.
// ----------------------------------------------------------------------------------------------
int desc_distance(const uint8_t* a, const uint8_t* b) {  // ORBMatcher.cc:941-956
  int dist = 0;
  for (int i = 0; i < 8; ++i) {
    uint32_t pa, pb;
    std::memcpy(&pa, a + 4 * i, 4);
    std::memcpy(&pb, b + 4 * i, 4);
    uint32_t v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555u);
    v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
    dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
  }
  return dist;
}

// ORBMatcher.cc:967-990 -- order dependent: the old best is NOT demoted to second (Q6).
void best_match(const uint8_t* desc, const uint8_t* cand_desc, const int64_t* cand_idx, int n, int64_t* min_idx_out,
                int* min_dist_out, int* second_out, float* ratio_out) {
  int min_d = INT_MAX, second = INT_MAX;
  int64_t min_idx = 0;
  for (int i = 0; i < n; ++i) {
    int64_t idx = cand_idx[i];
    int d = desc_distance(desc, cand_desc + (size_t)idx * 32);
    if (d < min_d) {
      min_d = d;
      min_idx = idx;
    } else if (d < second) {
      second = d;
    }
  }
  *min_idx_out = min_idx;
  *min_dist_out = min_d;
  *second_out = second;
  *ratio_out = (float)min_d / (float)second;
}

const int kMeanThreshold = 75;  // ORBMatcher::mnMeanThreshold (ORBMatcher.cc:1088)
const int kW = 5, kL = 5;       // mnW, mnL (ORBMatcher.cc:1089-1090)

// SAD of centre-subtracted 11x11 patches -- ORBMatcher.cc:893-905 (float math on exact integers).
float sad11(const Plane& i1, int x1, int y1, const Plane& i2, int x2, int y2) {
  float c1 = (float)i1.at(y1, x1), c2 = (float)i2.at(y2, x2);
  double acc = 0;
  for (int dy = -kW; dy <= kW; ++dy)
    for (int dx = -kW; dx <= kW; ++dx) {
      float a = (float)i1.at(y1 + dy, x1 + dx) - c1;
      float b = (float)i2.at(y2 + dy, x2 + dx) - c2;
      acc += std::fabs((double)(a - b));
    }
  return (float)acc;
}

// ORBMatcher.cc:841-881 + getPitch :1002-1011
float pixel_sad_match(const Plane& li, const Plane& ri, const KeyPoint& lk, const KeyPoint& rk, const std::vector<float>& sf) {
  std::vector<float> scores;
  float min_score = std::numeric_limits<float>::max();
  int best_l = 0;
  int lx = cv_floor(lk.x / sf[lk.octave]), ly = cv_floor(lk.y / sf[lk.octave]);
  int rx = cv_floor(rk.x / sf[rk.octave]), ry = cv_floor(rk.y / sf[rk.octave]);
  for (int l = -kL; l < kL + 1; ++l) {
    float score = sad11(li, lx, ly, ri, rx + l, ry);
    if (score < min_score) {
      min_score = score;
      best_l = l;
    }
    scores.push_back(score);
  }
  float delta_u = 0;
  best_l += kL;
  if (best_l > 0 && best_l < (int)scores.size() - 1) {
    const float s1 = scores[best_l - 1], s2 = scores[best_l], s3 = scores[best_l + 1];
    delta_u = (float)(0.5 * (s1 - s3) / (s1 + s3 - 2 * s2));
    if (delta_u < 1 && delta_u > -1) delta_u *= sf[rk.octave];
    else delta_u = 0;
  }
  return delta_u;
}

int stereo_match(const Extractor& exl, const Extractor& exr, const KeyPoint* lk, const uint8_t* ld, int nl,
                 const KeyPoint* rk, const uint8_t* rd, int nr, float fx, float bf, double* right_u, double* depth,
                 int32_t* best_right, int32_t* best_dist) {
  const std::vector<float>& sf = exl.sf;
  int rows = exl.pyr[0].h, cols = exl.pyr[0].w;
  // createRowIndexDB -- ORBMatcher.cc:915-932
  std::vector<std::vector<int64_t>> row_db(rows);
  for (int idx = 0; idx < nr; ++idx) {
    const KeyPoint& kp = rk[idx];
    float r = (float)(2.0 * sf[kp.octave]);
    unsigned row = (unsigned)cv_round(kp.y);
    unsigned max_row = (unsigned)std::min(rows, cv_round((float)row + r + 1));
    unsigned min_row = (unsigned)std::max(0, cv_round((float)row - r));
    for (unsigned rr = min_row; rr < max_row; ++rr) row_db[rr].push_back(idx);
  }
  int n_matches = 0;
  for (int l = 0; l < nl; ++l) {
    right_u[l] = -1.0;
    depth[l] = -1.0;
    if (best_right) best_right[l] = -1;
    if (best_dist) best_dist[l] = -1;
  }
  for (int ldx = 0; ldx < nl; ++ldx) {  // ORBMatcher.cc:35-79
    const KeyPoint& l = lk[ldx];
    float max_u = l.x - 0;
    float min_u = std::max(0.f, l.x - fx);
    const std::vector<int64_t>& ids = row_db[cv_round(l.y)];
    std::vector<int64_t> cand;
    for (int64_t idx : ids) {
      float rc = rk[idx].x;
      if (rc < max_u && rc > min_u) cand.push_back(idx);
    }
    if (cand.empty()) continue;
    int64_t bi;
    int bd, sd;
    float ratio;
    best_match(ld + (size_t)ldx * 32, rd, cand.data(), (int)cand.size(), &bi, &bd, &sd, &ratio);
    if (best_right) best_right[ldx] = (int32_t)bi;
    if (best_dist) best_dist[ldx] = bd;
    if (bd > kMeanThreshold) continue;
    const KeyPoint& r = rk[bi];
    if (l.octave > r.octave + 1 || l.octave < r.octave - 1) continue;
    float du = pixel_sad_match(exl.pyr[l.octave], exr.pyr[r.octave], l, r, sf);
    float ru = r.x + du;  // bestL is NOT added (Q7)
    ru = std::max(0.f, ru);
    ru = std::min(ru, (float)cols - 1);
    float delta = l.x - ru;
    if (delta <= 0) {
      ru = r.x;
      delta = l.x - ru;
      if (delta <= 0) continue;
    }
    right_u[ldx] = ru;
    depth[ldx] = bf / (l.x - ru);
    ++n_matches;
  }
  return n_matches;
}

}  // namespace

// ==================================================================================================
// C surface
// ==================================================================================================
extern "C" {

struct orc_keypoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};

// math_mode: 0 = libm atan2/cos/sin exactly as the reference calls them; 1 = shared deterministic routines.
// blur_variant: 0 (default) / 1, see gauss7_u8.  Returns NULL on the reference's ImageSizeError.
void* orc_extractor_create(const uint8_t* img, int w, int h, int stride, int n_features, int n_levels, float scale,
                           int th_hi, int th_lo, const int8_t* pattern, int blur_variant, int math_mode) {
  auto* e = new Extractor();
  e->n_features = n_features;
  e->n_levels = n_levels;
  e->scale_factor = scale;
  e->th_hi = th_hi;
  e->th_lo = th_lo;
  e->math_mode = math_mode;
  const int8_t* pat = pattern ? pattern : &kEmbeddedPattern[0][0];
  for (int i = 0; i < 256; ++i) e->tem.push_back({(float)pat[4 * i], (float)pat[4 * i + 1], (float)pat[4 * i + 2], (float)pat[4 * i + 3]});
  e->init_max_u();
  if (!e->init_pyramid(img, w, h, stride, kGaussTaps[blur_variant ? 1 : 0])) {
    delete e;
    return nullptr;
  }
  e->last_cands.assign(n_levels, {});
  e->last_splits.assign(n_levels, 0);
  return e;
}
void orc_extractor_destroy(void* h) { delete (Extractor*)h; }

int orc_extractor_level_info(void* h, int level, int* w, int* hh, float* sf, int* quota) {
  auto* e = (Extractor*)h;
  if (level < 0 || level >= e->n_levels) return -1;
  *w = e->pyr[level].w;
  *hh = e->pyr[level].h;
  *sf = e->sf[level];
  *quota = e->quota[level];
  return 0;
}
const uint8_t* orc_extractor_plane(void* h, int level, int blurred) {
  auto* e = (Extractor*)h;
  return blurred ? e->blur[level].d.data() : e->pyr[level].d.data();
}
int orc_extractor_umax(void* h, int32_t* out16) {
  auto* e = (Extractor*)h;
  for (int i = 0; i < 16; ++i) out16[i] = (int32_t)e->umax[i];
  return 16;
}
int orc_extractor_extract(void* h, orc_keypoint* kps, uint8_t* desc, int cap) {
  return ((Extractor*)h)->extract((KeyPoint*)kps, desc, cap);
}
// candidates of the last extract() (region coordinates, before the quadtree); returns the count
int orc_extractor_candidates(void* h, int level, float* xyr, int cap) {
  auto* e = (Extractor*)h;
  const auto& c = e->last_cands[level];
  int n = (int)std::min<size_t>(c.size(), (size_t)cap);
  for (int i = 0; i < n; ++i) {
    xyr[3 * i] = c[i].x;
    xyr[3 * i + 1] = c[i].y;
    xyr[3 * i + 2] = c[i].response;
  }
  return (int)c.size();
}
long orc_extractor_splits(void* h, int level) { return ((Extractor*)h)->last_splits[level]; }
int orc_extractor_thetas(void* h, double* out, int cap) {
  auto* e = (Extractor*)h;
  int n = (int)std::min<size_t>(e->last_theta.size(), (size_t)cap);
  for (int i = 0; i < n; ++i) out[i] = e->last_theta[i];
  return n;
}

// orientation + descriptor of one point of one level (level coordinates), for known-answer tests
double orc_extractor_describe(void* h, int level, float x, float y, uint8_t* desc32, int32_t* m10, int32_t* m01) {
  auto* e = (Extractor*)h;
  int a, b;
  e->gray_centroid_moments(e->pyr[level], x, y, a, b);
  if (m10) *m10 = a;
  if (m01) *m01 = b;
  return e->compute_brief_one(level, x, y, desc32);
}

// ---- standalone primitives (known-answer tests) ---------------------------------------------------
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orc_gauss7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int variant) {
  gauss7_u8(src, w, h, sstride, dst, dstride, kGaussTaps[variant ? 1 : 0]);
}
int orc_fast9_16(const uint8_t* img, int stride, int cols, int rows, int threshold, int nonmax, int32_t* xys, int cap) {
  std::vector<FastKp> out;
  fast9_16(img, stride, cols, rows, threshold, nonmax != 0, out);
  int n = (int)std::min<size_t>(out.size(), (size_t)cap);
  for (int i = 0; i < n; ++i) {
    xys[3 * i] = out[i].x;
    xys[3 * i + 1] = out[i].y;
    xys[3 * i + 2] = out[i].score;
  }
  return (int)out.size();
}
int orc_quadtree_select(int w, int h, const float* xyr, int n, int need, int32_t* out_idx, int64_t* n_splits) {
  std::vector<QCand> c(n);
  for (int i = 0; i < n; ++i) c[i] = {xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]};
  Quadtree qt(w, h, c, (unsigned)need);
  std::set<size_t> r = qt.run();
  int k = 0;
  for (size_t id : r) out_idx[k++] = (int32_t)id;
  if (n_splits) *n_splits = qt.n_splits;
  return k;
}
int orc_hamming256(const uint8_t* a, const uint8_t* b) { return desc_distance(a, b); }
void orc_best_match(const uint8_t* q, const uint8_t* train, const int64_t* cand, int n, int64_t* best_idx, int* best_dist,
                    int* second_dist, float* ratio) {
  best_match(q, train, cand, n, best_idx, best_dist, second_dist, ratio);
}
// dense form used by config 3: every query against all n_train in ascending order
void orc_match_bruteforce(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* best_idx, int32_t* best_dist,
                          int32_t* second_dist) {
  std::vector<int64_t> cand(nt);
  for (int i = 0; i < nt; ++i) cand[i] = i;
  for (int i = 0; i < nq; ++i) {
    int64_t bi;
    int bd, sd;
    float ratio;
    best_match(q + (size_t)i * 32, t, cand.data(), nt, &bi, &bd, &sd, &ratio);
    best_idx[i] = nt ? (int32_t)bi : -1;
    best_dist[i] = bd;
    second_dist[i] = sd;
  }
}
double orc_atan2(double y, double x, int math_mode) { return math_mode ? orbmath::det_atan2(y, x) : std::atan2(y, x); }
void orc_sincos(double t, int math_mode, double* s, double* c) {
  if (math_mode) orbmath::det_sincos(t, s, c);
  else {
    *s = std::sin(t);
    *c = std::cos(t);
  }
}

int orc_stereo_match(void* exl, void* exr, const orc_keypoint* lk, const uint8_t* ld, int nl, const orc_keypoint* rk,
                     const uint8_t* rd, int nr, float fx, float bf, double* right_u, double* depth, int32_t* best_right,
                     int32_t* best_dist) {
  return stereo_match(*(Extractor*)exl, *(Extractor*)exr, (const KeyPoint*)lk, ld, nl, (const KeyPoint*)rk, rd, nr, fx, bf,
                      right_u, depth, best_right, best_dist);
}

// One stereo frame exactly as Frame::Frame (Frame.cc:85-105) + Frame::createStereo (Frame.h:313-322) drive it:
// both extractors constructed sequentially (pyramid+blur in the ctor), extract() on two concurrent threads,
// then searchByStereo on the calling thread.  Used as the CPU baseline ("reference-shaped", 2 cores).
// Returns the match count, or -1 on ImageSizeError.  Any output pointer may be NULL.
int orc_stereo_frame(const uint8_t* left, const uint8_t* right, int w, int h, int stride, int n_features, int n_levels,
                     float scale, int th_hi, int th_lo, float fx, float bf, int math_mode, int threads, orc_keypoint* lk_out,
                     uint8_t* ld_out, int32_t* nl_out, orc_keypoint* rk_out, uint8_t* rd_out, int32_t* nr_out, double* right_u,
                     double* depth) {
  void* el = orc_extractor_create(left, w, h, stride, n_features, n_levels, scale, th_hi, th_lo, nullptr, 0, math_mode);
  void* er = orc_extractor_create(right, w, h, stride, n_features, n_levels, scale, th_hi, th_lo, nullptr, 0, math_mode);
  if (!el || !er) {
    if (el) orc_extractor_destroy(el);
    if (er) orc_extractor_destroy(er);
    return -1;
  }
  std::vector<KeyPoint> lk(n_features), rk(n_features);
  std::vector<uint8_t> ld((size_t)n_features * 32), rd((size_t)n_features * 32);
  int nl = 0, nr = 0;
  if (threads >= 2) {
    std::thread tl([&] { nl = ((Extractor*)el)->extract(lk.data(), ld.data(), n_features); });
    std::thread tr([&] { nr = ((Extractor*)er)->extract(rk.data(), rd.data(), n_features); });
    tl.join();
    tr.join();
  } else {
    nl = ((Extractor*)el)->extract(lk.data(), ld.data(), n_features);
    nr = ((Extractor*)er)->extract(rk.data(), rd.data(), n_features);
  }
  std::vector<double> ru(std::max(nl, 1)), dp(std::max(nl, 1));
  int m = stereo_match(*(Extractor*)el, *(Extractor*)er, lk.data(), ld.data(), nl, rk.data(), rd.data(), nr, fx, bf, ru.data(),
                       dp.data(), nullptr, nullptr);
  if (lk_out) std::memcpy(lk_out, lk.data(), sizeof(KeyPoint) * nl);
  if (ld_out) std::memcpy(ld_out, ld.data(), (size_t)32 * nl);
  if (rk_out) std::memcpy(rk_out, rk.data(), sizeof(KeyPoint) * nr);
  if (rd_out) std::memcpy(rd_out, rd.data(), (size_t)32 * nr);
  if (nl_out) *nl_out = nl;
  if (nr_out) *nr_out = nr;
  if (right_u) std::memcpy(right_u, ru.data(), sizeof(double) * nl);
  if (depth) std::memcpy(depth, dp.data(), sizeof(double) * nl);
  orc_extractor_destroy(el);
  orc_extractor_destroy(er);
  return m;
}

}  // extern "C"

// ---- guided matching: VirtualFrame::initGrid + findFeaturesInArea (src/Frame.cc:53-69, 286-311) + getBestMatch -------------
// The core every guided search of the reference shares (searchByProjection frame-to-frame, ORBMatcher.cc:265-347; map points to
// frame, :561-612): candidates = the features of the 64x48-px grid cells overlapping the box [x-r, x+r] x [y-r, y+r]
// (r = radius * getScaledFactor2(octave)), rows outer, columns inner, features of a cell in index order, filtered by octave
// range and by the caller's exclusion mask; then the order-dependent best / second-best scan.
// One deviation: the reference indexes mGrids[row][col] unchecked -- col == cols when maxX == width is a multiple of 64, and a
// negative / too large / non-finite undistorted keypoint coordinate in initGrid (Frame.cc:64-65) -- which is undefined behaviour
// there; cell indices are clamped to the grid at both ends here (non-finite -> cell 0), like csrc/k_guided.hip::grid_cell.
static int grid_cell(float v, int size, int n) {
  const float q = v / size;
  if (!(q > 0.0f)) return 0;
  if (q >= (float)(n - 1)) return n - 1;
  return cv_floor(q);
}
// bounds = {mfMinU, mfMaxU, mfMinV, mfMaxV} of the target frame (Frame.h:33-43): grid size as initGrid (Frame.cc:55-56), box clipped at
// (int)mfMaxU / (int)mfMaxV (:291-293).  excluded_hits[n] (nullable): occurrences of an excluded feature in a query's window -- what
// the copy_if of searchByProjection (ORBMatcher.cc:321-331) turns into addMatchInTrack calls.
extern "C" void orc_search_in_area_ex(const orc_keypoint* kps, const uint8_t* desc, int n, const float* bounds, int nq, const float* qxy,
                                      const float* radius, const int8_t* min_level, const int8_t* max_level, const uint8_t* q_desc,
                                      const uint8_t* exclude, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                                      int32_t* n_cand, int32_t* excluded_hits) {
  const int GW = 64, GH = 48;
  const int rows = cv_ceil((float)(bounds[3] - bounds[2]) / GH), cols = cv_ceil((float)(bounds[1] - bounds[0]) / GW);
  const int width = (int)bounds[1], height = (int)bounds[3];
  if (excluded_hits)
    for (int i = 0; i < n; ++i) excluded_hits[i] = 0;
  std::vector<std::vector<int64_t>> grid((size_t)rows * cols);
  for (int i = 0; i < n; ++i) {
    const int r = grid_cell(kps[i].y, GH, rows), c = grid_cell(kps[i].x, GW, cols);
    grid[(size_t)r * cols + c].push_back(i);
  }
  for (int q = 0; q < nq; ++q) {
    const float x = qxy[2 * q], y = qxy[2 * q + 1], rad = radius[q];
    const int min_x = std::max(0, cv_round(x - rad)), max_x = std::min(width, cv_round(x + rad));
    const int min_y = std::max(0, cv_round(y - rad)), max_y = std::min(height, cv_round(y + rad));
    const int c0 = std::max(0, std::min(cols - 1, cv_floor((float)min_x / GW))), c1 = std::min(cols - 1, cv_floor((float)max_x / GW));
    const int r0 = std::max(0, std::min(rows - 1, cv_floor((float)min_y / GH))), r1 = std::min(rows - 1, cv_floor((float)max_y / GH));
    std::vector<int64_t> cand;
    for (int r = r0; r <= r1; ++r)
      for (int c = c0; c <= c1; ++c)
        for (int64_t id : grid[(size_t)r * cols + c]) {
          const int oc = kps[id].octave;
          if (!(oc <= max_level[q] && oc >= min_level[q])) continue;
          if (exclude && exclude[id]) {
            if (excluded_hits) ++excluded_hits[id];
            continue;
          }
          cand.push_back(id);
        }
    n_cand[q] = (int32_t)cand.size();
    if (cand.empty()) {
      best_idx[q] = -1;
      best_dist[q] = INT_MAX;
      second_dist[q] = INT_MAX;
      continue;
    }
    int64_t bi;
    int bd, sd;
    float ratio;
    best_match(q_desc + (size_t)q * 32, desc, cand.data(), (int)cand.size(), &bi, &bd, &sd, &ratio);
    best_idx[q] = (int32_t)bi;
    best_dist[q] = bd;
    second_dist[q] = sd;
  }
}

extern "C" void orc_search_in_area(const orc_keypoint* kps, const uint8_t* desc, int n, int width, int height, int nq, const float* qxy,
                                   const float* radius, const int8_t* min_level, const int8_t* max_level, const uint8_t* q_desc,
                                   const uint8_t* exclude, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                                   int32_t* n_cand) {
  const float bounds[4] = {0.f, (float)width, 0.f, (float)height};
  orc_search_in_area_ex(kps, desc, n, bounds, nq, qxy, radius, min_level, max_level, q_desc, exclude, best_idx, best_dist, second_dist, n_cand,
                        nullptr);
}
