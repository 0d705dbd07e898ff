// k_guided.hip -- grid-guided descriptor matching against the device-resident features of one image slot.
//
// Replaces VirtualFrame::initGrid (src/ORB_SLAM2/src/Frame.cc:53-69) + VirtualFrame::findFeaturesInArea (:286-311) +
// ORBMatcher::getBestMatch (src/ORBMatcher.cc:967-990) as used by the guided searches (searchByProjection frame-to-frame
// ORBMatcher.cc:265-347, map points to frame :561-612): candidates of a query = the features of the 64x48-px grid cells that
// overlap its search box, rows outer / columns inner / index order inside a cell, filtered by octave range and an exclusion
// mask, scanned with the reference's order-dependent best / second-best rule.  SURVEY 8f, row f1.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"
#include "wave_ops.h"

namespace orbfe {

#define GRID_W 64
#define GRID_H 48
#define ORB_INT_MAX 2147483647

__device__ __forceinline__ int cvfloor_f(float v) {
  const int i = (int)v;
  return i - (i > v);
}

// cell of a coordinate: cvFloor(v / size) as Frame.cc:64-65, clamped to the grid at BOTH ends.  The reference indexes its vector of
// cells unchecked (a negative or too large undistorted coordinate is undefined behaviour there); here such a feature lands in the
// border cell, and a non-finite coordinate in cell 0, so that no index leaves the LDS counters or the scratch lists.
__device__ __forceinline__ int grid_cell(float v, int size, int n) {
  const float q = v / (float)size;
  if (!(q > 0.0f)) return 0;          // negative, -0, NaN
  if (q >= (float)(n - 1)) return n - 1;  // beyond the last cell, +inf
  return cvfloor_f(q);
}

// ---------------------------------------------------------------------------------------------
// grid build: one workgroup per call.  cell_off[ncells+1], cell_feat[n] (features of a cell in ascending index).
// ---------------------------------------------------------------------------------------------
#define GRID_NT 1024
__global__ __launch_bounds__(GRID_NT) void k_grid_build(const orbfe_keypoint* __restrict__ kps, const int32_t* __restrict__ n_kp_ptr,
                                                        int rows, int cols, int feat_in_lds, int32_t* __restrict__ cell_off,
                                                        int32_t* __restrict__ cell_feat) {
  // [ncells + 1] offsets, [ncells] counts / fill cursors, then (feat_in_lds) the [n] unordered feature lists and the [n] cells of the features
  extern __shared__ int32_t l_grid[];
  __shared__ int32_t l_scan[GRID_NT];
  const int tid = threadIdx.x;
  const int n = *n_kp_ptr;
  const int ncells = rows * cols;
  int32_t* l_off = l_grid;
  int32_t* l_cur = l_grid + ncells + 1;
  int32_t* feat = feat_in_lds ? l_cur + ncells : cell_feat;
  int32_t* l_cell = feat + n;  // (feat_in_lds only)
  for (int c = tid; c < ncells; c += GRID_NT) l_cur[c] = 0;
  __syncthreads();
  // The coordinates are read ONCE (a thread's loads independent of each other) and the cell kept in LDS: with 256 threads each of the
  // three passes below walked eight dependent trips to memory, ~1 us each -- 24 of this kernel's 27 us.
  for (int i = tid; i < n; i += GRID_NT) {
    const int cell = grid_cell(kps[i].y, GRID_H, rows) * cols + grid_cell(kps[i].x, GRID_W, cols);
    if (feat_in_lds) l_cell[i] = cell;
    atomicAdd(&l_cur[cell], 1);
  }
  __syncthreads();
  // exclusive prefix over the cells: a thread sums a run of consecutive cells, the run totals are scanned, the thread writes its run's offsets
  const int per = (ncells + GRID_NT - 1) / GRID_NT;
  const int c0 = min(tid * per, ncells), c1 = min(c0 + per, ncells);
  int local = 0;
  for (int c = c0; c < c1; ++c) local += l_cur[c];
  l_scan[tid] = local;
  __syncthreads();
  for (int o = 1; o < GRID_NT; o <<= 1) {
    const int v = tid >= o ? l_scan[tid - o] : 0;
    __syncthreads();
    l_scan[tid] += v;
    __syncthreads();
  }
  {
    int acc = l_scan[tid] - local;
    for (int c = c0; c < c1; ++c) {
      const int k = l_cur[c];
      l_off[c] = acc;
      l_cur[c] = 0;
      acc += k;
    }
    if (tid == GRID_NT - 1) l_off[ncells] = l_scan[GRID_NT - 1];
  }
  __syncthreads();
  for (int c = tid; c <= ncells; c += GRID_NT) cell_off[c] = l_off[c];
  for (int i = tid; i < n; i += GRID_NT) {
    const int cell = feat_in_lds ? l_cell[i] : grid_cell(kps[i].y, GRID_H, rows) * cols + grid_cell(kps[i].x, GRID_W, cols);
    feat[l_off[cell] + atomicAdd(&l_cur[cell], 1)] = i;
  }
  __syncthreads();
  // the reference pushes indices in ascending order (Frame.cc:61-68)
  if (feat_in_lds) {
    // every feature finds its place in its cell's list by counting the smaller indices there: the work of a sort, spread over the threads
    // (one thread sorting the densest cell of a clustered frame was the long pole of the per-cell sort below)
    for (int i = tid; i < n; i += GRID_NT) {
      const int cell = l_cell[i];
      const int base = l_off[cell], k = l_off[cell + 1] - base;
      int rank = 0;
      for (int j = 0; j < k; ++j) rank += feat[base + j] < i ? 1 : 0;
      cell_feat[base + rank] = i;
    }
    return;
  }
  for (int c = tid; c < ncells; c += GRID_NT) {
    int32_t* L = feat + l_off[c];
    const int k = l_off[c + 1] - l_off[c];
    for (int a = 1; a < k; ++a) {
      const int32_t v = L[a];
      int b = a - 1;
      while (b >= 0 && L[b] > v) {
        L[b + 1] = L[b];
        --b;
      }
      L[b + 1] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// search: one wave per query
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int hamming256w(const uint4 a0, const uint4 a1, const uint8_t* __restrict__ p) {
  const uint4 b0 = *(const uint4*)p;
  const uint4 b1 = *(const uint4*)(p + 16);
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
         __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

struct Best2g {
  int min_d, second, min_idx;
};
__device__ __forceinline__ void fold_chunk_g(Best2g& b, int d, int idx) {
  const int incl = wave_incl_scan_dpp<OpMinI>(d);
  const int excl = __builtin_amdgcn_update_dpp(ORB_INT_MAX, incl, ORBFE_DPP_WAVE_SHR1, 0xf, 0xf, false);
  const int pre = min(b.min_d, excl);
  const bool record = d < pre;
  b.second = min(b.second, wave_reduce_dpp<OpMinI>(record ? ORB_INT_MAX : d));
  const int cmin = __builtin_amdgcn_readlane(incl, 63);
  if (cmin < b.min_d) {
    const unsigned long long m = __ballot(d == cmin);
    b.min_d = cmin;
    b.min_idx = __builtin_amdgcn_readlane(idx, __ffsll((long long)m) - 1);
  }
}

__global__ __launch_bounds__(256) void k_search_area(const uint4* __restrict__ kpl, const uint8_t* __restrict__ desc, int width,
                                                     int height, int rows, int cols, const int32_t* __restrict__ cell_off,
                                                     const int32_t* __restrict__ cell_feat, int nq, const float* __restrict__ qxy,
                                                     const float* __restrict__ radius, const int8_t* __restrict__ min_level,
                                                     const int8_t* __restrict__ max_level, const uint8_t* __restrict__ q_desc,
                                                     const uint8_t* __restrict__ exclude, int32_t* __restrict__ best_idx,
                                                     int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist,
                                                     int32_t* __restrict__ n_cand, int32_t* __restrict__ excluded_hits) {
  __shared__ int32_t stage_all[4][128];  // per wave: filtered candidates waiting to fill a 64-lane chunk (order preserved)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int32_t* stage = stage_all[wv];
  const int q = blockIdx.x * 4 + wv;
  if (q >= nq) return;
  const float x = qxy[2 * q], y = qxy[2 * q + 1], rad = radius[q];
  if (rad < 0.f) {  // (the tracking chain marks a map point that takes no part in the search this way)
    if (lane == 0) n_cand[q] = 0, best_idx[q] = -1, best_dist[q] = ORB_INT_MAX, second_dist[q] = ORB_INT_MAX;
    return;
  }
  const int lo = min_level[q], hi = max_level[q];
  const int min_x = max(0, __float2int_rn(x - rad)), max_x = min(width, __float2int_rn(x + rad));
  const int min_y = max(0, __float2int_rn(y - rad)), max_y = min(height, __float2int_rn(y + rad));
  const int c0 = max(0, min(cols - 1, cvfloor_f((float)min_x / (float)GRID_W))), c1 = min(cols - 1, cvfloor_f((float)max_x / (float)GRID_W));
  const int r0 = max(0, min(rows - 1, cvfloor_f((float)min_y / (float)GRID_H))), r1 = min(rows - 1, cvfloor_f((float)max_y / (float)GRID_H));
  const uint4 a0 = *(const uint4*)(q_desc + (size_t)q * 32);
  const uint4 a1 = *(const uint4*)(q_desc + (size_t)q * 32 + 16);
  Best2g b = {ORB_INT_MAX, ORB_INT_MAX, 0};
  int staged = 0, total = 0;
  auto flush = [&](int count) {  // process stage[0..count) (count <= 64) as one chunk
    const int idx = (lane < count) ? stage[lane] : 0;
    const int d = (lane < count) ? hamming256w(a0, a1, desc + (size_t)idx * 32) : ORB_INT_MAX;
    fold_chunk_g(b, d, idx);
  };
  for (int r = r0; r <= r1; ++r)
    for (int c = c0; c <= c1; ++c) {
      const int beg = cell_off[r * cols + c], end = cell_off[r * cols + c + 1];
      for (int i0 = beg; i0 < end; i0 += 64) {
        const int i = i0 + lane;
        int id = 0;
        bool pass = false;
        if (i < end) {
          id = cell_feat[i];
          const int oc = (int)(kpl[id].y & 0xFFu);
          pass = oc <= hi && oc >= lo;
          if (pass && exclude && exclude[id]) {
            // a feature the caller excluded WAS in this query's window: searchByProjection bumps MapPoint::addMatchInTrack once per such
            // (query, feature) occurrence while it filters the candidates (src/ORBMatcher.cc:321-331) -- the counts let the caller do the same
            pass = false;
            if (excluded_hits) atomicAdd(&excluded_hits[id], 1);
          }
        }
        const unsigned long long m = __ballot(pass);
        if (pass) stage[staged + __popcll(m & ((1ull << lane) - 1ull))] = id;
        staged += __popcll(m);
        total += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (staged >= 64) {
          flush(64);
          const int keep = (lane < staged - 64) ? stage[64 + lane] : 0;
          __builtin_amdgcn_wave_barrier();
          if (lane < staged - 64) stage[lane] = keep;
          staged -= 64;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  if (staged > 0) flush(staged);
  if (lane == 0) {
    n_cand[q] = total;
    best_idx[q] = total ? b.min_idx : -1;
    best_dist[q] = b.min_d;
    second_dist[q] = b.second;
  }
}

// ---------------------------------------------------------------------------------------------
// MapPoint::isInVision + MapPoint::predictLevel (src/MapPoint.cc:141-201) for n map points against one frame: what
// ORBMatcher::searchByProjection(frame, mapPoints) evaluates per point before the area search (src/ORBMatcher.cc:575-580).
// One thread per map point; the float / double mix follows the cv::Mat expressions of the reference:
//   Rcw * X + tcw      3x3 * 3x1 gemm: float products summed left to right, then (float)(s * 1.0 + c * 1.0) in double
//   cv::norm, Mat::dot double accumulation of float elements
// Early exits keep the reference's order (z < 0, distance range, image bounds, cos < 0.5); a point that fails leaves visible = 0.
// ---------------------------------------------------------------------------------------------
struct ProjectParams {
  float R[9], t[3];
  float fx, fy, cx, cy;
  float min_u, max_u, min_v, max_v;
  float log_sf;  // std::log(ORBExtractor::mfScaledFactor) as float
  int max_level;
};

__global__ __launch_bounds__(256) void k_project_map_points(int n, const float* __restrict__ pos, const float* __restrict__ vdir,
                                                            const float* __restrict__ max_dist, const float* __restrict__ min_dist,
                                                            ProjectParams P, float* __restrict__ uv, float* __restrict__ dist_out,
                                                            float* __restrict__ cos_out, int8_t* __restrict__ level_out,
                                                            uint8_t* __restrict__ visible) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float X0 = pos[3 * i], X1 = pos[3 * i + 1], X2 = pos[3 * i + 2];
  float pc[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float s = P.R[3 * r] * X0 + P.R[3 * r + 1] * X1 + P.R[3 * r + 2] * X2;
    pc[r] = (float)((double)s + (double)P.t[r]);
  }
  uint8_t vis = 0;
  float u = 0.f, v = 0.f, distance = 0.f, cos_theta = 0.f;
  int level = 0;
  do {
    if (pc[2] < 0.f) break;
    const float x = pc[0], y = pc[1], z = pc[2];
    distance = sqrtf(x * x + y * y + z * z);  // (correctly rounded; __fsqrt_rn maps to the native approximation)
    if (!(distance < max_dist[i] && distance > min_dist[i])) break;  // isGoodDistance
    u = x / z * P.fx + P.cx;
    v = y / z * P.fy + P.cy;
    if (!(u < P.max_u && v < P.max_v && u > P.min_u && v > P.min_v)) break;  // VirtualFrame::isInImage
    const float D0 = vdir[3 * i], D1 = vdir[3 * i + 1], D2 = vdir[3 * i + 2];
    float vd[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) vd[r] = P.R[3 * r] * D0 + P.R[3 * r + 1] * D1 + P.R[3 * r + 2] * D2;
    const double nn = (double)vd[0] * (double)vd[0] + (double)vd[1] * (double)vd[1] + (double)vd[2] * (double)vd[2];
    const float vabs = (float)sqrt(nn);
    const double dot = (double)vd[0] * (double)pc[0] + (double)vd[1] * (double)pc[1] + (double)vd[2] * (double)pc[2];
    cos_theta = (float)(dot / (double)(distance * vabs));
    if (cos_theta < 0.5f) break;
    vis = 1;
    // predictLevel: cvRound(std::log(nMaxDis / distance) / std::log(mfScaledFactor)), clamped to [0, 7] in the reference (max_level here)
    const float lr = (float)log((double)(max_dist[i] / distance));
    level = __float2int_rn(lr / P.log_sf);
    level = level < 0 ? 0 : (level > P.max_level ? P.max_level : level);
  } while (false);
  uv[2 * i] = u, uv[2 * i + 1] = v;
  dist_out[i] = distance, cos_out[i] = cos_theta;
  level_out[i] = (int8_t)level;
  visible[i] = vis;
}

// ---------------------------------------------------------------------------------------------
// The tracking chain (Tracking::trackLocalMap, src/Tracking.cc:641-675): ORBMatcher::searchByProjection(frame, map points, th)
// (src/ORBMatcher.cc:561-612) -> Optimizer::OptimizePoseOnly(frame) (src/Optimizer.cc:33-178) with every intermediate on the device.
// k_track_queries: per map point, the search window the reference derives from isInVision / predictLevel (:575-582).
// k_track_claim / k_track_edges: the reference assigns in map-point order -- a feature that holds a (good) map point keeps it, a free
// feature goes to the FIRST map point whose best match it is (later ones find it taken): the minimum map-point index per feature,
// an atomicMin.  The pose-only edges are the features that hold a good map point afterwards, in feature order (Optimizer.cc:59-118).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_track_queries(int n, const uint8_t* __restrict__ mp_flags, const uint8_t* __restrict__ visible,
                                                       const float* __restrict__ cos_theta, const int8_t* __restrict__ level, float th,
                                                       const float* __restrict__ level_sigma2, int n_levels, float* __restrict__ radius,
                                                       int8_t* __restrict__ min_level, int8_t* __restrict__ max_level) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = -1.f;
  int lo = 0, hi = 0;
  if ((mp_flags[i] & 5) == 5 && visible[i]) {  // in the search list, !isBad, isInMap (ORBMatcher.cc:575-576)
    const int oc = level[i];
    const float base = cos_theta[i] > 0.998f ? 2.5f : 4.0f;
    r = (base * th) * level_sigma2[oc];  // findFeaturesInArea(kp, radius * th, ..): radius * getScaledFactor2(octave) (Frame.cc:289)
    lo = max(0, oc - 1), hi = min(n_levels - 1, oc + 1);
  }
  radius[i] = r;
  min_level[i] = (int8_t)lo, max_level[i] = (int8_t)hi;
}

__global__ __launch_bounds__(256) void k_track_claim(int n, const int32_t* __restrict__ n_cand, const int32_t* __restrict__ best_idx,
                                                     const int32_t* __restrict__ best_dist, const int32_t* __restrict__ second_dist,
                                                     int min_threshold, float ratio, int32_t* __restrict__ claim, int last_wins,
                                                     int32_t* __restrict__ n_accept, uint8_t* __restrict__ accepted) {
  // last_wins (the frame <- frame search, ORBMatcher.cc:265-347): every accepted query is a match and setMapPoints assigns them in query
  // order (:815-830) -- the LAST query that picked a feature keeps it (atomicMax over claims that start at -1), and the return value counts
  // the accepted queries (n_accept), not the features
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || n_cand[i] <= 0) return;
  const float fr = (float)best_dist[i] / (float)second_dist[i];  // getBestMatch's ratio (ORBMatcher.cc:988); second = INT_MAX -> ~0
  if (best_dist[i] < min_threshold && fr < ratio) {
    if (last_wins) {
      atomicMax(&claim[best_idx[i]], i);
      atomicAdd(n_accept, 1);
      if (accepted) accepted[i] = 1;  // (setMapPoints calls addMatchInTrack for every match, kept or overwritten: the caller needs them all)
    } else
      atomicMin(&claim[best_idx[i]], i);
  }
}

// One workgroup: final assignment per feature, the reference's match count, and the pose-only edge list in feature order.
// held[f]: map point the feature holds on entry (-1 none); mp_flags bit 0: !isBad && isInMap (the occupancy test, ORBMatcher.cc:595, and
// the search loop's own filter, :575), bit 1: !isBad (the initial count, :566-571, and the edge test, Optimizer.cc:64), bit 2: member of
// the list searchByProjection walks.
__global__ __launch_bounds__(1024) void k_track_edges(const orbfe_keypoint* __restrict__ kps, const int32_t* __restrict__ n_kp_ptr, int n_features,
                                                      const int32_t* __restrict__ held, const int32_t* __restrict__ claim,
                                                      const uint8_t* __restrict__ mp_flags, const float* __restrict__ mp_pos,
                                                      const double* __restrict__ right_u, const float* __restrict__ level_sigma2,
                                                      const float* __restrict__ level_inv_sigma2, int min_matches,
                                                      int32_t* __restrict__ assigned, int32_t* __restrict__ edge_of, double* __restrict__ Xw,
                                                      double* __restrict__ meas, double* __restrict__ info, float* __restrict__ sigma2,
                                                      int32_t* __restrict__ counts /*[0] n_matches, [1] n_edges (-1: below min_matches)*/,
                                                      int32_t unclaimed, const int32_t* __restrict__ n_accept, int base_matches) {
  // unclaimed: the value a claim starts with (0x7F7F7F7F under atomicMin, -1 under atomicMax).  n_accept (nullable, the frame <- frame
  // search): the matches of this pass are the accepted queries counted there, on top of base_matches from an earlier pass
  __shared__ int s_wave[16], s_match[16];
  __shared__ int s_base;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int n_kp = min(*n_kp_ptr, n_features);
  if (t == 0) s_base = 0;
  int total_matches = 0;
  __syncthreads();
  for (int f0 = 0; f0 < n_features; f0 += 1024) {
    const int f = f0 + t;
    int mp = -1;
    bool c_held = false, c_new = false, edge = false;
    if (f < n_kp) {
      const int h = held ? held[f] : -1;
      c_held = h >= 0 && (mp_flags[h] & 2);  // `if (pMp && !pMp->isBad()) ++nMatches` (ORBMatcher.cc:566-571)
      mp = h;
      const int cl = claim[f];
      if (cl != unclaimed && !(h >= 0 && (mp_flags[h] & 1))) mp = cl, c_new = true;  // setMapPoint + ++nMatches (:595-599)
      edge = mp >= 0 && (mp_flags[mp] & 2);
    }
    if (f < n_features) assigned[f] = mp;
    const unsigned long long me = __ballot(edge);
    const int wm = __popcll(__ballot(c_held)) + __popcll(__ballot(c_new));
    if (lane == 0) s_wave[wv] = __popcll(me), s_match[wv] = wm;
    __syncthreads();
    int before = s_base, chunk_edges = 0, chunk_matches = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < wv) before += s_wave[w];
      chunk_edges += s_wave[w], chunk_matches += s_match[w];
    }
    total_matches += chunk_matches;
    const int pos = before + __popcll(me & ((1ull << lane) - 1ull));
    if (f < n_features) edge_of[f] = edge ? pos : -1;
    if (edge) {
      const orbfe_keypoint k = kps[f];
      Xw[3 * pos] = (double)mp_pos[3 * mp], Xw[3 * pos + 1] = (double)mp_pos[3 * mp + 1], Xw[3 * pos + 2] = (double)mp_pos[3 * mp + 2];
      const double ru = right_u ? right_u[f] : -1.0;
      meas[3 * pos] = (double)k.x, meas[3 * pos + 1] = (double)k.y, meas[3 * pos + 2] = ru < 0 ? -1.0 : ru;
      const int oc = k.octave;
      info[pos] = (double)level_inv_sigma2[oc];
      sigma2[pos] = level_sigma2[oc];
    }
    __syncthreads();
    if (t == 0) s_base += chunk_edges;
    __syncthreads();
  }
  if (t == 0) {
    if (n_accept) total_matches = base_matches + *n_accept;
    counts[0] = total_matches;
    counts[1] = total_matches < min_matches ? -1 : s_base;  // Tracking::trackLocalMap returns before the optimisation (Tracking.cc:656-657)
  }
}

void launch_track_queries(hipStream_t s, int n, const uint8_t* d_flags, const uint8_t* d_visible, const float* d_cos, const int8_t* d_level, float th,
                          const float* d_sigma2, int n_levels, float* d_radius, int8_t* d_min_level, int8_t* d_max_level) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_track_queries, dim3((n + 255) / 256), dim3(256), 0, s, n, d_flags, d_visible, d_cos, d_level, th, d_sigma2, n_levels, d_radius,
                     d_min_level, d_max_level);
}
void launch_track_claim(hipStream_t s, int n, const int32_t* d_n_cand, const int32_t* d_best_idx, const int32_t* d_best_dist, const int32_t* d_second,
                        int min_threshold, float ratio, int32_t* d_claim, int last_wins, int32_t* d_n_accept, uint8_t* d_accepted) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_track_claim, dim3((n + 255) / 256), dim3(256), 0, s, n, d_n_cand, d_best_idx, d_best_dist, d_second, min_threshold, ratio, d_claim,
                     last_wins, d_n_accept, d_accepted);
}
void launch_track_edges(hipStream_t s, const orbfe_keypoint* d_kps, const int32_t* d_n_kp, int n_features, const int32_t* d_held, const int32_t* d_claim,
                        const uint8_t* d_mp_flags, const float* d_mp_pos, const double* d_right_u, const float* d_sigma2, const float* d_inv_sigma2,
                        int min_matches, int32_t* d_assigned, int32_t* d_edge_of, double* d_Xw, double* d_meas, double* d_info, float* d_sig,
                        int32_t* d_counts, int32_t unclaimed, const int32_t* d_n_accept, int base_matches) {
  hipLaunchKernelGGL(k_track_edges, dim3(1), dim3(1024), 0, s, d_kps, d_n_kp, n_features, d_held, d_claim, d_mp_flags, d_mp_pos, d_right_u, d_sigma2,
                     d_inv_sigma2, min_matches, d_assigned, d_edge_of, d_Xw, d_meas, d_info, d_sig, d_counts, unclaimed, d_n_accept, base_matches);
}

void launch_project_map_points(hipStream_t s, int n, const float* d_pos, const float* d_vdir, const float* d_max, const float* d_min,
                               const float* R, const float* t, const float* cam4, const float* bounds4, float log_sf, int max_level,
                               float* d_uv, float* d_dist, float* d_cos, int8_t* d_level, uint8_t* d_vis) {
  if (n <= 0) return;
  ProjectParams P;
  for (int k = 0; k < 9; ++k) P.R[k] = R[k];
  for (int k = 0; k < 3; ++k) P.t[k] = t[k];
  P.fx = cam4[0], P.fy = cam4[1], P.cx = cam4[2], P.cy = cam4[3];
  P.min_u = bounds4[0], P.max_u = bounds4[1], P.min_v = bounds4[2], P.max_v = bounds4[3];
  P.log_sf = log_sf, P.max_level = max_level;
  hipLaunchKernelGGL(k_project_map_points, dim3((n + 255) / 256), dim3(256), 0, s, n, d_pos, d_vdir, d_max, d_min, P, d_uv, d_dist, d_cos,
                     d_level, d_vis);
}

void launch_grid_build(hipStream_t s, const orbfe_keypoint* d_kps, const int32_t* d_n_kp, int n_cap, int rows, int cols,
                       int32_t* d_cell_off, int32_t* d_cell_feat) {
  // n_cap: upper bound of *d_n_kp; the cells' lists are filled and ordered in LDS when they fit there
  const size_t base = (size_t)(2 * rows * cols + 1) * sizeof(int32_t);
  const size_t lists = (size_t)n_cap * 2 * sizeof(int32_t);
  const bool in_lds = base + lists <= 56 * 1024;
  hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(GRID_NT), in_lds ? base + lists : base, s, d_kps, d_n_kp, rows, cols, in_lds ? 1 : 0,
                     d_cell_off, d_cell_feat);
}

void launch_search_area(hipStream_t s, const uint4* d_kpl, const uint8_t* d_desc, int width, int height, int rows, int cols,
                        const int32_t* d_cell_off, const int32_t* d_cell_feat, int nq, const float* d_qxy, const float* d_radius,
                        const int8_t* d_min_level, const int8_t* d_max_level, const uint8_t* d_q_desc, const uint8_t* d_exclude,
                        int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second, int32_t* d_n_cand, int32_t* d_excluded_hits) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(k_search_area, dim3((nq + 3) / 4), dim3(256), 0, s, d_kpl, d_desc, width, height, rows, cols, d_cell_off,
                     d_cell_feat, nq, d_qxy, d_radius, d_min_level, d_max_level, d_q_desc, d_exclude, d_best_idx, d_best_dist, d_second,
                     d_n_cand, d_excluded_hits);
}

}  // namespace orbfe
