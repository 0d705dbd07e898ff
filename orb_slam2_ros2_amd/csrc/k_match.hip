// k_match.hip -- Hamming-256 best/second-best matching on gfx950: masked brute force (one wavefront per
// query) and the stereo epipolar matcher with 11x11 SAD sub-pixel refinement (one wavefront per left
// keypoint).
//
// Replaces ORBMatcher::descDistance (src/ORB_SLAM2/src/ORBMatcher.cc:941-956), getBestMatch (:967-990),
// createRowIndexDB (:915-932), searchByStereo (:18-81), pixelSADMatch / SAD / getPitch (:841-905, :1002-1011).
//
// getBestMatch is order dependent (quirk Q6): when a new minimum is found the old minimum is NOT demoted
// to second best.  Equivalent order-free form used here, evaluated 64 candidates at a time in list order:
//   best   = first index attaining the global minimum
//   second = min over the candidates that are not strict prefix-minimum records at their position
// A candidate is a record iff d < min(all earlier d); the exclusive prefix-min inside a 64-chunk is a
// 6-step wave scan, the carry between chunks is a scalar.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"
#include "rowtable_body.h"
#include "wave_ops.h"

#ifndef STEREO4_FROM
#define STEREO4_FROM 4  // pairs per launch from which the batch matcher is used
#endif
#ifndef STEREO_ROWS
#ifndef STEREO_SAD_GRID
#define STEREO_SAD_GRID 32  // workgroups per pair of k_stereo_sad (each loops over the work list): 512 entries per sweep
#endif
#ifndef STEREO_SAD_WAVES
#define STEREO_SAD_WAVES 4  // waves per workgroup of k_stereo_sad (the waves never meet after the level table; the launch keeps 4 x STEREO_SAD_GRID waves per pair)
#endif
#define STEREO_ROWS 1   // the batch matcher: 1 = row-parallel (k_stereo_rows + k_stereo_sad), 0 = a wave per left keypoint everywhere (k_stereo)
#endif

namespace orbfe {

#define ORB_INT_MAX 2147483647

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_step(int v) {
  // lanes without a source keep INT_MAX (the identity of min)
  return min(v, __builtin_amdgcn_update_dpp(ORB_INT_MAX, v, CTRL, ROW_MASK, 0xf, false));
}
// inclusive prefix-min over the 64 lanes (all lanes active): Hillis-Steele inside each row of 16 with DPP row
// shifts, then the row totals are broadcast into the following rows.  Lane 63 holds the wave minimum.
__device__ __forceinline__ int wave_incl_prefix_min(int v) {
  v = dpp_min_step<0x111, 0xf>(v);  // row_shr:1
  v = dpp_min_step<0x112, 0xf>(v);  // row_shr:2
  v = dpp_min_step<0x114, 0xf>(v);  // row_shr:4
  v = dpp_min_step<0x118, 0xf>(v);  // row_shr:8
  v = dpp_min_step<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_min_step<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  return v;
}
__device__ __forceinline__ int wave_min_i(int v) { return __builtin_amdgcn_readlane(wave_incl_prefix_min(v), 63); }

struct Best2 {
  int min_d, second, min_idx;
};

// Fold one chunk of up to 64 candidates (lane order = list order) into the running (min, idx, second).
// d = distance of this lane's candidate or INT_MAX if the lane holds none; idx = its train index.
__device__ __forceinline__ void fold_chunk(Best2& b, int d, int idx, int lane) {
  const int incl = wave_incl_prefix_min(d);
  const int excl = __builtin_amdgcn_update_dpp(ORB_INT_MAX, incl, 0x138, 0xf, 0xf, false);  // wave_shr:1, lane 0 <- INT_MAX
  const int pre = min(b.min_d, excl);
  const bool record = d < pre;  // strict prefix-minimum record: becomes the new best, never the second best
  b.second = min(b.second, wave_min_i(record ? ORB_INT_MAX : d));
  const int cmin = __builtin_amdgcn_readlane(incl, 63);
  if (cmin < b.min_d) {
    const unsigned long long m = __ballot(d == cmin);
    b.min_d = cmin;
    b.min_idx = __builtin_amdgcn_readlane(idx, __ffsll((long long)m) - 1);
  }
}

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint8_t* __restrict__ p) {
  const uint4 b0 = *(const uint4*)p;
  const uint4 b1 = *(const uint4*)(p + 16);
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
         __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// ---------------------------------------------------------------------------------------------
// brute force: query i scans cand_idx[off[i]..off[i+1]) (or 0..nt-1) in order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_match_bruteforce(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t,
                                                          int nt, const uint32_t* __restrict__ off,
                                                          const uint32_t* __restrict__ cand, int32_t* __restrict__ best_idx,
                                                          int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nq) return;
  const uint4 a0 = *(const uint4*)(q + (size_t)i * 32);
  const uint4 a1 = *(const uint4*)(q + (size_t)i * 32 + 16);
  const int begin = off ? (int)off[i] : 0;
  const int end = off ? (int)off[i + 1] : nt;
  Best2 b = {ORB_INT_MAX, ORB_INT_MAX, 0};
  for (int c0 = begin; c0 < end; c0 += 64) {
    const int c = c0 + lane;
    int d = ORB_INT_MAX, idx = 0;
    if (c < end) {
      idx = off ? (int)cand[c] : c;
      d = hamming256(a0, a1, t + (size_t)idx * 32);
    }
    fold_chunk(b, d, idx, lane);
  }
  if (lane == 0) {
    best_idx[i] = (end > begin) ? b.min_idx : -1;
    best_dist[i] = b.min_d;
    second_dist[i] = b.second;
  }
}

// ---------------------------------------------------------------------------------------------
// stereo: one wave per left keypoint of pair blockIdx.y
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int cv_floor_f(float v) {
  const int i = (int)v;
  return i - (i > v);
}

// buffers of the row-parallel matcher of the batches (k_stereo_rows + k_stereo_sad), per pair; lrow_off == nullptr: not in use
struct StereoRows {
  uint32_t* lrow_off;   // [pairs][rows + 1]: the LEFT keypoints by image row (round(y)), offsets
  uint16_t* lrow_list;  // [pairs][n_features]: ... and the list
  uint4* work;          // [pairs][n_features]: the left keypoints whose best candidate goes on to the SAD refinement
  int32_t* work_n;      // [pairs]
  double *right_u, *depth;          // the pair outputs (k_rowtable writes their defaults)
  int32_t *best_right, *best_dist;
};

// ---------------------------------------------------------------------------------------------
// row table of one right image (createRowIndexDB, ORBMatcher.cc:915-932): for every image row the right keypoints whose band
// [row_min, row_max) holds it, as offsets[rows + 1] + one flat list (counting sort in LDS, one workgroup per image).  The order
// inside a row is whatever the atomics produce: the matcher below reduces (distance, index) keys, which does not depend on it.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rowtable(const KpAux* __restrict__ aux, const int32_t* __restrict__ n_kp, int n_features, int rows,
                                                  int list_cap, uint32_t* __restrict__ rowoff, uint16_t* __restrict__ rowlist, int slot_r0,
                                                  int slot_step, int pair0, int32_t* __restrict__ n_match, const orbfe_keypoint* __restrict__ kps,
                                                  int slot_l0, StereoRows sr) {
  extern __shared__ uint32_t s_rt[];  // cnt[rows] | part[256]
  uint32_t* cnt = s_rt;
  uint32_t* part = s_rt + rows;
  const int tid = threadIdx.x;
  const int slot = slot_r0 + blockIdx.x * slot_step;
  const int pair = pair0 + blockIdx.x;
  if (tid == 0) n_match[pair] = 0;  // k_stereo counts into it (a memset in front of this kernel was one more launch)
  const KpAux* A = aux + (size_t)slot * n_features;
  uint32_t* RO = rowoff + (size_t)pair * (rows + 1);
  uint16_t* RL = rowlist + (size_t)pair * list_cap;
  const int nr = min(n_kp[slot], n_features);
  rowtable_build(A, nr, rows, list_cap, RO, RL, cnt, part, tid);
  if (!sr.lrow_off) return;
  // ---- the row-parallel matcher's inputs (k_stereo_rows): the LEFT keypoints by image row -- LO[rows + 1] offsets + LL list, the same counting
  //      sort on round(y) (ORBMatcher.cc:38-41: rowIdxDB[cvRound(kp.pt.y)]) -- and the defaults of the pair's outputs: a left keypoint that no
  //      row wave reaches (its row is outside the image; slots past the count) and one without a match stay at -1
  __syncthreads();
  {
    const int sl = slot_l0 + blockIdx.x * slot_step;
    const orbfe_keypoint* LK = kps + (size_t)sl * n_features;
    const int nl = min(n_kp[sl], n_features);
    uint32_t* LO = sr.lrow_off + (size_t)pair * (rows + 1);
    uint16_t* LL = sr.lrow_list + (size_t)pair * n_features;
    if (tid == 0) sr.work_n[pair] = 0;
    for (int y = tid; y < rows; y += 256) cnt[y] = 0;
    __syncthreads();
    for (int i0 = tid; i0 < n_features; i0 += 4 * 256) {
      float y[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = LK[min(i0 + 256 * k, n_features - 1)].y;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + 256 * k;
        if (i < n_features) {
          const size_t o = (size_t)pair * n_features + i;
          sr.right_u[o] = -1.0, sr.depth[o] = -1.0, sr.best_right[o] = -1, sr.best_dist[o] = -1;
          const int row = __float2int_rn(y[k]);
          if (i < nl && (unsigned)row < (unsigned)rows) atomicAdd(&cnt[row], 1u);
        }
      }
    }
    __syncthreads();
    // exclusive prefix over the rows (as rowtable_build: a run of rows per thread, the run totals scanned per wave, four wave totals through LDS)
    const int per = (rows + 255) >> 8;
    const int y0 = tid * per, y1 = min(y0 + per, rows);
    uint32_t sum = 0;
    for (int y = y0; y < y1; ++y) sum += cnt[y];
    const uint32_t incl_w = (uint32_t)wave_incl_scan_dpp<OpAddI>((int)sum);
    if ((tid & 63) == 63) part[tid >> 6] = incl_w;
    __syncthreads();
    uint32_t wave_base = 0;
    for (int k = 0; k < (tid >> 6); ++k) wave_base += part[k];
    const uint32_t total_all = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
    uint32_t run = wave_base + incl_w - sum;
    for (int y = y0; y < y1; ++y) {
      const uint32_t c = cnt[y];
      cnt[y] = run;
      LO[y] = run;
      run += c;
    }
    if (tid == 255) LO[rows] = total_all;
    __syncthreads();
    for (int i0 = tid; i0 < nl; i0 += 4 * 256) {
      float y[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = LK[min(i0 + 256 * k, nl - 1)].y;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + 256 * k;
        const int row = __float2int_rn(y[k]);
        if (i < nl && (unsigned)row < (unsigned)rows) LL[atomicAdd(&cnt[row], 1u)] = (uint16_t)i;
      }
    }
  }
}

// results of ONE pair delivered to page-locked host memory by the kernel itself (all nullable; index = left keypoint): the single-pair
// entry point queued up to five device-to-host copies behind the kernel otherwise
struct StereoHost {
  double *right_u, *depth;
  int32_t *best_right, *best_dist;
};
__global__ __launch_bounds__(256) void k_stereo(const LevelDev* __restrict__ lv, int n_levels, const uint8_t* __restrict__ pyr, size_t img_pitch,
                                                const orbfe_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                const KpX* __restrict__ kx, const uint32_t* __restrict__ rowoff,
                                                const uint16_t* __restrict__ rowlist, int rows, int list_cap,
                                                const int32_t* __restrict__ n_kp,
                                                int n_features, float fx, float bf, int cols0, int mean_threshold,
                                                double* __restrict__ right_u, double* __restrict__ depth, int32_t* __restrict__ n_match,
                                                int32_t* __restrict__ best_right, int32_t* __restrict__ best_dist, int slot_l0,
                                                int slot_r0, int slot_step, int pair0, StereoHost host, int table_of_slot) {
  // table_of_slot: the row tables are indexed by the RIGHT slot (built at extraction: k_brief), not by the pair
#pragma clang fp contract(off)
  __shared__ uint32_t s_sad[4][11 * 4 + 11 * 7 + 7];  // per wave: left 11 rows x 4 words, right 11 rows x 7 words
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = blockIdx.x * 4 + wv;
  const int pair = pair0 + blockIdx.y;
  const int sl = slot_l0 + blockIdx.y * slot_step, sr = slot_r0 + blockIdx.y * slot_step;
  const size_t out_i = (size_t)pair * n_features + li;
  if (li >= n_features) return;
  const orbfe_keypoint* LK = kps + (size_t)sl * n_features;
  const orbfe_keypoint* RK = kps + (size_t)sr * n_features;
  const uint8_t* LD = desc + (size_t)sl * n_features * 32;
  const uint8_t* RD = desc + (size_t)sr * n_features * 32;
  const KpX* RX = kx + (size_t)sr * n_features;
  const size_t tab = table_of_slot ? (size_t)sr : (size_t)pair;
  const uint32_t* RO = rowoff + tab * (rows + 1);
  const uint16_t* RL = rowlist + tab * list_cap;
  // first round trip: the count, the left keypoint, its descriptor and -- lane = level -- the three per-level constants the SAD
  // stage will want for the two octaves it does not know yet
  // (slot li always exists in the buffers; whether it holds a keypoint is decided after the loads are in flight)
  const int nl = n_kp[sl];
  const orbfe_keypoint l = LK[li];
  const uint4 a0 = *(const uint4*)(LD + (size_t)li * 32);
  const uint4 a1 = *(const uint4*)(LD + (size_t)li * 32 + 16);
  const LevelDev& Lmine = lv[min(lane, n_levels - 1)];
  const float lv_sf = Lmine.sf;
  const uint32_t lv_off = Lmine.plane_off;
  const int lv_stride = Lmine.stride;
  if (li >= nl) {
    if (lane == 0) {
      right_u[out_i] = -1.0;
      depth[out_i] = -1.0;
      best_right[out_i] = -1;
      best_dist[out_i] = -1;
      if (host.right_u) host.right_u[li] = -1.0;
      if (host.depth) host.depth[li] = -1.0;
      if (host.best_right) host.best_right[li] = -1;
      if (host.best_dist) host.best_dist[li] = -1;
    }
    return;
  }
  const float max_u = l.x - 0;
  const float min_u = fmaxf(0.f, l.x - fx);
  const int row = __builtin_amdgcn_readfirstlane(__float2int_rn(l.y));

  // candidates = rowIdxDB[row] filtered by the u-range (ORBMatcher.cc:38-48); the reference takes the FIRST minimum of the list,
  // which is in ascending right index: the minimum of (distance << 16 | index) over the candidates, in any order.  A lane keeps the
  // minimum of the candidates it has seen (one per 64 list entries); one wave reduction after the loop.
  int key = ORB_INT_MAX;
  if ((unsigned)row < (unsigned)rows) {
    const int beg = (int)RO[row], end = (int)RO[row + 1];
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int g = beg; g < end; g += 64) {
      const int c = g + lane;
      const bool has = c < end;
      const uint32_t idx = RL[has ? c : beg];
      const float x = RX[idx].x;
      const int d = hamming256(a0, a1, RD + (idx << 5));
      const bool pass = has && x < max_u && x > min_u;
      key = min(key, pass ? (int)(((uint32_t)d << 16) | idx) : ORB_INT_MAX);
    }
  }
  key = wave_min_i(key);
  const bool any = key != ORB_INT_MAX;
  Best2 b = {key >> 16, ORB_INT_MAX, key & 0xFFFF};
  double out_u = -1.0, out_depth = -1.0;
  int matched = 0;
  if (any && b.min_d <= mean_threshold) {
    const orbfe_keypoint r = RK[b.min_idx];
    if (!(l.octave > r.octave + 1 || l.octave < r.octave - 1)) {
      // ---- pixelSADMatch (ORBMatcher.cc:841-881): 11 SADs of centre-subtracted 11x11 patches ----
      const int ol = __builtin_amdgcn_readfirstlane(l.octave), orr = __builtin_amdgcn_readfirstlane(r.octave);
      const float sf_l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lv_sf), ol));
      const float sf_r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lv_sf), orr));
      const int stride_l = __builtin_amdgcn_readlane(lv_stride, ol), stride_r = __builtin_amdgcn_readlane(lv_stride, orr);
      const uint8_t* IL = pyr + (size_t)sl * img_pitch + (uint32_t)__builtin_amdgcn_readlane((int)lv_off, ol);
      const uint8_t* IR = pyr + (size_t)sr * img_pitch + (uint32_t)__builtin_amdgcn_readlane((int)lv_off, orr);
      const int lx = cv_floor_f(l.x / sf_l), ly = cv_floor_f(l.y / sf_l);  // getPitch (:1004-1006)
      const int rx = cv_floor_f(r.x / sf_r), ry = cv_floor_f(r.y / sf_r);
      // stage the left 11x11 patch (x in [lx-5, lx+5]) and the right 11x21 window (x in [rx-10, rx+10]) in LDS as words
      uint32_t* wl = s_sad[wv];
      uint32_t* wr = s_sad[wv] + 44;
      const int lxa = (lx - 5) & ~3, rxa = (rx - 10) & ~3;
      for (int t = lane; t < 44 + 77; t += 64) {
        if (t < 44) {
          const int rr = t >> 2, cc = t & 3;
          wl[t] = *(const uint32_t*)(IL + (uint32_t)mad24u(ly - 5 + rr, stride_l, lxa + 4 * cc));
        } else {
          const int u = t - 44;
          const int rr = u / 7, cc = u - rr * 7;
          wr[u] = *(const uint32_t*)(IR + (uint32_t)mad24u(ry - 5 + rr, stride_r, rxa + 4 * cc));
        }
      }
      __builtin_amdgcn_wave_barrier();  // LDS accesses of one wave execute in order; this only pins the compiler
      const uint8_t* bl = (const uint8_t*)wl + ((lx - 5) - lxa);
      const uint8_t* br = (const uint8_t*)wr + ((rx - 10) - rxa);
      const int c1 = bl[5 * 16 + 5];
      // lane = part*11 + Lidx, part 0..4 sums rows {part, part+5, (part+10 if part==0)}
      int partial = 0;
      if (lane < 55) {
        const int Lidx = lane % 11, part = lane / 11;
        // each shifted right patch subtracts ITS OWN centre pixel (SAD(), ORBMatcher.cc:901-903)
        const uint32_t c2 = br[5 * 28 + 5 + Lidx];
        uint32_t acc = 0;
        for (int rr = part; rr < 11; rr += 5) {
          const uint8_t* pl = bl + rr * 16;
          const uint8_t* pr = br + rr * 28 + Lidx;
#pragma unroll
          for (int cc = 0; cc < 11; ++cc)  // |(pl - c1) - (pr - c2)| = |(pl + c2) - (pr + c1)|, both sides non-negative: one v_sad_u32
            acc = __builtin_amdgcn_sad_u16((uint32_t)pl[cc] + c2, (uint32_t)pr[cc] + (uint32_t)c1, acc);
        }
        partial = (int)acc;
      }
      int sad = partial;
      sad += __shfl(partial, (lane + 11) & 63);
      sad += __shfl(partial, (lane + 22) & 63);
      sad += __shfl(partial, (lane + 33) & 63);
      sad += __shfl(partial, (lane + 44) & 63);
      // lanes 0..10 now hold SAD(L = lane-5); first minimum wins (strict <, :856)
      const int mine = lane < 11 ? sad : ORB_INT_MAX;
      const int smin = wave_min_i(mine);
      const int best_l = __ffsll((long long)__ballot(mine == smin)) - 1;  // 0..10 == bestL + mnL
      float delta_u = 0.f;
      if (best_l > 0 && best_l < 10) {
        const float s1 = (float)__shfl(sad, best_l - 1);
        const float s2 = (float)__shfl(sad, best_l);
        const float s3 = (float)__shfl(sad, best_l + 1);
        delta_u = (float)(0.5 * (double)(s1 - s3) / (double)(s1 + s3 - 2 * s2));
        if (delta_u < 1 && delta_u > -1) delta_u *= sf_r;
        else delta_u = 0.f;
      }
      float ru = r.x + delta_u;  // bestL itself is not added (quirk Q7)
      ru = fmaxf(0.f, ru);
      ru = fminf(ru, (float)cols0 - 1);
      float delta = l.x - ru;
      bool ok = true;
      if (delta <= 0) {
        ru = r.x;
        delta = l.x - ru;
        if (delta <= 0) ok = false;
      }
      if (ok) {
        out_u = (double)ru;
        out_depth = (double)(bf / (l.x - ru));
        matched = 1;
      }
    }
  }
  if (lane == 0) {
    right_u[out_i] = out_u;
    depth[out_i] = out_depth;
    best_right[out_i] = any ? b.min_idx : -1;
    best_dist[out_i] = any ? b.min_d : -1;
    if (host.right_u) host.right_u[li] = out_u;
    if (host.depth) host.depth[li] = out_depth;
    if (host.best_right) host.best_right[li] = any ? b.min_idx : -1;
    if (host.best_dist) host.best_dist[li] = any ? b.min_d : -1;
    if (matched) atomicAdd(&n_match[pair], 1);
  }
}

#ifndef ST4_PAD
#define ST4_PAD 8
#endif
// 128 words of patches per keypoint + ST4_PAD: with a pitch of exactly 128 words the four keypoints of a wave read the same LDS banks at every
// step of the SAD loop (a four-way conflict on every one of its 242 byte reads per lane)
#define ST4_WORDS (11 * 4 + 11 * 7 + 7 + ST4_PAD)
// k_stereo_sad: 121 words of patches + 11 rows x 8 words of 16-bit left values, + 7: a pitch of 216 words puts the wave's four entries 24 banks apart
#define ST5_WORDS (124 + 88 + 4)

__device__ __forceinline__ int row_min_i(int v, int lane) {  // minimum over the lane's row of 16, in every lane of the row
  v = dpp_min_step<0x111, 0xf>(v);
  v = dpp_min_step<0x112, 0xf>(v);
  v = dpp_min_step<0x114, 0xf>(v);
  v = dpp_min_step<0x118, 0xf>(v);
  return __shfl(v, lane | 15);
}

// ---------------------------------------------------------------------------------------------
// The batches' matcher, ROW-PARALLEL (r5).  searchByStereo's candidates of a left keypoint are rowIdxDB[round(y)] (ORBMatcher.cc:38-48):
// every left keypoint of one image row scans the SAME list.  One wave per image row: its lanes hold the row's right candidates (index, x,
// octave / patch centre, descriptor: loaded ONCE per row), and the row's left keypoints -- the counting sort of k_rowtable -- are taken one
// after the other: the left descriptor travels by v_readlane from the lane that loaded it, the wave reduces distance << 16 | index with one
// DPP minimum (the reference's first minimum of the list in ascending right index, whatever the order of the lanes).  With a wave per four
// left keypoints (k_stereo4) every keypoint fetched its ~43 candidates' descriptors itself -- 86 k gathered records per pair against 16 k
// here -- and the kernel was bound by the rate at which a CU looks up the cache lines of such gathers (each lane another line), not by
// bandwidth or arithmetic.  A left keypoint whose best candidate passes the distance threshold and the octave test (ORBMatcher.cc:50-58) is
// appended to the pair's work list with everything the refinement needs; k_stereo_sad takes it from there.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stereo_rows(const KpX* __restrict__ kx, const uint8_t* __restrict__ desc, const uint32_t* __restrict__ rowoff,
                                                     const uint16_t* __restrict__ rowlist, int rows, int list_cap, int n_features, float fx,
                                                     int mean_threshold, StereoRows sr, int slot_l0, int slot_r0, int slot_step, int pair0) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int y = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wv);
  if (y >= rows) return;
  const int pair = pair0 + blockIdx.y;
  const int sl = slot_l0 + blockIdx.y * slot_step, srt = slot_r0 + blockIdx.y * slot_step;
  const uint32_t* LO = sr.lrow_off + (size_t)pair * (rows + 1);
  const uint32_t* RO = rowoff + (size_t)pair * (rows + 1);
  const int lbeg = (int)LO[y], lend = (int)LO[y + 1];
  if (lbeg >= lend) return;  // no left keypoint in this row
  const int cbeg = (int)RO[y], cend = (int)RO[y + 1];
  const uint16_t* LL = sr.lrow_list + (size_t)pair * n_features;
  const uint16_t* RL = rowlist + (size_t)pair * list_cap;
  const KpX* LX = kx + (size_t)sl * n_features;
  const KpX* RX = kx + (size_t)srt * n_features;
  const uint8_t* LD = desc + (size_t)sl * n_features * 32;
  const uint8_t* RD = desc + (size_t)srt * n_features * 32;
  for (int lg = lbeg; lg < lend; lg += 64) {
    const int nlg = min(64, lend - lg);  // uniform
    // lane j: the j-th left keypoint of the group -- its record and descriptor (one gather per group, then v_readlane per keypoint)
    const uint32_t my_li = (lane < nlg) ? (uint32_t)LL[lg + lane] : (uint32_t)LL[lg];
    const KpX lxq = LX[my_li];
    const uint4 la0 = *(const uint4*)(LD + (my_li << 5));
    const uint4 la1 = *(const uint4*)(LD + (my_li << 5) + 16);
    int my_best = ORB_INT_MAX;
    uint32_t my_wx = 0u, my_wq = 0u;  // x (bits) and q of lane j's best candidate
    for (int c0 = cbeg; c0 < cend; c0 += 64) {
      const bool has = c0 + lane < cend;
      const uint32_t idx = RL[has ? c0 + lane : c0];
      const KpX xq = RX[idx];
      const uint4 b0 = *(const uint4*)(RD + (idx << 5));
      const uint4 b1 = *(const uint4*)(RD + (idx << 5) + 16);
      for (int j = 0; j < nlg; ++j) {
        const uint32_t a0x = (uint32_t)__builtin_amdgcn_readlane((int)la0.x, j), a0y = (uint32_t)__builtin_amdgcn_readlane((int)la0.y, j);
        const uint32_t a0z = (uint32_t)__builtin_amdgcn_readlane((int)la0.z, j), a0w = (uint32_t)__builtin_amdgcn_readlane((int)la0.w, j);
        const uint32_t a1x = (uint32_t)__builtin_amdgcn_readlane((int)la1.x, j), a1y = (uint32_t)__builtin_amdgcn_readlane((int)la1.y, j);
        const uint32_t a1z = (uint32_t)__builtin_amdgcn_readlane((int)la1.z, j), a1w = (uint32_t)__builtin_amdgcn_readlane((int)la1.w, j);
        const float l_x = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(lxq.x), j));
        const int d = __popc(a0x ^ b0.x) + __popc(a0y ^ b0.y) + __popc(a0z ^ b0.z) + __popc(a0w ^ b0.w) + __popc(a1x ^ b1.x) + __popc(a1y ^ b1.y) +
                      __popc(a1z ^ b1.z) + __popc(a1w ^ b1.w);
        const float max_u = l_x - 0;                 // ORBMatcher.cc:42-43 (minD = 0, maxD = fx: the reference's own constants)
        const float min_u = fmaxf(0.f, l_x - fx);
        const bool pass = has && xq.x < max_u && xq.x > min_u;
        const int key = pass ? (int)(((uint32_t)d << 16) | idx) : ORB_INT_MAX;
        const int m = wave_min_i(key);
        if (m < ORB_INT_MAX) {  // uniform
          const int w = __ffsll((long long)__ballot(key == m)) - 1;  // the one lane that holds it (indices are distinct)
          const uint32_t wx = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(xq.x), w);
          const uint32_t wq = (uint32_t)__builtin_amdgcn_readlane((int)xq.q, w);
          if (lane == j && m < my_best) my_best = m, my_wx = wx, my_wq = wq;
        }
      }
    }
    // lane j: its keypoint's result
    const bool mine = lane < nlg;
    const bool any = mine && my_best != ORB_INT_MAX;
    const int min_d = my_best >> 16, min_idx = my_best & 0xFFFF;
    const size_t o = (size_t)pair * n_features + my_li;
    if (any) sr.best_right[o] = min_idx, sr.best_dist[o] = min_d;
    const int l_oct = ORBFE_KPX_OCT(lxq.q), r_oct = ORBFE_KPX_OCT(my_wq);
    const bool go = any && min_d <= mean_threshold && !(l_oct > r_oct + 1 || l_oct < r_oct - 1);
    const unsigned long long gm = __ballot(go);
    if (gm) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&sr.work_n[pair], __popcll(gm));
      base = __builtin_amdgcn_readfirstlane(base);
      if (go) sr.work[(size_t)pair * n_features + base + __popcll(gm & ((1ull << lane) - 1ull))] = make_uint4(my_li | ((uint32_t)min_idx << 16), lxq.q, my_wx, my_wq);
    }
  }
}

// pixelSADMatch (ORBMatcher.cc:841-881) + the sub-pixel / disparity bookkeeping of searchByStereo (:59-78) for the entries of the pair's work
// list: FOUR entries per wave, one per row of 16 lanes -- the two patches staged in LDS, a keypoint's 11 SADs as 11 lanes with a whole
// patch each (k_stereo4's second half; the entry carries the patch centres and octaves, so the windows are the first thing requested).
__global__ __launch_bounds__(64 * STEREO_SAD_WAVES) void k_stereo_sad(const LevelDev* __restrict__ lv, int n_levels, const uint8_t* __restrict__ pyr, size_t img_pitch,
                                                    const KpX* __restrict__ kx, int n_features, float bf, int cols0, StereoRows sr,
                                                    int32_t* __restrict__ n_match, int slot_l0, int slot_r0, int slot_step, int pair0) {
#pragma clang fp contract(off)
  // per wave, per entry: left 11 rows x 4 words | right 11 rows x 7 words | the left patch again as 16-bit values (pl - c1 + 255), 11 rows of 8 words
  __shared__ __attribute__((aligned(16))) uint32_t s_sad[STEREO_SAD_WAVES][4][ST5_WORDS];
  __shared__ float s_sf[16];
  __shared__ uint32_t s_off[16];
  __shared__ int s_stride[16];
  const int pair = pair0 + blockIdx.y;
  const int n_work = sr.work_n[pair];
  if ((int)blockIdx.x * (4 * STEREO_SAD_WAVES) >= n_work) return;  // (uniform for the block)
  // A workgroup takes sixteen entries and then the sixteen gridDim.x workgroups further on: the launch is sized for a fraction of the
  // worst case (every feature matched) -- sized for all of it, three quarters of its workgroups found nothing to do, each holding 14 KB
  // of LDS and four wave slots for the round trip of the counter above, beside the next batch's FAST, whose throughput follows the
  // waves it can keep resident.
  if ((int)threadIdx.x < min(n_levels, 16)) {
    s_sf[threadIdx.x] = lv[threadIdx.x].sf;
    s_off[threadIdx.x] = lv[threadIdx.x].plane_off;
    s_stride[threadIdx.x] = lv[threadIdx.x].stride;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, grp = lane >> 4, sub = lane & 15;
  const int sl = slot_l0 + blockIdx.y * slot_step, srt = slot_r0 + blockIdx.y * slot_step;
  __syncthreads();  // (the level table; every wave of the block passes here)
  for (int bx = blockIdx.x; bx * (4 * STEREO_SAD_WAVES) < n_work; bx += gridDim.x) {  // (uniform for the block)
  const int wi = (bx * STEREO_SAD_WAVES + wv) * 4 + grp;
  const bool go = wi < n_work;
  const uint4 e = sr.work[(size_t)pair * n_features + (go ? wi : 0)];
  const uint32_t li = e.x & 0xFFFFu;
  const float l_x = kx[(size_t)sl * n_features + li].x;  // (needed at the very end: it travels with the windows)
  const float r_x = __uint_as_float(e.z);
  const uint32_t l_q = e.y, r_q = e.w;
  double out_u = -1.0, out_depth = -1.0;
  int matched = 0;
  {
    const int ol = go ? ORBFE_KPX_OCT(l_q) : 0, orr = go ? ORBFE_KPX_OCT(r_q) : 0;
    const float sf_r = s_sf[orr];
    const int stride_l = s_stride[ol], stride_r = s_stride[orr];
    const uint8_t* IL = pyr + (size_t)sl * img_pitch;  // wave-uniform bases; the rest of the address is a 32-bit offset
    const uint8_t* IR = pyr + (size_t)srt * img_pitch;
    const uint32_t off_l = s_off[ol], off_r = s_off[orr];
    // (an idle row of lanes reads around (16, 16) of level 0: harmless)
    const int lx = go ? ORBFE_KPX_QX(l_q) : 16, ly = go ? ORBFE_KPX_QY(l_q) : 16;  // getPitch (:1004-1006), computed with the keypoint (KpX)
    const int rx = go ? ORBFE_KPX_QX(r_q) : 16, ry = go ? ORBFE_KPX_QY(r_q) : 16;
    uint32_t* wl = s_sad[wv][grp];
    uint32_t* wr = wl + 44;
    const int lxa = (lx - 5) & ~3, rxa = (rx - 10) & ~3;
    uint32_t wreg[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int t = min(sub + 16 * it, 120);
      if (t < 44) {
        const int rr = t >> 2, cc = t & 3;
        wreg[it] = *(const uint32_t*)(IL + (off_l + (uint32_t)mad24u(ly - 5 + rr, stride_l, lxa + 4 * cc)));
      } else {
        const int u = t - 44;
        const int rr = (u * 37) >> 8, cc = u - rr * 7;  // u / 7 for u < 77
        wreg[it] = *(const uint32_t*)(IR + (off_r + (uint32_t)mad24u(ry - 5 + rr, stride_r, rxa + 4 * cc)));
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int t = sub + 16 * it;
      if (t < 121) wl[t] = wreg[it];  // (wr = wl + 44: one index space)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // LDS accesses of one wave execute in order; this only pins the compiler
    const uint8_t* bl = (const uint8_t*)wl + ((lx - 5) - lxa);
    const int c1 = bl[5 * 16 + 5];
    // SAD() subtracts each patch's OWN centre pixel (ORBMatcher.cc:893-905): sum |(pl - c1) - (pr - c2)| = sum |(pl - c1 + 255) - (pr - c2 + 255)|,
    // both sides in 0 .. 510.  The left side does not depend on the shift: the row's 16 lanes write it ONCE as 16-bit values, two per word
    // (a row = 11 values + a zero = 6 words, pitch 8), so that a lane's row is three LDS reads on the left, two on the right and
    // v_sad_u16 on PAIRS of pixels -- byte by byte a lane issued 242 LDS reads and 363 vector instructions per patch (r5: this kernel is
    // what is left of the match on images with many stereo matches).
    uint16_t* lp16 = (uint16_t*)(wl + 124);  // (16-byte aligned: the rows are read as 16 + 8 bytes)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int t = sub + 16 * it;
      if (t < 121) {
        const int rr = (t * 373) >> 12, cc = t - rr * 11;  // t / 11 for t < 121
        lp16[rr * 16 + cc] = (uint16_t)((int)bl[rr * 16 + cc] - c1 + 255);
      }
    }
    if (sub < 11) lp16[sub * 16 + 11] = 0;  // the twelfth value of a row: zero on both sides
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int sad = ORB_INT_MAX;
    if (sub < 11) {
      const int s_off_b = ((rx - 10) - rxa) + sub;       // byte offset of this lane's shifted patch inside a staged right row (0 .. 13)
      const uint8_t* brc = (const uint8_t*)wr + s_off_b;
      const uint32_t k2 = 255u - (uint32_t)brc[5 * 28 + 5];  // 255 - c2
      const uint32_t K = k2 | (k2 << 16);
      const uint32_t* rw = wr + (s_off_b >> 2);
      const uint32_t sh = (uint32_t)(s_off_b & 3);
      const uint32_t* lw = wl + 124;
      uint32_t acc = 0;
#pragma unroll
      for (int rr = 0; rr < 11; ++rr) {
        const uint32_t r0 = rw[rr * 7], r1 = rw[rr * 7 + 1], r2 = rw[rr * 7 + 2], r3 = rw[rr * 7 + 3];
        const uint32_t R0 = __builtin_amdgcn_alignbyte(r1, r0, sh), R1 = __builtin_amdgcn_alignbyte(r2, r1, sh), R2 = __builtin_amdgcn_alignbyte(r3, r2, sh);
        const uint4 q03 = *(const uint4*)(lw + rr * 8);
        const uint2 q45 = *(const uint2*)(lw + rr * 8 + 4);
        acc = __builtin_amdgcn_sad_u16(q03.x, __builtin_amdgcn_perm(0u, R0, 0x0c010c00u) + K, acc);
        acc = __builtin_amdgcn_sad_u16(q03.y, __builtin_amdgcn_perm(0u, R0, 0x0c030c02u) + K, acc);
        acc = __builtin_amdgcn_sad_u16(q03.z, __builtin_amdgcn_perm(0u, R1, 0x0c010c00u) + K, acc);
        acc = __builtin_amdgcn_sad_u16(q03.w, __builtin_amdgcn_perm(0u, R1, 0x0c030c02u) + K, acc);
        acc = __builtin_amdgcn_sad_u16(q45.x, __builtin_amdgcn_perm(0u, R2, 0x0c010c00u) + K, acc);
        acc = __builtin_amdgcn_sad_u16(q45.y, __builtin_amdgcn_perm(0u, R2, 0x0c0c0c02u) + k2, acc);  // pixel 10 | the zero
      }
      sad = (int)acc;
    }
    // lanes 0..10 of a row hold SAD(L = lane-5); first minimum wins (strict <, :856)
    const int smin = row_min_i(sad, lane);
    const unsigned long long mm = __ballot(sad == smin && sub < 11);
    const int best_l = __ffs((int)((mm >> (16 * grp)) & 0xFFFFull)) - 1;  // 0..10 == bestL + mnL
    float delta_u = 0.f;
    const int base = lane & ~15;
    const int bl_c = min(max(best_l, 1), 9);
    const float s1 = (float)__shfl(sad, base + bl_c - 1);
    const float s2 = (float)__shfl(sad, base + bl_c);
    const float s3 = (float)__shfl(sad, base + bl_c + 1);
    if (best_l > 0 && best_l < 10) {
      delta_u = (float)(0.5 * (double)(s1 - s3) / (double)(s1 + s3 - 2 * s2));
      if (delta_u < 1 && delta_u > -1) delta_u *= sf_r;
      else delta_u = 0.f;
    }
    float ru = r_x + delta_u;  // bestL itself is not added (quirk Q7)
    ru = fmaxf(0.f, ru);
    ru = fminf(ru, (float)cols0 - 1);
    float delta = l_x - ru;
    bool ok = true;
    if (delta <= 0) {
      ru = r_x;
      delta = l_x - ru;
      if (delta <= 0) ok = false;
    }
    if (go && ok) {
      out_u = (double)ru;
      out_depth = (double)(bf / (l_x - ru));
      matched = 1;
    }
  }
  if (sub == 0 && matched) {
    const size_t out_i = (size_t)pair * n_features + li;
    sr.right_u[out_i] = out_u;
    sr.depth[out_i] = out_depth;
  }
  const int n_m = __popcll(__ballot(matched && sub == 0));
  if (lane == 0 && n_m > 0) atomicAdd(&n_match[pair], n_m);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // (the next entries' windows go where these were read from: a wave's LDS accesses execute in order)
  }
}

void launch_match_bruteforce(hipStream_t s, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, const uint32_t* d_off,
                             const uint32_t* d_cand, int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(k_match_bruteforce, dim3((nq + 3) / 4), dim3(256), 0, s, d_q, nq, d_t, nt, d_off, d_cand, d_best_idx,
                     d_best_dist, d_second);
}

void launch_stereo(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint8_t* d_pyr, size_t img_pitch, const orbfe_keypoint* d_kps,
                   const uint8_t* d_desc, const KpAux* d_aux, const KpX* d_kx, uint32_t* d_rowoff, uint16_t* d_rowlist, int rows, int list_cap,
                   const int32_t* d_n_kp, int n_features, float fx, float bf, int cols0, int mean_threshold, double* d_right_u, double* d_depth, int32_t* d_n_match, int32_t* d_best_right,
                   int32_t* d_best_dist, int slot_l0, int slot_r0, int slot_step, int pair0, int n_pairs, double* h_right_u, double* h_depth,
                   int32_t* h_best_right, int32_t* h_best_dist, bool table_ready, const StereoRowsBuf* rowsbuf) {
  // h_*: page-locked host arrays [n_features] for the results of ONE pair (n_pairs == 1), or null
  // table_ready: d_rowoff / d_rowlist are the per-SLOT tables the extraction of the right image left behind (k_brief) and the match
  // counter is zero: no table launch
  if (n_pairs <= 0 || n_features <= 0) return;
  StereoHost host = {nullptr, nullptr, nullptr, nullptr};
  if (n_pairs == 1) host = {h_right_u, h_depth, h_best_right, h_best_dist};
  // batches: the row-parallel matcher (k_stereo_rows + k_stereo_sad); a pair or two (and STEREO_ROWS = 0): a wave per left keypoint (k_stereo)
  // (k_stereo4 -- four left keypoints per wave, the step between the two in r5 -- is kept as tools/exp/patches/k_stereo4.applied_r5.patch)
  StereoRows sr = {nullptr, nullptr, nullptr, nullptr, d_right_u, d_depth, d_best_right, d_best_dist};
  const bool batch = !table_ready && n_pairs >= STEREO4_FROM && n_levels <= 16;
  const bool by_rows = batch && STEREO_ROWS && rowsbuf && rowsbuf->lrow_off;
  if (by_rows) sr.lrow_off = rowsbuf->lrow_off, sr.lrow_list = rowsbuf->lrow_list, sr.work = rowsbuf->work, sr.work_n = rowsbuf->work_n;
  if (!table_ready)
  hipLaunchKernelGGL(k_rowtable, dim3(n_pairs), dim3(256), (size_t)(rows + 256) * sizeof(uint32_t), s, d_aux, d_n_kp, n_features, rows, list_cap,
                     d_rowoff, d_rowlist, slot_r0, slot_step, pair0, d_n_match, d_kps, slot_l0, sr);
#ifdef EXP_SKIP_STEREO  // (tools/exp: timing only -- what a free match would buy the step)
  if (n_pairs >= STEREO4_FROM) return;
#endif
  if (by_rows) {
    hipLaunchKernelGGL(k_stereo_rows, dim3((rows + 3) / 4, n_pairs), dim3(256), 0, s, d_kx, d_desc, d_rowoff, d_rowlist, rows, list_cap, n_features, fx,
                       mean_threshold, sr, slot_l0, slot_r0, slot_step, pair0);
    hipLaunchKernelGGL(k_stereo_sad, dim3(std::min((n_features + 4 * STEREO_SAD_WAVES - 1) / (4 * STEREO_SAD_WAVES), STEREO_SAD_GRID * 4 / STEREO_SAD_WAVES), n_pairs), dim3(64 * STEREO_SAD_WAVES), 0, s, d_lv, n_levels, d_pyr, img_pitch, d_kx, n_features, bf, cols0,
                       sr, d_n_match, slot_l0, slot_r0, slot_step, pair0);
    return;
  }
  hipLaunchKernelGGL(k_stereo, dim3((n_features + 3) / 4, n_pairs), dim3(256), 0, s, d_lv, n_levels, d_pyr, img_pitch, d_kps, d_desc,
                     d_kx, d_rowoff, d_rowlist, rows, list_cap, d_n_kp, n_features, fx, bf, cols0, mean_threshold, d_right_u, d_depth, d_n_match, d_best_right,
                     d_best_dist, slot_l0, slot_r0, slot_step, pair0, host, table_ready ? 1 : 0);
}

}  // namespace orbfe
