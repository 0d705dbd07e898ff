// rowtable_body.h -- the row table of one right image (ORBMatcher::createRowIndexDB, src/ORBMatcher.cc:915-932) by one 256-thread
// workgroup: the body of k_rowtable (k_match.hip), shared with k_brief (k_brief.hip), whose launches of a frame or two carry one such
// workgroup per image so that a stereo match that follows needs no table launch of its own.
#pragma once
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"
#include "wave_ops.h"

#ifndef RT_STAMP
#define RT_STAMP(k)
#endif
#ifndef RTP_STAMP
#define RTP_STAMP(k)
#endif
namespace orbfe {

// A: the image's row bands, nr keypoints; RO[rows + 1] offsets, RL[list_cap] entries; cnt[rows] and part[4] in LDS
__device__ __forceinline__ void rowtable_build(const KpAux* __restrict__ A, int nr, int rows, int list_cap, uint32_t* __restrict__ RO,
                                               uint16_t* __restrict__ RL, uint32_t* cnt, uint32_t* part, int tid, uint16_t* stage = nullptr,
                                               int stage_cap = 0) {
  // stage / stage_cap (k_brief's launches of a frame or two): an LDS buffer the list is assembled in when it fits -- the scatter's 14 k
  // two-byte stores to 14 k different places in global memory were most of this workgroup's 13 us (r6); from LDS the list leaves in
  // 16-byte units, one after the other
  for (int y = tid; y < rows; y += 256) cnt[y] = 0;
  __syncthreads();
  RT_STAMP(0)
  // The bands are read EIGHT per thread at a time, the eight loads independent of each other: one keypoint per trip of a strided loop
  // was eight dependent trips to memory per pass (~1 us each -- two thirds of this workgroup's 31 us, and the long pole of the descriptor
  // launch that carries it for a frame or two).
  static_assert(sizeof(KpAux) == 4, "KpAux");
  for (int i0 = tid; i0 < nr; i0 += 8 * 256) {
    KpAux a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = A[min(i0 + 256 * k, nr - 1)];
    // (r6) the eight bands of a thread are walked TOGETHER, one row of each per trip: eight independent LDS atomics in flight where a
    // band after the other was eight chains of dependent ones -- the scatter pass below (returning atomics, a store behind each) was
    // most of this workgroup's 14 us, and the workgroup the long pole of a pair's descriptor launch
    int len = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) len = max(len, (i0 + 256 * k < nr) ? (int)a[k].row_max - (int)a[k].row_min : 0);
    for (int t = 0; t < len; ++t) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int y = (int)a[k].row_min + t;
        if (i0 + 256 * k < nr && y < (int)a[k].row_max) atomicAdd(&cnt[y], 1u);  // (k_orient clips the band to [0, rows])
      }
    }
  }
  __syncthreads();
  RT_STAMP(1)  // counted
  // exclusive prefix sum over the rows: a run of rows per thread, the 256 run totals scanned in LDS
  const int per = (rows + 255) >> 8;
  const int y0 = tid * per, y1 = min(y0 + per, rows);
  uint32_t sum = 0;
  for (int y = y0; y < y1; ++y) sum += cnt[y];
  // inclusive scan of the 256 run totals: DPP scan inside each wave, the four wave totals through LDS (two barriers, not sixteen)
  const uint32_t incl_w = (uint32_t)wave_incl_scan_dpp<OpAddI>((int)sum);
  if ((tid & 63) == 63) part[tid >> 6] = incl_w;
  __syncthreads();
  uint32_t wave_base = 0;
  for (int k = 0; k < (tid >> 6); ++k) wave_base += part[k];
  const uint32_t incl = wave_base + incl_w;
  const uint32_t total_all = part[0] + part[1] + part[2] + part[3];
  __syncthreads();
  uint32_t run = incl - sum;
  for (int y = y0; y < y1; ++y) {
    const uint32_t c = cnt[y];
    cnt[y] = run;  // from here on: the row's write cursor
    RO[y] = run;
    run += c;
  }
  if (tid == 255) RO[rows] = total_all;
  RT_STAMP(2)  // prefix
  const bool staged = stage != nullptr && (int)total_all <= stage_cap && (int)total_all <= list_cap;
  __syncthreads();
  for (int i0 = tid; i0 < nr; i0 += 8 * 256) {
    KpAux a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = A[min(i0 + 256 * k, nr - 1)];
    int len = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) len = max(len, (i0 + 256 * k < nr) ? (int)a[k].row_max - (int)a[k].row_min : 0);
    for (int t = 0; t < len; ++t) {
      uint32_t pos[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int y = (int)a[k].row_min + t;
        pos[k] = (i0 + 256 * k < nr && y < (int)a[k].row_max) ? atomicAdd(&cnt[y], 1u) : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (pos[k] < (uint32_t)list_cap) {  // (list_cap = n_features x the widest band: always true for a real position)
          if (staged) stage[pos[k]] = (uint16_t)(i0 + 256 * k);
          else RL[pos[k]] = (uint16_t)(i0 + 256 * k);
        }
    }
  }
  RT_STAMP(3)  // scattered
  if (staged) {  // (uniform)
    __syncthreads();
    const int n16 = ((int)total_all + 7) >> 3;  // 16-byte units (RL is 16-byte aligned: list_cap is a multiple of 8; the buffers' tails are spare)
    for (int u = tid; u < n16; u += 256) ((uint4*)RL)[u] = ((const uint4*)stage)[u];
  }
}

// The same table by EIGHT workgroups of a launch (k_brief's launches of a frame or two: the eight spare workgroups per image), part r the
// rows [r RPW, (r + 1) RPW).  A CU's LDS serves about one atomic lane per clock: the 14 k increments of the count pass and the 14 k
// cursor fetches of the scatter pass were 5.6 + 7.9 us on one workgroup (stamps build, r6) -- the long pole of a pair's descriptor
// launch once its descriptor waves were down to 4 us.  Every part reads all the bands and keeps the rows of its range; its list begins
// where the lower parts' lists end: each part publishes its total in flags[r] (+ 1; zeroed by the launch before this one) and waits for
// the parts below it -- workgroups of SMALLER index, dispatched before it, so the wait cannot deadlock.
// cnt[RPW + 1] and part[4] in LDS; RPW = ceil(rows / 8) <= the caller's LDS budget.
__device__ __forceinline__ void rowtable_build_part(const KpAux* __restrict__ A, int nr, int rows, int list_cap, uint32_t* __restrict__ RO,
                                                    uint16_t* __restrict__ RL, uint32_t* cnt, uint32_t* part, int tid, int r, int32_t* flags,
                                                    uint2* lst = nullptr, int lst_cap = 0) {
  // lst / lst_cap: LDS for the bands that touch this part's rows (index | first row << 16, end row).  Only about a sixth of the bands do,
  // and a thread that walks its eight bands in lock step spends fifteen trips of eight predicated atomics on them either way -- 5 us a
  // pass whatever the part's share (stamps build).  With the list a thread takes one or two real bands.  cnt[rpw + 8]: the list's length.
  const int rpw = (rows + 7) >> 3;
  const int y_lo = r * rpw, y_hi = min(rows, y_lo + rpw), ny = max(y_hi - y_lo, 0);
  for (int y = tid; y < rpw + 9; y += 256) cnt[y] = 0;
  __syncthreads();
  RTP_STAMP(0)
  static_assert(sizeof(KpAux) == 4, "KpAux");
  bool listed = lst != nullptr && nr <= 65535;
  if (listed) {
    for (int i = tid; i < nr; i += 256) {
      const KpAux a = A[i];
      const int lo = max((int)a.row_min, y_lo), hi = min((int)a.row_max, y_hi);
      if (lo < hi) {
        const uint32_t p = atomicAdd(&cnt[rpw + 8], 1u);
        if (p < (uint32_t)lst_cap) lst[p] = make_uint2((uint32_t)i | ((uint32_t)lo << 16), (uint32_t)hi);
      }
    }
    __syncthreads();
    listed = cnt[rpw + 8] <= (uint32_t)lst_cap;  // (uniform; a list that does not fit: the lock-step form below)
  }
  const int n_list = listed ? (int)cnt[rpw + 8] : 0;
  if (listed) {
    for (int e = tid; e < n_list; e += 256) {
      const uint2 v = lst[e];
      for (int y = (int)(v.x >> 16); y < (int)v.y; ++y) atomicAdd(&cnt[y - y_lo], 1u);
    }
  } else
  for (int i0 = tid; i0 < nr; i0 += 8 * 256) {
    KpAux a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = A[min(i0 + 256 * k, nr - 1)];
    int lo[8], hi[8], len = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      lo[k] = max((int)a[k].row_min, y_lo), hi[k] = (i0 + 256 * k < nr) ? min((int)a[k].row_max, y_hi) : 0;
      len = max(len, hi[k] - lo[k]);
    }
    for (int t = 0; t < len; ++t) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (lo[k] + t < hi[k]) atomicAdd(&cnt[lo[k] + t - y_lo], 1u);
    }
  }
  __syncthreads();
  RTP_STAMP(1)  // counted
  // exclusive prefix over this part's rows (a run of rows per thread, as in rowtable_build)
  const int per = (rpw + 255) >> 8;
  const int q0 = min(tid * per, ny), q1 = min(q0 + per, ny);
  uint32_t sum = 0;
  for (int q = q0; q < q1; ++q) sum += cnt[q];
  const uint32_t incl_w = (uint32_t)wave_incl_scan_dpp<OpAddI>((int)sum);
  if ((tid & 63) == 63) part[tid >> 6] = incl_w;
  __syncthreads();
  uint32_t wave_base = 0;
  for (int k = 0; k < (tid >> 6); ++k) wave_base += part[k];
  const uint32_t total = part[0] + part[1] + part[2] + part[3];
  // publish this part's total, collect the lower parts'
  if (tid == 0) __hip_atomic_store(&flags[r], (int32_t)total + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t below = 0;
  if (tid < r) {
    int32_t f;
    while ((f = __hip_atomic_load(&flags[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(1);
    below = (uint32_t)(f - 1);
  }
  __syncthreads();  // (part[] has been read by everyone)
  if (tid < 8) cnt[rpw + tid] = below;  // (eight spare words behind the counters)
  __syncthreads();
  RTP_STAMP(2)  // lower parts known
  uint32_t base = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) base += cnt[rpw + k];
  uint32_t run = base + wave_base + incl_w - sum;
  for (int q = q0; q < q1; ++q) {
    const uint32_t c = cnt[q];
    cnt[q] = run;  // from here on: the row's write cursor
    RO[y_lo + q] = run;
    run += c;
  }
  if (r == 7 && tid == 0) RO[rows] = base + total;
  __syncthreads();
  if (listed) {
    for (int e = tid; e < n_list; e += 256) {
      const uint2 v = lst[e];
      for (int y = (int)(v.x >> 16); y < (int)v.y; ++y) {
        const uint32_t p = atomicAdd(&cnt[y - y_lo], 1u);
        if (p < (uint32_t)list_cap) RL[p] = (uint16_t)(v.x & 0xFFFFu);
      }
    }
  } else
  for (int i0 = tid; i0 < nr; i0 += 8 * 256) {
    KpAux a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = A[min(i0 + 256 * k, nr - 1)];
    int lo[8], hi[8], len = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      lo[k] = max((int)a[k].row_min, y_lo), hi[k] = (i0 + 256 * k < nr) ? min((int)a[k].row_max, y_hi) : 0;
      len = max(len, hi[k] - lo[k]);
    }
    for (int t = 0; t < len; ++t) {
      uint32_t pos[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) pos[k] = (lo[k] + t < hi[k]) ? atomicAdd(&cnt[lo[k] + t - y_lo], 1u) : 0xFFFFFFFFu;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (pos[k] < (uint32_t)list_cap) RL[pos[k]] = (uint16_t)(i0 + 256 * k);
    }
  }
  RTP_STAMP(3)  // scattered
}

}  // namespace orbfe
