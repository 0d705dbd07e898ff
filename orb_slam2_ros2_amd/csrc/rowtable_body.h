// rowtable_body.h -- the row table of one right image (ORBMatcher::createRowIndexDB, src/ORBMatcher.cc:915-932) by one 256-thread
// workgroup: the body of k_rowtable (k_match.hip), shared with k_brief (k_brief.hip), whose launches of a frame or two carry one such
// workgroup per image so that a stereo match that follows needs no table launch of its own.
#pragma once
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"
#include "wave_ops.h"

namespace orbfe {

// A: the image's row bands, nr keypoints; RO[rows + 1] offsets, RL[list_cap] entries; cnt[rows] and part[4] in LDS
__device__ __forceinline__ void rowtable_build(const KpAux* __restrict__ A, int nr, int rows, int list_cap, uint32_t* __restrict__ RO,
                                               uint16_t* __restrict__ RL, uint32_t* cnt, uint32_t* part, int tid) {
  for (int y = tid; y < rows; y += 256) cnt[y] = 0;
  __syncthreads();
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
        if (pos[k] < (uint32_t)list_cap) RL[pos[k]] = (uint16_t)(i0 + 256 * k);  // (list_cap = n_features x the widest band: always true for a real position)
    }
  }
}

}  // namespace orbfe
