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
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + 256 * k < nr)
        for (int y = a[k].row_min; y < a[k].row_max; ++y) atomicAdd(&cnt[y], 1u);  // (k_orient clips the band to [0, rows])
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
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + 256 * k;
      if (i < nr)
        for (int y = a[k].row_min; y < a[k].row_max; ++y) {
          const uint32_t p = atomicAdd(&cnt[y], 1u);
          if (p < (uint32_t)list_cap) RL[p] = (uint16_t)i;  // (list_cap = n_features x the widest band: always true)
        }
    }
  }
}

}  // namespace orbfe
