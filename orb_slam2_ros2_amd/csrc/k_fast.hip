// k_fast.hip -- cell-tiled FAST-9/16 with score, in-cell 3x3 NMS and the per-cell threshold fallback.
//
// Replaces the cell loop of ORBExtractor::extractFast (src/ORB_SLAM2/src/ORBExtractor.cc:346-375):
//   cv::FAST(patch, kps, iniThFAST, true);  if (kps.empty()) cv::FAST(patch, kps, minThFAST, true);
// one workgroup per cell patch (what the reference hands to cv::FAST as a ROI view), so NMS and the
// fallback see exactly the pixels cv::FAST would see (seams between cells are NOT suppressed).
//
// Formulation (proved equivalent to OpenCV's FAST_t<16> + cornerScore<16> in DESIGN.md):
//   d_k = v - ring_k;  A = max over the 16 arcs of 9 contiguous ring pixels of min(d);  B = same for -d
//   V = max(A, B)            (threshold free; cornerScore = V - 1)
//   corner at threshold t  <=>  V > t
//   kept by NMS at t       <=>  V > t  and  V > V(q) for the 8 neighbours q inside the patch interior
// so one V map serves both thresholds; the cell emits {V > hi} if that set is non-empty, else {V > lo}.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"

namespace orbfe {

#define FAST_PW ORBFE_MAX_CELL

__device__ __forceinline__ uint32_t rot16(uint32_t m, int k) { return ((m >> k) | (m << (16 - k))) & 0xFFFFu; }

__device__ __forceinline__ bool has_arc9(uint32_t m) {
  uint32_t m2 = m & rot16(m, 1);
  uint32_t m4 = m2 & rot16(m2, 2);
  uint32_t m8 = m4 & rot16(m4, 4);
  return (m8 & rot16(m, 8)) != 0;
}

// max over the 16 circular windows of length 9 of the window minimum
__device__ __forceinline__ int max_arc_min(const int (&d)[16]) {
  int m2[16], m4[16], m8[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) m2[i] = min(d[i], d[(i + 1) & 15]);
#pragma unroll
  for (int i = 0; i < 16; ++i) m4[i] = min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
  for (int i = 0; i < 16; ++i) m8[i] = min(m4[i], m4[(i + 4) & 15]);
  int best = -1000;
#pragma unroll
  for (int i = 0; i < 16; ++i) best = max(best, min(m8[i], d[(i + 8) & 15]));
  return best;
}

__global__ __launch_bounds__(256) void k_fast(const LevelDev* __restrict__ lv, const CellDev* __restrict__ cells,
                                              const uint8_t* __restrict__ pyr, size_t img_pitch, int t_hi, int t_lo,
                                              uint32_t* __restrict__ slots, size_t slots_pitch, uint16_t* __restrict__ counts,
                                              int n_cells_total) {
  __shared__ uint8_t P[FAST_PW * FAST_PW];
  __shared__ uint8_t V[FAST_PW * FAST_PW];
  __shared__ uint8_t F[FAST_PW * FAST_PW];
  __shared__ int s_cnt_hi;
  const int tid = threadIdx.x;
  const int img = blockIdx.y;
  const CellDev cell = cells[blockIdx.x];
  const LevelDev& L = lv[cell.level];
  const int pw = cell.pw, ph = cell.ph;
  const uint8_t* src = pyr + (size_t)img * img_pitch + L.plane_off + (size_t)cell.y0 * L.stride + cell.x0;
  for (int i = tid; i < pw * ph; i += 256) {
    const int r = i / pw, c = i - r * pw;
    P[r * FAST_PW + c] = src[(size_t)r * L.stride + c];
    V[r * FAST_PW + c] = 0;
    F[r * FAST_PW + c] = 0;
  }
  if (tid == 0) s_cnt_hi = 0;
  __syncthreads();

  const int iw = pw - 6, ih = ph - 6;  // interior cv::FAST scans: rows/cols 3 .. size-4
  const int n_int = (iw > 0 && ih > 0) ? iw * ih : 0;
  const int t_min = min(t_hi, t_lo);
  for (int i = tid; i < n_int; i += 256) {
    const int iy = i / iw + 3, ix = i - (i / iw) * iw + 3;
    const uint8_t* c = &P[iy * FAST_PW + ix];
    const int v = c[0];
    int ring[16];
    ring[0] = c[3 * FAST_PW];
    ring[1] = c[3 * FAST_PW + 1];
    ring[2] = c[2 * FAST_PW + 2];
    ring[3] = c[1 * FAST_PW + 3];
    ring[4] = c[3];
    ring[5] = c[-1 * FAST_PW + 3];
    ring[6] = c[-2 * FAST_PW + 2];
    ring[7] = c[-3 * FAST_PW + 1];
    ring[8] = c[-3 * FAST_PW];
    ring[9] = c[-3 * FAST_PW - 1];
    ring[10] = c[-2 * FAST_PW - 2];
    ring[11] = c[-1 * FAST_PW - 3];
    ring[12] = c[-3];
    ring[13] = c[1 * FAST_PW - 3];
    ring[14] = c[2 * FAST_PW - 2];
    ring[15] = c[3 * FAST_PW - 1];
    uint32_t dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      dark |= (uint32_t)(ring[k] < v - t_min) << k;
      bright |= (uint32_t)(ring[k] > v + t_min) << k;
    }
    if (has_arc9(dark) || has_arc9(bright)) {
      int d[16], nd[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        d[k] = v - ring[k];
        nd[k] = -d[k];
      }
      const int a = max_arc_min(d), b = max_arc_min(nd);
      V[iy * FAST_PW + ix] = (uint8_t)min(255, max(a, b));  // > t_min >= 0 here
    }
  }
  __syncthreads();

  for (int i = tid; i < n_int; i += 256) {
    const int iy = i / iw + 3, ix = i - (i / iw) * iw + 3;
    const uint8_t* c = &V[iy * FAST_PW + ix];
    const int v = c[0];
    if (v == 0) continue;
    const bool is_max = v > c[-1] && v > c[1] && v > c[-FAST_PW - 1] && v > c[-FAST_PW] && v > c[-FAST_PW + 1] &&
                        v > c[FAST_PW - 1] && v > c[FAST_PW] && v > c[FAST_PW + 1];
    if (!is_max) continue;
    const int f = ((v > t_hi) ? 2 : 0) | ((v > t_lo) ? 1 : 0);
    F[iy * FAST_PW + ix] = (uint8_t)f;
    if (f & 2) atomicAdd(&s_cnt_hi, 1);
  }
  __syncthreads();

  if (tid < 64) {  // wave 0: ordered (raster) compaction of the kept maxima
    const int want = (s_cnt_hi > 0) ? 2 : 1;
    const int lane = tid;
    int cnt = 0;
    uint32_t* out = slots + (size_t)img * slots_pitch + cell.slot_off;
    const int cap = L.cell_cap;
    for (int base = 0; base < n_int; base += 64) {
      const int i = base + lane;
      bool keep = false;
      int ix = 0, iy = 0;
      if (i < n_int) {
        iy = i / iw + 3;
        ix = i - (i / iw) * iw + 3;
        keep = (F[iy * FAST_PW + ix] & want) != 0;
      }
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < cap) out[pos] = ORBFE_PACK_XYR(ix + cell.offx, iy + cell.offy, V[iy * FAST_PW + ix] - 1);
      }
      cnt += __popcll(m);
    }
    if (lane == 0) counts[(size_t)img * n_cells_total + blockIdx.x] = (uint16_t)min(cnt, cap);
  }
}

void launch_fast(hipStream_t s, const LevelDev* d_lv, const CellDev* d_cells, int n_cells_total, const uint8_t* d_pyr,
                 size_t img_pitch, int t_hi, int t_lo, uint32_t* d_slots, size_t slots_pitch, uint16_t* d_counts, int n_img) {
  if (n_cells_total <= 0 || n_img <= 0) return;
  hipLaunchKernelGGL(k_fast, dim3(n_cells_total, n_img), dim3(256), 0, s, d_lv, d_cells, d_pyr, img_pitch, t_hi, t_lo, d_slots,
                     slots_pitch, d_counts, n_cells_total);
}

}  // namespace orbfe
