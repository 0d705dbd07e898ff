// k_fast.hip -- cell-tiled FAST-9/16 with score, in-cell 3x3 NMS and the per-cell threshold fallback.
//
// Replaces the cell loop of ORBExtractor::extractFast (src/ORB_SLAM2/src/ORBExtractor.cc:346-375):
//   cv::FAST(patch, kps, iniThFAST, true);  if (kps.empty()) cv::FAST(patch, kps, minThFAST, true);
// One wavefront per cell patch (what the reference hands to cv::FAST as a ROI view), so NMS and the
// fallback see exactly the pixels cv::FAST would see (seams between cells are NOT suppressed).
//
// Formulation (proved equivalent to OpenCV's FAST_t<16> + cornerScore<16> in DESIGN.md):
//   d_k = v - ring_k;  A = max over the 16 arcs of 9 contiguous ring pixels of min(d);  B = same for -d
//   V = max(A, B)            (threshold free; cornerScore = V - 1)
//   corner at threshold t  <=>  V > t
//   kept by NMS at t       <=>  V > t  and  V > V(q) for the 8 neighbours q inside the patch interior
// so one V map serves both thresholds; the cell emits {V > hi} if that set is non-empty, else {V > lo}.
//
// Work-efficient schedule inside the wave (most pixels are not corners):
//   1. patch -> LDS with aligned 32-bit loads;
//   2. every interior pixel takes a 9-read necessary test (a 9-arc contains one pixel of each opposite ring
//      pair, so min over 4 pairs of max(pair) must exceed v+t, or max of min(pair) be below v-t); survivors
//      are appended -- in raster order -- to an LDS queue with ballot/prefix;
//   3. the queue is processed densely: exact 16-bit arc masks, then V for the true corners, written to a
//      zero-bordered V map;
//   4. NMS, the hi/lo decision and the ordered compaction run over the queue only (it is already in raster
//      order), never over the whole patch again.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"

namespace orbfe {

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

// floor(i / d) for 0 <= i < 8192, 1 <= d <= 128 with inv = ceil(2^20 / d)
__device__ __forceinline__ int fdiv20(int i, uint32_t inv) { return (int)(((uint32_t)i * inv) >> 20); }

__global__ __launch_bounds__(64) void k_fast(const LevelDev* __restrict__ lv, const CellDev* __restrict__ cells,
                                             const uint8_t* __restrict__ pyr, size_t img_pitch, int t_hi, int t_lo,
                                             uint32_t* __restrict__ slots, size_t slots_pitch, uint16_t* __restrict__ counts,
                                             int n_cells_total, int lds_v_off, int lds_q_off, int lds_f_off) {
  extern __shared__ uint32_t lds_w[];
  uint8_t* P = (uint8_t*)lds_w;              // patch rows, pitch = pitch_p bytes, pixel (r,c) at P[r*pitch_p + xa + c]
  uint8_t* V = (uint8_t*)lds_w + lds_v_off;  // (ih+2) x (iw+2) scores with a zero border, pixel (iy,ix) at V[(iy+1)*pv + ix+1]
  uint16_t* Q = (uint16_t*)((uint8_t*)lds_w + lds_q_off);  // survivors: interior index i = iy*iw + ix, raster order
  uint8_t* F = (uint8_t*)lds_w + lds_f_off;  // per queue entry: bit0 = max & V>lo, bit1 = max & V>hi

  const int lane = threadIdx.x;
  const int img = blockIdx.y;
  const CellDev cell = cells[blockIdx.x];
  const LevelDev& L = lv[cell.level];
  const int pw = cell.pw, ph = cell.ph;
  const int iw = pw - 6, ih = ph - 6;  // interior cv::FAST scans: rows/cols 3 .. size-4
  const int n_int = (iw > 0 && ih > 0) ? iw * ih : 0;
  const int t_min = min(t_hi, t_lo);

  // ---- 1. patch -> LDS (aligned words) ----
  const int xa = cell.x0 & 3;
  const int nwords = (xa + pw + 3) >> 2;
  const int pitch_p = nwords * 4;
  {
    const uint8_t* src = pyr + (size_t)img * img_pitch + L.plane_off + (size_t)cell.y0 * L.stride + (cell.x0 - xa);
    const uint32_t inv = ((1u << 20) + nwords - 1) / nwords;
    const int total = ph * nwords;
    for (int k = lane; k < total; k += 64) {
      const int r = fdiv20(k, inv), c = k - r * nwords;
      lds_w[k] = *(const uint32_t*)(src + (size_t)r * L.stride + 4 * c);
    }
  }
  // ---- zero the V map (with border) ----
  const int pv = iw + 2;
  {
    const int vwords = ((ih + 2) * pv + 3) >> 2;
    uint32_t* V32 = (uint32_t*)V;
    for (int k = lane; k < vwords; k += 64) V32[k] = 0;
  }
  __syncthreads();

  // ---- 2. necessary test on every interior pixel, survivors -> queue (raster order) ----
  int nq = 0;
  {
    const uint32_t inv = iw > 0 ? ((1u << 20) + iw - 1) / iw : 0;
    for (int base = 0; base < n_int; base += 64) {
      const int i = base + lane;
      bool pass = false;
      if (i < n_int) {
        const int iy = fdiv20(i, inv), ix = i - iy * iw;
        const uint8_t* c = P + (iy + 3) * pitch_p + xa + ix + 3;
        const int v = c[0];
        const int r0 = c[3 * pitch_p], r8 = c[-3 * pitch_p], r4 = c[3], r12 = c[-3];
        const int r2 = c[2 * pitch_p + 2], r10 = c[-2 * pitch_p - 2], r6 = c[-2 * pitch_p + 2], r14 = c[2 * pitch_p - 2];
        const int lo_of_hi = min(min(max(r0, r8), max(r4, r12)), min(max(r2, r10), max(r6, r14)));
        const int hi_of_lo = max(max(min(r0, r8), min(r4, r12)), max(min(r2, r10), min(r6, r14)));
        pass = (lo_of_hi > v + t_min) || (hi_of_lo < v - t_min);
      }
      const unsigned long long m = __ballot(pass);
      if (pass) Q[nq + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
      nq += __popcll(m);
    }
  }
  __syncthreads();

  // ---- 3. exact test + score for the survivors ----
  {
    const uint32_t inv = iw > 0 ? ((1u << 20) + iw - 1) / iw : 0;
    for (int q = lane; q < nq; q += 64) {
      const int i = Q[q];
      const int iy = fdiv20(i, inv), ix = i - iy * iw;
      const uint8_t* c = P + (iy + 3) * pitch_p + xa + ix + 3;
      const int v = c[0];
      int ring[16];
      ring[0] = c[3 * pitch_p];
      ring[1] = c[3 * pitch_p + 1];
      ring[2] = c[2 * pitch_p + 2];
      ring[3] = c[1 * pitch_p + 3];
      ring[4] = c[3];
      ring[5] = c[-1 * pitch_p + 3];
      ring[6] = c[-2 * pitch_p + 2];
      ring[7] = c[-3 * pitch_p + 1];
      ring[8] = c[-3 * pitch_p];
      ring[9] = c[-3 * pitch_p - 1];
      ring[10] = c[-2 * pitch_p - 2];
      ring[11] = c[-1 * pitch_p - 3];
      ring[12] = c[-3];
      ring[13] = c[1 * pitch_p - 3];
      ring[14] = c[2 * pitch_p - 2];
      ring[15] = c[3 * pitch_p - 1];
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
        V[(iy + 1) * pv + ix + 1] = (uint8_t)min(255, max(a, b));  // > t_min >= 0 here
      }
    }
  }
  __syncthreads();

  // ---- 4. NMS over the queue, hi/lo decision, ordered compaction ----
  bool any_hi = false;
  {
    const uint32_t inv = iw > 0 ? ((1u << 20) + iw - 1) / iw : 0;
    for (int q0 = 0; q0 < nq; q0 += 64) {
      const int q = q0 + lane;
      int f = 0;
      if (q < nq) {
        const int i = Q[q];
        const int iy = fdiv20(i, inv), ix = i - iy * iw;
        const uint8_t* c = V + (iy + 1) * pv + ix + 1;
        const int v = c[0];
        if (v != 0) {
          const bool is_max = v > c[-1] && v > c[1] && v > c[-pv - 1] && v > c[-pv] && v > c[-pv + 1] && v > c[pv - 1] &&
                              v > c[pv] && v > c[pv + 1];
          if (is_max) f = ((v > t_hi) ? 2 : 0) | ((v > t_lo) ? 1 : 0);
        }
        F[q] = (uint8_t)f;
      }
      any_hi = any_hi || (__ballot((f & 2) != 0) != 0ull);
    }
  }
  __syncthreads();
  {
    const uint32_t inv = iw > 0 ? ((1u << 20) + iw - 1) / iw : 0;
    const int want = any_hi ? 2 : 1;
    int cnt = 0;
    uint32_t* out = slots + (size_t)img * slots_pitch + cell.slot_off;
    const int cap = L.cell_cap;
    for (int q0 = 0; q0 < nq; q0 += 64) {
      const int q = q0 + lane;
      const bool keep = (q < nq) && ((F[q] & want) != 0);
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int i = Q[q];
        const int iy = fdiv20(i, inv), ix = i - iy * iw;
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < cap) out[pos] = ORBFE_PACK_XYR(ix + 3 + cell.offx, iy + 3 + cell.offy, V[(iy + 1) * pv + ix + 1] - 1);
      }
      cnt += __popcll(m);
    }
    if (lane == 0) counts[(size_t)img * n_cells_total + blockIdx.x] = (uint16_t)min(cnt, cap);
  }
}

// LDS carve-up for the largest cell patch of a context (host side helper)
void fast_lds_layout(int max_pw, int max_ph, int* v_off, int* q_off, int* f_off, int* total) {
  const int pitch_p = ((3 + max_pw + 3) >> 2) * 4 + 4;
  const int p_bytes = (max_ph * pitch_p + 15) & ~15;
  const int iw = max_pw - 6, ih = max_ph - 6;
  const int v_bytes = (((ih + 2) * (iw + 2) + 3 + 15) & ~15);
  const int n_int = iw * ih;
  const int q_bytes = (2 * n_int + 15) & ~15;
  const int f_bytes = (n_int + 15) & ~15;
  *v_off = p_bytes;
  *q_off = p_bytes + v_bytes;
  *f_off = p_bytes + v_bytes + q_bytes;
  *total = p_bytes + v_bytes + q_bytes + f_bytes;
}

void launch_fast(hipStream_t s, const LevelDev* d_lv, const CellDev* d_cells, int n_cells_total, const uint8_t* d_pyr,
                 size_t img_pitch, int t_hi, int t_lo, uint32_t* d_slots, size_t slots_pitch, uint16_t* d_counts, int n_img,
                 int max_pw, int max_ph) {
  if (n_cells_total <= 0 || n_img <= 0) return;
  int v_off, q_off, f_off, total;
  fast_lds_layout(max_pw, max_ph, &v_off, &q_off, &f_off, &total);
  hipLaunchKernelGGL(k_fast, dim3(n_cells_total, n_img), dim3(64), total, s, d_lv, d_cells, d_pyr, img_pitch, t_hi, t_lo, d_slots,
                     slots_pitch, d_counts, n_cells_total, v_off, q_off, f_off);
}

}  // namespace orbfe
