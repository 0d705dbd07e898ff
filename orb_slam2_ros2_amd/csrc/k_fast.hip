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
//   3. the queue is processed densely: V for both polarities at once with packed-i16 min/max (v_pk_min_i16 /
//      v_pk_max_i16 on (d, -d) pairs), written to a zero-bordered V map;
//   4. NMS and the hi/lo decision run over the queue only, never over the whole patch again; the kept maxima
//      are appended to the level's candidate list (one atomic reservation per cell).
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"

namespace orbfe {

typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 as_s2(uint32_t u) { return __builtin_bit_cast(s2, u); }

// floor(i / d) for 0 <= i < 8192, 1 <= d <= 128 with inv = ceil(2^20 / d)
__device__ __forceinline__ int fdiv20(int i, uint32_t inv) { return (int)(((uint32_t)i * inv) >> 20); }

__global__ __launch_bounds__(64) void k_fast(const LevelDev* __restrict__ lv, const CellDev* __restrict__ cells,
                                             const uint8_t* __restrict__ pyr, size_t img_pitch, int t_hi, int t_lo,
                                             uint32_t* __restrict__ cand, size_t cand_pitch, int32_t* __restrict__ n_cand,
                                             int n_levels, int n_cells_total, int lds_v_off, int lds_q_off, int lds_f_off, int lds_wave_bytes) {
  extern __shared__ uint32_t lds_all[];
  // one cell per single-wave workgroup (four cells per workgroup measured 10 % slower: the LDS of a workgroup stays
  // allocated until its slowest cell is done).  WAVE_SYNC: the LDS accesses of one wave execute in order, the fence only
  // pins the compiler -- no s_barrier needed.
#define WAVE_SYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)
  uint32_t* lds_w = lds_all;
  uint8_t* P = (uint8_t*)lds_w;              // patch rows, pitch = pitch_p bytes, pixel (r,c) at P[r*pitch_p + xa + c]
  uint8_t* V = (uint8_t*)lds_w + lds_v_off;  // (ih+2) x (iw+2) scores with a zero border, pixel (iy,ix) at V[(iy+1)*pv + ix+1]
  uint16_t* Q = (uint16_t*)((uint8_t*)lds_w + lds_q_off);  // survivors: interior index i = iy*iw + ix, raster order
  uint8_t* F = (uint8_t*)lds_w + lds_f_off;  // per queue entry: bit0 = max & V>lo, bit1 = max & V>hi

  const int lane = threadIdx.x & 63;
  const int img = blockIdx.y;
  const int cell_id = blockIdx.x;
  if (cell_id >= n_cells_total) return;
  const CellDev cell = cells[cell_id];
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
  WAVE_SYNC();

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
  WAVE_SYNC();

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
      // V = max(A, B) for both polarities at once: lane-local packed i16 pairs (d_k, -d_k); the 16 circular
      // 9-windows by doubling (min over 2, 4, 8, then +1), then the max over the windows.  Corner at t <=> V > t.
      const uint32_t vhi = (uint32_t)v << 16;
      s2 d[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uint32_t a = ((uint32_t)ring[k] << 16) | (uint32_t)v;  // (v, r)
        const uint32_t b = (uint32_t)ring[k] | vhi;                  // (r, v)
        d[k] = as_s2(a) - as_s2(b);                                  // (v - r, r - v)
      }
      s2 m2[16], m4[16], m8[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) m2[k] = __builtin_elementwise_min(d[k], d[(k + 1) & 15]);
#pragma unroll
      for (int k = 0; k < 16; ++k) m4[k] = __builtin_elementwise_min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
      for (int k = 0; k < 16; ++k) m8[k] = __builtin_elementwise_min(m4[k], m4[(k + 4) & 15]);
      s2 best = __builtin_elementwise_min(m8[0], d[8]);
#pragma unroll
      for (int k = 1; k < 16; ++k) best = __builtin_elementwise_max(best, __builtin_elementwise_min(m8[k], d[(k + 8) & 15]));
      const int vv = max((int)best.x, (int)best.y);
      if (vv > t_min) V[(iy + 1) * pv + ix + 1] = (uint8_t)min(255, vv);
    }
  }
  WAVE_SYNC();

  // ---- 4. NMS over the queue, hi/lo decision, append to the level's candidate list ----
  int n_hi = 0, n_lo = 0;
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
      n_hi += __popcll(__ballot((f & 2) != 0));
      n_lo += __popcll(__ballot((f & 1) != 0));
    }
  }
  WAVE_SYNC();
  {
    // cv::FAST(hi) result if non-empty, else cv::FAST(lo) (ORBExtractor.cc:365-367).  The list of a level is a SET:
    // the quadtree orders candidates by (cell, y, x) recomputed from the coordinates, so cells may append in any
    // order -- one atomic reservation per cell, then the wave writes its records.
    const uint32_t inv = iw > 0 ? ((1u << 20) + iw - 1) / iw : 0;
    const int want = n_hi > 0 ? 2 : 1;
    const int total = n_hi > 0 ? n_hi : n_lo;
    if (total == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&n_cand[(size_t)img * n_levels + cell.level], total);
    base = __builtin_amdgcn_readfirstlane(base);
    uint32_t* out = cand + (size_t)img * cand_pitch + L.cand_base + base;
    int cnt = 0;
    for (int q0 = 0; q0 < nq; q0 += 64) {
      const int q = q0 + lane;
      const bool keep = (q < nq) && ((F[q] & want) != 0);
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int i = Q[q];
        const int iy = fdiv20(i, inv), ix = i - iy * iw;
        out[cnt + __popcll(m & ((1ull << lane) - 1ull))] =
            ORBFE_PACK_XYR(ix + 3 + cell.offx, iy + 3 + cell.offy, V[(iy + 1) * pv + ix + 1] - 1);
      }
      cnt += __popcll(m);
    }
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
                 size_t img_pitch, int t_hi, int t_lo, uint32_t* d_cand, size_t cand_pitch, int32_t* d_n_cand, int n_levels,
                 int n_img, int max_pw, int max_ph) {
  if (n_cells_total <= 0 || n_img <= 0) return;
  int v_off, q_off, f_off, total;
  fast_lds_layout(max_pw, max_ph, &v_off, &q_off, &f_off, &total);
  total = (total + 15) & ~15;
  hipLaunchKernelGGL(k_fast, dim3(n_cells_total, n_img), dim3(64), total, s, d_lv, d_cells, d_pyr, img_pitch, t_hi, t_lo,
                     d_cand, cand_pitch, d_n_cand, n_levels, n_cells_total, v_off, q_off, f_off, total);
}

}  // namespace orbfe
