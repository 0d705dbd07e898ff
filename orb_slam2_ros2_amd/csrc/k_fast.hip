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
//      pair, so min over 4 pairs of max(pair) must exceed v+t, or max of min(pair) be below v-t); survivors are
//      appended -- in raster order, one entry per polarity that passed -- to an LDS queue with ballot/prefix;
//   3. the queue is processed densely, one entry per lane: the entry's polarity is scored in 32-bit registers (v_mad_i32_i24 applies
//      the sign, v_min3_i32 / v_max3_i32 the arcs) into a zero-bordered V map, V = max(A, B) where both polarities were queued;
//   4. NMS runs over the queue only, never over the whole patch again; the kept maxima are appended to the level's
//      candidate list (one atomic reservation per cell).
// Steps 2-4 run with the HIGH threshold first: a pixel with V <= hi can never suppress one with V > hi (V(p) > hi >= V(q)), so the
// map of the hi-survivors alone gives cv::FAST(hi) exactly -- and only a cell where that comes out empty (4 % of the cells of level
// 0 on the synthetic frames, none on the coarse levels) repeats them with the LOW threshold (ORBExtractor.cc:365-367).  The test at
// hi passes 13 ... 37 % of the pixels against 16 ... 47 % at lo: 3.4 instead of 4.45 scoring trips per cell.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "orbfe_internal.h"
#include "wave_ops.h"

#ifndef FAST_PP40_MAX
#define FAST_PP40_MAX 37  // xa + pw <= 40
#endif
#ifndef FAST_PV36_MAX
#define FAST_PV36_MAX 40  // iw + 2 <= 36
#endif

namespace orbfe {

// (the necessary test takes one row step per loop trip.  Measured on one box, per 128 pairs: 1 -> 2.252 ms, 2 -> 2.278, 4 -> 2.287: the
// reads of the extra pixels only lengthen the trip, eight waves per SIMD already cover the LDS latency)

// queue entry: interior column | interior row << 7 | polarity to score | dual marker
#define Q_IX(e) ((int)((e)&0x7Fu))
#define Q_IY(e) ((int)(((e) >> 7) & 0x7Fu))
#define Q_XY(e) ((e)&0x3FFFu)
#define Q_BRIGHT 0x4000u
#define Q_DUAL 0x8000u  // both polarities passed the necessary test: the main entry scores the dark one

__device__ __forceinline__ int mbcnt64(unsigned long long m, int acc) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)acc));
}

// A = max over the 16 arcs of 9 contiguous ring pixels of min(sgn * (v - ring)): sgn = +1 scores the dark-ring polarity (A of the
// header), sgn = -1 the bright-ring one (B).  `a` points at the top-left corner of the pixel's 7x7 window in the LDS patch (compile-time
// pitch PP, so all 17 reads are immediate offsets of one address register).  One pixel per lane in 32-bit registers: the 16 circular
// 9-windows as 3 x 3 (min over 3 neighbours, then over three of those at distance 3), which the three-operand v_min3_i32 /
// v_max3_i32 cover in 16 + 16 + 8 instructions.  (Until r1_v15 two pixels per lane in packed 16-bit halves with the doubling
// scheme 2-4-8-9: 16 x 4 v_pk_min + 15 v_pk_max + 17 byte -> half packs per PAIR, i.e. no fewer instructions per pixel, and a
// trip of 128 entries: a cell's ~150 survivors took two expensive trips, the second mostly empty, instead of three cheap ones.)
template <int PP>
__device__ __forceinline__ int arc_score1(const uint8_t* a, int sgn) {
  constexpr int off[16] = {6 * PP + 3, 6 * PP + 4, 5 * PP + 5, 4 * PP + 6, 3 * PP + 6, 2 * PP + 6, 1 * PP + 5, 0 * PP + 4,
                           0 * PP + 3, 0 * PP + 2, 1 * PP + 1, 2 * PP + 0, 3 * PP + 0, 4 * PP + 0, 5 * PP + 1, 6 * PP + 2};
  int r[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) r[k] = a[off[k]];
  const int v = a[3 * PP + 3];
  int vs, d[16];
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(vs) : "v"(v), "v"(sgn));
  const int negs = -sgn;
#pragma unroll
  for (int k = 0; k < 16; ++k) asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d[k]) : "v"(r[k]), "v"(negs), "v"(vs));  // sgn * (v - r)
  int m3[16], m9[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) asm("v_min3_i32 %0, %1, %2, %3" : "=v"(m3[k]) : "v"(d[k]), "v"(d[(k + 1) & 15]), "v"(d[(k + 2) & 15]));
#pragma unroll
  for (int k = 0; k < 16; ++k)  // (spelled out: left to itself the compiler re-associates the 32 three-way minima into 24 + 14 two-way ones)
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(m9[k]) : "v"(m3[k]), "v"(m3[(k + 3) & 15]), "v"(m3[(k + 6) & 15]));
  const int a0 = max(max(m9[0], m9[1]), m9[2]), a1 = max(max(m9[3], m9[4]), m9[5]), a2 = max(max(m9[6], m9[7]), m9[8]);
  const int a3 = max(max(m9[9], m9[10]), m9[11]), a4 = max(max(m9[12], m9[13]), m9[14]);
  return max(max(max(a0, a1), a2), max(max(a3, a4), m9[15]));
}

// PP / PV: compile-time pitches of the LDS patch and of the score map (48 / 40 for patches up to 44 px wide -- every
// cell of the 30-px grid --, 80 / 72 for the largest patch the context accepts)
template <int PP, int PV>
__global__ __launch_bounds__(64) void k_fast(const LevelDev* __restrict__ lv, const CellDev* __restrict__ cells,
                                             const uint8_t* __restrict__ pyr, size_t img_pitch, int t_hi, int t_lo,
                                             uint32_t* __restrict__ cand, size_t cand_pitch, int32_t* __restrict__ n_cand,
                                             int n_levels, int cell_first, int n_cells, int lds_v_off, int lds_q_off, int q_cap) {
  extern __shared__ uint32_t lds_all[];
  // One cell per single-wave workgroup (four cells per workgroup measured 10 % slower: the LDS of a workgroup stays
  // allocated until its slowest cell is done); the kernel is VALU-bound and gains from every extra resident wave
  // (21 -> 26 waves per CU: -7 %), so the LDS carve-up is per level and as tight as the level's largest patch allows.
  // WAVE_SYNC: the LDS accesses of one wave execute in order, the fence only pins the compiler -- no s_barrier needed.
#define WAVE_SYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)
  uint32_t* lds_w = lds_all;
  uint8_t* P = (uint8_t*)lds_w;              // patch rows, pitch PP, pixel (r,c) at P[r*PP + xa + c]
  uint8_t* V = (uint8_t*)lds_w + lds_v_off;  // (ih+2) rows of scores with a zero border, pitch PV, pixel (iy,ix) at V[(iy+1)*PV + ix+1]
  uint16_t* Q = (uint16_t*)((uint8_t*)lds_w + lds_q_off);  // [q_cap] survivors from the front; pixels that need the second
                                                           // polarity too are listed again from the back
  // (the NMS verdict of a queue entry goes into bit 14 of the entry itself: its polarity / dual tags are spent by then)

  const int lane = threadIdx.x & 63;
  const int img = blockIdx.y;
  if ((int)blockIdx.x >= n_cells) return;
  const CellDev cell = cells[cell_first + blockIdx.x];
  const LevelDev& L = lv[cell.level];
  const int pw = cell.pw, ph = cell.ph;
  const int iw = pw - 6, ih = ph - 6;  // interior cv::FAST scans: rows/cols 3 .. size-4
  if (iw <= 0 || ih <= 0) return;
  // ---- 1. patch -> LDS: 16-byte units (PP / 16 per row: 3 for the 30-px grid's patches), 21 rows per pass, every pass of a lane
  //         requested before the first is parked (rows clamped, so the loads are unconditional): two loads and two stores per lane
  //         for a 36-row patch where 32-bit words took nine of each plus their address arithmetic (~ 60 of the kernel's ~1080 VALU
  //         instructions per wave).  The global address is only 4-byte aligned (x0 - xa); gfx950 takes that for dwordx4.
  const int xa = cell.x0 & 3;
  {
    constexpr int UPR = (PP + 15) / 16;             // 16-byte units per patch row (pitch 40: the third one is half a unit)
    constexpr int RPP = 64 / UPR;                   // rows per pass
    constexpr int MUL = (128 + UPR - 1) / UPR;      // lane / UPR == (lane * MUL) >> 7 for lane < 64 (UPR = 3: 43, UPR = 5: 26)
    static_assert(PP % 8 == 0 && (UPR == 3 || UPR == 5), "patch pitch");
    const int row0 = (lane * MUL) >> 7, part = lane - row0 * UPR;
    const int nbytes = (xa + pw + 3) & ~3;          // bytes of a patch row that are needed
    const uint8_t* src = pyr + (size_t)img * img_pitch + L.plane_off + (size_t)cell.y0 * L.stride + (cell.x0 - xa);
    const uint32_t stride = (uint32_t)L.stride;
    constexpr int BATCH = 2;
    if (row0 < RPP && 16 * part < nbytes) {         // (a unit that starts past the needed bytes is never read: it may lie past the row)
      for (int rb = row0; rb < ph; rb += RPP * BATCH) {
        uint4 wv[BATCH];
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
          const int r = min(rb + RPP * it, ph - 1);
          wv[it] = *(const uint4*)(src + (uint32_t)mad24u(r, (int)stride, 16 * part));  // full-rate 24-bit product: rows and strides < 2^13
        }
        // both passes in flight before the first is consumed (left alone, the compiler sinks the second load into the branch that
        // guards its store and the wave pays two memory round trips)
#pragma unroll
        for (int it = 0; it < BATCH; ++it) asm volatile("" : "+v"(wv[it].x), "+v"(wv[it].y), "+v"(wv[it].z), "+v"(wv[it].w));
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
          const int r = rb + RPP * it;
          if (r < ph) {
            if (PP % 16 == 0) {
              ((uint4*)lds_all)[r * UPR + part] = wv[it];
            } else {  // rows 8-byte aligned only: two 8-byte stores, the half unit at the end of a row one
              uint2* d = (uint2*)(P + r * PP + 16 * part);
              d[0] = make_uint2(wv[it].x, wv[it].y);
              if (part < UPR - 1) d[1] = make_uint2(wv[it].z, wv[it].w);
            }
          }
        }
      }
    }
  }
  // ---- zero the V map (with border), 16 bytes per store (its carve-up is rounded to 16) ----
  {
    const int vq = ((ih + 2) * PV + 3 + 15) >> 4;
    uint4* V128 = (uint4*)V;
    for (int k = lane; k < vq; k += 64) V128[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  WAVE_SYNC();

  int nq = 0, n_keep = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const int t_pass = pass == 0 ? t_hi : t_lo;  // cv::FAST(patch, hi); if that yields nothing: cv::FAST(patch, lo)
    if (pass == 1) {
      if (t_lo >= t_hi) break;  // the second call could only return a subset of the (empty) first
      const int vq = ((ih + 2) * PV + 3 + 15) >> 4;
      uint4* V128 = (uint4*)V;
      for (int k = lane; k < vq; k += 64) V128[k] = make_uint4(0u, 0u, 0u, 0u);
      WAVE_SYNC();
    }
    // ---- 2. necessary test on every interior pixel; survivors -> queue, tagged with the polarity to score.  The wave covers
    //         a (64/lw rows) x (lw columns) tile per step, lw = 16/32/64 by cell width, so addresses advance by a constant.
    //         (The queue order is free: NMS does not depend on it and the candidate list of a level is a set.  Two pixels
    //         per lane in packed halves would need unaligned 16-bit LDS reads: measured ~20 cycles each on gfx950.)
    nq = 0;
    int nd = 0;
    bool d_overflow = false;
    {
      const int shift = iw <= 16 ? 4 : (iw <= 32 ? 5 : 6);
      const int lw = 1 << shift, rpi = 64 >> shift;
      const int lx = lane & (lw - 1), ly = lane >> shift;
      for (int x0 = 0; x0 < iw; x0 += lw) {
        const int ix = x0 + lx;
        const uint8_t* a0 = P + ly * PP + xa + ix;  // top-left corner of the pixel's 7x7 window
        const uint32_t e_lane = (uint32_t)ix | ((uint32_t)ly << 7);
        const int thr_x = ix < iw ? t_pass : 0x7FFF;
        const uint8_t* a = a0;
        // every predicate is ONE vector compare whose lane mask feeds the ballot and the branch directly (a predicate built from
        // several masks is expanded to 0 / 1 per lane and compared again before a ballot: two more VALU instructions each)
        auto trip = [&](const int y0, const int thr) __attribute__((always_inline)) {
          const int v = a[3 * PP + 3];
          const int r0 = a[6 * PP + 3], r8 = a[3], r4 = a[3 * PP + 6], r12 = a[3 * PP];
          const int r2 = a[5 * PP + 5], r10 = a[PP + 1], r6 = a[PP + 5], r14 = a[5 * PP + 1];
          const int lo_of_hi = min(min(max(r0, r8), max(r4, r12)), min(max(r2, r10), max(r6, r14)));
          const int hi_of_lo = max(max(min(r0, r8), min(r4, r12)), max(min(r2, r10), min(r6, r14)));
          const int sb = lo_of_hi - v;  // > t: every opposite pair has a pixel brighter than v + t
          const int sd = v - hi_of_lo;  // > t: ... darker than v - t
          // two compares; their lane masks are combined on the SCALAR unit and handed back as lane predicates (inverse ballot: the
          // mask register is used as it is) -- max / min of the two margins and a compare each were four vector instructions
          const bool pd = sd > thr;
          const unsigned long long mb = __ballot(sb > thr), md = __ballot(pd);
          const unsigned long long m = mb | md, m2 = mb & md;
          const bool any = __builtin_amdgcn_inverse_ballot_w64(m), dual = __builtin_amdgcn_inverse_ballot_w64(m2);
          const uint32_t eh = e_lane + ((uint32_t)y0 << 7);
          if (any) Q[nq + mbcnt64(m, 0)] = (uint16_t)(eh | (pd ? 0u : Q_BRIGHT) | (dual ? Q_DUAL : 0u));
          nq += __popcll(m);
          if (m2) {  // both polarities passed: the main entry scores the dark one, the list at the back of Q the bright one
            const int k = __popcll(m2);
            if (nq + nd + k <= q_cap) {
              if (dual) Q[q_cap - 1 - (nd + mbcnt64(m2, 0))] = (uint16_t)eh;
              nd += k;
            } else {
              d_overflow = true;  // (pathological cell) the tagged entries are re-scanned one by one below
            }
          }
        };
        // lanes right of the interior can never pass (thr_x), so the column test costs nothing per trip; the row test is only needed in
        // the last, partial trip of a column block (its rows past the interior are read -- they lie inside this wave's carve-up -- but
        // cannot pass either)
        int y0 = 0;
        for (; y0 + rpi <= ih; y0 += rpi, a += rpi * PP) trip(y0, thr_x);
        if (y0 < ih) trip(y0, y0 + ly < ih ? thr_x : 0x7FFF);
      }
    }
    if (nq + nd > q_cap) {  // the front grew into the back list after it was written: drop the list, re-scan instead
      nd = 0;
      d_overflow = true;
    }
    WAVE_SYNC();

    // ---- 3. exact test + score for the survivors, one queue entry per lane (trips of 64).  Only about
    //         half of the survivors are corners: those (and the dual-tagged entries, whose second polarity is still to come) are
    //         compacted in place at the front of Q, so that the NMS and the output pass touch no entry that cannot matter.
    int nc = 0;
    for (int j0 = 0; j0 < nq; j0 += 64) {
      const bool act = j0 + lane < nq;
      const uint32_t q = Q[min(j0 + lane, nq - 1)];
      const int ix = Q_IX(q), iy = Q_IY(q);
      const int A = arc_score1<PP>(P + iy * PP + xa + ix, (q & Q_BRIGHT) ? -1 : 1);
      const bool c = act && A > t_pass;
      if (c) V[(iy + 1) * PV + ix + 1] = (uint8_t)min(255, A);
      const bool keep = c || (act && (q & Q_DUAL));
      const unsigned long long m = __ballot(keep);
      if (keep) Q[nc + mbcnt64(m, 0)] = (uint16_t)q;  // (in place: this trip's entries were all read above, later trips read further back)
      nc += __popcll(m);
    }
    nq = nc;
    if (nd > 0 || d_overflow) {
      WAVE_SYNC();
      for (int j = lane; j < nd; j += 64) {  // the back list: V = max(A, B)
        const uint32_t q = Q[q_cap - 1 - j];
        const int ix = Q_IX(q), iy = Q_IY(q);
        const int B = arc_score1<PP>(P + iy * PP + xa + ix, -1);
        uint8_t* vp = V + (iy + 1) * PV + ix + 1;
        if (B > t_pass) *vp = (uint8_t)max((int)*vp, min(255, B));
      }
      if (d_overflow) {  // entries that did not fit the back list (max is idempotent, so re-scoring listed ones is harmless)
        for (int q = lane; q < nq; q += 64) {
          const uint32_t e = Q[q];
          if (e & Q_DUAL) {
            const int ix = Q_IX(e), iy = Q_IY(e);
            const uint8_t* a = P + iy * PP + xa + ix;
            const int B = arc_score1<PP>(a, -1);
            uint8_t* vp = V + (iy + 1) * PV + ix + 1;
            if (B > t_pass) *vp = (uint8_t)max((int)*vp, min(255, B));
          }
        }
      }
    }
    WAVE_SYNC();

    // ---- 4. NMS over the queue: an entry is kept iff its score beats its 8 neighbours' (0 where nothing was scored) ----
    n_keep = 0;
    for (int q0 = 0; q0 < nq; q0 += 64) {
      const int q = q0 + lane;
      bool is_max = false;
      if (q < nq) {
        const uint32_t e = Q[q];
        const uint8_t* c = V + Q_IY(e) * PV + Q_IX(e);  // top-left corner of the 3x3 neighbourhood
        // all nine reads requested at once, one compare against the largest neighbour (v > max >= 0 also says v != 0): written as a
        // chain of && the compiler made nine dependent LDS round trips of it, each behind its own branch
        const int v = c[PV + 1];
        const int n0 = c[0], n1 = c[1], n2 = c[2], n3 = c[PV], n4 = c[PV + 2], n5 = c[2 * PV], n6 = c[2 * PV + 1], n7 = c[2 * PV + 2];
        is_max = v > max(max(max(n0, n1), n2), max(max(max(n3, n4), n5), max(n6, n7)));
        Q[q] = (uint16_t)(Q_XY(e) | (is_max ? Q_BRIGHT : 0u));  // (scores in the map are > t_pass by construction)
      }
      n_keep += __popcll(__ballot(is_max));
    }
    WAVE_SYNC();
    if (n_keep > 0) break;
  }
  {
    // The list of a level is a SET: the quadtree orders candidates by (cell, y, x) recomputed from the coordinates, so cells may
    // append in any order -- one atomic reservation per cell, then the wave writes its records.
    const int total = n_keep;
    if (total == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&n_cand[(size_t)img * n_levels + cell.level], total);
    base = __builtin_amdgcn_readfirstlane(base);
    uint32_t* out = cand + (size_t)img * cand_pitch + L.cand_base + base;
    int cnt = 0;
    for (int q0 = 0; q0 < nq; q0 += 64) {
      const int q = q0 + lane;
      const uint32_t e = q < nq ? (uint32_t)Q[q] : 0u;
      const bool keep = (e & Q_BRIGHT) != 0;
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int ix = Q_IX(e), iy = Q_IY(e);
        out[cnt + mbcnt64(m, 0)] = ORBFE_PACK_XYR(ix + 3 + cell.offx, iy + 3 + cell.offy, V[(iy + 1) * PV + ix + 1] - 1);
      }
      cnt += __popcll(m);
    }
  }
}


// LDS carve-up for cell patches up to max_pw x max_ph with pitches pp / pv (host side helper)
void fast_lds_layout(int max_pw, int max_ph, int pp, int pv, int* v_off, int* q_off, int* q_cap, int* total) {
  const int p_bytes = (max_ph * pp + 15) & ~15;
  const int iw = max_pw - 6, ih = max_ph - 6;
  const int v_bytes = ((ih + 2) * pv + 3 + 15) & ~15;
  const int n_int = iw * ih;
  const int q_bytes = (2 * n_int + 15) & ~15;  // one entry per interior pixel (+ the back list in what the front leaves)
  *v_off = p_bytes;
  *q_off = p_bytes + v_bytes;
  *q_cap = q_bytes / 2;
  *total = p_bytes + v_bytes + q_bytes;
#ifdef FAST_LDS_PAD
  *total += FAST_LDS_PAD;  // (tools/exp: the occupancy experiment of the launcher's comment)
#endif
}

// One kernel launch for cells [cell_first, cell_first + n_cells) whose patches are at most max_pw x max_ph: the tightest pitches the
// patches allow.  The kernel's throughput follows the resident waves almost one to one (r3, same-box A/B: 520 bytes of padding per
// wave, 30 -> 27 waves per CU on level 0, +10 % time), so the patch pitch is 40 where xa + pw <= 40 and the score map's 36 where
// iw + 2 <= 36: levels 0..2 of 1241x376 fit 32 waves per CU (5120 bytes each) instead of 30 (worth ~1 %: the knee is just below).
// Built, bit-exact and dropped in r3: the necessary test on TWO pixels per lane -- the patch parked row-paired (dword = pixel (r, c) |
// pixel (r + ih/2, c) << 16, every ring read one aligned 32-bit LDS read for both; misaligned LDS reads cost 28 aligned ones,
// tools/exp/lds_misaligned.hip, so horizontal pairs are out), v_pk_min/max_u16 + v_pk_sub_i16 and sign-bit verdicts, the score map in
// the unused upper bytes, Q capped to keep 5120 bytes per wave with a dense fallback for cells whose survivors do not fit: 8 trips of 34
// vector instructions instead of 15 of ~30, but the interleaving stores, the half-select of every window / score address and the
// dual list moved into the scoring trips gave back all of it (892 vector instructions per wave against 813, SQ counters of
// tools/exp/fast_sq.sh) with 3.5 x the LDS bank-conflict cycles: 2.22 ms against 2.07.
static void launch_fast_cells(hipStream_t s, const LevelDev* d_lv, const CellDev* d_cells, int max_pw, int max_ph, const uint8_t* d_pyr,
                              size_t img_pitch, int t_hi, int t_lo, uint32_t* d_cand, size_t cand_pitch, int32_t* d_n_cand, int n_levels,
                              int cell_first, int n_cells, int n_img) {
  int v_off, q_off, q_cap, total;
  const int pp = max_pw <= FAST_PP40_MAX ? 40 : (max_pw <= 44 ? 48 : 80), pv = max_pw <= FAST_PV36_MAX ? 36 : (max_pw <= 44 ? 40 : 72);
  fast_lds_layout(max_pw, max_ph, pp, pv, &v_off, &q_off, &q_cap, &total);
#define FAST_GO(K) hipLaunchKernelGGL(K, dim3(n_cells, n_img), dim3(64), total, s, d_lv, d_cells, d_pyr, img_pitch, t_hi, t_lo, d_cand, \
                                      cand_pitch, d_n_cand, n_levels, cell_first, n_cells, v_off, q_off, q_cap)
  if (pp == 40) FAST_GO((k_fast<40, 36>));
  else if (pp == 48 && pv == 36) FAST_GO((k_fast<48, 36>));
  else if (pp == 48) FAST_GO((k_fast<48, 40>));
  else FAST_GO((k_fast<80, 72>));
#undef FAST_GO
}

// One launch per pyramid level: the cells of a level have (almost) one size, so each launch reserves just the LDS its
// patches need.  Measured: merging levels 0..3 into one launch with their common carve-up is 5 % slower than the four separate launches.
void launch_fast(hipStream_t s, const LevelDev* d_lv, const CellDev* d_cells, const LevelDev* h_lv, const int* lvl_max_pw,
                 const int* lvl_max_ph, const uint8_t* d_pyr, size_t img_pitch, int t_hi, int t_lo, uint32_t* d_cand, size_t cand_pitch,
                 int32_t* d_n_cand, int n_levels, int n_img, hipStream_t side, hipEvent_t ev_go, hipEvent_t ev_done, int side_from) {
  // side (nullable): the launches of levels >= side_from go to this stream (after ev_go, recorded on s here; ev_done joins them
  // back into s): the small levels do not fill the machine and their waves run in the tails the large levels' launches leave
  if (n_img <= 0) return;
  // a frame or two (the drop-in path): the whole sweep fits the machine at once, and eight back-to-back launches would each
  // cost a full wave lifetime (~18 us): one launch with the common carve-up instead (0.144 -> 0.03 ms for one stereo pair)
  // (Built, parity-tested and dropped in r3 for this launch: four waves per cell, each a BAND of the cell's rows with one halo row on
  //  each side and the cell-wide high / low verdict through LDS.  A band wave lives 2.6 us instead of ~18, but with one list
  //  reservation per band the kernel took 98 us instead of 21 -- 4264 atomics on the two level-0 counters at ~45 ns each -- and with
  //  sixteen-wave workgroups of four cells x four bands and ONE reservation per workgroup 24.8 us: three barriers, a serial prefix
  //  and the atomic's round trip per workgroup cost what the shorter waves save.)
  int total_cells = 0, max_pw = 0, max_ph = 0;
  for (int l = 0; l < n_levels; ++l) {
    total_cells += h_lv[l].n_cells;
    max_pw = std::max(max_pw, lvl_max_pw[l]);
    max_ph = std::max(max_ph, lvl_max_ph[l]);
  }
  if ((long long)total_cells * n_img <= 16384 && max_pw > 6 && max_ph > 6) {
    launch_fast_cells(s, d_lv, d_cells, max_pw, max_ph, d_pyr, img_pitch, t_hi, t_lo, d_cand, cand_pitch, d_n_cand, n_levels, 0, total_cells,
                      n_img);
    return;
  }
  const bool split = side && ev_go && ev_done && side_from > 0 && side_from < n_levels;
  if (split) {
    (void)hipEventRecord(ev_go, s);
    (void)hipStreamWaitEvent(side, ev_go, 0);
  }
  for (int l = 0; l < n_levels; ++l) {
    const int n_cells = h_lv[l].n_cells;
    if (n_cells <= 0 || lvl_max_pw[l] <= 6 || lvl_max_ph[l] <= 6) continue;
    launch_fast_cells((split && l >= side_from) ? side : s, d_lv, d_cells, lvl_max_pw[l], lvl_max_ph[l], d_pyr, img_pitch, t_hi, t_lo, d_cand,
                      cand_pitch, d_n_cand, n_levels, (int)h_lv[l].cell_base, n_cells, n_img);
  }
  if (split) {
    (void)hipEventRecord(ev_done, side);
    (void)hipStreamWaitEvent(s, ev_done, 0);
  }
}

}  // namespace orbfe
