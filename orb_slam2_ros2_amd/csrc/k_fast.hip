// k_fast.hip -- cell-tiled FAST-9/16 with score, in-cell 3x3 NMS and the per-cell threshold fallback.
//
// Replaces the cell loop of ORBExtractor::extractFast (src/ORB_SLAM2/src/ORBExtractor.cc:346-375):
//   cv::FAST(patch, kps, iniThFAST, true);  if (kps.empty()) cv::FAST(patch, kps, minThFAST, true);
// One wavefront works on ONE cell patch at a time (what the reference hands to cv::FAST as a ROI view), so NMS and the
// fallback see exactly the pixels cv::FAST would see (seams between cells are NOT suppressed); in large launches a wave takes
// several cells in turn, the next one's patch travelling while the current one is worked on (k_fast's own comment).
//
// Formulation (proved equivalent to OpenCV's FAST_t<16> + cornerScore<16> in DESIGN.md):
//   d_k = v - ring_k;  A = max over the 16 arcs of 9 contiguous ring pixels of min(d);  B = same for -d
//   V = max(A, B)            (threshold free; cornerScore = V - 1)
//   corner at threshold t  <=>  V > t
//   kept by NMS at t       <=>  V > t  and  V > V(q) for the 8 neighbours q inside the patch interior
// so one V map serves both thresholds; the cell emits {V > hi} if that set is non-empty, else {V > lo}.
//
// Work-efficient schedule inside the wave (most pixels are not corners):
//   1. patch -> registers -> LDS in 16-byte units;
//   2. every interior pixel takes a 9-read necessary test (a 9-arc contains one pixel of each opposite ring
//      pair, so min over 4 pairs of max(pair) must exceed v+t, or max of min(pair) be below v-t); survivors are
//      appended -- one entry per pixel, tagged with the polarity to score and whether both passed -- to an LDS queue with ballot/prefix;
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
#include <cstdio>
#include <cstring>

#include "orbfe_internal.h"
#include "wave_ops.h"

#ifndef FAST_CPW
#define FAST_CPW 4
#endif
#ifndef FAST_PP40_MAX
#define FAST_PP40_MAX 37  // xa + pw <= 40
#endif
#ifndef FAST_PV36_MAX
#define FAST_PV36_MAX 40  // iw + 2 <= 36
#endif

namespace orbfe {

#ifdef FAST_STAMPS  // diagnostic build only (tools/exp/fast_stamps.sh): where a one-cell wave's life goes, in 100 MHz ticks of s_memrealtime
__device__ unsigned long long g_fs_rec[8192][8];  // per wave: six phase durations, start, end (no atomics: 2462 waves on one address take 30 us each)
#define FS_DECL unsigned long long fs_t0 = __builtin_amdgcn_s_memrealtime(), fs_last = fs_t0, fs_d[6] = {0, 0, 0, 0, 0, 0};
#define FS(k)                                                                        \
  {                                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                      \
    const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                \
    fs_d[k] += now_ - fs_last;                                                       \
    fs_last = now_;                                                                  \
  }
#define FS_END                                                                                         \
  if (lane == 0) {                                                                                     \
    const unsigned w_ = (blockIdx.y * gridDim.x + blockIdx.x) & 8191u;                                 \
    for (int k_ = 0; k_ < 6; ++k_) g_fs_rec[w_][k_] = fs_d[k_];                                        \
    g_fs_rec[w_][6] = fs_t0, g_fs_rec[w_][7] = __builtin_amdgcn_s_memrealtime();                       \
  }
#else
#define FS_DECL
#define FS(k)
#define FS_END
#endif

// (the necessary test takes one row step per loop trip.  Measured on one box, per 128 pairs: 1 -> 2.252 ms, 2 -> 2.278, 4 -> 2.287: the
// reads of the extra pixels only lengthen the trip, eight waves per SIMD already cover the LDS latency)

// queue entry: interior column | interior row << 7 | polarity to score | dual marker
#define Q_IX(e) ((int)((e)&0x7Fu))
#define Q_IY(e) ((int)(((e) >> 7) & 0x7Fu))
#define Q_XY(e) ((e)&0x3FFFu)
#define Q_BRIGHT 0x4000u
#define Q_DUAL 0x8000u  // both polarities passed the necessary test: the main entry scores the dark one

#ifndef FAST_OPS16
#define FAST_OPS16 1
#endif
#ifndef FAST_SLIDE
#define FAST_SLIDE 3  // the necessary test walks down its columns with register rings: 1 = window column 3 (7 LDS reads per pixel), 2 = columns 3 and
                      // 1 (6 reads), 3 = columns 3, 1 and 5 (5 reads; the default: 63 registers, no spill); 0 = r5's two rows per trip, 9 reads
#endif
#if FAST_OPS16
__device__ __forceinline__ int fast_min16(int a, int b) {
  int d;
  asm("v_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int fast_max16(int a, int b) {
  int d;
  asm("v_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
#define FAST_MIN16(a, b) fast_min16(a, b)
#define FAST_MAX16(a, b) fast_max16(a, b)
#else
#define FAST_MIN16(a, b) min(a, b)
#define FAST_MAX16(a, b) max(a, b)
#endif

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

// PP / PV: compile-time pitches of the LDS patch and of the score map (40 / 36, 48 / 36 and 48 / 40 for the patches of the 30-px grid,
// 80 / 72 for the largest patch the context accepts).
//
// A wave works through cells ci = blockIdx.x, + n_groups, + 2 n_groups ... of its launch (n_groups = n_cells: one cell per wave, the
// drop-in path's launches; a quarter of that for the batches).  Of a one-cell wave's ~10.7 us a fifth is spent at s_waitcnt for global
// memory (SQ counters, r3) -- the patch at its start, the list reservation's round trip at its end -- with nothing to issue, and eight
// such waves per SIMD leave the vector unit idle a fifth of the time.  In the loop both are off the critical path: the NEXT cell's patch
// is requested into registers before the current cell is worked on, and a cell's records (almost always <= 64: one per lane, in a
// register) are stored only after the NEXT cell's work, when the reservation issued before it has long returned.
//
// SH (a frame or two, r6): a level's candidates go to n_shards lists (shard = cell index mod n_shards, part shard of the level's region,
// counters n_cand_sh[image][level][shard]) instead of one.  A pair's launch is 2462 one-cell waves whose lifetime is ONE round: each ends
// with a reservation on one of sixteen (image, level) counters, and same-address atomics are served one after the other (~25 ns): the
// 880 reservations of a level-0 counter alone take 22 us -- stamps build, tools/exp/fast_stamps.sh: a wave's own work 6.3 us, the wait for
// its reservation 19.9 us on average, the launch 32 us.  The small-launch quadtree reads the shards as one list (k_quadtree.hip).  The
// batches keep one list per level: their counters are 8 per image and the reservation is issued a whole cell before it is used.
template <int PP, int PV, bool SH = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fast(const LevelDev* __restrict__ lv, const CellDev* __restrict__ cells,
                                             const uint8_t* __restrict__ pyr, size_t img_pitch, int t_hi, int t_lo,
                                             uint32_t* __restrict__ cand, size_t cand_pitch, int32_t* __restrict__ n_cand,
                                             int n_levels, int cell_first, int n_cells, int lds_v_off, int lds_q_off, int q_cap,
                                             int n_groups, int32_t* __restrict__ n_cand_sh, int n_shards) {
  extern __shared__ uint32_t lds_all[];
  // One wave per workgroup (four waves per workgroup measured 10 % slower: the LDS of a workgroup stays allocated until its slowest
  // wave is done); the kernel gains from every extra resident wave (21 -> 26 waves per CU: -7 %), so the LDS carve-up is per launch
  // and as tight as the launch's largest patch allows.
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

  const int lane = threadIdx.x & 63;
  const int img = blockIdx.y;
  int ci = __builtin_amdgcn_readfirstlane(blockIdx.x);
  if (ci >= n_cells) return;
  FS_DECL
  // ---- patch -> registers: 16-byte units (3 per row for the 30-px grid's patches: 21 rows per pass of the wave), every pass of a lane
  //      requested at once (rows clamped, so the loads are unconditional).  The global address is only 4-byte aligned (x0 - xa);
  //      gfx950 takes that for dwordx4.
  constexpr int UPR = (PP + 15) / 16;             // 16-byte units per patch row (pitch 40: the third one is half a unit)
  constexpr int RPP = 64 / UPR;                   // rows per pass
  constexpr int MUL = (128 + UPR - 1) / UPR;      // lane / UPR == (lane * MUL) >> 7 for lane < 64 (UPR = 3: 43, UPR = 5: 26)
  constexpr int NB = 2;                          // passes kept in flight: 42 rows at three units per row
  static_assert(PP % 8 == 0 && (UPR == 3 || UPR == 5), "patch pitch");
  // (a lane's row / unit recomputed where needed, behind an opaque copy of the lane number: held across the cell loop the two values cost
  //  the registers that tip the kernel over 64 -- a spill whose reload is a vector memory operation pending at the loop's back edge)
#define FAST_ROW_PART()           \
  int lane_o = lane;              \
  asm volatile("" : "+v"(lane_o)); \
  const int row0 = mul24u(lane_o, MUL) >> 7, part = lane_o - mul24u(row0, UPR) /* (24-bit products: full rate; the 32-bit ones are quarter rate) */
  const uint8_t* img_base = pyr + (size_t)img * img_pitch;
  auto patch_src = [&](const CellDev& c) __attribute__((always_inline)) -> const uint8_t* {
    const LevelDev& Lc = lv[c.level];
    return img_base + Lc.plane_off + (size_t)c.y0 * Lc.stride + (c.x0 - (c.x0 & 3));
  };
  // (every lane loads, rows and units clamped into the patch: a load behind a lane condition leaves the compiler merging old and new
  //  register values right behind it -- a wait for the data where it was only meant to be requested)
  // (dry, wave-uniform: the request of a wave's last cell, which has no successor -- every lane asks for the patch's first 16 bytes, one
  //  cache line per load instead of a patch that would be dropped: the whole patch again cost 16 % more fetched bytes per sweep)
  auto fetch = [&](const CellDev& c, uint4 (&wv)[NB], bool dry) __attribute__((always_inline)) {
    FAST_ROW_PART();
    const int nbytes = ((c.x0 & 3) + c.pw + 3) & ~3;  // bytes of a patch row that are needed
    const uint8_t* src = patch_src(c);
    const int stride = dry ? 0 : lv[c.level].stride;
    const int xoff = (16 * part < nbytes && !dry) ? 16 * part : 0;  // (a unit that starts past the needed bytes may lie past the row: it loads the row's first instead and is never parked)
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int r = min(row0 + RPP * it, c.ph - 1);
      wv[it] = *(const uint4*)(src + (uint32_t)mad24u(r, stride, xoff));  // full-rate 24-bit product: rows and strides < 2^13
    }
  };
  // the cell's record through the scalar cache (16 bytes, 16-byte aligned: left as a struct of 16-bit fields it is fetched with vector
  // loads -- a memory round trip before the patch can even be requested)
  auto load_cell = [&](int idx) __attribute__((always_inline)) -> CellDev {
    static_assert(sizeof(CellDev) == 16, "CellDev");
    const uint4 raw = *(const uint4*)(cells + (cell_first + idx));
    CellDev c;
    c.level = (int16_t)(raw.x & 0xFFFFu);
    c.x0 = (int16_t)(raw.x >> 16);
    c.y0 = (int16_t)(raw.y & 0xFFFFu);
    c.pw = (int16_t)(raw.y >> 16);
    c.ph = (int16_t)(raw.z & 0xFFFFu);
    c.offx = (int16_t)(raw.z >> 16);
    c.offy = (int16_t)(raw.w & 0xFFFFu);
    c.pad = 0;
    return c;
  };
  auto park_unit = [&](int r, int part, const uint4& w) __attribute__((always_inline)) {
    if (PP % 16 == 0) {
      ((uint4*)lds_all)[r * UPR + part] = w;
    } else {  // rows 8-byte aligned only: two 8-byte stores, the half unit at the end of a row one
      uint2* d = (uint2*)(P + (uint32_t)mad24u(r, PP, 16 * part));  // (as r * PP + ... the offset was a 64-bit quarter-rate multiply-add)
      d[0] = make_uint2(w.x, w.y);
      if (part < UPR - 1) d[1] = make_uint2(w.z, w.w);
    }
  };

  // registers -> LDS (both passes were in flight before the first is consumed), the rows past the prefetched ones straight from memory
  // (patches taller than 42 / 24 rows)
  auto park = [&](const CellDev& c, uint4 (&wv)[NB]) __attribute__((always_inline)) {
    FAST_ROW_PART();
    const int nbytes = ((c.x0 & 3) + c.pw + 3) & ~3;
    if (c.pw > 6 && c.ph > 6 && row0 < RPP && 16 * part < nbytes) {
#pragma unroll
      for (int it = 0; it < NB; ++it) {
        const int r = row0 + RPP * it;
        if (r < c.ph) park_unit(r, part, wv[it]);
      }
      if (c.ph > RPP * NB) {  // wave-uniform
        const uint8_t* src = patch_src(c);
        const int stride = lv[c.level].stride;
        for (int r = row0 + RPP * NB; r < c.ph; r += RPP) park_unit(r, part, *(const uint4*)(src + (uint32_t)mad24u(r, stride, 16 * part)));
      }
    }
  };

  // every lane "uses" the prefetched registers at ONE place that every path through the cell loop crosses (the last cell's included): a
  // wait left on some paths only leaves the loads pending on the others as far as the compiler can tell, and it then waits -- for
  // everything in flight, the reservation included -- wherever those registers are next written
#define FAST_PATCH_ARRIVED() \
  _Pragma("unroll") for (int it = 0; it < NB; ++it) asm volatile("" : "+v"(wv[it].x), "+v"(wv[it].y), "+v"(wv[it].z), "+v"(wv[it].w))

  CellDev cell = load_cell(ci);
  uint4 wv[NB];
  fetch(cell, wv, false);
  FAST_PATCH_ARRIVED();
  FS(0)  // cell record + patch arrived
  park(cell, wv);
  // the previous cell's records, one per lane, waiting for its list reservation (lane 0 of base_prev) to return
  uint32_t rec_prev = 0u;
  int n_prev = 0, base_prev = 0;
  uint32_t* out_prev = cand;

  for (;;) {
    const LevelDev& L = lv[cell.level];
    const int pw = cell.pw, ph = cell.ph;
    const int iw = pw - 6, ih = ph - 6;  // interior cv::FAST scans: rows/cols 3 .. size-4
    const int xa = cell.x0 & 3;
    const int ci_next = ci + n_groups;
    const bool has_next = ci_next < n_cells;
    const bool live = iw > 0 && ih > 0;
    int nq = 0, n_keep = 0;
    if (live) {
      // ---- zero the V map (with border), 16 bytes per store (its carve-up is rounded to 16) ----
      {
        const int vq = ((ih + 2) * PV + 3 + 15) >> 4;
        uint4* V128 = (uint4*)V;
        for (int k = lane; k < vq; k += 64) V128[k] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    // the next cell's patch travels while this one is worked on (unconditionally -- a wave's last cell makes a dry request --: behind
    // `if (has_next)` the registers become a merge of old and new values, copied, and so waited for, at once)
    const CellDev cell_next = load_cell(has_next ? ci_next : ci);
    fetch(cell_next, wv, !has_next);
    WAVE_SYNC();
    FS(1)  // parked, map zeroed, next requested (waits for it in this build)

    if (live) {
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
#if !FAST_SLIDE
          const int thr_x = ix < iw ? t_pass : 0x7FFF;
          const uint8_t* a0 = P + ly * PP + xa + ix;  // top-left corner of the pixel's 7x7 window
          const uint32_t e_lane = (uint32_t)ix | ((uint32_t)ly << 7);
          const uint8_t* a = a0;
#endif
#if FAST_SLIDE
          // r6: a lane walks DOWN its column (row group ly takes rows ly H2 .. ly H2 + H2 - 1, one row per trip), so the ring pixels it
          // reads recur: column 3 of the window is read at rows t, t + 3, t + 6 (r8, v, r0) and columns 1 / 5 at rows t + 1, t + 5 (r10 /
          // r14, r6 / r2) -- each value fetched ONCE into a 7-slot register ring per column and used again three / four trips later; only
          // columns 0 and 6 (r12, r4) are read once anyway.  With all three rings 5 LDS reads per trip instead of 9 (+ 14 per column block to
          // fill them).  Unrolled by 7 so that the ring slots are register names.  What it buys (profiles/NOTES_r6.md, same-box A/B): LDS
          // instructions per wave 1151 -> 927, LDS-array cycles 2680 -> 2438, k_fast alone -0.7 % (rect) / -2.2 % (camera-like) -- and the step
          // within +-0.2 %: the kernel's LDS was not what held the step either.
          {
            // rows per row group (rpi = 64 >> shift groups).  (With the 30-px grid's patches -- 31 interior rows, pitch 40 -- the two groups sit
            // 640 bytes apart, i.e. on the same LDS banks, and the scoring trips, which read the survivors of both groups together, take 70 % more
            // bank-conflict cycles than with r5's row order: 483 against 288 of ~2600 LDS cycles per wave, tools/exp/lds_sq.sh.  One row more in
            // the first group (17 + 14: 680 bytes apart) does not change that and costs a seventeenth trip per cell: measured slower, dropped.)
            const int H2 = (ih + rpi - 1) >> (6 - shift);
            const int row0 = mul24u(ly, H2);
            const uint8_t* aw = P + row0 * PP + xa + ix;      // window of the lane's first row
            const uint32_t e_row0 = (uint32_t)ix | ((uint32_t)row0 << 7);
            // which lanes hold an interior pixel in trip t -- as SCALAR masks: the columns right of the interior never do (col_mask), the last row
            // group runs out of rows at t_full = ih - (rpi - 1) H2 (the other groups have all H2), so the verdict masks are cut on the scalar unit
            // and the threshold stays the wave-uniform t_pass: no per-lane threshold, no vector instruction for the bounds
            const unsigned long long col_mask = __ballot(ix < iw);
            const unsigned long long not_last = rpi > 1 ? ((1ull << (64 - lw)) - 1ull) : 0ull;
            const int t_full = ih - (rpi - 1) * H2;
            int w3[7], w1[7], w5[7];
#pragma unroll
            for (int k = 0; k < 6; ++k) w3[k] = aw[k * PP + 3];
#if FAST_SLIDE >= 2
#pragma unroll
            for (int k = 1; k < 5; ++k) w1[k] = aw[k * PP + 1];
#endif
#if FAST_SLIDE >= 3
#pragma unroll
            for (int k = 1; k < 5; ++k) w5[k] = aw[k * PP + 5];
#endif
            // (every predicate is ONE vector compare whose lane mask feeds the ballot and the branch directly; the two masks are combined on the
            //  scalar unit and handed back as lane predicates -- see the FAST_SLIDE = 0 form below for the measurements behind that)
            auto trip_s = [&](const int t, const int k) __attribute__((always_inline)) {
              const uint8_t* a = aw + t * PP;
              w3[(k + 6) % 7] = a[6 * PP + 3];
              const int r12 = a[3 * PP], r4 = a[3 * PP + 6];
              const int r8 = w3[k % 7], v = w3[(k + 3) % 7], r0 = w3[(k + 6) % 7];
#if FAST_SLIDE >= 2
              w1[(k + 5) % 7] = a[5 * PP + 1];
              const int r10 = w1[(k + 1) % 7], r14 = w1[(k + 5) % 7];
#else
              (void)w1;
              const int r10 = a[PP + 1], r14 = a[5 * PP + 1];
#endif
#if FAST_SLIDE >= 3
              w5[(k + 5) % 7] = a[5 * PP + 5];
              const int r6 = w5[(k + 1) % 7], r2 = w5[(k + 5) % 7];
#else
              (void)w5;
              const int r2 = a[5 * PP + 5], r6 = a[PP + 5];
#endif
              const int lo_of_hi = FAST_MIN16(FAST_MIN16(FAST_MAX16(r0, r8), FAST_MAX16(r4, r12)), FAST_MIN16(FAST_MAX16(r2, r10), FAST_MAX16(r6, r14)));
              const int hi_of_lo = FAST_MAX16(FAST_MAX16(FAST_MIN16(r0, r8), FAST_MIN16(r4, r12)), FAST_MAX16(FAST_MIN16(r2, r10), FAST_MIN16(r6, r14)));
              const int sb = lo_of_hi - v, sd = v - hi_of_lo;
              const unsigned long long valid = t < t_full ? col_mask : (col_mask & not_last);
              const unsigned long long mb = __ballot(sb > t_pass) & valid, md = __ballot(sd > t_pass) & valid;
              const unsigned long long m = mb | md, m2 = mb & md;
              const bool any = __builtin_amdgcn_inverse_ballot_w64(m), dual = __builtin_amdgcn_inverse_ballot_w64(m2);
              const bool pd = __builtin_amdgcn_inverse_ballot_w64(md);
              const uint32_t eh = e_row0 + ((uint32_t)t << 7);
              if (any) Q[nq + mbcnt64(m, 0)] = (uint16_t)(eh | (pd ? 0u : Q_BRIGHT) | (dual ? Q_DUAL : 0u));
              nq += __popcll(m);
            };
            for (int t0 = 0; t0 < H2; t0 += 7) {
#pragma unroll
              for (int k = 0; k < 7; ++k) {
                const int t = t0 + k;
                if (t >= H2) break;  // wave-uniform
                trip_s(t, k);
              }
            }
          }
#else
          // every predicate is ONE vector compare whose lane mask feeds the ballot and the branch directly (a predicate built from
          // several masks is expanded to 0 / 1 per lane and compared again before a ballot: two more VALU instructions each)
          auto trip = [&](const int y0, const int thr) __attribute__((always_inline)) {
            const int v = a[3 * PP + 3];
            const int r0 = a[6 * PP + 3], r8 = a[3], r4 = a[3 * PP + 6], r12 = a[3 * PP];
            const int r2 = a[5 * PP + 5], r10 = a[PP + 1], r6 = a[PP + 5], r14 = a[5 * PP + 1];
            // (the 14 minima / maxima as 16-BIT instructions: on gfx950 the non-packed 16-bit VOP2 arithmetic goes through a SIMD at ~1.8 x
            //  the rate of the 32-bit v_min / v_max / v_min3 at eight waves per SIMD -- profiles/r5_valu_census.txt -- and the operands are
            //  bytes; the results' upper halves are zero on this generation, so the 32-bit subtractions below read them as they are)
            const int lo_of_hi = FAST_MIN16(FAST_MIN16(FAST_MAX16(r0, r8), FAST_MAX16(r4, r12)), FAST_MIN16(FAST_MAX16(r2, r10), FAST_MAX16(r6, r14)));
            const int hi_of_lo = FAST_MAX16(FAST_MAX16(FAST_MIN16(r0, r8), FAST_MIN16(r4, r12)), FAST_MAX16(FAST_MIN16(r2, r10), FAST_MIN16(r6, r14)));
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
          };
          // lanes right of the interior can never pass (thr_x), so the column test costs nothing per trip; the row test is only needed in
          // the last, partial trip of a column block (its rows past the interior are read -- they lie inside this wave's carve-up -- but
          // cannot pass either)
          int y0 = 0;
          for (; y0 + rpi <= ih; y0 += rpi, a += rpi * PP) trip(y0, thr_x);
          if (y0 < ih) trip(y0, y0 + ly < ih ? thr_x : 0x7FFF);
#endif
        }
      }
      WAVE_SYNC();
      FS(2)  // necessary test

      // ---- 3. exact test + score for the survivors, one queue entry per lane (trips of 64).  Only about
      //         half of the survivors are corners: those (and the dual-tagged entries, whose second polarity is still to come) are
      //         compacted in place at the front of Q, so that the NMS and the output pass touch no entry that cannot matter.  The
      //         dual-tagged ones (both polarities passed: the main entry scores the dark one) are listed once more from the back of
      //         Q for the bright one -- here, in 3.4 trips per cell, not in the test's 15.
      int nc = 0;
      for (int j0 = 0; j0 < nq; j0 += 64) {
        const bool act = j0 + lane < nq;
        const uint32_t q = Q[min(j0 + lane, nq - 1)];
        const int ix = Q_IX(q), iy = Q_IY(q);
        const int A = arc_score1<PP>(P + iy * PP + xa + ix, (q & Q_BRIGHT) ? -1 : 1);
        const bool c = act && A > t_pass;
        if (c) V[(iy + 1) * PV + ix + 1] = (uint8_t)min(255, A);
        const bool dual = act && (q & Q_DUAL);
        const bool keep = c || dual;
        const unsigned long long m = __ballot(keep), m2 = __ballot(dual);
        if (keep) Q[nc + mbcnt64(m, 0)] = (uint16_t)q;  // (in place: this trip's entries were all read above, later trips read further back)
        nc += __popcll(m);
        if (m2) {
          const int k = __popcll(m2);
          if (nq + nd + k <= q_cap) {  // (the unread part of the front ends at nq)
            if (dual) Q[q_cap - 1 - (nd + mbcnt64(m2, 0))] = (uint16_t)q;
            nd += k;
          } else {
            d_overflow = true;  // (pathological cell) the tagged entries are re-scanned one by one below
          }
        }
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
      FS(3)  // scoring

        // ---- 4. NMS over the queue: an entry is kept iff its score beats its 8 neighbours' (0 where nothing was scored); the kept ones
        //         are compacted at the front of Q (in place: a trip's entries are all read before it writes, and it writes no further
        //         than it has read)
        n_keep = 0;
        for (int q0 = 0; q0 < nq; q0 += 64) {
          const int q = q0 + lane;
          bool is_max = false;
          uint32_t e = 0u;
          if (q < nq) {
            e = Q[q];
            const uint8_t* c = V + Q_IY(e) * PV + Q_IX(e);  // top-left corner of the 3x3 neighbourhood
            // all nine reads requested at once, one compare against the largest neighbour (v > max >= 0 also says v != 0): written as a
            // chain of && the compiler made nine dependent LDS round trips of it, each behind its own branch
            const int v = c[PV + 1];
            const int n0 = c[0], n1 = c[1], n2 = c[2], n3 = c[PV], n4 = c[PV + 2], n5 = c[2 * PV], n6 = c[2 * PV + 1], n7 = c[2 * PV + 2];
            is_max = v > max(max(max(n0, n1), n2), max(max(max(n3, n4), n5), max(n6, n7)));  // (scores in the map are > t_pass by construction)
          }
          const unsigned long long m = __ballot(is_max);
          if (is_max) Q[n_keep + mbcnt64(m, 0)] = (uint16_t)Q_XY(e);
          n_keep += __popcll(m);
        }
        WAVE_SYNC();
        FS(4)  // NMS
        if (n_keep > 0) break;
      }
    }
    // ---- 5. output.  The list of a level is a SET: the quadtree orders candidates by (cell, y, x) recomputed from the coordinates, so
    //         cells may append in any order -- one atomic reservation per cell, then the wave writes its records.
    if (n_prev > 0) {  // the previous cell's: its reservation was issued a whole cell ago
      const int b = __builtin_amdgcn_readfirstlane(base_prev);
      if (lane < n_prev) out_prev[b + lane] = rec_prev;
      n_prev = 0;
    }
    // A cell's records (almost always <= 64) go to registers, the NEXT cell's patch is parked -- that waits for the patch alone: the last
    // reservation has just been consumed above --, and only then is this cell's reservation issued; its value is looked at after
    // the next cell's work.
    bool defer = false;
    const int shard = SH ? (ci & (n_shards - 1)) : 0;  // (n_shards: a power of two)
    uint32_t* const out = cand + (size_t)img * cand_pitch + L.cand_base + (SH ? (size_t)shard * L.shard_cap : (size_t)0);
    int32_t* const counter = SH ? &n_cand_sh[((size_t)img * n_levels + cell.level) * n_shards + shard] : &n_cand[(size_t)img * n_levels + cell.level];
    auto record = [&](int j) __attribute__((always_inline)) -> uint32_t {
      const uint32_t e = Q[min(j, n_keep - 1)];
      const int ix = Q_IX(e), iy = Q_IY(e);
      return ORBFE_PACK_XYR(ix + 3 + cell.offx, iy + 3 + cell.offy, V[(iy + 1) * PV + ix + 1] - 1);
    };
    // (this file is compiled with -mllvm -amdgpu-atomic-optimizer-strategy=None, see the Makefile: the optimizer rewrites an atomicAdd on
    //  a uniform address into a wave reduction that waits for the returned value on the spot; left alone the instruction returns into
    //  lane 0's register and the compiler waits where that is first read)
    auto reserve = [&]() __attribute__((always_inline)) -> int {
      int base = 0;
      if (lane == 0) base = atomicAdd(counter, n_keep);
      return base;
    };
    if (n_keep > 0) {
      if (has_next && n_keep <= 64) {
        rec_prev = record(lane);
        defer = true;
      } else {
        const int b = __builtin_amdgcn_readfirstlane(reserve());
        for (int j0 = 0; j0 < n_keep; j0 += 64) {
          const uint32_t r = record(j0 + lane);
          if (j0 + lane < n_keep) out[b + j0 + lane] = r;
        }
      }
    }
    FAST_PATCH_ARRIVED();
    FS(5)  // output
    if (!has_next) break;
    WAVE_SYNC();  // (the records above were read from Q / V before the next cell's patch and map overwrite them)
    park(cell_next, wv);
    if (defer) {
      base_prev = reserve();
      n_prev = n_keep;
      out_prev = out;
    }
    ci = ci_next;
    cell = cell_next;
  }
  FS_END
}

#ifdef FAST_STAMPS
}  // namespace orbfe
extern "C" void orbfe_debug_fast_stamps() {
  static unsigned long long rec[8192][8];
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(rec, HIP_SYMBOL(orbfe::g_fs_rec), sizeof(rec));
  const char* names[6] = {"arrive", "park", "necessary", "scoring", "nms", "output"};
  unsigned long long first = ~0ull, last = 0, n = 0, sum[7] = {0}, mx[7] = {0};
  for (int w = 0; w < 8192; ++w) {
    if (!rec[w][7]) continue;
    ++n;
    first = first < rec[w][6] ? first : rec[w][6], last = last > rec[w][7] ? last : rec[w][7];
    for (int k = 0; k < 6; ++k) sum[k] += rec[w][k], mx[k] = mx[k] > rec[w][k] ? mx[k] : rec[w][k];
    const unsigned long long life = rec[w][7] - rec[w][6];
    sum[6] += life, mx[6] = mx[6] > life ? mx[6] : life;
  }
  printf("FAST stamps (last launch): %llu waves, first start .. last end %.2f us\n", n, (last - first) * 0.01);
  for (int k = 0; k < 6; ++k) printf("  %-10s mean %7.2f us  max %7.2f us\n", names[k], n ? sum[k] * 0.01 / n : 0.0, mx[k] * 0.01);
  printf("  %-10s mean %7.2f us  max %7.2f us\n", "wave", n ? sum[6] * 0.01 / n : 0.0, mx[6] * 0.01);
  // start times by decile of the launch, end times
  unsigned long long smax = 0;
  for (int w = 0; w < 8192; ++w) if (rec[w][7]) smax = smax > rec[w][6] - first ? smax : rec[w][6] - first;
  printf("  last wave started %.2f us after the first\n", smax * 0.01);
  memset(rec, 0, sizeof(rec));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(orbfe::g_fs_rec), rec, sizeof(rec));
}
namespace orbfe {
#endif

#undef FAST_PATCH_ARRIVED
#undef FAST_ROW_PART
#undef WAVE_SYNC

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
                              int cell_first, int n_cells, int n_img, int cpw_force, int32_t* d_n_cand_sh = nullptr, int n_shards = 1) {
  int v_off, q_off, q_cap, total;
  const int pp = max_pw <= FAST_PP40_MAX ? 40 : (max_pw <= 44 ? 48 : 80), pv = max_pw <= FAST_PV36_MAX ? 36 : (max_pw <= 44 ? 40 : 72);
  fast_lds_layout(max_pw, max_ph, pp, pv, &v_off, &q_off, &q_cap, &total);
  // cells per wave: one for the drop-in path's launches (every cell its own wave: the launch is a single wave lifetime), FAST_CPW where the
  // launch holds many rounds of waves anyway (tools/exp/cpw_by_batch.sh, ms per step of 16 / 64 / 128 / 256 pairs: this rule 0.394 / 0.909 / 1.522 /
  // 2.654, one cell everywhere 0.403 / 0.910 / 1.542 / 2.695, four everywhere 0.476 / 0.933 / 1.541 / 2.683)
  // (late r4, after the other kernels' diets: eight cells per wave in the launches of 128 k cells and more -- 5.18 -> 5.14 ms per 512 pairs)
  const long long work = (long long)n_cells * n_img;
  const int cpw = cpw_force > 0 ? cpw_force : (work >= 131072 ? 2 * FAST_CPW : (work >= 65536 ? FAST_CPW : 1));  // (ORBFE_FAST_CPW: the tests' way into the cell loop with small inputs)
  // a multiple of 8: the cell table is in XCD order (orbfe_create: table position = strip (mod 8), workgroups go round-robin to the 8 XCDs),
  // so a wave's cells ci, ci + n_groups ... stay on its XCD's strip and the launch's rows of workgroups start on XCD 0 for every image
  const int n_groups = ((n_cells + cpw - 1) / cpw + 7) & ~7;
#define FAST_GO(K) hipLaunchKernelGGL(K, dim3(n_groups, n_img), dim3(64), total, s, d_lv, d_cells, d_pyr, img_pitch, t_hi, t_lo, d_cand, \
                                      cand_pitch, d_n_cand, n_levels, cell_first, n_cells, v_off, q_off, q_cap, n_groups, d_n_cand_sh, n_shards)
  if (d_n_cand_sh && n_shards > 1) {  // a frame or two: sharded lists
    if (pp == 40) FAST_GO((k_fast<40, 36, true>));
    else if (pp == 48 && pv == 36) FAST_GO((k_fast<48, 36, true>));
    else if (pp == 48) FAST_GO((k_fast<48, 40, true>));
    else FAST_GO((k_fast<80, 72, true>));
  } else if (pp == 40) FAST_GO((k_fast<40, 36>));
  else if (pp == 48 && pv == 36) FAST_GO((k_fast<48, 36>));
  else if (pp == 48) FAST_GO((k_fast<48, 40>));
  else FAST_GO((k_fast<80, 72>));
#undef FAST_GO
}

bool fast_single_launch(const LevelDev* h_lv, const int* lvl_max_pw, const int* lvl_max_ph, int n_levels, int n_img) {
  int total_cells = 0, max_pw = 0, max_ph = 0;
  for (int l = 0; l < n_levels; ++l) {
    total_cells += h_lv[l].n_cells;
    max_pw = std::max(max_pw, lvl_max_pw[l]);
    max_ph = std::max(max_ph, lvl_max_ph[l]);
  }
  return n_img > 0 && (long long)total_cells * n_img <= 16384 && max_pw > 6 && max_ph > 6;
}

// One launch per pyramid level: the cells of a level have (almost) one size, so each launch reserves just the LDS its
// patches need.  Measured: merging levels 0..3 into one launch with their common carve-up is 5 % slower than the four separate launches.
void launch_fast(hipStream_t s, const LevelDev* d_lv, const CellDev* d_cells, const LevelDev* h_lv, const int* lvl_max_pw,
                 const int* lvl_max_ph, const uint8_t* d_pyr, size_t img_pitch, int t_hi, int t_lo, uint32_t* d_cand, size_t cand_pitch,
                 int32_t* d_n_cand, int n_levels, int n_img, int cpw_force, uint32_t level_mask, bool merge_masked, int32_t* d_n_cand_sh, int n_shards) {
  // d_n_cand_sh / n_shards > 1: the caller has checked fast_single_launch() -- only the one-launch form shards its lists
  // level_mask: the levels this call launches (bit l; ~0u: all) -- an experiment of r6 puts some levels' launches on a second stream (run_extract)
  // (the launches of the small levels on a second stream beside the large ones, or alternating levels on two streams, were measured in
  //  rounds 2-3 and dropped: profiles/NOTES_r1-r3.md)
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
  if ((long long)total_cells * n_img <= 16384 && max_pw > 6 && max_ph > 6 && level_mask == ~0u) {
    launch_fast_cells(s, d_lv, d_cells, max_pw, max_ph, d_pyr, img_pitch, t_hi, t_lo, d_cand, cand_pitch, d_n_cand, n_levels, 0, total_cells,
                      n_img, cpw_force, d_n_cand_sh, n_shards);
    return;
  }
  if (merge_masked && level_mask != ~0u) {
    // (experiment, r6) the masked levels as ONE launch: they must be consecutive in the cell table; the carve-up is their largest patch's
    int first = -1, last = -1, cells = 0, mpw = 0, mph = 0;
    bool contiguous = true;
    for (int l = 0; l < n_levels; ++l) {
      if (!((level_mask >> l) & 1u) || h_lv[l].n_cells <= 0 || lvl_max_pw[l] <= 6 || lvl_max_ph[l] <= 6) continue;
      if (first < 0) first = l;
      if (last >= 0 && l != last + 1) contiguous = false;
      last = l, cells += h_lv[l].n_cells, mpw = std::max(mpw, lvl_max_pw[l]), mph = std::max(mph, lvl_max_ph[l]);
    }
    if (first >= 0 && contiguous && (int)h_lv[last].cell_base + h_lv[last].n_cells - (int)h_lv[first].cell_base == cells) {
      launch_fast_cells(s, d_lv, d_cells, mpw, mph, d_pyr, img_pitch, t_hi, t_lo, d_cand, cand_pitch, d_n_cand, n_levels, (int)h_lv[first].cell_base, cells,
                        n_img, cpw_force);
      return;
    }
  }
  for (int l = 0; l < n_levels; ++l) {
    const int n_cells = h_lv[l].n_cells;
    if (n_cells <= 0 || lvl_max_pw[l] <= 6 || lvl_max_ph[l] <= 6 || !((level_mask >> l) & 1u)) continue;
    launch_fast_cells(s, d_lv, d_cells, lvl_max_pw[l], lvl_max_ph[l], d_pyr, img_pitch, t_hi, t_lo, d_cand, cand_pitch, d_n_cand, n_levels,
                      (int)h_lv[l].cell_base, n_cells, n_img, cpw_force);
  }
}

}  // namespace orbfe
