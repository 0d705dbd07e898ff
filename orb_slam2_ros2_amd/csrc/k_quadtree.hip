// k_quadtree.hip -- spatially uniform top-N keypoint selection, one wavefront per (image, level).
//
// Replaces Quadtree / QuadtreeNode (include/ORB_SLAM2/ORBExtractor.h:18-93,
// src/ORB_SLAM2/src/ORBExtractor.cc:19-192) as driven by extractFast (:376-386).
//
// The algorithm is inherently sequential (best-first expansion of a priority queue), so the parallelism
// is (a) across the (image, level) pairs of a batch -- one wave each -- and (b) inside one expansion step.
// The cost that matters is the LATENCY of one pop+split (a level needs 40-150 of them, one after the
// other), so everything a step touches lives in LDS:
//   * node table in LDS, 16 bytes per node: key = count<<32 | ~insertion_seq, segment begin | pre-partition code, and the node's
//     PATH (strip, depth, column / row index at that depth: two bits per halving).  The fp64 bounds are not stored: a split replays
//     the reference's own halvings (b + e) / 2 from the strip bounds down the path (node_mid) -- a node with two or more records spans
//     more than a pixel, so it is at most 11 halvings deep (coordinates are 12-bit) and its children 12;
//   * the candidate records (x:12 | y:12 | response:8, 4 bytes) in ONE LDS array; a node owns a contiguous
//     segment; a node with <= 512 records is split IN PLACE through registers (8 records per lane, ballots
//     give the stable 4-way partition); only the few big nodes near the root bounce through a global
//     scratch buffer; levels whose candidates do not fit the LDS array fall back to global memory for
//     everything (same code, generic pointers);
//   * pop = arg-max of the key = std::multimap<size_t,...,greater>::begin() with insertion-order ties:
//     7 LDS reads per lane + two 32-bit DPP wave reductions.
// Membership is the reference's strict test against double-precision bounds (ORBExtractor.h:55-62),
// midpoints (b+e)/2 in fp64, so points on a split line are dropped exactly as in the reference.  No
// "single point => stop" rule, exact min(quota, nodes) truncation, per-node first-maximum response,
// output ordered by candidate index (std::set) -- quirks Q3/Q4 of SURVEY.md.  The candidate index itself
// is not carried: candidate order is (cell row, cell column, y, x), recomputed from the coordinates.
#include <hip/hip_runtime.h>

#include <mutex>

#include "orbfe_internal.h"
#include "wave_ops.h"
#include "blur_body.h"

#include <cmath>
#include <vector>

namespace orbfe {

#define QT_INPLACE_CHUNKS 4  // nodes up to QT_INPLACE_CHUNKS*64 records are partitioned in registers

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t identity, uint32_t v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
// full-wave max / min of a u32 (all 64 lanes active); result broadcast to every lane
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, dpp_u32<0x111, 0xf>(0u, v));  // row_shr:1
  v = max(v, dpp_u32<0x112, 0xf>(0u, v));  // row_shr:2
  v = max(v, dpp_u32<0x114, 0xf>(0u, v));  // row_shr:4
  v = max(v, dpp_u32<0x118, 0xf>(0u, v));  // row_shr:8
  v = max(v, dpp_u32<0x142, 0xa>(0u, v));  // row_bcast:15 -> rows 1,3
  v = max(v, dpp_u32<0x143, 0xc>(0u, v));  // row_bcast:31 -> rows 2,3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return ~wave_max_u32(~v); }
// full-wave sum (the row_shr steps leave a row's total in its lane 15, the broadcasts chain the rows)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  v += dpp_u32<0x111, 0xf>(0u, v);
  v += dpp_u32<0x112, 0xf>(0u, v);
  v += dpp_u32<0x114, 0xf>(0u, v);
  v += dpp_u32<0x118, 0xf>(0u, v);
  v += dpp_u32<0x142, 0xa>(0u, v);
  v += dpp_u32<0x143, 0xc>(0u, v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// max over the lanes BELOW this one (0 for lane 0; the values are >= 0).  Scans by data-parallel-primitive moves (wave_ops.h): the
// __shfl_up versions were six / seven dependent ds_bpermute round trips each, ~1.5 k cycles per batched step and 11 k in the
// sixteen-row prefix of the pre-partition (stamps build, tools/exp/qt_stamps.sh).
__device__ __forceinline__ int wave_excl_max(int v, int lane) {
  const int incl = wave_incl_scan_dpp<OpMaxI>(v);
  const int prev = __builtin_amdgcn_update_dpp(0, incl, 0x138, 0xf, 0xf, true);  // wave_shr:1 (lane 0 reads 0)
  return lane > 0 ? prev : 0;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
  (void)lane;
  return wave_incl_scan_dpp<OpAddI>(v);
}

// Integer form of the strict fp64 membership test: coordinates are integers, so  x < mid  <=>  x <= ceil(mid) - 1  and
// x > mid  <=>  x >= floor(mid) + 1  (a point with x == mid, possible only for an integral mid, belongs to neither child).
struct SplitInt {
  int x_lt, x_gt, y_lt, y_gt;  // x <= x_lt: left child; x >= x_gt: right child (likewise rows)
};
__device__ __forceinline__ SplitInt make_split(double midx, double midy) {
  SplitInt s;
  s.x_lt = (int)ceil(midx) - 1;
  s.x_gt = (int)floor(midx) + 1;
  s.y_lt = (int)ceil(midy) - 1;
  s.y_gt = (int)floor(midy) + 1;
  return s;
}
__device__ __forceinline__ int quadrant_of(uint32_t p, const SplitInt& s) {
  const int x = (int)ORBFE_REC_X(p), y = (int)ORBFE_REC_Y(p);
  const int qx = (x <= s.x_lt) ? 0 : ((x >= s.x_gt) ? 1 : -1);
  const int qy = (y <= s.y_lt) ? 0 : ((y >= s.y_gt) ? 1 : -1);
  return (qx >= 0 && qy >= 0) ? (qy * 2 + qx) : -1;  // rows outer, cols inner (ORBExtractor.cc:60-72)
}

// ---- node path -------------------------------------------------------------------------------------------------------------------
// strip (4 bits) | depth (4 bits) | column index at that depth (12 bits) | row index (12 bits).  The reference's child bounds are
// (b, (b + e) / 2) and ((b + e) / 2, e) per axis (ORBExtractor.cc:60-72), rows and columns halved independently, so the bounds of a
// node are a function of its strip and the halves taken: they are REPLAYED with the same fp64 operations when the node is split,
// never stored (32 of a node's 44 bytes were four doubles; at 16 bytes a tree's table is 7 KB and sixteen trees fit a CU).
// Depth <= 12 is enough: only nodes with two or more records are ever split (a table whose largest count is one can never reach
// the quota: every pop replaces a one-record node by at most one one-record child until all have vanished -- tree_body returns the
// empty selection at once), two distinct pixels strictly inside a node need it wider or higher than one pixel, and coordinates are
// below 4096.
#define QT_PATH(s, d, ix, iy) (((uint32_t)(s) << 28) | ((uint32_t)(d) << 24) | ((uint32_t)(ix) << 12) | (uint32_t)(iy))
#define QT_PATH_CHILD(p, q) \
  (((p)&0xF0000000u) | ((((p) >> 24) & 15u) + 1u) << 24 | ((((p) >> 11) & 0xFFEu) | ((uint32_t)(q)&1u)) << 12 | ((((p) << 1) & 0xFFEu) | ((uint32_t)(q) >> 1)))
__device__ __forceinline__ void node_mid(const LevelDev& L, uint32_t path, double& midx, double& midy) {
  const int s = (int)(path >> 28), d = (int)((path >> 24) & 15u);
  const uint32_t ix = (path >> 12) & 0xFFFu, iy = path & 0xFFFu;
  double cb = L.strips[s], ce = L.strips[s + 1], rb = 0.0, re = (double)L.reg_h;
  for (int k = d - 1; k >= 0; --k) {
    const double mx = (cb + ce) / 2, my = (rb + re) / 2;
    const bool hx = (ix >> k) & 1u, hy = (iy >> k) & 1u;
    cb = hx ? mx : cb;
    ce = hx ? ce : mx;
    rb = hy ? my : rb;
    re = hy ? re : my;
  }
  midx = (cb + ce) / 2;
  midy = (rb + re) / 2;
}

// candidate order of the reference = (cell row, cell column, y, x): cells are visited row-major and cv::FAST emits
// a patch in raster order (ORBExtractor.cc:346-373).  39-bit key, smaller = earlier.
__device__ __forceinline__ unsigned long long order_key(uint32_t rec, const LevelDev& L) {
  const uint32_t x = ORBFE_REC_X(rec), y = ORBFE_REC_Y(rec);
  // (coordinates are >= 3 and < 2^12, the reciprocals <= 2^20, cell counts < 2^8: full-rate 24-bit products; the plain ones compile
  // to the quarter-rate v_mul_lo_u32 / v_mad_u64_u32 -- this function sits in the winner scan of every node)
  const uint32_t jdx = min((uint32_t)mul24u((int)(x - 3u), (int)L.inv_w_cell) >> 20, (uint32_t)L.n_cols - 1u);
  const uint32_t idx = min((uint32_t)mul24u((int)(y - 3u), (int)L.inv_h_cell) >> 20, (uint32_t)L.n_rows - 1u);
  return ((unsigned long long)(uint32_t)mad24u((int)idx, L.n_cols, (int)jdx) << 24) | ((unsigned long long)y << 12) | (unsigned long long)x;
}

// the same order in 32 bits: cell << 14 | row inside the cell << 7 | column inside the cell (cells are at most 74 pixels wide and
// high -- the largest patch the context accepts is 80 -- and a level has < 2^18 of them).  64-bit integer compares are quarter-rate.
__device__ __forceinline__ uint32_t order_key32(uint32_t rec, const LevelDev& L) {
  const uint32_t x = ORBFE_REC_X(rec), y = ORBFE_REC_Y(rec);
  const uint32_t jdx = min((uint32_t)mul24u((int)(x - 3u), (int)L.inv_w_cell) >> 20, (uint32_t)L.n_cols - 1u);
  const uint32_t idx = min((uint32_t)mul24u((int)(y - 3u), (int)L.inv_h_cell) >> 20, (uint32_t)L.n_rows - 1u);
  const uint32_t dy = y - 3u - (uint32_t)mul24u((int)idx, L.h_cell), dx = x - 3u - (uint32_t)mul24u((int)jdx, L.w_cell);
  return ((uint32_t)mad24u((int)idx, L.n_cols, (int)jdx) << 14) | ((dy & 127u) << 7) | (dx & 127u);
}

// root strip of a record (Quadtree::initSplit children, strict membership) or -1
__device__ __forceinline__ int strip_of(uint32_t rec, const LevelDev& L) {
  const double x = (double)ORBFE_REC_X(rec), y = (double)ORBFE_REC_Y(rec);
  int s = -1;
  if (y > 0.0 && y < (double)L.reg_h) {
    for (int k = 0; k < L.n_ini; ++k)
      if (x > L.strips[k] && x < L.strips[k + 1]) s = k;
  }
  return s;
}

// ---- pre-partition of the top four tree levels ----------------------------------------------------------------------
// The first pops of a level split nodes of hundreds to thousands of records, one wave moving 64 records per step: measured,
// these partitions are half (records in LDS) to three quarters (records in global memory) of a tree's time.  But WHICH
// child a record falls into does not depend on the pop order, only on the midpoints, which are fixed by the strip bounds.
// So the records are scattered ONCE, in parallel, into the layout four levels of splitting would produce:
//   node segment = [child 0 | child 1 | child 2 | child 3 | records on its split lines]   (recursively, four levels deep)
// and popping a strip or one of its descendants down to the great-grandchildren only reads four precomputed totals (a level of
// 1241x376 with quota 434 never pops deeper: its pops are 4 + 16 + 64 + ~60 of those nodes).  A node that is never popped keeps
// its whole segment, split-line records included, exactly like the reference's un-split node.  The order of the records
// inside a segment is irrelevant (the per-node winner is the maximum response with ties broken by a key recomputed from
// the coordinates).  Group index inside a strip (layout order): leaf (q1,q2,q3,q4) = 85 q1 + 21 q2 + 5 q3 + q4, lines of
// (q1,q2,q3) = 85 q1 + 21 q2 + 5 q3 + 4, lines of (q1,q2) = 85 q1 + 21 q2 + 20, lines of (q1) = 85 q1 + 84, lines of the strip
// = 340.  Internal node index: (q1,q2,q3) = 16 q1 + 4 q2 + q3, (q1,q2) = 64 + 4 q1 + q2, (q1) = 80 + q1, strip = 84.  Totals in LDS
// (parked in the node table's tail): (q1) = q1, (q1,q2) = 4 + 4 q1 + q2, (q1,q2,q3) = 20 + 16 q1 + 4 q2 + q3; the 256 totals of the
// fourth level, (q1,q2,q3,q4) = 4 (16 q1 + 4 q2 + q3) + q4, live in global memory behind the level's bounce buffer (a batch of pops
// reads them in one round trip).
#define QT_PP_GROUPS 341
#define QT_PP_INTERNAL 85
#define QT_PP_TOTALS 84    // levels 1-3, LDS
#define QT_PP_TOTALS4 256  // level 4, global memory
#define QT_PP_MAX_STRIPS 4
#define QT_PP_HALF 704       // 11 rows of 64 groups: see the packed counters in tree_body
#define QT_CODE_SHIFT 22
#define QT_BEG_MASK 0x3FFFFFu  // n_beg = segment begin | pre-partition code << 22 (0: a node split the ordinary way)

// Integer thresholds of the pre-partition.  The x and the y splits of a node are independent, and all strips share the rows
// (0, reg_h): seven y thresholds (1 + 2 + 4 midpoints) are uniform, seven x thresholds per strip sit in LDS.
struct Thr {
  int lt, gt;  // v <= lt: first half, v >= gt: second half, else on the line
};
__host__ __device__ __forceinline__ Thr make_thr(double mid) {
  Thr t;
  t.lt = (int)ceil(mid) - 1;
  t.gt = (int)floor(mid) + 1;
  return t;
}
// the fifteen midpoints of four halvings of (lo, hi): index 0 | 1 + b1 | 3 + 2 b1 + b2 | 7 + 4 b1 + 2 b2 + b3   (same fp64 operations
// as the pop loop)
__host__ __device__ __forceinline__ void thr15(double lo, double hi, Thr* out) {
  const double m1 = (lo + hi) / 2;
  out[0] = make_thr(m1);
#pragma unroll
  for (int b1 = 0; b1 < 2; ++b1) {
    const double l1 = b1 ? m1 : lo, h1 = b1 ? hi : m1;
    const double m2 = (l1 + h1) / 2;
    out[1 + b1] = make_thr(m2);
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
      const double l2 = b2 ? m2 : l1, h2 = b2 ? h1 : m2;
      const double m3 = (l2 + h2) / 2;
      out[3 + 2 * b1 + b2] = make_thr(m3);
#pragma unroll
      for (int b3 = 0; b3 < 2; ++b3) {
        const double l3 = b3 ? m3 : l2, h3 = b3 ? h2 : m3;
        out[7 + 4 * b1 + 2 * b2 + b3] = make_thr((l3 + h3) / 2);
      }
    }
  }
}
__host__ __device__ __forceinline__ int half_of(int v, const Thr& t) { return v <= t.lt ? 0 : (v >= t.gt ? 1 : -1); }

struct PpGeom {
  int ns, y_max;
  int s_lo[QT_PP_MAX_STRIPS], s_hi[QT_PP_MAX_STRIPS];  // integer form of the strict strip membership
};

// The x and the y halves of a record's group are functions of ONE coordinate each, so they are tabulated once per tree (a few
// hundred entries) and a record costs two independent LDS reads plus a dozen integer operations instead of ~90 with four
// dependent threshold reads.  Code (16 bits): bit 15 = inside, bits 12-13 = strip (x table only), bits 4-6 = number of levels the
// coordinate passes without sitting on a split line (0..4), bits 0-3 = the halves taken (level 1 in bit 3).
__host__ __device__ __forceinline__ uint32_t pp_axis_code(int v, const Thr* t /*[15]*/) {
  const int b1 = half_of(v, t[0]);
  const int c1 = b1 > 0 ? b1 : 0;
  const int b2 = half_of(v, t[1 + c1]);
  const int c2 = b2 > 0 ? b2 : 0;
  const int b3 = half_of(v, t[3 + 2 * c1 + c2]);
  const int c3 = b3 > 0 ? b3 : 0;
  const int b4 = half_of(v, t[7 + 4 * c1 + 2 * c2 + c3]);
  const int nv = b1 < 0 ? 0 : (b2 < 0 ? 1 : (b3 < 0 ? 2 : (b4 < 0 ? 3 : 4)));
  return (uint32_t)(nv << 4) | (uint32_t)(c1 << 3) | (uint32_t)(c2 << 2) | (uint32_t)(c3 << 1) | (uint32_t)(b4 > 0 ? b4 : 0);
}
__device__ __forceinline__ int pp_group_codes(uint32_t cx, uint32_t cy) {
  const int nv = (int)min((cx >> 4) & 7u, (cy >> 4) & 7u);
  const int q1 = (int)(((cy >> 3) & 1u) * 2u + ((cx >> 3) & 1u));  // rows outer, cols inner (ORBExtractor.cc:60-72)
  const int q2 = (int)(((cy >> 2) & 1u) * 2u + ((cx >> 2) & 1u));
  const int q3 = (int)(((cy >> 1) & 1u) * 2u + ((cx >> 1) & 1u));
  const int q4 = (int)((cy & 1u) * 2u + (cx & 1u));
  const int l3 = nv >= 3 ? q3 * 5 + (nv >= 4 ? q4 : 4) : 20;
  const int l2 = nv >= 2 ? q2 * 21 + l3 : 84;
  const int l1 = nv >= 1 ? q1 * 85 + l2 : 340;
  return ((cx & cy) & 0x8000u) ? (int)((cx >> 12) & 3u) * QT_PP_GROUPS + l1 : -1;
}

// Bitonic sort of ROWS x 64 (key, payload) pairs held in registers, element index = r * 64 + lane, ascending by key (the keys are distinct
// or padding).  Strides >= 64 pair two registers of a lane; strides 1 and 2 exchange through DPP quad permutes (vector ALU, no LDS round
// trip: 17 of the 39 cross-lane stages of 512 keys); the others through ds_bpermute -- ALL rows' exchanges of a stage are requested
// before the first is consumed (r5: with one wait per row a level-0 tree spent 53 k cycles here, eight dependent LDS round trips per
// stage).  ROWS is a template parameter so that no stage touches a row of padding.
template <int ROWS>
__device__ __forceinline__ void bitonic_rows(uint32_t (&key)[8], uint32_t (&pay)[8], int lane) {
  constexpr int CAP = ROWS * 64;
#pragma unroll 1
  for (int k = 2; k <= CAP; k <<= 1) {
#pragma unroll 1
    for (int st = k >> 1; st > 0; st >>= 1) {
      if (st >= 64) {
        const int rs = st >> 6;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
#pragma unroll
          for (int q = 1; q < ROWS; q <<= 1) {
            if (rs == q && (r & q) == 0) {
              const bool up = ((r * 64) & k) == 0;
              const uint32_t a = key[r], b2 = key[r | q], pa = pay[r], pb = pay[r | q];
              const bool sw = (a > b2) == up;
              key[r] = sw ? b2 : a;
              key[r | q] = sw ? a : b2;
              pay[r] = sw ? pb : pa;
              pay[r | q] = sw ? pa : pb;
            }
          }
        }
      } else {
        uint32_t ok[ROWS], op[ROWS];
        if (st == 1) {
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            ok[r] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key[r], 0xB1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]
            op[r] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pay[r], 0xB1, 0xf, 0xf, false);
          }
        } else if (st == 2) {
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            ok[r] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key[r], 0x4E, 0xf, 0xf, false);  // quad_perm [2, 3, 0, 1]
            op[r] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pay[r], 0x4E, 0xf, 0xf, false);
          }
        } else {
          const int src = (lane ^ st) << 2;
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            ok[r] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)key[r]);
            op[r] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)pay[r]);
          }
        }
        const bool lower = (lane & st) == 0;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const bool up = (((r * 64) + lane) & k) == 0;
          const bool other = (ok[r] < key[r]) == (lower == up);  // take the partner's pair: it holds the minimum and this lane keeps minima, or the reverse
          pay[r] = other ? op[r] : pay[r];
          key[r] = other ? ok[r] : key[r];
        }
      }
    }
  }
}

// Everything after the strip counts: scatter into the strip segments, best-first expansion, selection, ordering.
// IN_LDS selects the address space of the record home H at compile time (ds_* instead of flat_* accesses: a flat access
// costs several hundred cycles even when it lands in LDS, and a pop is a chain of dependent accesses).
// NW = waves per tree.  1: the batch path (one wave per tree: a 1024-image launch has more trees than the chip has wave slots to spare).
// 4: small launches (a frame or two: 16 trees on 256 CUs) -- the best-first expansion itself stays on wave 0, but the phases around it
// are data-parallel: the pre-partition (two passes over the candidates, ~45 us of a level-0 tree's 123 us on one wave) and the
// per-node winner scan (~40 us) are spread over all four waves.  WSYNC: ordering inside the part only wave 0 runs (a wave's own LDS
// traffic executes in order; its global traffic needs the wait) -- a workgroup barrier there when the wave is the whole workgroup.
template <int NW>
__device__ __forceinline__ void qt_wsync() {
  if (NW == 1) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// NODES_LDS = false: the node table and the sort buffer live in GLOBAL memory (a level whose quota one CU's LDS cannot hold: the
// reference has no limit on nFeatures, ORBExtractor.cc:291-301).  The records are global then too (IN_LDS = false), so every step
// already ends in the wait that orders a wave's global stores before its next loads; the pre-partition (whose tables borrow the LDS
// node arrays) is off and the strips are counted with ballots.  Slow (a pop is a chain of global round trips) but exact.
#ifdef QT_STAMPS  // diagnostic build only: timeline of one level-0 tree (device printf), see tools/exp/qt_stamps.sh
__device__ int g_qt_stamp_count;
#define QTS_DECL long long qts_t0 = __builtin_amdgcn_s_memtime(), qts_last = qts_t0; int qts_n = 0; long long qts_d[96]; int qts_i[96]; int qts_single = 0, qts_single_n = 0, qts_fail = 0;
#define QTS(info) { const long long now_ = __builtin_amdgcn_s_memtime(); if (qts_n < 96) { qts_d[qts_n] = now_ - qts_last; qts_i[qts_n] = (info); ++qts_n; } qts_last = now_; }
#define QTS_PRINT                                                                                                                   \
  if (w0 && lane == 0 && need > 400) {                                                                                              \
    const int c_ = atomicAdd(&g_qt_stamp_count, 1);                                                                                 \
    if (c_ == 5) {                                                                                                                  \
      printf("QT tree N %d need %d n_act %d single pops %d (records %d) batched attempts without a commit %d\n", N, need, n_act, qts_single, qts_single_n, qts_fail);                                                                   \
      for (int k_ = 0; k_ < qts_n; ++k_)                                                                                            \
        printf("  %2d: %7lld cycles  info %d (pops %d nmax %d pp %d)\n", k_, qts_d[k_], qts_i[k_], qts_i[k_] >> 16, (qts_i[k_] >> 8) & 255, qts_i[k_] & 255); \
    }                                                                                                                               \
  }
#else
#define QTS_DECL
#define QTS(info)
#define QTS_PRINT
#endif
template <bool IN_LDS, int NW, bool NODES_LDS = true>
__device__ __forceinline__ void tree_body(const LevelDev& L, const uint32_t* __restrict__ A, int N, uint32_t* H, uint32_t* __restrict__ T,
                                          unsigned long long* n_key, uint2* n_bp /* = n_key + node_cap: x = begin | code, y = path */,
                                          unsigned long long* sortbuf, unsigned long long* bkey, uint32_t* bj,
                                          uint32_t* shared_ints, int batch_on, int node_cap, int need,
                                          int sort_cap, uint32_t* __restrict__ out_sel, int32_t* __restrict__ sel_count_out, int lane, int wv,
                                          const uint16_t* __restrict__ qt_tabs, uint16_t* tot4_lds = nullptr, const uint32_t* shP = nullptr) {
  constexpr int NT = 64 * NW;
  // record i of the level's candidate set.  A frame or two: k_fast leaves the set as ORBFE_FAST_SHARDS lists in equal parts of the level's
  // region (k_fast.hip); shP = the exclusive prefix of their sizes in LDS, [ORBFE_FAST_SHARDS + 1] (shards without records share their
  // successor's prefix and come before it: the LAST shard whose prefix is <= i holds record i)
  auto A_at = [&](int i) __attribute__((always_inline)) -> uint32_t {
    if (NW == 1 || !shP) return A[i];
    int sh = 0;
#pragma unroll
    for (int step = ORBFE_FAST_SHARDS / 2; step; step >>= 1)
      if ((int)shP[sh + step] <= i) sh += step;
    return A[(size_t)sh * L.shard_cap + (size_t)(i - (int)shP[sh])];
  };
  const int tid = wv * 64 + lane;
  const bool w0 = wv == 0;
  QTS_DECL
  int n_act = 0;
  uint32_t next_seq = 0;
  // ---- first pop: the root, whose children are the initSplit strips (ORBExtractor.cc:81-96, 147-170) ----
  const int ns = L.n_ini;
  const int n_tot = (ns * QT_PP_TOTALS + 3) & ~3;
  // The pre-partition borrows the (still empty) node table, node_cap * 16 bytes from n_key on: group counters, then cursors, as 16-bit
  // halves of 32-bit words (N <= 65535; group g < QT_PP_HALF in the low half of word g, the others in the high half of word g - QT_PP_HALF,
  // so that a lane of the cursor scan owns both halves of the words it writes) | the coordinate -> code tables (uint16) | ... | the
  // totals of levels 1-3 in the table's tail, which stay until the table grows into them.
  uint8_t* const tab_base = (uint8_t*)n_key;
  uint16_t* tot = (uint16_t*)(tab_base + (size_t)node_cap * 16) - n_tot;
  const int pp_limit = node_cap - (n_tot * 2 + 7) / 8;       // first node slot whose n_bp entry overlaps the parked totals
  const int n4_off = (N + 3) & ~3;                           // the fourth level's totals: global, behind the records of the bounce buffer
  // (tot4_lds: launches of a frame or two have the LDS to spare -- three of a level-0 tree's seven batched steps read these totals, and
  //  from global memory each read was a ~3 k-cycle round trip on the critical path)
  uint16_t* tot4 = tot4_lds ? tot4_lds : (uint16_t*)(T + n4_off);
  const int tab_w = (int)ceil(L.strips[ns]) + 1, tab_h = (int)ceil((double)L.reg_h) + 1;
  const int tab_w2 = (tab_w + 1) & ~1;
  const int ng = ns * QT_PP_GROUPS;
  const int cur_words = min(ng, QT_PP_HALF);
  const int tab_words = (tab_w2 + tab_h + 1) >> 1;
  bool pp_ok = NODES_LDS && N > 0 && N <= 65535 && ns <= QT_PP_MAX_STRIPS && tab_w <= 4096 && tab_h <= 4096 &&
               (cur_words + tab_words) * 4 + n_tot * 2 <= node_cap * 16 &&
               (tot4_lds != nullptr || n4_off + ns * (QT_PP_TOTALS4 / 2) <= (int)L.cand_cap);
  if (pp_ok) {
    uint32_t* cur = (uint32_t*)tab_base;  // packed group sizes, then group cursors
    uint16_t* xtab = (uint16_t*)(cur + cur_words);
    uint16_t* ytab = xtab + tab_w2;
    auto cur_w = [&](int g) -> int { return g < QT_PP_HALF ? g : g - QT_PP_HALF; };
    auto cur_sh = [&](int g) -> int { return g < QT_PP_HALF ? 0 : 16; };
    auto cur_get = [&](int g) -> uint32_t { return (cur[cur_w(g)] >> cur_sh(g)) & 0xFFFFu; };
    // PU records per lane per trip: sixteen with helper waves (a frame or two: registers are free), eight in the one-wave batch kernel,
    // which is compiled for four waves per SIMD (128 VGPRs).  With helper waves a trip takes only as many records per lane as it takes
    // to cover the level ONCE with all waves (pu_eff, in uniform groups of four slots): 5087 candidates on eight waves are ten records a
    // lane on every wave -- with whole trips of sixteen, five waves took one trip each and three had nothing to do.
    constexpr int PU = NW == 1 ? 8 : 16;
    // Sharded candidate lists (a frame or two, k_fast.hip): the set's order is free, so wave w simply takes shards [w SPW, (w + 1) SPW) back to
    // back -- equal numbers of cells each, so the waves are balanced -- and no record needs the search of A_at: local index t -> shard k by
    // SPW - 1 compares against wave-uniform prefixes.  (With the search in the loads the passes were 3.5 us longer per pair.)
    constexpr int SPW = (NW > 1 && ORBFE_FAST_SHARDS / NW > 0) ? ORBFE_FAST_SHARDS / NW : 1;
    static_assert(NW == 1 || NW > ORBFE_FAST_SHARDS || SPW * NW == ORBFE_FAST_SHARDS, "shards per wave");
    const bool shw = NW > 1 && NW <= ORBFE_FAST_SHARDS && shP != nullptr;  // (uniform)
    int lp[SPW + 1];  // this wave's shards: exclusive prefix of their sizes
#pragma unroll
    for (int k = 0; k <= SPW; ++k) lp[k] = shw ? (int)(shP[wv * SPW + k] - shP[wv * SPW]) : 0;
    const int n_loc = lp[SPW];
    const int pu_eff = NW == 1 ? PU : min(PU, max(1, shw ? (n_loc + 63) / 64 : (N + NT - 1) / NT));
    const int pch = 64 * pu_eff;
    const int t_beg = shw ? 0 : wv * pch, t_end = shw ? n_loc : N, t_step = shw ? pch : NW * pch;  // this wave's trips: records [b0, b0 + pch)
#define QT_SLOT_ON(u) (NW == 1 || ((u) & ~3) < pu_eff)  // (wave-uniform: a group of four slots is worked on or skipped as a whole)
#define QT_REC_AT(b0, u) ((b0) + (u) * 64 + lane)        // slot-major: a wave instruction's 64 records are consecutive (coalesced loads)
    // (measured and dropped, eight waves: LANE-major slots -- the candidates arrive cell by cell, 64 consecutive records fall into four to six
    //  groups of the pre-partition and the lanes' LDS atomics serialise a dozen deep -- pass 2 5.97 -> 5.1 k cycles, pass 1 8.8 -> 10.2 k: its
    //  strided loads)
    auto rec_fetch = [&](int i) __attribute__((always_inline)) -> uint32_t {  // record i of this wave's range (clamped by the caller)
      if (!shw) return A[i];
      int k = 0;
#pragma unroll
      for (int q = 1; q < SPW; ++q) k += i >= lp[q] ? 1 : 0;
      int base = lp[0];
#pragma unroll
      for (int q = 1; q < SPW; ++q) base = k >= q ? lp[q] : base;
      return A[(size_t)(wv * SPW + k) * L.shard_cap + (size_t)(i - base)];
    };
    // pass 1: group sizes.  The next trip's records are requested before this trip's are classified: a lone wave sees every global round
    // trip, so the loads of trip k+1 fly under the work of trip k.
    auto load16 = [&](int b0, uint32_t* r) {
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int i = QT_REC_AT(b0, u);
        r[u] = QT_SLOT_ON(u) ? rec_fetch(max(min(i, t_end - 1), 0)) : 0u;  // (clamped: unconditional loads, validity is checked when the record is used)
      }
    };
    auto slot_off = [&](int b0, int u) -> bool { return u >= pu_eff || QT_REC_AT(b0, u) >= t_end; };  // not a record of this trip
    // (r6) the first trip's records are requested HERE, before the tables are copied: a tree's first global round trip -- the candidates
    // k_fast has just written -- flies under the copy and its barrier instead of opening pass 1
    uint32_t nxt0[PU];
    if (NW > 1) load16(t_beg, nxt0);  // (the one-wave batch kernel keeps its request in pass 1: eight more live registers across the copy buy it nothing)
    for (int g = tid; g < cur_words; g += NT) cur[g] = 0;
    // The coordinate -> code tables depend on the LEVEL's geometry alone (strip bounds, region height): the host builds them once per
    // context (quadtree_build_tables, the same fp64 operations) and a tree only copies its level's ~3 KB into LDS.  Built here, per tree,
    // they were 24 k of a level-0 tree's 224 k cycles with four waves (stamps build) -- and every tree wave of a batch paid them alone.
    {
      const uint32_t* src = (const uint32_t*)(qt_tabs + L.qt_tab_off);  // [tab_w2 | tab_h] uint16, 4-byte aligned
      uint32_t* dst = (uint32_t*)xtab;
      for (int i0 = 0; i0 < tab_words; i0 += 4 * NT) {  // four independent loads per trip (the loop was a chain of single round trips)
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[min(i0 + u * NT + tid, tab_words - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + u * NT + tid < tab_words) dst[i0 + u * NT + tid] = v[u];
      }
    }
    __syncthreads();
    QTS(-10)  // tables
    // a trip's records -> groups: ALL table reads of the trip first (two per record), then the arithmetic -- the compiler keeps the order
    // it is given here, and read-by-read each record waited for its own two LDS round trips
    auto groups_of = [&](const uint32_t* r, int* g, int n) {
      uint32_t cx[16], cy[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (u < n && QT_SLOT_ON(u)) {  // (candidates lie inside the region; the clamp only guards the table)
          cx[u] = xtab[min(ORBFE_REC_X(r[u]), (uint32_t)tab_w - 1u)];
          cy[u] = ytab[min(ORBFE_REC_Y(r[u]), (uint32_t)tab_h - 1u)];
        }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (u < n) g[u] = QT_SLOT_ON(u) ? pp_group_codes(cx[u], cy[u]) : -1;
    };
    // (the records of the FIRST trip and their groups stay in registers for the scatter pass: up to a trip's records per wave need no
    //  second load -- an exposed global round trip -- and no second classification)
    uint32_t rec0[PU];
    int g0[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) rec0[u] = 0u, g0[u] = -1;
    {
      uint32_t nxt[PU];
      if (NW > 1) {
#pragma unroll
        for (int u = 0; u < PU; ++u) nxt[u] = nxt0[u];
      } else {
        load16(t_beg, nxt);
      }
      for (int b0 = t_beg; b0 < t_end; b0 += t_step) {
        uint32_t rec[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) rec[u] = nxt[u];
        if (b0 + t_step < t_end) load16(b0 + t_step, nxt);
        int g[PU];  // (all groups first, then the atomics: the table reads must not queue behind the atomics they may alias)
        // (classified UNCONDITIONALLY -- the loads are clamped, so every register holds a real record -- and masked afterwards: written as
        //  `i < N ? group_of(rec) : -1` each record became a branch of its own whose two table reads were waited for on the spot, eight
        //  dependent LDS round trips per trip instead of sixteen reads in flight)
#ifdef QT_STAMPS
        { uint32_t x_ = 0;
#pragma unroll
          for (int u = 0; u < PU; ++u) x_ |= rec[u];
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (x_ == 0x12345u) g[0] = 0; QTS(-40) }
#endif
        groups_of(rec, g, PU);
#ifdef QT_STAMPS
        { int x_ = 0;
#pragma unroll
          for (int u = 0; u < PU; ++u) x_ += g[u];
          asm volatile("" :: "v"(x_)); QTS(-41) }
#endif
#pragma unroll
        for (int u = 0; u < PU; ++u) g[u] |= -(int)slot_off(b0, u);
        if (b0 == t_beg) {  // wave-uniform
#pragma unroll
          for (int u = 0; u < PU; ++u) rec0[u] = rec[u], g0[u] = g[u];
        }
#pragma unroll
        for (int u = 0; u < PU; ++u)
          if (g[u] >= 0) atomicAdd(&cur[cur_w(g[u])], 1u << cur_sh(g[u]));  // LDS atomic; lanes of one group serialise, a chunk spans a handful of groups
#ifdef QT_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); QTS(-42)
#endif
      }
    }
    __syncthreads();
    QTS(-11)  // pass 1
    // totals of the 84 + 256 nodes below every strip, then sizes -> cursors (exclusive prefix in layout order).  Bottom-up: a node's total
    // is its four children's totals plus its own split-line group -- five reads per entry on each of the three levels.  (Until r5 every
    // entry summed its groups directly: 85 dependent LDS reads for a child of the strip, 39 k of a level-0 tree's 337 k cycles.)
    for (int t = tid; t < ns * 64; t += NT) {  // (q1, q2, q3): four leaves + the lines group
      const int st = t >> 6, m = t & 63;
      const int b = st * QT_PP_GROUPS + (m >> 4) * 85 + ((m >> 2) & 3) * 21 + (m & 3) * 5;
      tot[st * QT_PP_TOTALS + 20 + m] = (uint16_t)(cur_get(b) + cur_get(b + 1) + cur_get(b + 2) + cur_get(b + 3) + cur_get(b + 4));
    }
    __syncthreads();
    for (int t = tid; t < ns * 16; t += NT) {  // (q1, q2)
      const int st = t >> 4, k = t & 15;
      const uint16_t* t3 = tot + st * QT_PP_TOTALS + 20 + 4 * k;
      tot[st * QT_PP_TOTALS + 4 + k] = (uint16_t)((uint32_t)t3[0] + t3[1] + t3[2] + t3[3] + cur_get(st * QT_PP_GROUPS + (k >> 2) * 85 + (k & 3) * 21 + 20));
    }
    __syncthreads();
    for (int t = tid; t < ns * 4; t += NT) {  // (q1)
      const int st = t >> 2, q1 = t & 3;
      const uint16_t* t2 = tot + st * QT_PP_TOTALS + 4 + 4 * q1;
      tot[st * QT_PP_TOTALS + q1] = (uint16_t)((uint32_t)t2[0] + t2[1] + t2[2] + t2[3] + cur_get(st * QT_PP_GROUPS + q1 * 85 + 84));
    }
    for (int t = tid; t < ns * QT_PP_TOTALS4; t += NT) {
      const int st = t / QT_PP_TOTALS4, m = t - QT_PP_TOTALS4 * st;  // m = 64 q1 + 16 q2 + 4 q3 + q4
      tot4[t] = (uint16_t)cur_get(st * QT_PP_GROUPS + (m >> 6) * 85 + ((m >> 4) & 3) * 21 + ((m >> 2) & 3) * 5 + (m & 3));
    }
    __syncthreads();
    QTS(-12)  // totals
    int carry = 0, strip_base = 0, strip_cnt = 0;  // lane st keeps the segment of strip st
    if (NW > 1) {
      // (r6) several waves per tree: the rows of 64 group sizes are dealt to the waves (row r to wave r mod NW), each scans its rows, the row
      // totals meet in LDS (the head list of the batched pops is idle until the expansion) and every wave adds up the rows before its own.
      // One wave scanning all 22 rows was 6.5 k of a level-0 tree's 93 k cycles.  Rows r and r + ROWS / 2 share their counter WORDS and may
      // belong to different waves: the cursors go back as 16-bit stores.
      constexpr int ROWS = (QT_PP_MAX_STRIPS * QT_PP_GROUPS + 63) / 64, RPW = (ROWS + NW - 1) / NW;
      uint32_t* rowtot = (uint32_t*)bkey;  // [ROWS] row totals | [QT_PP_MAX_STRIPS] strip begins | [QT_PP_MAX_STRIPS] strip ends
      int vals[RPW], incs[RPW];
#pragma unroll
      for (int k = 0; k < RPW; ++k) {
        const int r = wv + k * NW, g = r * 64 + lane;
        vals[k] = (r < ROWS && r * 64 < ng && g < ng) ? (int)cur_get(g) : 0;
        incs[k] = wave_incl_scan(vals[k], lane);
        if (r < ROWS && lane == 63) rowtot[r] = (uint32_t)incs[k];
      }
      __syncthreads();
      const int rt = lane < ROWS ? (int)rowtot[lane] : 0;
      const int rt_excl = wave_incl_scan(rt, lane) - rt;  // lane r: the groups in the rows before row r
#pragma unroll
      for (int k = 0; k < RPW; ++k) {
        const int r = wv + k * NW, g0 = r * 64;
        if (r < ROWS && g0 < ng) {  // wave-uniform
          const int excl = __builtin_amdgcn_readlane(rt_excl, r) + incs[k] - vals[k];
          for (int st = 0; st < ns; ++st) {  // strip st starts at group 341 st and ends where strip st + 1 starts
            const int first = st * QT_PP_GROUPS, last = first + QT_PP_GROUPS - 1;
            if (first >= g0 && first < g0 + 64) {
              const int bb = __builtin_amdgcn_readlane(excl, first - g0);
              if (lane == 0) rowtot[ROWS + st] = (uint32_t)bb;
            }
            if (last >= g0 && last < g0 + 64) {
              const int ee = __builtin_amdgcn_readlane(excl + vals[k], last - g0);
              if (lane == 0) rowtot[ROWS + QT_PP_MAX_STRIPS + st] = (uint32_t)ee;
            }
          }
          const int g = g0 + lane;
          if (g < ng) ((uint16_t*)cur)[2 * cur_w(g) + (cur_sh(g) ? 1 : 0)] = (uint16_t)excl;
        }
      }
      __syncthreads();
      if (lane < ns) strip_base = (int)rowtot[ROWS + lane], strip_cnt = (int)rowtot[ROWS + QT_PP_MAX_STRIPS + lane];
    } else
    if (w0) {
      // one wave: an exclusive scan over the <= 22 rows of 64 group sizes.  All rows are read first, scanned in registers (the row
      // scans are independent: only the carry is a chain, and it is scalar) and written back at the end -- row by row with a
      // synchronisation between a row's read and its write this was 19 k of a level-0 tree's cycles.
      constexpr int ROWS = (QT_PP_MAX_STRIPS * QT_PP_GROUPS + 63) / 64;
      int vals[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const int g = r * 64 + lane;
        vals[r] = (r * 64 < ng && g < ng) ? (int)cur_get(g) : 0;
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const int g0 = r * 64;
        if (g0 < ng) {  // wave-uniform
          const int v = vals[r];
          const int incl = wave_incl_scan(v, lane);
          const int excl = carry + incl - v;
          for (int st = 0; st < ns; ++st) {  // strip st starts at group 341 st and ends where strip st + 1 starts
            const int first = st * QT_PP_GROUPS, last = first + QT_PP_GROUPS - 1;
            if (first >= g0 && first < g0 + 64) {
              const int bb = __builtin_amdgcn_readlane(excl, first - g0);
              if (lane == st) strip_base = bb;
            }
            if (last >= g0 && last < g0 + 64) {
              const int ee = __builtin_amdgcn_readlane(excl + v, last - g0);
              if (lane == st) strip_cnt = ee;
            }
          }
          carry += __builtin_amdgcn_readlane(incl, 63);
          vals[r] = excl;
        }
      }
      qt_wsync<NW>();
      static_assert(ROWS == 2 * (QT_PP_HALF / 64), "a lane owns both halves of a counter word");
#pragma unroll
      for (int r = 0; r < ROWS / 2; ++r) {  // word r * 64 + lane = groups (r, lane) and (r + ROWS / 2, lane)
        const int g = r * 64 + lane;
        if (g < cur_words) cur[g] = (uint32_t)vals[r] | ((uint32_t)vals[r + ROWS / 2] << 16);  // (rows past the last group hold 0)
      }
    }
    strip_cnt -= strip_base;
    __syncthreads();
    QTS(-13)  // scan
    // pass 2: scatter
    {
      uint32_t nxt[PU];
      if (t_beg + t_step < t_end) load16(t_beg + t_step, nxt);  // (the first trip's records are still in registers)
      // (r5, measured and dropped: the groups of pass 1 kept in the bounce buffer for this pass instead of classifying every record a
      //  second time -- ~35 vector instructions per record saved here, one store and one load added: pass 1 45 k -> 59 k cycles, this
      //  pass 53 k -> 49 k on a level-0 tree: it is not the classification that bounds this pass)
      for (int b0 = t_beg; b0 < t_end; b0 += t_step) {
        uint32_t rec[PU];
        int g[PU];
        if (b0 == t_beg) {  // wave-uniform
#pragma unroll
          for (int u = 0; u < PU; ++u) rec[u] = rec0[u], g[u] = g0[u];
        } else {
#pragma unroll
          for (int u = 0; u < PU; ++u) rec[u] = nxt[u];
          if (b0 + t_step < t_end) load16(b0 + t_step, nxt);
          groups_of(rec, g, PU);
#pragma unroll
          for (int u = 0; u < PU; ++u) g[u] |= -(int)slot_off(b0, u);
        }
        uint32_t pos[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) pos[u] = (g[u] >= 0) ? ((atomicAdd(&cur[cur_w(g[u])], 1u << cur_sh(g[u])) >> cur_sh(g[u])) & 0xFFFFu) : 0u;
#pragma unroll
        for (int u = 0; u < PU; ++u)
          if (g[u] >= 0) H[pos[u]] = rec[u];  // the order inside a group is irrelevant (see above)
      }
    }
    __syncthreads();
    for (int st = 0; w0 && st < ns; ++st) {
      const int c = __builtin_amdgcn_readlane(strip_cnt, st);
      const int off = __builtin_amdgcn_readlane(strip_base, st);
      if (c > 0) {
        if (lane == 0) {
          n_key[n_act] = ((unsigned long long)c << 32) | (unsigned long long)(0xFFFFFFFFu - next_seq);
          n_bp[n_act] = make_uint2((uint32_t)off | ((uint32_t)(1 + st * QT_PP_INTERNAL + 84) << QT_CODE_SHIFT), QT_PATH(st, 0, 0, 0));
        }
        ++n_act;
        ++next_seq;
      }
    }
#undef QT_SLOT_ON
#undef QT_REC_AT
  } else if (w0) {
    // Pass 1 counts the records of each strip (lane s keeps strip s's counter), pass 2 scatters them into the
    // strip segments of H.  Four records per lane are in flight per step to hide the global-load latency.
    int my_cnt = 0;
    for (int b0 = 0; b0 < N; b0 += 256) {
      uint32_t rec[4];
      int sid[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = b0 + u * 64 + lane;
        rec[u] = (i < N) ? A_at(i) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = b0 + u * 64 + lane;
        sid[u] = (i < N) ? strip_of(rec[u], L) : -1;
      }
      for (int st = 0; st < L.n_ini; ++st) {
        int c = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) c += __popcll(__ballot(sid[u] == st));
        if (lane == st) my_cnt += c;
      }
    }
    const int incl = wave_incl_scan(my_cnt, lane);  // lanes >= n_ini hold 0
    int run = incl - my_cnt;                         // lane s: write cursor of strip s
    for (int st = 0; st < L.n_ini; ++st) {
      const int c = __builtin_amdgcn_readlane(my_cnt, st);
      const int off = __builtin_amdgcn_readlane(run, st);
      if (c > 0) {
        if (lane == 0) {
          n_key[n_act] = ((unsigned long long)c << 32) | (unsigned long long)(0xFFFFFFFFu - next_seq);
          n_bp[n_act] = make_uint2((uint32_t)off, QT_PATH(st, 0, 0, 0));
        }
        ++n_act;
        ++next_seq;
      }
    }
    for (int b0 = 0; b0 < N; b0 += 256) {
      uint32_t rec[4];
      int sid[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = b0 + u * 64 + lane;
        rec[u] = (i < N) ? A_at(i) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = b0 + u * 64 + lane;
        sid[u] = (i < N) ? strip_of(rec[u], L) : -1;
      }
      for (int st = 0; st < L.n_ini; ++st) {
        int base = __builtin_amdgcn_readlane(run, st);
        int c = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned long long m = __ballot(sid[u] == st);
          if (sid[u] == st) H[base + c + __popcll(m & ((1ull << lane) - 1ull))] = rec[u];
          c += __popcll(m);
        }
        if (lane == st) run += c;
      }
    }
  }
  __syncthreads();

  // ---- best-first expansion ----
  // A lone wave pays tens of cycles for every taken branch (nothing hides the instruction fetch), so the step is
  // written as straight-line predicated code: fixed 8-way unrolled arg-max, uniform-address LDS traffic executed by
  // all lanes instead of "if (lane == 0)" blocks, the four children written by lanes 0..3 at once.
  QTS(-1)  // build (pre-partition) done
  // ---- NW > 1: the ORDINARY members of a batched step split by the whole workgroup (r6) ----------------------------------------------
  // The expansion lives on wave 0, one node per lane -- and an ordinary member (a node below the pre-partition's four levels) was split by
  // its lane alone, record after record, twice (count, then scatter): a step of 19 such nodes of up to 31 records was 17 k of a level-0
  // tree's 106 k cycles with nineteen lanes of one wave at work and seven waves asleep at the barrier behind the loop.  Now wave 0 posts
  // the members (segment, size, prefix of the sizes, integer split) in the idle sort area and every thread of the workgroup takes a
  // record or two: member by binary search over the prefix, quadrant, an LDS atomic on the member's quadrant counter (its return value is
  // the record's place inside the quadrant); wave 0 reads the counts, decides how many of the pops are valid exactly as before, posts the
  // quadrants' offsets, and the threads store their records.  Four workgroup barriers per step, executed by every wave in the same order:
  // the helper waves loop on coop_count / coop_scatter until wave 0 posts the empty job (the barrier that used to end the expansion).
  uint32_t* const coop = (uint32_t*)sortbuf;  // [64][4] beg | n << 22, prefix, x_lt | x_gt << 16, y_lt | y_gt << 16; [64][4] counters, then offsets
  constexpr int KR = (64 * 64 + NT - 1) / NT;   // records per thread: a head of 64 members of up to 64 records always fits
  uint32_t crec[KR], cslot[KR];
  int cmem[KR], cquad[KR];
  auto coop_count = [&]() __attribute__((always_inline)) -> bool {
    __syncthreads();  // (1) the job is posted
    const int R = (int)shared_ints[1];
    if (R == 0) return false;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int g = tid + k * NT;
      cmem[k] = -1, cquad[k] = 0, crec[k] = 0u, cslot[k] = 0u;
      if (g < R) {
        int lo = 0;  // the last member whose prefix is <= g (members without records share their successor's prefix and come before it)
#pragma unroll
        for (int step = 32; step; step >>= 1)
          if ((int)coop[(lo + step) * 4 + 1] <= g) lo += step;
        const uint32_t d0 = coop[lo * 4], d1 = coop[lo * 4 + 1], d2 = coop[lo * 4 + 2], d3 = coop[lo * 4 + 3];
        const uint32_t rec = H[(int)(d0 & QT_BEG_MASK) + (g - (int)d1)];
        SplitInt s;
        s.x_lt = (int)(int16_t)(d2 & 0xFFFFu), s.x_gt = (int)(d2 >> 16), s.y_lt = (int)(int16_t)(d3 & 0xFFFFu), s.y_gt = (int)(d3 >> 16);
        const int q = quadrant_of(rec, s);
        crec[k] = rec, cquad[k] = q;
        if (q >= 0) {
          cmem[k] = lo;
          cslot[k] = atomicAdd(&coop[256 + lo * 4 + q], 1u);
        }
      }
    }
    __syncthreads();  // (2) the counts are complete
    return true;
  };
  auto coop_scatter = [&]() __attribute__((always_inline)) {
    __syncthreads();  // (3) offsets and the number of valid pops are posted
    const int vv = (int)shared_ints[2];
#pragma unroll
    for (int k = 0; k < KR; ++k)
      if (cmem[k] >= 0 && cmem[k] < vv) H[coop[256 + cmem[k] * 4 + cquad[k]] + cslot[k]] = crec[k];  // (every record was read before barrier 2)
    __syncthreads();  // (4)
  };
  if (NW > 1 && !w0) {
    while (coop_count()) coop_scatter();
  }
  const long long max_iter = 80ll * (long long)N + 1024;  // each point survives < ~64 halvings (fp64); hard stop for safety
  long long iter = 0;
  while (n_act < need && n_act > 0 && iter < max_iter) {
    ++iter;
    // ---- batched pops ------------------------------------------------------------------------------------------------------
    // Pops come in non-increasing count order (a child never outnumbers its parent), so the next pops are simply the head of the
    // table sorted by key -- as long as no child created on the way outnumbers a later member of that head (it would have to be
    // popped first).  One lane takes one node: the up-to-64 nodes with the largest keys are sorted into pop order, every lane
    // splits its node (four precomputed totals, or the node's <= 64 records held in registers), and a prefix scan finds how many
    // of these pops the reference would really make in this order: up to the first member outnumbered by an earlier member's
    // child, up to the pop that reaches the quota, and never across a node that vanishes (the table would need compacting) or
    // holds more than 64 records; those take the one-node path below.  Sequence numbers and table slots are handed out in pop
    // order, so the table afterwards is exactly what the same pops made one at a time leave behind.  (A level needs 40-150
    // pops; measured on the synthetic frames the valid prefix is almost always the whole head: ~10 batches per tree.)
    if (batch_on && n_act <= 512) {
      unsigned long long lk[8];
      uint32_t lmax = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        lk[u] = 0ull;
        if (u * 64 < n_act) {
          const int jj = u * 64 + lane;
          lk[u] = (jj < n_act) ? n_key[jj] : 0ull;
        }
        lmax = max(lmax, (uint32_t)(lk[u] >> 32));
      }
      const uint32_t mc = wave_max_u32(lmax);
      if (mc <= 1u) {  // only one-record nodes are left and the quota is not reached: it never will be (see QT_PATH) -- the reference
        n_act = 0;     // halves every point until it sits on a split line and returns the empty set (quirk Q3)
        break;
      }
      // (r6) several waves per tree: the head from a QUARTER of the largest count, not half -- a level-0 tree of the bench frames takes six
      // batched steps instead of seven (a pair's trees 43.9 -> 42.0 us; an eighth: no fewer).  The one-wave batch kernel keeps half: same-box,
      // rect 0.255 -> 0.250 ms but saturated (38 k candidates: larger heads of large ordinary nodes, more of them invalid) 0.432 -> 0.470.
      constexpr int QT_HEAD_SHIFT = NW == 1 ? 1 : 2;
      uint32_t C = max((mc >> QT_HEAD_SHIFT) + 1u, 2u);  // the head = every node with count >= C: a prefix of the pop order whatever C is
      int B;
      for (;;) {
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) cnt += ((uint32_t)(lk[u] >> 32) >= C) ? 1u : 0u;
        B = (int)wave_sum_u32(cnt);
        if (B <= 64 || C >= mc) break;
        C = (C + mc + 1) >> 1;
      }
      if (B <= 64) {
        QTS(-30)  // head threshold found
        // members -> dense list (table order), then one member per lane, sorted into pop order (key descending)
        int base = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (u * 64 < n_act) {
            const bool mem = (uint32_t)(lk[u] >> 32) >= C;
            const unsigned long long m = __ballot(mem);
            if (mem) {
              const int idx = base + __popcll(m & ((1ull << lane) - 1ull));
              bkey[idx] = lk[u];
              bj[idx] = (uint32_t)(u * 64 + lane);
            }
            base += __popcll(m);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // pop order = keys descending.  Rank by counting (the B keys are read back with wave-uniform addresses, all reads
        // independent) instead of a 64-lane bitonic network (21 dependent cross-lane exchanges of ~100 cycles each).
        unsigned long long mykey = (lane < B) ? bkey[lane] : 0ull;
        uint32_t myj = (lane < B) ? bj[lane] : 0u;
        {
          int rank = 0;
          // (sixteen keys requested per trip before the first is compared: four per trip, each trip waiting for its own reads, the ranking
          //  of a 38-member head was 4.8 k cycles of a lone wave -- r5 stamps, tools/exp/qt_stamps.sh)
          for (int i = 0; i < B; i += 16) {
            unsigned long long o[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) o[u] = bkey[min(i + u, 63)];
#pragma unroll
            for (int u = 0; u < 16; ++u) rank += (i + u < B && o[u] > mykey) ? 1 : 0;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (lane < B) {
            bkey[rank] = mykey;  // keys are distinct (unique sequence numbers): the ranks are a permutation
            bj[rank] = myj;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          mykey = (lane < B) ? bkey[lane] : 0ull;
          myj = (lane < B) ? bj[lane] : 0u;
        }
        QTS(-31)  // head ranked
        const bool act = lane < B;
        const int j = (int)myj;
        const int n = (int)(mykey >> 32);
        const uint2 bp = n_bp[j];
        const uint32_t beg_raw = bp.x, path = bp.y;
        const int beg = (int)(beg_raw & QT_BEG_MASK), code = (int)(beg_raw >> QT_CODE_SHIFT);
        const bool is_pp = pp_ok && code != 0;
        SplitInt sp = {0, 0, 0, 0};
        if (__ballot(act && !is_pp)) {  // (a pre-partitioned node reads its totals: no bounds)
          double midx, midy;
          node_mid(L, (act && !is_pp) ? path : 0u, midx, midy);
          sp = make_split(midx, midy);
        }
        // (BR: the records of an ordinary member its lane holds in registers; larger nodes take the one-node path)
        constexpr int BR = NW == 1 ? 32 : 64;
        const unsigned long long big = __ballot(act && !is_pp && n > BR);
        const int b_eff = big ? min(B, __ffsll((long long)big) - 1) : B;
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0, child_code = 0;
        if (is_pp) {
          const int st = (code - 1) / QT_PP_INTERNAL, k = (code - 1) - QT_PP_INTERNAL * st;
          if (k >= 64) {  // strip, child, grandchild: totals in LDS
            const uint16_t* t = tot + st * QT_PP_TOTALS;
            int ti;
            if (k == 84) {
              ti = 0;
              child_code = 1 + st * QT_PP_INTERNAL + 80;
            } else if (k >= 80) {
              ti = 4 + (k - 80) * 4;
              child_code = 1 + st * QT_PP_INTERNAL + 64 + (k - 80) * 4;
            } else {
              ti = 20 + (k - 64) * 4;
              child_code = 1 + st * QT_PP_INTERNAL + (k - 64) * 4;
            }
            c0 = t[ti], c1 = t[ti + 1], c2 = t[ti + 2], c3 = t[ti + 3];
          } else {  // great-grandchild: its four leaf totals come from global memory, its children are ordinary nodes
            const uint32_t* t4 = (const uint32_t*)(tot4 + st * QT_PP_TOTALS4 + 4 * k);  // (the buffer is only 4-byte aligned)
            const uint32_t ta = t4[0], tb = t4[1];
            c0 = (int)(ta & 0xFFFFu), c1 = (int)(ta >> 16), c2 = (int)(tb & 0xFFFFu), c3 = (int)(tb >> 16);
          }
        }
        // the ordinary members: all records of a node into the registers of its lane (one round trip for the whole head)
        const bool ld = lane < b_eff && !is_pp;
        const int nmax_all = (int)wave_max_u32(ld ? (uint32_t)n : 0u);
        // (cooperative from eight records in the largest member: below that the lane-serial form is over before four barriers are)
        const int r_tot = NW > 1 ? (int)wave_sum_u32(ld ? (uint32_t)n : 0u) : 0;
        const bool coop_go = NW > 1 && IN_LDS && nmax_all >= 8 && r_tot <= KR * NT;  // (wave-uniform)
        if (NW > 1 && coop_go) {
          const int nn = ld ? n : 0;
          const int pm = wave_incl_scan(nn, lane) - nn;
          *(uint4*)(coop + lane * 4) = make_uint4((uint32_t)beg | ((uint32_t)nn << 22), (uint32_t)pm, ((uint32_t)sp.x_lt & 0xFFFFu) | ((uint32_t)sp.x_gt << 16),
                                                  ((uint32_t)sp.y_lt & 0xFFFFu) | ((uint32_t)sp.y_gt << 16));
          *(uint4*)(coop + 256 + lane * 4) = make_uint4(0u, 0u, 0u, 0u);
          if (lane == 0) shared_ints[1] = (uint32_t)r_tot;
          (void)coop_count();
          if (ld) {
            const uint4 cc = *(const uint4*)(coop + 256 + lane * 4);
            c0 = (int)cc.x, c1 = (int)cc.y, c2 = (int)cc.z, c3 = (int)cc.w;
          }
        }
        const int nmax = coop_go ? 0 : nmax_all;  // (the register form below: skipped)
        uint32_t* seg = H + beg;
        uint32_t rec[BR];
#pragma unroll
        for (int c8 = 0; c8 < BR / 8; ++c8) {
          if (c8 * 8 < nmax) {  // (nmax = 0 -- a head of pre-partitioned nodes only, most steps of a tree -- skips all of it)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int i = c8 * 8 + u;
              rec[i] = (ld && i < n) ? seg[i] : 0u;
            }
          } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) rec[c8 * 8 + u] = 0u;
          }
        }
        QTS(-32)  // records requested
        if (ld && !coop_go) {
          int k0 = 0, k1 = 0, k2 = 0, k3 = 0;
#pragma unroll
          for (int c8 = 0; c8 < BR / 8; ++c8) {
            if (c8 * 8 < nmax) {
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int i = c8 * 8 + u;
                const int q = (i < n) ? quadrant_of(rec[i], sp) : -1;
                k0 += q == 0, k1 += q == 1, k2 += q == 2, k3 += q == 3;
              }
            }
          }
          c0 = k0, c1 = k1, c2 = k2, c3 = k3;
        }
        QTS(-33)  // quadrants counted
        const int ne0 = c0 > 0, ne1 = c1 > 0, ne2 = c2 > 0, ne3 = c3 > 0;
        const int added = ne0 + ne1 + ne2 + ne3;
        const bool in = lane < b_eff;
        const int d = in ? added - 1 : 0;
        const int incl = wave_incl_scan(d, lane);              // growth of the table up to and including this pop
        const int em = wave_excl_max(in ? max(max(c0, c1), max(c2, c3)) : 0, lane);
        const unsigned long long bad = __ballot(in && (em > n || added == 0));   // a child of an earlier member comes first | the node vanishes
        const unsigned long long stop = __ballot(in && (n_act + incl >= need));  // the quota is reached by this pop
        int v = b_eff;
        if (bad) v = min(v, __ffsll((long long)bad) - 1);
        if (stop) v = min(v, __ffsll((long long)stop));
        QTS(-34)  // prefix of valid pops
        if (NW > 1 && coop_go) {  // the quadrants' places, the number of valid pops; every thread stores its records (members past v keep theirs)
          if (ld) *(uint4*)(coop + 256 + lane * 4) = make_uint4((uint32_t)beg, (uint32_t)(beg + c0), (uint32_t)(beg + c0 + c1), (uint32_t)(beg + c0 + c1 + c2));
          if (lane == 0) shared_ints[2] = (uint32_t)v;
          coop_scatter();
        }
        if (v > 0) {
          const bool cm = lane < v;
          const int ex = incl - d;  // table growth of the earlier pops = (children of the earlier pops) - (earlier pops)
          const uint32_t seq0 = next_seq + (uint32_t)(ex + lane);
          const int slot0 = n_act + ex;
          if (cm) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int cc = q == 0 ? c0 : (q == 1 ? c1 : (q == 2 ? c2 : c3));
              const int offc = beg + (q == 0 ? 0 : (q == 1 ? c0 : (q == 2 ? c0 + c1 : c0 + c1 + c2)));
              const int rank = q == 0 ? 0 : (q == 1 ? ne0 : (q == 2 ? ne0 + ne1 : ne0 + ne1 + ne2));
              if (cc > 0) {
                const int slot = rank == 0 ? j : slot0 + rank - 1;  // first child into the popped node's slot, the others appended
                n_key[slot] = ((unsigned long long)cc << 32) | (unsigned long long)(0xFFFFFFFFu - (seq0 + (uint32_t)rank));
                n_bp[slot] = make_uint2((uint32_t)offc | ((uint32_t)(child_code ? child_code + q : 0) << QT_CODE_SHIFT), QT_PATH_CHILD(path, q));
              }
            }
            if (!is_pp && !coop_go) {  // in-place 4-way partition straight from the registers (the node's segment belongs to this lane alone)
              int p0 = 0, p1 = c0, p2 = c0 + c1, p3 = c0 + c1 + c2;
#pragma unroll
              for (int c8 = 0; c8 < BR / 8; ++c8) {
                if (c8 * 8 < nmax) {
#pragma unroll
                  for (int u = 0; u < 8; ++u) {
                    const int i = c8 * 8 + u;
                    const int q = (i < n) ? quadrant_of(rec[i], sp) : -1;
                    if (q >= 0) {
                      const int pos = q == 0 ? p0 : (q == 1 ? p1 : (q == 2 ? p2 : p3));
                      seg[pos] = rec[i];
                      p0 += q == 0, p1 += q == 1, p2 += q == 2, p3 += q == 3;
                    }
                  }
                }
              }
            }
          }
          const int grow = __builtin_amdgcn_readlane(incl, v - 1);
          n_act += grow;
          next_seq += (uint32_t)(grow + v);
          if (n_act + 3 > pp_limit) pp_ok = false;  // the table has reached the totals parked in its tail
          if (!IN_LDS) qt_wsync<NW>();
          QTS((v << 16) | (nmax << 8) | (int)__popcll(__ballot(act && is_pp)))
          continue;
        }
      }
    }
    // pop: arg-max of (count desc, seq asc) = max of the 64-bit key
    uint32_t bc = 0, bs = 0;
    int bj = 0;
    for (int j0 = 0; j0 < n_act; j0 += 128) {  // two table rows per trip; the table is short for most of a tree's life
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = j0 + u * 64 + lane;
        const unsigned long long key = (j < n_act) ? n_key[j] : 0ull;
        const uint32_t kc = (uint32_t)(key >> 32), ks = (uint32_t)key;
        const bool better = kc > bc || (kc == bc && ks > bs);
        bc = better ? kc : bc;
        bs = better ? ks : bs;
        bj = better ? j : bj;
      }
    }
    const uint32_t mc = wave_max_u32(bc);
    if (mc <= 1u) {  // (as above: quirk Q3)
      n_act = 0;
      break;
    }
    const uint32_t ms = wave_max_u32(bc == mc ? bs : 0u);
    const unsigned long long win = __ballot(bc == mc && bs == ms);
    const int j = __builtin_amdgcn_readlane(bj, __ffsll((long long)win) - 1);
    const int n = (int)mc;
#ifdef QT_STAMPS
    ++qts_single, qts_single_n += n;
#endif
    const uint2 bp = n_bp[j];
    const uint32_t beg_raw = bp.x, path = bp.y;
    const int beg = (int)(beg_raw & QT_BEG_MASK), code = (int)(beg_raw >> QT_CODE_SHIFT);
    // (the popped node's slot is reused by its first child below: no erase-and-compact round trip through the table)
    SplitInt sp = {0, 0, 0, 0};
    if (!(pp_ok && code != 0)) {  // wave-uniform
      double midx, midy;
      node_mid(L, path, midx, midy);
      sp = make_split(midx, midy);
    }
    uint32_t* seg = H + beg;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int child_code = 0;  // pre-partition code of child 0 (children q get child_code + q), 0: none
    if (pp_ok && code != 0) {
      // a strip, a child or a grandchild whose records were laid out by the pre-partition: nothing moves
      const int st = (code - 1) / QT_PP_INTERNAL, k = (code - 1) - QT_PP_INTERNAL * st;
      if (k >= 64) {
        const uint16_t* t = tot + st * QT_PP_TOTALS;
        int ti;
        if (k == 84) {
          ti = 0;
          child_code = 1 + st * QT_PP_INTERNAL + 80;
        } else if (k >= 80) {
          ti = 4 + (k - 80) * 4;
          child_code = 1 + st * QT_PP_INTERNAL + 64 + (k - 80) * 4;
        } else {
          ti = 20 + (k - 64) * 4;
          child_code = 1 + st * QT_PP_INTERNAL + (k - 64) * 4;
        }
        c0 = t[ti], c1 = t[ti + 1], c2 = t[ti + 2], c3 = t[ti + 3];
      } else {
        const uint32_t* t4 = (const uint32_t*)(tot4 + st * QT_PP_TOTALS4 + 4 * k);
        const uint32_t ta = t4[0], tb = t4[1];
        c0 = (int)(ta & 0xFFFFu), c1 = (int)(ta >> 16), c2 = (int)(tb & 0xFFFFu), c3 = (int)(tb >> 16);
      }
    } else if (n <= 64) {
      // the common case: one record per lane, in-place 4-way partition
      const uint32_t rec = (lane < n) ? seg[lane] : 0u;
      const int q = (lane < n) ? quadrant_of(rec, sp) : -1;
      const unsigned long long m0 = __ballot(q == 0), m1 = __ballot(q == 1), m2 = __ballot(q == 2), m3 = __ballot(q == 3);
      c0 = __popcll(m0);
      c1 = __popcll(m1);
      c2 = __popcll(m2);
      c3 = __popcll(m3);
      const unsigned long long below = (1ull << lane) - 1ull;
      const unsigned long long mq = q == 0 ? m0 : (q == 1 ? m1 : (q == 2 ? m2 : m3));
      const int baseq = q == 0 ? 0 : (q == 1 ? c0 : (q == 2 ? c0 + c1 : c0 + c1 + c2));
      if (q >= 0) seg[baseq + __popcll(mq & below)] = rec;
    } else if (n <= 64 * QT_INPLACE_CHUNKS) {
      // up to 512 records: in place through registers
      uint32_t rec[QT_INPLACE_CHUNKS];
      int q[QT_INPLACE_CHUNKS];
      int c4[4] = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < QT_INPLACE_CHUNKS; ++c) {
        const int i = c * 64 + lane;
        rec[c] = (i < n) ? seg[i] : 0u;
        q[c] = (i < n) ? quadrant_of(rec[c], sp) : -1;
      }
#pragma unroll
      for (int c = 0; c < QT_INPLACE_CHUNKS; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) c4[k] += __popcll(__ballot(q[c] == k));
      if (!IN_LDS) qt_wsync<NW>();  // every record is in a register before the segment is overwritten
      int run4[4] = {0, c4[0], c4[0] + c4[1], c4[0] + c4[1] + c4[2]};
#pragma unroll
      for (int c = 0; c < QT_INPLACE_CHUNKS; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned long long m = __ballot(q[c] == k);
          if (q[c] == k) seg[run4[k] + __popcll(m & ((1ull << lane) - 1ull))] = rec[c];
          run4[k] += __popcll(m);
        }
      c0 = c4[0];
      c1 = c4[1];
      c2 = c4[2];
      c3 = c4[3];
    } else {
      // big node: count, scatter to the bounce buffer, copy back
      int c4[4] = {0, 0, 0, 0};
      for (int b0 = 0; b0 < n; b0 += 64) {
        const int i = b0 + lane;
        const int q = (i < n) ? quadrant_of(seg[i], sp) : -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) c4[k] += __popcll(__ballot(q == k));
      }
      int run4[4] = {0, c4[0], c4[0] + c4[1], c4[0] + c4[1] + c4[2]};
      uint32_t* tmp = T + beg;
      for (int b0 = 0; b0 < n; b0 += 64) {
        const int i = b0 + lane;
        uint32_t rec = 0;
        int q = -1;
        if (i < n) {
          rec = seg[i];
          q = quadrant_of(rec, sp);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned long long m = __ballot(q == k);
          if (q == k) tmp[run4[k] + __popcll(m & ((1ull << lane) - 1ull))] = rec;
          run4[k] += __popcll(m);
        }
      }
      qt_wsync<NW>();
      const int kept = c4[0] + c4[1] + c4[2] + c4[3];
      for (int i = lane; i < kept; i += 64) seg[i] = tmp[i];
      c0 = c4[0];
      c1 = c4[1];
      c2 = c4[2];
      c3 = c4[3];
    }
    // insert the non-empty children in order TL, TR, BL, BR (ORBExtractor.cc:161-170): lane c writes child c
    {
      const int cc = lane == 0 ? c0 : (lane == 1 ? c1 : (lane == 2 ? c2 : c3));
      const int offc = beg + (lane == 0 ? 0 : (lane == 1 ? c0 : (lane == 2 ? c0 + c1 : c0 + c1 + c2)));
      const int ne0 = c0 > 0, ne1 = c1 > 0, ne2 = c2 > 0, ne3 = c3 > 0;
      const int rank = lane == 0 ? 0 : (lane == 1 ? ne0 : (lane == 2 ? ne0 + ne1 : ne0 + ne1 + ne2));
      if (n_act + 3 > pp_limit) pp_ok = false;  // the table reaches the totals parked in its tail: split the ordinary way from now on
      if (lane < 4 && cc > 0) {
        const int slot = rank == 0 ? j : n_act + rank - 1;  // first child into the popped node's slot, the others appended
        n_key[slot] = ((unsigned long long)cc << 32) | (unsigned long long)(0xFFFFFFFFu - (next_seq + (uint32_t)rank));
        n_bp[slot] = make_uint2((uint32_t)offc | ((uint32_t)(child_code ? child_code + lane : 0) << QT_CODE_SHIFT), QT_PATH_CHILD(path, lane));
      }
      const int added = ne0 + ne1 + ne2 + ne3;
      if (added == 0) {  // every record sat on a split line: the node disappears, the last entry fills its slot
        --n_act;
        const unsigned long long t4 = n_key[n_act];
        const uint2 t5 = n_bp[n_act];
        n_key[j] = t4;
        n_bp[j] = t5;
      } else {
        n_act += added - 1;
      }
      next_seq += (uint32_t)added;
    }
    if (!IN_LDS) qt_wsync<NW>();  // LDS traffic of one wave is executed in order; global needs the wait
  }
  if (NW == 1) {
    __syncthreads();
  } else if (w0) {  // the empty job: the helper waves leave their loop at this barrier (their coop_count's first)
    if (lane == 0) shared_ints[1] = 0u;
    __syncthreads();
  }

  QTS(-2)  // expansion done (single pops included in the last interval)
  // ---- nodes2kpoints (ORBExtractor.cc:182-192): keep the first min(need, size) nodes in map order ----
  // (r6) The last pop overshoots the quota by at most three nodes.  Dropping them one at a time -- a scan of the whole table, two wave
  // reductions and two waits each -- was 5.3 k of a level-0 tree's 94 k cycles, with every helper wave waiting.  Where counts and sequence
  // numbers fit 16 bits each (always, on tables the batched steps handle) the keys are read ONCE as 32-bit surrogates count << 16 | ~seq
  // (the same order), every drop is one lane-minimum over registers and one wave reduction, and the holes are filled from the table's
  // tail at the end (any order: the selection below ranks by candidate order).
  if (NODES_LDS && n_act > need && n_act - need <= 4 && n_act <= 512 && next_seq < 65536u && N <= 65535) {
    const int d = n_act - need, new_n = need;
    uint32_t k32[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = u * 64 + lane;
      k32[u] = 0xFFFFFFFFu;
      if (u * 64 < n_act && j < n_act) {
        const unsigned long long key = n_key[j];
        k32[u] = ((uint32_t)(key >> 32) << 16) | ((uint32_t)key & 0xFFFFu);
      }
    }
    int hole[4] = {-1, -1, -1, -1};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < d) {  // uniform
        uint32_t lm = k32[0];
#pragma unroll
        for (int u = 1; u < 8; ++u) lm = min(lm, k32[u]);
        const uint32_t m = wave_min_u32(lm);
        int ju = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool hit = k32[u] == m;  // (keys are distinct: one register of one lane)
          ju = hit ? u * 64 + lane : ju;
          k32[u] = hit ? 0xFFFFFFFFu : k32[u];
        }
        const unsigned long long win = __ballot(lm == m);
        hole[q] = __builtin_amdgcn_readlane(ju, __ffsll((long long)win) - 1);
      }
    }
    // survivors of the tail [new_n, n_act) -> the holes below new_n (as many of the one as of the other)
    int hi = 0;
    for (int t = new_n; t < n_act; ++t) {  // d trips, everything uniform
      const bool dropped = t == hole[0] || t == hole[1] || t == hole[2] || t == hole[3];
      if (!dropped) {
        int h = -1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (h < 0 && q >= hi && hole[q] >= 0 && hole[q] < new_n) h = hole[q], hi = q + 1;
        if (lane == 0 && h >= 0) {
          n_key[h] = n_key[t];
          n_bp[h] = n_bp[t];
        }
      }
    }
    n_act = new_n;
    qt_wsync<NW>();
  }
  while (n_act > need) {
    // drop the last node in map order = arg-min of the key
    uint32_t bc = 0xFFFFFFFFu, bs = 0xFFFFFFFFu;
    int bj = -1;
    for (int j = lane; j < n_act; j += 64) {
      const unsigned long long key = n_key[j];
      const uint32_t kc = (uint32_t)(key >> 32), ks = (uint32_t)key;
      if (bj < 0 || kc < bc || (kc == bc && ks < bs)) {
        bc = kc;
        bs = ks;
        bj = j;
      }
    }
    const uint32_t mc = wave_min_u32(bj >= 0 ? bc : 0xFFFFFFFFu);
    const uint32_t ms = wave_min_u32((bj >= 0 && bc == mc) ? bs : 0xFFFFFFFFu);
    const unsigned long long win = __ballot(bj >= 0 && bc == mc && bs == ms);
    const int j = __builtin_amdgcn_readlane(bj, __ffsll((long long)win) - 1);
    qt_wsync<NW>();
    --n_act;
    if (lane == 0 && j != n_act) {
      n_key[j] = n_key[n_act];
      n_bp[j] = n_bp[n_act];
    }
    qt_wsync<NW>();
  }
  if (NW > 1) {  // the helper waves join in again: they need the node count
    if (w0 && lane == 0) shared_ints[0] = (uint32_t)n_act;
    __syncthreads();
    n_act = (int)shared_ints[0];
  }
  QTS(-20)  // drop loop + rejoin

  // per node: first maximum response (ORBExtractor.cc:103-117).  Sort key = candidate order recomputed from the coordinates:
  // (cell row, cell col, y, x), then the response.  One lane per node, eight records requested per trip (with the records in
  // global memory every trip is a round trip the lone wave waits for).
  int sc = 2;
  while (sc < need) sc <<= 1;  // need <= sort_cap by construction (host side)
  sort_cap = min(sort_cap, sc);
  auto node_key = [&](int j) -> unsigned long long {
    if (j >= n_act) return ~0ull;
    const uint32_t* p = H + (n_bp[j].x & QT_BEG_MASK);
    const int n = (int)(n_key[j] >> 32);
    uint32_t best = 0, br = 0;
    unsigned long long bk = ~0ull;
    for (int i0 = 0; i0 < n; i0 += 8) {
      uint32_t rec[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) rec[u] = p[min(i0 + u, n - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t r = ORBFE_REC_R(rec[u]);
        if (i0 + u < n && (bk == ~0ull || r >= br)) {  // maximum response; the reference keeps the FIRST maximum in candidate order
          const unsigned long long k = order_key(rec[u], L);
          if (bk == ~0ull || r > br || k < bk) {
            best = rec[u];
            br = r;
            bk = k;
          }
        }
      }
    }
    return (bk << 8) | (unsigned long long)ORBFE_REC_R(best);
  };
  // The same winner as node_key, for the sorts of up to 512 keys: the 32-bit order key (order_key32) and the winning record itself.
  // The first 32 records of a lane's node are requested AT ONCE (one round trip; a node of the final table holds 10 - 25 records on the
  // bench frames): eight at a time, every eight a dependent trip to memory, the scan of a level-0 tree's 434 nodes was ~30 round trips
  // of ~2.5 k cycles -- most of the 91 k cycles of the selection phase (r5 stamps, tools/exp/qt_stamps.sh with QT_BATCH).  All compares
  // are 32-bit (the 64-bit integer compares of the old keys are quarter-rate instructions).
  auto node_win = [&](int j, uint32_t& k32, uint32_t& wrec) {
    k32 = 0xFFFFFFFFu, wrec = 0u;
    const bool live = j < n_act;
    const uint32_t* p = H + (live ? (n_bp[j].x & QT_BEG_MASK) : 0u);
    const int n = live ? (int)(n_key[j] >> 32) : 0;
    const int nmax = (int)wave_max_u32((uint32_t)n);
    uint32_t br = 0;
    auto consider = [&](uint32_t rc) {
      const uint32_t r = ORBFE_REC_R(rc);
      if (r >= br) {  // maximum response; the reference keeps the FIRST maximum in candidate order
        const uint32_t k = order_key32(rc, L);
        if (r > br || k < k32) wrec = rc, br = r, k32 = k;
      }
    };
    uint32_t rec[32];
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      if (c8 * 8 < nmax) {  // wave-uniform
#pragma unroll
        for (int u = 0; u < 8; ++u) rec[c8 * 8 + u] = p[min(c8 * 8 + u, max(n - 1, 0))];
      }
    }
    // (r6) The order key -- eighteen instructions -- is formed only for the records AT the node's maximum response, not for every record
    // that is at least the running maximum (across 64 lanes some lane always is, so the key was formed for every index): one pass keeps the
    // maximum and a bit mask of the positions that hold it (six instructions a record), then the lanes walk their masks, one record read
    // back per trip, as many trips as the wave's most-tied node has ties (one to three).
    // (Several waves per tree only: 9.4 -> 5.7 k cycles of a level-0 tree.  In the one-wave batch kernel -- 128 registers -- the same form
    //  took the batch's trees from 0.255 to 0.320 ms, same box; it keeps the running-maximum form.)
    if (NW > 1) {
      uint32_t tie = 0u;
#pragma unroll
      for (int c8 = 0; c8 < 4; ++c8) {
        if (c8 * 8 < nmax) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = c8 * 8 + u;
            const uint32_t r = ORBFE_REC_R(rec[i]);
            const bool in = i < n, gt = in && r > br, eq = in && r == br;
            tie = gt ? (1u << i) : (eq ? (tie | (1u << i)) : tie);
            br = gt ? r : br;
          }
        }
      }
      const int tmax = (int)wave_max_u32((uint32_t)__popc(tie));
      for (int t = 0; t < tmax; ++t) {
        if (tie) {
          const int idx = __ffs((int)tie) - 1;
          tie &= tie - 1u;
          const uint32_t rc = p[idx];
          const uint32_t k = order_key32(rc, L);
          if (k < k32) wrec = rc, k32 = k;
        }
      }
    } else {
#pragma unroll
      for (int c8 = 0; c8 < 4; ++c8) {
        if (c8 * 8 < nmax) {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (c8 * 8 + u < n) consider(rec[c8 * 8 + u]);
        }
      }
    }
    for (int i0 = 32; i0 < nmax; i0 += 8) {  // larger nodes (a table that ended before the quota was reached, a clustered level)
      uint32_t rr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) rr[u] = p[min(i0 + u, max(n - 1, 0))];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < n) consider(rr[u]);
    }
  };
  if (sort_cap <= 512) {
    // bitonic sort of up to 512 keys IN REGISTERS, key index = r * 64 + lane: strides >= 64 pair two registers of a lane,
    // smaller strides exchange with lane ^ stride (a first version through LDS with a barrier per stage took 94 k cycles,
    // a quarter of the pop loop)
    uint32_t key[8], pay[8];  // order key (order_key32) | the node's winning record
    if (NW == 1) {
      // (measured and dropped in r4: chunk-major over four or eight of a lane's nodes at a time, so that their record loads share round
      //  trips -- no gain; what r5 does instead is one deep request per node, see node_win)
      // The scan is a LOOP over the table's rows (not unrolled: eight copies of it were a third of the kernel's code, and the kernel does not
      // fit the instruction cache as it is); a lane parks its node's result in the node's own key slot -- nobody else reads that slot --
      // and the sort below collects eight per lane.
#pragma unroll 1
      for (int j0 = 0; j0 < n_act; j0 += 64) {
        uint32_t k32, wrec;
        node_win(j0 + lane, k32, wrec);
        if (j0 + lane < n_act) n_key[j0 + lane] = ((unsigned long long)wrec << 32) | (unsigned long long)k32;
      }
      QTS(-22)  // winner scan
      qt_wsync<NW>();  // (the table may live in global memory: NODES_LDS = false)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        key[r] = 0xFFFFFFFFu, pay[r] = 0u;
        if (r * 64 < sort_cap && r * 64 + lane < n_act) {
          const unsigned long long kv = n_key[r * 64 + lane];
          key[r] = (uint32_t)kv, pay[r] = (uint32_t)(kv >> 32);
        }
      }
    } else {
      // all waves scan nodes (one node per thread per trip); the keys meet in LDS (the fp64 bound arrays are dead) and every thread
      // RANKS its own keys by counting the smaller ones -- the keys are read back with wave-uniform addresses (one LDS read serves
      // the wave), they are distinct (the order key is the candidate's position), so the ranks are the sorted positions and the
      // keypoints go straight to their output slots.  (Wave 0 alone sorting the keys in registers, 36 cross-lane exchange stages
      // of eight keys, was a quarter of the 98 us of a level-0 tree: stamps build, tools/exp/qt_stamps.sh.)
      __syncthreads();
      constexpr int KPT = (512 + NT - 1) / NT;
      uint32_t mine[KPT];
      uint32_t k32[KPT];
      uint32_t* sort32 = (uint32_t*)sortbuf;  // the ranking compares 32-bit order keys (order_key32)
      // r6: the ranks by BUCKETS of cells instead of by counting all smaller keys (every thread against every key: with eight waves that
      // was 8.6 k of a level-0 tree's 112 k cycles, two waves per SIMD issuing ~1000 vector instructions each).  The order key is
      // cell << 14 | position inside the cell, so a key's rank = the keys in the buckets of earlier cells (a histogram and its prefix) +
      // the smaller keys of its own bucket (one to three on the bench frames).  Histogram and prefix borrow the node table, which is dead
      // once every node's winner is known; a table too small for them (a context of a few dozen features) keeps the counting form.
      constexpr int NBK = NT > 512 ? NT : 512, BPT = NBK / NT;
      const bool by_bucket = NODES_LDS && (size_t)node_cap * 16 >= (size_t)(2 * NBK + NW) * sizeof(uint32_t);  // (uniform)
#pragma unroll
      for (int u = 0; u < KPT; ++u) {
        const int j = tid + u * NT;
        k32[u] = 0xFFFFFFFFu, mine[u] = 0u;
        if (u * NT < sort_cap) node_win(j, k32[u], mine[u]);  // (uniform per wave: every lane of a wave shares u; lanes past n_act get padding)
        if (!by_bucket && j < max(sort_cap, 16)) sort32[j] = k32[u];
      }
      __syncthreads();
      QTS(-21)  // node keys
      if (by_bucket) {
        uint32_t* hist = (uint32_t*)n_key;  // [NBK] keys per bucket | [NBK] exclusive prefix | [NW] wave totals
        uint32_t* pref = hist + NBK;
        uint32_t* wsum = pref + NBK;
        const int cells = L.n_cols * L.n_rows;
        int bsh = 0;
        while (((cells - 1) >> bsh) >= NBK) ++bsh;
        for (int b = tid; b < NBK; b += NT) hist[b] = 0u;
        __syncthreads();
        uint32_t slot[KPT];
        int bk[KPT];
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
          bk[u] = 0, slot[u] = 0u;
          if (tid + u * NT < n_act) {
            bk[u] = (int)((k32[u] >> 14) >> bsh);
            slot[u] = atomicAdd(&hist[bk[u]], 1u);  // arrival order inside the bucket: arbitrary, only used to park the key
          }
        }
        __syncthreads();
        uint32_t loc[BPT], tsum = 0u;  // thread t owns buckets [t BPT, (t + 1) BPT)
#pragma unroll
        for (int q = 0; q < BPT; ++q) loc[q] = hist[tid * BPT + q], tsum += loc[q];
        const uint32_t incl = (uint32_t)wave_incl_scan((int)tsum, lane);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t base = incl - tsum;
        for (int w = 0; w < NW; ++w) base += w < wv ? wsum[w] : 0u;  // (wave-uniform reads)
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
          pref[tid * BPT + q] = base;
          base += loc[q];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < KPT; ++u)
          if (tid + u * NT < n_act) sort32[pref[bk[u]] + slot[u]] = k32[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < KPT; ++u)
          if (tid + u * NT < n_act) {
            const uint32_t b0 = pref[bk[u]], c = hist[bk[u]];
            uint32_t r = b0;
            for (uint32_t i = 0; i < c; ++i) r += sort32[b0 + i] < k32[u] ? 1u : 0u;  // (distinct keys: the ranks are a permutation)
            out_sel[r] = mine[u];
          }
      } else {
      int rank[KPT];
#pragma unroll
      for (int u = 0; u < KPT; ++u) rank[u] = 0;
      // (sixteen keys per trip, all requested before the first is compared: with four the loop ran at the LDS latency, 385 cycles a trip.
      //  The slots past n_act hold ~0: they never count.)
      for (int i = 0; i < sort_cap; i += 16) {
        if (i >= n_act) break;  // wave-uniform
        uint32_t o[16];
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
          const uint4 w = *(const uint4*)(sort32 + i + q);  // (slots up to max(sort_cap, 16) are written)
          o[q] = w.x, o[q + 1] = w.y, o[q + 2] = w.z, o[q + 3] = w.w;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int u = 0; u < KPT; ++u) rank[u] += (o[q] < k32[u]) ? 1 : 0;
      }
#pragma unroll
      for (int u = 0; u < KPT; ++u) {
        const int j = tid + u * NT;
        if (j < n_act) out_sel[rank[u]] = mine[u];
      }
      }
      QTS(-4)
      QTS_PRINT
      if (w0 && lane == 0) *sel_count_out = n_act;
      return;
    }
    if (sort_cap > 256) bitonic_rows<8>(key, pay, lane);  // (wave-uniform: sort_cap is the power of two above the level's quota)
    else if (sort_cap > 128) bitonic_rows<4>(key, pay, lane);
    else if (sort_cap > 64) bitonic_rows<2>(key, pay, lane);
    else bitonic_rows<1>(key, pay, lane);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int j = r * 64 + lane;
      if (j < n_act) out_sel[j] = pay[r];
    }
  } else {
    for (int j = tid; j < sort_cap; j += NT) {
      const unsigned long long key = node_key(j);
      __syncthreads();
      sortbuf[j] = key;
    }
    __syncthreads();
    for (int k = 2; k <= sort_cap; k <<= 1) {
      for (int st = k >> 1; st > 0; st >>= 1) {
        for (int i = tid; i < sort_cap; i += NT) {
          const int p = i ^ st;
          if (p > i) {
            const unsigned long long a = sortbuf[i], b2 = sortbuf[p];
            const bool up = (i & k) == 0;
            if ((a > b2) == up) {
              sortbuf[i] = b2;
              sortbuf[p] = a;
            }
          }
        }
        __syncthreads();
      }
    }
    for (int j = tid; j < n_act; j += NT) {
      const unsigned long long key = sortbuf[j];
      const uint32_t y = (uint32_t)(key >> 20) & 0xFFFu, x = (uint32_t)(key >> 8) & 0xFFFu, r = (uint32_t)key & 0xFFu;  // key = order<<8 | r
      out_sel[j] = ORBFE_PACK_XYR(x, y, r);
    }
  }
  if (w0 && lane == 0) *sel_count_out = n_act;
  QTS(-3)
  QTS_PRINT
}


template <int NW>
__device__ __forceinline__ void quadtree_levels(const LevelDev* __restrict__ lv, int n_levels, const uint32_t* __restrict__ cand,
                                                 uint32_t* __restrict__ scratch_b, uint32_t* __restrict__ scratch_c,
                                                 size_t scratch_pitch,
                                                 uint32_t* __restrict__ sel, int32_t* __restrict__ sel_count, int n_features,
                                                 const int32_t* __restrict__ n_cand, int node_cap, int sort_cap, int rec_cap, int batch,
                                                 QtGroups groups, uint8_t* __restrict__ big_base, size_t big_pitch,
                                                 const uint16_t* __restrict__ qt_tabs, int32_t* __restrict__ qt_next,
                                                 const int32_t* __restrict__ n_cand_sh = nullptr, int n_shards = 1) {
  extern __shared__ unsigned long long lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int img = blockIdx.y;
  // One wave works through the trees of a GROUP of levels, one after the other (blockIdx.x = group; the host balances the groups by
  // quota: the level-0 tree alone, the small levels together).  With one wave per level a 1024-image launch filled every wave slot
  // of the chip with tree waves (8 levels x 1024 images = 8 per SIMD) that mostly wait, and the blur issued beside it on the second
  // stream could only start as trees finished: the two kernels ran back to back, not side by side (timeline).
  // Which trees a wave works through: its group's levels (fixed masks), or -- launches with several waves per image -- the next level nobody
  // has taken yet, most expensive first (groups.order), from the image's counter qt_next[img] (zeroed before the launch): the longest wave of
  // a launch is its critical path, and how long a level's tree takes follows its candidate count, which the host does not know (r5: with
  // masks dealt by quota the wave holding levels {0, 7} or {1, 6} ran 15 - 20 % longer than the others).
#ifdef QT_STAMPS
  const long long qtw_begin = __builtin_amdgcn_s_memtime();
  int qtw_trees = 0;
  unsigned qtw_mask = 0;
  long long qtw_t[8];
#endif
  const bool dyn = NW == 1 && groups.n_order > 0 && qt_next != nullptr;
  uint32_t todo = dyn ? 0u : groups.mask[blockIdx.x];
  for (;;) {
  int level;
  if (dyn) {
    int t = 0;
    if (lane == 0) t = atomicAdd(&qt_next[img], 1);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= groups.n_order) break;
    level = groups.order[t];
  } else {
    if (!todo) break;
    level = __builtin_ctz(todo);
    todo &= todo - 1;
  }
  const LevelDev& L = lv[level];
  // LDS carve-up: u64 keys | (begin, path) pairs | head list of the batched pops | sort buffer (quotas above 512, or several waves per tree) | records.
  unsigned long long* n_key = (unsigned long long*)lds;
  uint2* n_bp = (uint2*)(n_key + node_cap);
  unsigned long long* bkey = (unsigned long long*)(n_bp + node_cap);  // head list of the batched pops: 64 keys + 64 slots
  uint32_t* bj = (uint32_t*)(bkey + 64);
  uint32_t* shared_ints = bj + 64;  // [4]: what wave 0 tells the helper waves of a tree
  unsigned long long* sortbuf = (unsigned long long*)(shared_ints + 4);
  uint32_t* lds_recs = (uint32_t*)((uint8_t*)sortbuf + (sort_cap > 512 ? (size_t)sort_cap * 8 : (NW > 1 ? 2048 : 0)));
  // several waves per tree (launches of a frame or two): the level-4 totals of the pre-partition behind the record cache
  uint16_t* tot4_lds = NW > 1 ? (uint16_t*)(lds_recs + ((rec_cap + 3) & ~3)) : nullptr;

  uint32_t* out_sel = sel + (size_t)img * n_features + L.quota_off;
  const int need = L.quota;
  // the level's candidate SET, appended by k_fast in arbitrary order
  const uint32_t* A = cand + (size_t)img * scratch_pitch + L.cand_base;
  int N;
  const uint32_t* shP = nullptr;
  if (NW > 1 && n_cand_sh && n_shards > 1) {
    // a frame or two: the set arrives in shards (k_fast.hip); their sizes -> N and the exclusive prefix in LDS (the sort area: idle until the
    // expansion, and tree_body reads its records before that)
    const int32_t* cs = n_cand_sh + ((size_t)img * n_levels + level) * n_shards;
    const int cnt = lane < n_shards ? min(max(cs[lane], 0), (int)L.shard_cap) : 0;
    const int incl = wave_incl_scan(cnt, lane);
    N = __builtin_amdgcn_readlane(incl, 63);
    uint32_t* pfx = (uint32_t*)sortbuf;
    if (wv == 0 && lane <= ORBFE_FAST_SHARDS) pfx[lane] = (uint32_t)(incl - cnt);
    shP = pfx;
    __syncthreads();
  } else {
    N = min(n_cand[(size_t)img * n_levels + level], (int)L.cand_cap);
  }
  auto A_rec = [&](int i) __attribute__((always_inline)) -> uint32_t {  // (tree_body's A_at, for the one-feature level below)
    if (!shP) return A[i];
    int sh = 0;
#pragma unroll
    for (int step = ORBFE_FAST_SHARDS / 2; step; step >>= 1)
      if ((int)shP[sh + step] <= i) sh += step;
    return A[(size_t)sh * L.shard_cap + (size_t)(i - (int)shP[sh])];
  };
  const bool in_lds = N <= rec_cap;
  uint32_t* gb = scratch_b + (size_t)img * scratch_pitch + L.cand_base;
  uint32_t* gc = scratch_c + (size_t)img * scratch_pitch + L.cand_base;

  if (need <= 1) {
    // while (mnNodes < mnNeedNodes ...) never runs: the map holds only the root (ORBExtractor.cc:151)
    if (wv != 0) continue;  // (one wave's work)
    if (need == 1 && N > 0) {
      // root->getFeature(): maximum response, first in candidate order on ties
      uint32_t br = 0;
      unsigned long long bk = ~0ull;
      uint32_t brec = 0;
      for (int i = lane; i < N; i += 64) {
        const uint32_t rec = A_rec(i);
        const uint32_t r = ORBFE_REC_R(rec);
        const unsigned long long k = order_key(rec, L);
        if (r > br || (r == br && k < bk)) {
          br = r;
          bk = k;
          brec = rec;
        }
      }
      const uint32_t mr = wave_max_u32(br);
      const uint32_t khi = wave_min_u32(br == mr ? (uint32_t)(bk >> 24) : 0xFFFFFFFFu);
      const uint32_t klo = wave_min_u32((br == mr && (uint32_t)(bk >> 24) == khi) ? (uint32_t)(bk & 0xFFFFFFu) : 0xFFFFFFFFu);
      const unsigned long long win = __ballot(br == mr && (uint32_t)(bk >> 24) == khi && (uint32_t)(bk & 0xFFFFFFu) == klo);
      const uint32_t rec = (uint32_t)__builtin_amdgcn_readlane((int)brec, __ffsll((long long)win) - 1);
      if (lane == 0) {
        out_sel[0] = rec;
        sel_count[(size_t)img * n_levels + level] = 1;
      }
    } else if (lane == 0) {
      sel_count[(size_t)img * n_levels + level] = 0;
    }
    continue;
  }

  if (N < need) {
    // Fewer candidates than the quota: the selection is EMPTY whatever the points are (quirk Q3).  Every node in the map holds at least
    // one record, so mnNodes <= N < mnNeedNodes for ever and `while (mnNodes < mnNeedNodes && !toSplitNodes.empty())`
    // (ORBExtractor.cc:151) can only end with the map empty -- the reference halves every point until it sits on a split line.
    // tree_body reaches the same result by simulating those pops down to one-record nodes: thousands of steps for a few hundred
    // clustered records (the "sparse" content class: 0.57 ms per 1024 images against 0.31 for full levels).
    if (wv == 0 && lane == 0) sel_count[(size_t)img * n_levels + level] = 0;
    continue;
  }

  if (L.qt_big_cap > 0) {  // the level's node table does not fit the LDS: everything in global memory
    const size_t cap = (size_t)L.qt_big_cap;
    unsigned long long* g_key = (unsigned long long*)(big_base + (size_t)img * big_pitch + L.qt_big_off);
    uint2* g_bp = (uint2*)(g_key + cap);
    unsigned long long* g_sort = (unsigned long long*)(g_bp + cap);
    tree_body<false, NW, false>(L, A, N, gb, gc, g_key, g_bp, g_sort, bkey, bj, shared_ints, batch, (int)cap, need,
                                L.qt_big_sort, out_sel, sel_count + (size_t)img * n_levels + level, lane, wv, qt_tabs, nullptr, shP);
  } else if (in_lds)
    tree_body<true, NW>(L, A, N, lds_recs, gb, n_key, n_bp, sortbuf, bkey, bj, shared_ints, batch, node_cap, need, sort_cap,
                        out_sel, sel_count + (size_t)img * n_levels + level, lane, wv, qt_tabs, tot4_lds, shP);
  else
    tree_body<false, NW>(L, A, N, gb, gc, n_key, n_bp, sortbuf, bkey, bj, shared_ints, batch, node_cap, need, sort_cap,
                         out_sel, sel_count + (size_t)img * n_levels + level, lane, wv, qt_tabs, tot4_lds, shP);
  // the next tree reuses the LDS: the accesses of one wave execute in order, the fence only pins the compiler
  if (NW > 1) __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#ifdef QT_STAMPS
  qtw_t[qtw_trees & 7] = __builtin_amdgcn_s_memtime();
  ++qtw_trees;
  qtw_mask |= 1u << level;
#endif
  }
#ifdef QT_STAMPS
  if (NW == 1 && lane == 0 && (blockIdx.y % 100) == 7)
    printf("QTW img %d wave %d trees %d mask %u begin %lld d0 %lld d1 %lld d2 %lld\n", (int)blockIdx.y, (int)blockIdx.x, qtw_trees, qtw_mask, qtw_begin,
           qtw_t[0] - qtw_begin, qtw_trees > 1 ? qtw_t[1] - qtw_t[0] : 0ll, qtw_trees > 2 ? qtw_t[2] - qtw_t[1] : 0ll);
#endif
}

#define QT_KERNEL_ARGS                                                                                                                 \
  const LevelDev *__restrict__ lv, int n_levels, const uint32_t *__restrict__ cand, uint32_t *__restrict__ scratch_b,                   \
      uint32_t *__restrict__ scratch_c, size_t scratch_pitch, uint32_t *__restrict__ sel, int32_t *__restrict__ sel_count, int n_features, \
      const int32_t *__restrict__ n_cand, int node_cap, int sort_cap, int rec_cap, int batch, QtGroups groups, uint8_t *__restrict__ big_base, \
      size_t big_pitch, const uint16_t *__restrict__ qt_tabs, int32_t *__restrict__ qt_next
#define QT_KERNEL_PASS \
  lv, n_levels, cand, scratch_b, scratch_c, scratch_pitch, sel, sel_count, n_features, n_cand, node_cap, sort_cap, rec_cap, batch, groups, big_base, big_pitch, qt_tabs, qt_next
// One wave per tree (batches): compiled for FOUR waves per SIMD (128 VGPRs) -- with 16-byte nodes sixteen trees fit a CU's LDS, and the
// launch deals an image's levels to as many waves as make sixteen per CU, so that a SIMD has four dependent chains to interleave.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_quadtree(QT_KERNEL_ARGS) { quadtree_levels<1>(QT_KERNEL_PASS); }
// Four waves per tree (a frame or two on an otherwise empty chip): registers are free.  The launch also carries the BLUR of the
// frame(s) as extra workgroups (blockIdx.x >= n_groups: one blur tile each, blur.n_tiles of them): only the descriptors read the blurred
// planes, so a pair's 15 us of blur run beside its 60 us of trees instead of as a launch of its own in front of FAST.
struct QtBlur {
  const uint8_t* pyr;  // nullptr: no blur tiles in this launch
  uint8_t* blur;
  size_t img_pitch;
  BlurTaps taps;
  int n_groups;
  const int32_t* n_cand_sh;  // != nullptr: the candidate sets arrive in n_shards lists per level (k_fast.hip, SH)
  int n_shards;
};
#ifndef QT_SMALL_WAVES
#define QT_SMALL_WAVES 8  // (a power of two: 4 -> 8 takes a pair's trees from 60 to 50 us, 16 -> 69; tools/exp/qt_waves.sh)
#endif
__global__ __launch_bounds__(64 * QT_SMALL_WAVES) void k_quadtree_w4(QT_KERNEL_ARGS, QtBlur bl) {
  if (bl.pyr && (int)blockIdx.x >= bl.n_groups) {
    if (QT_SMALL_WAVES > 4 && threadIdx.x >= 256) return;  // (a blur tile is 256 threads' work and has no workgroup barrier)
    if (blur_taps_saturate(bl.taps.t))  // (uniform)
      blur_tile<true>(lv, n_levels, bl.pyr, bl.blur, bl.img_pitch, bl.taps, (int)blockIdx.x - bl.n_groups, (int)blockIdx.y, (int)threadIdx.x);
    else
      blur_tile<false>(lv, n_levels, bl.pyr, bl.blur, bl.img_pitch, bl.taps, (int)blockIdx.x - bl.n_groups, (int)blockIdx.y, (int)threadIdx.x);
    return;
  }
  quadtree_levels<QT_SMALL_WAVES>(QT_KERNEL_PASS, bl.n_cand_sh, bl.n_shards);
}

// The pre-partition's coordinate -> code tables of one level (tree_body): x table [tab_w2] then y table [tab_h], uint16 codes as
// described at pp_axis_code.  Host side, once per context; returns false when the level's geometry rules the pre-partition out.
bool quadtree_build_tables(const LevelDev& L, std::vector<uint16_t>& out) {
  const int ns = L.n_ini;
  out.clear();
  if (ns < 1 || ns > QT_PP_MAX_STRIPS) return false;
  const int tab_w = (int)std::ceil(L.strips[ns]) + 1, tab_h = (int)std::ceil((double)L.reg_h) + 1;
  if (tab_w > 4096 || tab_h > 4096 || tab_w < 1 || tab_h < 1) return false;
  const int tab_w2 = (tab_w + 1) & ~1;
  out.assign((size_t)tab_w2 + tab_h + 1, 0);
  const int y_max = (int)std::ceil((double)L.reg_h) - 1;  // y > 0 && y < reg_h
  Thr yt[15];
  thr15(0.0, (double)L.reg_h, yt);
  for (int k = 0; k < ns; ++k) {
    const double lo = L.strips[k], hi = L.strips[k + 1];
    const int s_lo = (int)std::floor(lo) + 1, s_hi = (int)std::ceil(hi) - 1;  // x > lo, x < hi
    Thr t[15];
    thr15(lo, hi, t);
    for (int x = std::max(s_lo, 0); x <= s_hi && x < tab_w; ++x) out[x] = (uint16_t)(0x8000u | ((uint32_t)k << 12) | pp_axis_code(x, t));
  }
  for (int y = 0; y < tab_h; ++y) out[tab_w2 + y] = (uint16_t)(((y >= 1 && y <= y_max) ? 0x8000u : 0u) | pp_axis_code(y, yt));
  return true;
}

// node table (16 bytes a node) + head list of the batched pops + flags + the sort buffer of quotas above 512 + record cache
size_t quadtree_lds_bytes(int node_cap, int rec_cap, int sort_cap) {
  return (size_t)node_cap * 16 + 64 * (sizeof(unsigned long long) + sizeof(uint32_t)) + 16 + (sort_cap > 512 ? (size_t)sort_cap * 8 : 0) +
         (size_t)rec_cap * sizeof(uint32_t);
}

// The dynamic-LDS limit of a kernel is state of the process and the device, not of a context: contexts of different geometries come
// and go (on several threads), so the limit is only ever RAISED, under a lock -- a later, smaller context must not lower it under an
// earlier one whose launches still ask for more.
hipError_t quadtree_configure(size_t lds_bytes) {
  static std::mutex mu;
  static size_t current[64] = {0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(mu);
  const int slot = dev >= 0 && dev < 64 ? dev : 63;
  if (lds_bytes <= current[slot]) return hipSuccess;
  e = hipFuncSetAttribute((const void*)k_quadtree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_quadtree_w4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e == hipSuccess) current[slot] = lds_bytes;
  return e;
}

void launch_quadtree(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint32_t* d_cand, uint32_t* d_scr_b, uint32_t* d_scr_c,
                     size_t scratch_pitch, uint32_t* d_sel, int32_t* d_sel_count, int n_features, const int32_t* d_n_cand, int node_cap, int sort_cap,
                     int rec_cap, int n_img, int batch, const QtGroups& groups, int n_groups, int waves_per_tree, uint8_t* d_big, size_t big_pitch,
                     const uint16_t* d_qt_tabs, const uint8_t* blur_pyr, uint8_t* blur_out, size_t img_pitch, const int* blur_taps, int blur_tiles,
                     int32_t* d_qt_next, bool next_zeroed, const int32_t* d_n_cand_sh, int n_shards) {
  // d_qt_next != nullptr and groups.n_order > 0 (one-wave launches only): the image's waves pull levels from d_qt_next[img], zeroed here
  // unless the caller has had them zeroed already (next_zeroed: run_extract lets the resize kernel do it)
  // blur_pyr != nullptr (four-wave launches only): the blur of the same images rides in this launch, blur_tiles workgroups per image
  if (n_img <= 0) return;
  const size_t lds = quadtree_lds_bytes(node_cap, rec_cap, sort_cap);
  QtBlur bl{};
  bl.n_groups = n_groups;
  bl.n_cand_sh = (waves_per_tree >= 4 && n_shards > 1) ? d_n_cand_sh : nullptr;  // (the caller shards only where this launch is the several-waves one)
  bl.n_shards = bl.n_cand_sh ? n_shards : 1;
  if (waves_per_tree >= 4 && blur_pyr && blur_tiles > 0) {
    bl.pyr = blur_pyr, bl.blur = blur_out, bl.img_pitch = img_pitch;
    for (int i = 0; i < 7; ++i) bl.taps.t[i] = blur_taps[i];
  }
  if (waves_per_tree >= 4)
    hipLaunchKernelGGL(k_quadtree_w4, dim3(n_groups + (bl.pyr ? blur_tiles : 0), n_img), dim3(64 * QT_SMALL_WAVES), lds + 2048 + 16 + QT_PP_MAX_STRIPS * QT_PP_TOTALS4 * sizeof(uint16_t), s, d_lv, n_levels, d_cand, d_scr_b, d_scr_c, scratch_pitch, d_sel,
                       d_sel_count, n_features, d_n_cand, node_cap, sort_cap, rec_cap, batch, groups, d_big, big_pitch, d_qt_tabs, (int32_t*)nullptr, bl);
  else {
    int32_t* next = (groups.n_order > 0 && n_groups > 1) ? d_qt_next : nullptr;
    if (next && !next_zeroed) (void)hipMemsetAsync(next, 0, (size_t)n_img * sizeof(int32_t), s);
    hipLaunchKernelGGL(k_quadtree, dim3(n_groups, n_img), dim3(64), lds, s, d_lv, n_levels, d_cand, d_scr_b, d_scr_c, scratch_pitch, d_sel,
                       d_sel_count, n_features, d_n_cand, node_cap, sort_cap, rec_cap, batch, groups, d_big, big_pitch, d_qt_tabs, next);
  }
}

}  // namespace orbfe
