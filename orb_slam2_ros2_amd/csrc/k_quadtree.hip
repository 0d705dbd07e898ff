// k_quadtree.hip -- spatially uniform top-N keypoint selection, one wavefront per (image, level).
//
// Replaces Quadtree / QuadtreeNode (include/ORB_SLAM2/ORBExtractor.h:18-93,
// src/ORB_SLAM2/src/ORBExtractor.cc:19-192) as driven by extractFast (:376-386).
//
// The algorithm is inherently sequential (best-first expansion of a priority queue), so the parallelism
// is (a) across the (image, level) pairs of a batch -- one wave each -- and (b) inside one expansion step:
//   * pop  = 64-lane arg-max over the active node table in LDS, key (count desc, insertion seq asc),
//            which is exactly std::multimap<size_t,...,greater> begin() with insertion-order ties;
//   * split = two-pass stable 4-way partition of the node's record segment with wave ballots
//            (records ping-pong between two scratch buffers at the same offsets, so memory is 2N);
//   * membership is the reference's strict test against double-precision bounds (ORBExtractor.h:55-62),
//     midpoints (b+e)/2 in fp64, so points on a split line are dropped exactly as in the reference.
// No "single point => stop" rule, exact min(quota, nodes) truncation, per-node first-maximum response,
// output ordered by candidate index (std::set) -- quirks Q3/Q4 of SURVEY.md.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"

namespace orbfe {

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

struct NodeTab {
  double* rb;
  double* re;
  double* cb;
  double* ce;
  uint32_t* cnt;
  uint32_t* seq;
  uint32_t* beg;
  uint32_t* buf;
};

// Copy the records of `src[0..n)` that lie strictly inside (cb,ce) x (rb,re) to dst[0..), stable.  Wave-uniform args.
__device__ int filter_box(const uint64_t* __restrict__ src, int n, uint64_t* __restrict__ dst, double rb, double re, double cb,
                          double ce, int lane) {
  int out = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    uint64_t rec = 0;
    bool in = false;
    if (i < n) {
      rec = src[i];
      const uint32_t p = (uint32_t)rec;
      const double x = (double)ORBFE_REC_X(p), y = (double)ORBFE_REC_Y(p);
      in = x > cb && x < ce && y > rb && y < re;
    }
    const unsigned long long m = __ballot(in);
    if (in) dst[out + __popcll(m & ((1ull << lane) - 1ull))] = rec;
    out += __popcll(m);
  }
  return out;
}

__global__ __launch_bounds__(64) void k_quadtree(const LevelDev* __restrict__ lv, int n_levels,
                                                 const uint16_t* __restrict__ counts, int n_cells_total,
                                                 const uint32_t* __restrict__ slots, size_t slots_pitch,
                                                 uint64_t* __restrict__ scratch_a, uint64_t* __restrict__ scratch_b,
                                                 size_t scratch_pitch, uint32_t* __restrict__ sel, int32_t* __restrict__ sel_count,
                                                 int n_features, int32_t* __restrict__ n_cand, int node_cap, int sort_cap) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int level = blockIdx.x, img = blockIdx.y;
  const LevelDev& L = lv[level];
  NodeTab T;
  T.rb = lds;
  T.re = T.rb + node_cap;
  T.cb = T.re + node_cap;
  T.ce = T.cb + node_cap;
  uint64_t* sortbuf = (uint64_t*)(T.ce + node_cap);
  T.cnt = (uint32_t*)(sortbuf + sort_cap);
  T.seq = T.cnt + node_cap;
  T.beg = T.seq + node_cap;
  T.buf = T.beg + node_cap;

  uint64_t* bufs[2] = {scratch_a + (size_t)img * scratch_pitch + L.cand_base, scratch_b + (size_t)img * scratch_pitch + L.cand_base};
  const uint16_t* cnt = counts + (size_t)img * n_cells_total + L.cell_base;
  const uint32_t* sl = slots + (size_t)img * slots_pitch + L.slot_base;
  uint32_t* out_sel = sel + (size_t)img * n_features + L.quota_off;
  const int need = L.quota;

  // ---- gather the level's candidates in the reference's order (cell-row-major, in-cell raster) ----
  int N = 0;
  for (int c0 = 0; c0 < L.n_cells; c0 += 64) {
    const int c = c0 + lane;
    const int k = (c < L.n_cells) ? (int)cnt[c] : 0;
    const int incl = wave_incl_scan(k, lane);
    const int excl = incl - k;
    const int total = __shfl(incl, 63);
    const int kmax = wave_max_i(k);
    for (int j = 0; j < kmax; ++j)
      if (j < k) {
        const uint32_t p = sl[(size_t)c * L.cell_cap + j];
        const uint32_t idx = (uint32_t)(N + excl + j);
        bufs[0][idx] = ((uint64_t)idx << 32) | p;
      }
    N += total;
  }
  if (lane == 0) n_cand[(size_t)img * n_levels + level] = N;
  __syncthreads();

  int n_act = 0;
  if (need <= 1) {
    // while (mnNodes < mnNeedNodes ...) never runs: the map holds only the root (ORBExtractor.cc:151)
    if (need == 1 && N > 0) {
      // root->getFeature(): first maximum response over all candidates
      int best_r = -1;
      uint32_t best_i = 0xFFFFFFFFu;
      uint64_t best_rec = 0;
      for (int i = lane; i < N; i += 64) {
        const uint64_t rec = bufs[0][i];
        const int r = (int)ORBFE_REC_R((uint32_t)rec);
        if (r > best_r) {
          best_r = r;
          best_i = (uint32_t)(rec >> 32);
          best_rec = rec;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int r2 = __shfl_xor(best_r, o);
        const uint32_t i2 = __shfl_xor(best_i, o);
        const uint64_t rec2 = __shfl_xor(best_rec, o);
        if (r2 > best_r || (r2 == best_r && i2 < best_i)) {
          best_r = r2;
          best_i = i2;
          best_rec = rec2;
        }
      }
      if (lane == 0) {
        out_sel[0] = (uint32_t)best_rec;
        sel_count[(size_t)img * n_levels + level] = 1;
      }
    } else if (lane == 0) {
      sel_count[(size_t)img * n_levels + level] = 0;
    }
    return;
  }

  // ---- first pop: the root, whose children are the initSplit strips (ORBExtractor.cc:81-96, 147-170) ----
  uint32_t next_seq = 0;
  {
    int off = 0;
    for (int s = 0; s < L.n_ini; ++s) {
      const int c = filter_box(bufs[0], N, bufs[1] + off, 0.0, (double)L.reg_h, L.strips[s], L.strips[s + 1], lane);
      if (c > 0) {
        if (lane == 0) {
          T.rb[n_act] = 0.0;
          T.re[n_act] = (double)L.reg_h;
          T.cb[n_act] = L.strips[s];
          T.ce[n_act] = L.strips[s + 1];
          T.cnt[n_act] = (uint32_t)c;
          T.seq[n_act] = next_seq;
          T.beg[n_act] = (uint32_t)off;
          T.buf[n_act] = 1u;
        }
        ++n_act;
        ++next_seq;
        off += c;
      }
    }
  }
  __syncthreads();

  // ---- best-first expansion ----
  const long long max_iter = 80ll * (long long)N + 1024;  // each point survives < ~64 halvings (fp64); hard stop for safety
  long long iter = 0;
  while (n_act < need && n_act > 0 && iter < max_iter) {
    ++iter;
    // pop: arg-max of (count desc, seq asc)
    unsigned long long best_key = 0ull;
    int best_j = -1;
    for (int j = lane; j < n_act; j += 64) {
      const unsigned long long key = ((unsigned long long)T.cnt[j] << 32) | (unsigned long long)(0xFFFFFFFFu - T.seq[j]);
      if (best_j < 0 || key > best_key) {
        best_key = key;
        best_j = j;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long k2 = __shfl_xor(best_key, o);
      const int j2 = __shfl_xor(best_j, o);
      if (j2 >= 0 && (best_j < 0 || k2 > best_key)) {
        best_key = k2;
        best_j = j2;
      }
    }
    const int j = best_j;
    const double rb = T.rb[j], re = T.re[j], cb = T.cb[j], ce = T.ce[j];
    const int n = (int)T.cnt[j];
    const int beg = (int)T.beg[j];
    const int sb = (int)T.buf[j];
    __syncthreads();
    // erase from the active table (order inside the table is irrelevant, the key decides)
    --n_act;
    if (lane == 0 && j != n_act) {
      T.rb[j] = T.rb[n_act];
      T.re[j] = T.re[n_act];
      T.cb[j] = T.cb[n_act];
      T.ce[j] = T.ce[n_act];
      T.cnt[j] = T.cnt[n_act];
      T.seq[j] = T.seq[n_act];
      T.beg[j] = T.beg[n_act];
      T.buf[j] = T.buf[n_act];
    }
    // split (ORBExtractor.cc:60-72): rows outer, cols inner; child q = 2*row_half + col_half
    const double midy = (rb + re) / 2, midx = (cb + ce) / 2;
    const uint64_t* src = bufs[sb] + beg;
    uint64_t* dst = bufs[sb ^ 1] + beg;
    int c4[4] = {0, 0, 0, 0};
    if (n <= 64) {
      uint64_t rec = 0;
      int q = -1;
      if (lane < n) {
        rec = src[lane];
        const uint32_t p = (uint32_t)rec;
        const double x = (double)ORBFE_REC_X(p), y = (double)ORBFE_REC_Y(p);
        const int qx = (x < midx) ? 0 : ((x > midx) ? 1 : -1);
        const int qy = (y < midy) ? 0 : ((y > midy) ? 1 : -1);
        q = (qx >= 0 && qy >= 0) ? (qy * 2 + qx) : -1;
      }
      unsigned long long m[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        m[c] = __ballot(q == c);
        c4[c] = __popcll(m[c]);
      }
      int base = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (q == c) dst[base + __popcll(m[c] & ((1ull << lane) - 1ull))] = rec;
        base += c4[c];
      }
    } else {
      for (int b0 = 0; b0 < n; b0 += 64) {
        const int i = b0 + lane;
        int q = -1;
        if (i < n) {
          const uint32_t p = (uint32_t)src[i];
          const double x = (double)ORBFE_REC_X(p), y = (double)ORBFE_REC_Y(p);
          const int qx = (x < midx) ? 0 : ((x > midx) ? 1 : -1);
          const int qy = (y < midy) ? 0 : ((y > midy) ? 1 : -1);
          q = (qx >= 0 && qy >= 0) ? (qy * 2 + qx) : -1;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) c4[c] += __popcll(__ballot(q == c));
      }
      int basec[4];
      basec[0] = 0;
      basec[1] = c4[0];
      basec[2] = c4[0] + c4[1];
      basec[3] = c4[0] + c4[1] + c4[2];
      int run[4] = {0, 0, 0, 0};
      for (int b0 = 0; b0 < n; b0 += 64) {
        const int i = b0 + lane;
        int q = -1;
        uint64_t rec = 0;
        if (i < n) {
          rec = src[i];
          const uint32_t p = (uint32_t)rec;
          const double x = (double)ORBFE_REC_X(p), y = (double)ORBFE_REC_Y(p);
          const int qx = (x < midx) ? 0 : ((x > midx) ? 1 : -1);
          const int qy = (y < midy) ? 0 : ((y > midy) ? 1 : -1);
          q = (qx >= 0 && qy >= 0) ? (qy * 2 + qx) : -1;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned long long m = __ballot(q == c);
          if (q == c) dst[basec[c] + run[c] + __popcll(m & ((1ull << lane) - 1ull))] = rec;
          run[c] += __popcll(m);
        }
      }
    }
    // insert the non-empty children in order TL, TR, BL, BR (ORBExtractor.cc:161-170)
    int off = beg;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c4[c] > 0) {
        if (lane == 0) {
          T.rb[n_act] = (c & 2) ? midy : rb;
          T.re[n_act] = (c & 2) ? re : midy;
          T.cb[n_act] = (c & 1) ? midx : cb;
          T.ce[n_act] = (c & 1) ? ce : midx;
          T.cnt[n_act] = (uint32_t)c4[c];
          T.seq[n_act] = next_seq;
          T.beg[n_act] = (uint32_t)off;
          T.buf[n_act] = (uint32_t)(sb ^ 1);
        }
        ++n_act;
        ++next_seq;
      }
      off += c4[c];
    }
    __syncthreads();
  }

  // ---- nodes2kpoints (ORBExtractor.cc:182-192): keep the first min(need, size) nodes in map order ----
  while (n_act > need) {
    // drop the last node in map order = arg-min of (count, then latest insertion)
    unsigned long long worst_key = ~0ull;
    int worst_j = -1;
    for (int j = lane; j < n_act; j += 64) {
      const unsigned long long key = ((unsigned long long)T.cnt[j] << 32) | (unsigned long long)(0xFFFFFFFFu - T.seq[j]);
      if (worst_j < 0 || key < worst_key) {
        worst_key = key;
        worst_j = j;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long k2 = __shfl_xor(worst_key, o);
      const int j2 = __shfl_xor(worst_j, o);
      if (j2 >= 0 && (worst_j < 0 || k2 < worst_key)) {
        worst_key = k2;
        worst_j = j2;
      }
    }
    __syncthreads();
    --n_act;
    if (lane == 0 && worst_j != n_act) {
      const int j = worst_j;
      T.cnt[j] = T.cnt[n_act];
      T.seq[j] = T.seq[n_act];
      T.beg[j] = T.beg[n_act];
      T.buf[j] = T.buf[n_act];
    }
    __syncthreads();
  }

  // per node: first maximum response (ORBExtractor.cc:103-117); records are in candidate order inside a node
  int sc = 2;
  while (sc < need) sc <<= 1;  // need <= sort_cap by construction (host side)
  sort_cap = min(sort_cap, sc);
  for (int j = lane; j < sort_cap; j += 64) {
    uint64_t best_rec = ~0ull;
    if (j < n_act) {
      const uint64_t* p = bufs[T.buf[j]] + T.beg[j];
      const int n = (int)T.cnt[j];
      int best_r = -1;
      for (int i = 0; i < n; ++i) {
        const uint64_t rec = p[i];
        const int r = (int)ORBFE_REC_R((uint32_t)rec);
        if (r > best_r) {
          best_r = r;
          best_rec = rec;
        }
      }
    }
    sortbuf[j] = best_rec;
  }
  __syncthreads();
  // bitonic sort ascending on the candidate index (high word) = std::set<size_t> iteration order
  for (int k = 2; k <= sort_cap; k <<= 1) {
    for (int s = k >> 1; s > 0; s >>= 1) {
      for (int i = lane; i < sort_cap; i += 64) {
        const int p = i ^ s;
        if (p > i) {
          const uint64_t a = sortbuf[i], b = sortbuf[p];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            sortbuf[i] = b;
            sortbuf[p] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int j = lane; j < n_act; j += 64) out_sel[j] = (uint32_t)sortbuf[j];
  if (lane == 0) sel_count[(size_t)img * n_levels + level] = n_act;
}

size_t quadtree_lds_bytes(int node_cap, int sort_cap) {
  return (size_t)node_cap * (4 * sizeof(double) + 4 * sizeof(uint32_t)) + (size_t)sort_cap * sizeof(uint64_t);
}

void launch_quadtree(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint16_t* d_counts, int n_cells_total,
                     const uint32_t* d_slots, size_t slots_pitch, uint64_t* d_scr_a, uint64_t* d_scr_b, size_t scratch_pitch,
                     uint32_t* d_sel, int32_t* d_sel_count, int n_features, int32_t* d_n_cand, int node_cap, int sort_cap,
                     int n_img) {
  if (n_img <= 0) return;
  const size_t lds = quadtree_lds_bytes(node_cap, sort_cap);
  hipLaunchKernelGGL(k_quadtree, dim3(n_levels, n_img), dim3(64), lds, s, d_lv, n_levels, d_counts, n_cells_total, d_slots,
                     slots_pitch, d_scr_a, d_scr_b, scratch_pitch, d_sel, d_sel_count, n_features, d_n_cand, node_cap, sort_cap);
}

}  // namespace orbfe
