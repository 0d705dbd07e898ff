// k_blur_mfma.hip -- cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of the batches on the INTEGER matrix cores
// (src/ORB_SLAM2/src/ORBExtractor.cc:319; OpenCV's fixed-point path: SURVEY A.3).
//
// Why: the step of a batch is bound by VECTOR ISSUE chip-wide, the blur runs beside FAST on the second stream and k_blur
// (register sliding window, v_dot4 / v_dot2) costs ~176 vector instructions per 768 pixels at its real tile efficiency.  The filter is
// exact integer arithmetic -- out = (sum_dy sum_dx t[dy] t[dx] p + 2^15) >> 16 with 8-bit taps --, i.e. two banded matrix products, and
// the matrix cores are idle.  Here both passes are v_mfma_i32_16x16x32_i8; the vector unit only re-packs bytes (~60 instructions per 768
// pixels).  Not a GEMM dressed up to reach MFMA: the same byte stream at a third of the vector instructions.
//
//   row pass     H'[r][c] = sum_k (P[r][16b - 8 + k] - 128) Tx_b[k][c - 16b]          A = 16 image rows x 32 columns (bytes ^ 0x80 = p - 128),
//                                                                                       B = the block's 32 x 16 band of taps
//   column pass  V'[c][r] = sum_k H'(row R(k), c) Ty_j[k][r - 16j]                    A = 16 columns x 32 rows of H' (as TWO byte planes:
//                H' = 256 hi + lo, lo - 128 signed), B = the row block's 32 x 16 band
//   out = (256 acc_hi + acc_lo + 2^23 + 2^15 + 2^15) >> 16  (the constants undo the three -128 offsets; the accumulator starts at them)
//
// BORDER_REFLECT_101 lives in the BANDS: the host folds the taps of reflected positions onto the pixels they mirror (per 16-column block
// Tx_b, per 16-row block Ty_j; interior blocks all get the same band), positions outside the image get weight zero, so whatever bytes
// are loaded there (addresses are clamped into the plane) cannot matter.  No per-pixel border code at all.
//
// The row pass leaves a 16 x 16 tile in the accumulator layout (lane = column, registers = rows 4 (lane >> 4) + i); the column pass
// takes exactly that as its A operand (lane = "row" of A = image column, bytes = K = image rows) when the K slots are ORDERED like the
// registers -- slot (lane >> 4, 4 t + i) = row 16 (j + t) - 8 + 4 (lane >> 4) + i -- and Ty_j is built in that order: no lane movement,
// no LDS between the passes.  Its result has lane = image row, registers = four consecutive columns.
//
// One wave = a strip of 48 columns (three 16-column blocks), walking down the plane in 16-row tiles, the next tile's operands requested
// a tile ahead; the four strips of a workgroup hand their row blocks to shared LDS and leave as 192 contiguous bytes per row.  Taps
// must sum to 256 with every folded weight <= 127 (variant 0); other tap sets and the launches of a frame or two keep k_blur.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "orbfe_internal.h"
#include "wave_ops.h"

namespace orbfe {

#ifndef MB_WAVES
#define MB_WAVES 4  // waves (= neighbouring strips) per workgroup: 4 or 8
#endif
#define MB_COLS 48

typedef int mb_v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int mb_reflect101(int p, int n) {
  while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
  return p;
}

__global__ __launch_bounds__(64 * MB_WAVES) void k_blur_mfma(const LevelDev* __restrict__ lv, int n_levels, MbGeom g, const uint8_t* __restrict__ pyr,
                                                   uint8_t* __restrict__ blur, size_t img_pitch, const uint2* __restrict__ tx_tab,
                                                   const uint2* __restrict__ ty_tab) {
  __shared__ __attribute__((aligned(16))) uint8_t s_out[2][16 * MB_WAVES * MB_COLS];  // the workgroup's row block on its way out: 16 rows x 192 bytes, two of them in turn
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  int l = 0;
  while (l + 1 < n_levels && (int)blockIdx.x >= g.lv[l + 1].wg_base) ++l;  // wave-uniform
  // the four waves of a workgroup = four neighbouring strips; a wave past the row's last strip has nothing to compute but keeps the
  // workgroup's barriers company (the row blocks leave through shared LDS, below)
  const int wgx = (int)blockIdx.x - g.lv[l].wg_base;
  const int strip = wgx * MB_WAVES + wv;
  const bool wave_active = strip < g.lv[l].strips;
  const LevelDev& L = lv[l];
  const int w = __builtin_amdgcn_readfirstlane(L.w), h = __builtin_amdgcn_readfirstlane(L.h), stride = __builtin_amdgcn_readfirstlane(L.stride);
  const int img = blockIdx.y;
  const uint8_t* P = pyr + (size_t)img * img_pitch + L.plane_off;
  uint8_t* D = blur + (size_t)img * img_pitch + L.plane_off;

  const int q = lane >> 4, m = lane & 15;

  // the bands of the strip's three column blocks
  long txb[3];
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) {
    const uint2 t = tx_tab[(size_t)(g.lv[l].tx_off + 3 * min(strip, g.lv[l].strips - 1) + nb) * 64 + lane];
    txb[nb] = (long)(((unsigned long long)t.y << 32) | t.x);
  }
  // The row pass's A operand straight from memory in its lane layout: lane (q, m) takes 8 bytes of image row 16 tau - 8 + m at column
  // 16 b - 8 + 8 q of each of its three blocks (the blocks' 32-column windows overlap by half: the second read of a byte hits the L1).
  // Every address is clamped into the plane; a unit that starts left of the row or past its end holds no pixel of the image and meets
  // zero weights.  (First version: the tile's 16 x 64-byte window staged in the wave's LDS by coalesced loads and read back from there
  // -- the same step on `rect`, +0.7 % on `camera`, and 1.3 KB of LDS per wave that FAST beside it cannot use.  Measured and dropped,
  // each +1.5 % on the step although it removes vector instructions: a scalar row offset for the tiles inside the plane -- two code
  // paths --, the ^ 0x80 applied once per staged byte; tiles requested two steps ahead: no change.)
  uint32_t dx[3];
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) dx[nb] = (uint32_t)min(max(MB_COLS * strip - 8 + 16 * nb + 8 * q, 0), stride - 8);
  auto request = [&](int tau, uint2 (&v)[3]) __attribute__((always_inline)) {
    const int gy = min(max(16 * tau - 8 + m, 0), h - 1);
    const uint32_t ro = (uint32_t)(gy * stride);
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) v[nb] = *(const uint2*)(P + (ro + dx[nb]));
  };
  const int n_blocks = (h + 15) >> 4;  // row blocks of the output; tiles tau = 0 .. n_blocks (tile tau = rows 16 tau - 8 .. 16 tau + 7)
  const uint2* ty_l = ty_tab + (size_t)g.lv[l].ty_off * 64 + lane;
  uint2 cur[3];
  uint2 ty_nx = make_uint2(0u, 0u);
  request(0, cur);
  uint32_t hi_t[2][3], lo_t[2][3];  // the packed planes of the tile in K slot 0 / 1 (= tile parity), per column block
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) hi_t[s][nb] = lo_t[s][nb] = 0u;
  // The column pass leaves lane (q, m) four bytes of row m: stored from there, neighbouring LANES would write neighbouring ROWS -- 64
  // separate 4-byte writes per instruction (measured: the kernel 1.8 ms with them, 0.5 without any store) --, and a strip's 48 bytes of a
  // row are a third of a cache line: written strip by strip as the waves come by, 1.35 ms; with the four strips of a workgroup written
  // together 1.06.  So the row block of the whole workgroup meets in shared LDS and wave v stores rows 4 v .. 4 v + 3 of it, twelve
  // 16-byte units = 192 contiguous bytes per row.
  constexpr int UPR = 3 * MB_WAVES, RPW = 16 / MB_WAVES;  // 16-byte units per row of the workgroup, rows per wave
  const int o_rr = UPR == 12 ? (lane * 171) >> 11 : (lane * 171) >> 12, o_unit = lane - UPR * o_rr;  // lane / UPR, lane % UPR (lanes 48 .. 63: nothing to store)
  const int o_row = RPW * wv + min(o_rr, RPW - 1);
  const uint32_t spare_off = g.spare_off - L.plane_off + 16u * (uint32_t)(lane & 15);  // (from D: the 256 spare bytes behind the last plane of the image's block)
  const bool o_lane = lane < 48 && MB_WAVES * MB_COLS * wgx + 16 * o_unit < w;
  const uint32_t st_off0 = (uint32_t)(o_row * stride + MB_WAVES * MB_COLS * wgx + 16 * o_unit);
  const uint32_t o_lds = (uint32_t)(o_row * (MB_WAVES * MB_COLS) + 16 * o_unit);
  const mb_v4i c_init = {8454144, 8454144, 8454144, 8454144};  // 2^23 + 2^15 + 2^15
  const mb_v4i c_zero = {0, 0, 0, 0};

  // tile tau (compile-time parity PAR = tau & 1): stage, row pass, and -- from the second tile on -- the column pass of row block tau - 1
  auto step = [&](const int tau, auto par) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par)::value;
    const uint2 aop[3] = {cur[0], cur[1], cur[2]};
    // the next tile and the next row block's band travel under this tile's work.  (The band FIRST: requested after the rows and used in
    // this very step it made the compiler wait for everything outstanding -- the rows just requested included -- at every step.)
    const uint2 ty = ty_nx;
    if (tau < n_blocks) {  // wave-uniform
      ty_nx = ty_l[(size_t)tau * 64];  // row block tau's, used by tile tau + 1
      request(tau + 1, cur);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // a wave's LDS accesses execute in order; this only pins the compiler
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
      uint2 a = aop[nb];
      a.x ^= 0x80808080u, a.y ^= 0x80808080u;  // p - 128 as a signed byte
      const mb_v4i dh =
          __builtin_amdgcn_mfma_i32_16x16x32_i8((long)(((unsigned long long)a.y << 32) | a.x), txb[nb], c_zero, 0, 0, 0);
      // H' fits 16 bits: bytes 0 / 1 of each register -> the lo and hi planes of four rows
      const uint32_t A = __builtin_amdgcn_perm((uint32_t)dh[1], (uint32_t)dh[0], 0x05010400u);  // lo0 lo1 hi0 hi1
      const uint32_t B = __builtin_amdgcn_perm((uint32_t)dh[3], (uint32_t)dh[2], 0x05010400u);  // lo2 lo3 hi2 hi3
      lo_t[PAR][nb] = __builtin_amdgcn_perm(B, A, 0x05040100u) ^ 0x80808080u;
      hi_t[PAR][nb] = __builtin_amdgcn_perm(B, A, 0x07060302u);
    }
    if (tau >= 1) {  // wave-uniform
      const long tyb = (long)(((unsigned long long)ty.y << 32) | ty.x);
      const int j = tau - 1;
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) {
        const long ah = (long)(((unsigned long long)hi_t[1][nb] << 32) | hi_t[0][nb]);
        const long al = (long)(((unsigned long long)lo_t[1][nb] << 32) | lo_t[0][nb]);
        const mb_v4i vh = __builtin_amdgcn_mfma_i32_16x16x32_i8(ah, tyb, c_zero, 0, 0, 0);
        const mb_v4i vl = __builtin_amdgcn_mfma_i32_16x16x32_i8(al, tyb, c_init, 0, 0, 0);
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ((uint32_t)vh[i] << 8) + (uint32_t)vl[i];
        const uint32_t o = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u) | __builtin_amdgcn_perm(v[3], v[2], 0x06020c0cu);
        *(uint32_t*)(s_out[PAR] + m * (MB_WAVES * MB_COLS) + MB_COLS * wv + 16 * nb + 4 * q) = o;
      }
      __syncthreads();  // (one per row block: the buffer of the block after next is this one again, a barrier further on)
      // every lane stores, every time: the lanes below the image or right of it (and lanes 48 .. 63) write the image block's spare bytes
      // instead.  (Behind a lane condition the store sits in a branch, the compiler can no longer count the stores in flight and waits
      // for ALL memory operations -- the tile requested a moment ago included -- where it needs the oldest loads only.)
      const uint4 ov = *(const uint4*)(s_out[PAR] + o_lds);
      const bool ok = o_lane && 16 * j + o_row < h;
      *(uint4*)(D + (ok ? st_off0 + (uint32_t)(16 * j * stride) : spare_off)) = ov;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  if (!wave_active) {  // (wave-uniform) nothing to compute: the barriers of row blocks 0 .. n_blocks - 1, and this wave's share of the stores
    for (int tau = 1; tau <= n_blocks; ++tau) {
      __syncthreads();
      const int j = tau - 1;
      const uint4 ov = *(const uint4*)(s_out[tau & 1] + o_lds);
      const bool ok = o_lane && 16 * j + o_row < h;
      *(uint4*)(D + (ok ? st_off0 + (uint32_t)(16 * j * stride) : spare_off)) = ov;
    }
    return;
  }
  for (int tau = 0; tau <= n_blocks; tau += 2) {
    step(tau, std::integral_constant<int, 0>{});
    if (tau + 1 <= n_blocks) step(tau + 1, std::integral_constant<int, 1>{});
  }
}

// ---- host: the bands -------------------------------------------------------------------------------------------------------------
// Tx of column block b (columns 16 b .. 16 b + 15 of a row of width w): lane (q, n) holds K slots k = 8 q + jj, column 16 b - 8 + k.
// Ty of row block j: lane (q, n) holds K slots (q, 4 s + i), s = parity of the tile, row 16 tau - 8 + 4 q + i with tau = j or j + 1.
// Returns false when a folded weight does not fit a signed byte.
static bool mb_band(const int taps[7], int n, int blk, bool by_rows, uint8_t out[512]) {
  std::memset(out, 0, 512);
  for (int lane = 0; lane < 64; ++lane) {
    const int q = lane >> 4, o = lane & 15;
    const int pos = 16 * blk + o;  // output row / column
    if (pos >= n) continue;
    for (int jj = 0; jj < 8; ++jj) {
      int src;
      if (by_rows) {
        const int s = jj >> 2, i = jj & 3;
        const int tau = ((blk & 1) == s) ? blk : blk + 1;
        src = 16 * tau - 8 + 4 * q + i;
      } else {
        src = 16 * blk - 8 + 8 * q + jj;
      }
      if (src < 0 || src >= n) continue;
      int wsum = 0;
      for (int d = 0; d < 7; ++d) {
        int p = pos + d - 3;
        while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
        if (p == src) wsum += taps[d];
      }
      if (wsum > 127) return false;
      out[lane * 8 + jj] = (uint8_t)wsum;
    }
  }
  return true;
}

// Builds the geometry and the band tables of a pyramid; ok = false: this tap set / geometry keeps k_blur.
bool mb_build(const LevelDev* lv, int n_levels, const int taps[7], uint32_t spare_off, MbGeom* g, std::vector<uint8_t>* tx, std::vector<uint8_t>* ty) {
  g->spare_off = spare_off;
  int sum = 0;
  for (int d = 0; d < 7; ++d) {
    if (taps[d] < 0 || taps[d] > 127) return false;
    sum += taps[d];
  }
  if (sum != 256 || n_levels > ORBFE_MAX_LEVELS) return false;
  tx->clear();
  ty->clear();
  int wg = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (lv[l].w < 16 || lv[l].h < 16 || lv[l].stride < 64) return false;
    const int strips = (lv[l].w + MB_COLS - 1) / MB_COLS, blocks_y = (lv[l].h + 15) / 16;
    g->lv[l].wg_base = wg;
    g->lv[l].strips = strips;
    g->lv[l].tx_off = (int)(tx->size() / 512);
    g->lv[l].ty_off = (int)(ty->size() / 512);
    wg += (strips + MB_WAVES - 1) / MB_WAVES;
    for (int b = 0; b < 3 * strips; ++b) {
      uint8_t band[512];
      if (!mb_band(taps, lv[l].w, b, false, band)) return false;
      tx->insert(tx->end(), band, band + 512);
    }
    for (int j = 0; j < blocks_y; ++j) {
      uint8_t band[512];
      if (!mb_band(taps, lv[l].h, j, true, band)) return false;
      ty->insert(ty->end(), band, band + 512);
    }
  }
  g->n_wg = wg;
  return true;
}

void launch_blur_mfma(hipStream_t s, const LevelDev* d_lv, int n_levels, const MbGeom& g, const uint8_t* d_pyr, uint8_t* d_blur, size_t img_pitch,
                      const uint8_t* d_tx, const uint8_t* d_ty, int n_img) {
  if (g.n_wg <= 0 || n_img <= 0) return;
  hipLaunchKernelGGL(k_blur_mfma, dim3(g.n_wg, n_img), dim3(64 * MB_WAVES), 0, s, d_lv, n_levels, g, d_pyr, d_blur, img_pitch, (const uint2*)d_tx,
                     (const uint2*)d_ty);
}

}  // namespace orbfe
