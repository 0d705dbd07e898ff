// blur_body.h -- cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of one pyramid tile by one workgroup of four independent waves
// (src/ORB_SLAM2/src/ORBExtractor.cc:319).  The body of k_blur (k_pyramid.hip), shared with k_quadtree_w4 (k_quadtree.hip), whose
// launches of a frame or two carry the blur's tiles as extra workgroups: the blurred planes are read by the descriptors only, so the
// 15 us of a pair's blur run beside the trees instead of in front of FAST.
#pragma once
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"
#include "wave_ops.h"

namespace orbfe {

#define BLUR_WORDS 62
#ifndef BLUR_DOT2
#define BLUR_DOT2 1
#endif

struct BlurTaps {
  int t[7];
};

__device__ __forceinline__ int reflect101(int p, int n) {
  // n >= 38 and |p| < n + 80 here, so at most two reflections
  while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
  return p;
}

__device__ __forceinline__ uint32_t mad24(uint32_t tap_uniform, uint32_t v, uint32_t acc) {
  uint32_t d;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(tap_uniform), "v"(v), "v"(acc));
  return d;
}
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x138, 0xf, 0xf, true); }  // (bound_ctrl: the lane without a source reads 0, no "old" value to set up)
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v) { return __builtin_amdgcn_update_dpp(0u, v, 0x130, 0xf, 0xf, true); }

// tile: flattened blur tile of the pyramid (LevelDev::bl_tile_base), img: image slot, tid: thread of a 256-thread workgroup
// SAT = false: taps that sum to <= 256 (variant 0) can reach neither the 16-bit saturation of the row pass (255 * 256 < 65536) nor the
// 8-bit one of the column pass (256 * 65280 + 0x8000 < 2^24 + 2^16): the clamps and the byte-by-byte packing are left out at COMPILE
// time (as a run-time flag the compiler kept both and selected per lane: 8 of a row's 54 vector instructions).  blur_taps_saturate()
// tells the caller which instance a tap set needs.
__host__ __device__ inline bool blur_taps_saturate(const int* t) { return (t[0] & 255) + (t[1] & 255) + (t[2] & 255) + (t[3] & 255) + (t[4] & 255) + (t[5] & 255) + (t[6] & 255) > 256; }
template <bool SAT>
__device__ __forceinline__ void blur_tile(const LevelDev* __restrict__ lv, int n_levels, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                          size_t img_pitch, const BlurTaps& taps, int tile, int img, int tid) {
  int l = 0;
  while (l + 1 < n_levels && tile >= lv[l + 1].bl_tile_base) ++l;
  const LevelDev& L = lv[l];
  const int t = tile - L.bl_tile_base;
  // (the wave index is uniform, but only a readfirstlane tells the compiler: the row arithmetic below -- reflections, row * stride --
  // then runs on the scalar unit instead of as quarter-rate 64-bit vector multiplies)
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = t % L.bl_tiles_x;
  const int y0 = ((t / L.bl_tiles_x) * 4 + wv) * BLUR_ROWS;
  if (y0 >= L.h) return;  // wave-uniform
  // (the level's geometry is wave-uniform, but it arrives through vector loads: readfirstlane puts it where the row arithmetic is scalar)
  const int w = __builtin_amdgcn_readfirstlane(L.w), h = __builtin_amdgcn_readfirstlane(L.h), stride = __builtin_amdgcn_readfirstlane(L.stride);
  // Words of a row: n_words hold at least one pixel, the last one (W) q of them.  The last strip of a row is RIGHT-ALIGNED (it starts at
  // word n_words - 62 and recomputes a few words of its left neighbour: the same values) so that word W sits at lane 62 with the
  // three words before it in the same wave; a row narrower than a strip is one strip from word 0.
  const int n_words = (w + 3) >> 2, W = n_words - 1, q = w - 4 * W;
  const int o = min(strip * BLUR_WORDS, max(0, n_words - BLUR_WORDS));
  const int x4 = (o + lane - 1) * 4;  // lane 0 / 63 = left / right halo word
  const uint8_t* P = pyr + (size_t)img * img_pitch + L.plane_off;
  uint8_t* D = blur + (size_t)img * img_pitch + L.plane_off;
  // (the right-aligned last strip computes the words it shares with its left neighbour but leaves their stores to that strip)
  const bool writer = (lane >= 1) && (lane <= BLUR_WORDS) && (x4 < w) && (o + lane - 1 >= strip * BLUR_WORDS);
  // BORDER_REFLECT_101 in x without a single per-byte load: every lane loads an aligned word (address clamped into the row) and the
  // two or three lanes that hold pixels outside the image rebuild their word from their neighbours' with one v_perm_b32 --
  //   left  (strips that start at word 0): lane 0 = pixels -4..-1 = pixels 4, 3, 2, 1: bytes 3, 2, 1 of lane 1's word (byte 0 feeds nothing);
  //   right (strips that hold word W at lane lw <= 62): pixel w - 1 + k = pixel w - 1 - k, i.e. byte j of word W + d is byte
  //         2 q - 2 - 4 d - j counted from byte 0 of word W: lane lw takes it from (W - 1 | W), lane lw + 1 from (W - 1 | W) when q >= 3
  //         and from (W - 2 | W - 1) otherwise; the words to the left come by three DPP wave shifts.
  // (The per-byte path -- four byte loads for every lane of a border wave, a third of all waves -- cost a quarter of the kernel.)
  const int x4c = min(max(x4, 0), stride - 4);
  const bool fix_l = o == 0;
  const int lw = W - o + 1;           // lane of word W (>= 2: rows are at least 10 words wide)
  const bool fix_r = lw <= BLUR_WORDS + 1;  // (lw = 63: word W is this strip's right halo lane -- its pixels past the border still feed lane 62's outputs)
  const bool lane_l = lane == 0, lane_r = lane == lw || lane == lw + 1, lane_w = lane == lw;
  const bool q_hi = q >= 3;
  uint32_t sel_r;
  {
    const uint32_t sel_a = q == 1 ? 0x01020304u : q == 2 ? 0x03040504u : q == 3 ? 0x05060504u : 0x07060504u;
    const uint32_t sel_b = (q & 1) ? 0x01020304u : 0x03040506u;
    sel_r = lane_w ? sel_a : sel_b;
  }
  // taps are 8.8 fixed-point fractions (< 256): masking tells the compiler that 24-bit multiplies suffice
  const uint32_t t0 = taps.t[0] & 255u, t1 = taps.t[1] & 255u, t2 = taps.t[2] & 255u, t3 = taps.t[3] & 255u, t4 = taps.t[4] & 255u,
                 t5 = taps.t[5] & 255u, t6 = taps.t[6] & 255u;
  const uint32_t T03 = t0 | (t1 << 8) | (t2 << 16) | (t3 << 24), T46 = t4 | (t5 << 8) | (t6 << 16);
  constexpr bool no_sat = !SAT;

  uint32_t win[7][4];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) win[i][j] = 0;
#if BLUR_DOT2
  // column pass on PAIRS of rows: slot r % 7 holds (row sum r-1 | row sum r << 16) -- the row sums fit 16 bits (ufixedpoint16) -- and
  // v_dot2_u32_u16 takes two taps per instruction: 1 pack + 4 dot2 per output pixel instead of 7 v_mad_u32_u24
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  uint32_t hprev[4] = {0u, 0u, 0u, 0u};
  const us2 T01 = {(unsigned short)t0, (unsigned short)t1}, T23 = {(unsigned short)t2, (unsigned short)t3},
            T45 = {(unsigned short)t4, (unsigned short)t5}, T6 = {(unsigned short)0, (unsigned short)t6};
#endif

  const int n_out = min(BLUR_ROWS, h - y0);
  const int n_in = n_out + 6;
  uint32_t out_off = mul24u(y0, stride) + (uint32_t)x4;  // byte offset of this lane's word in the output row being produced
  // (loading a group of seven rows, working through it and only then requesting the next group left one exposed memory round trip per
  //  group -- six per wave, and the kernel's waves spent two thirds of their time at s_waitcnt: hence the rotating prefetch below)
  auto load_row = [&](int r) __attribute__((always_inline)) -> uint32_t {
    // (the row is wave-uniform: its byte offset is one scalar multiply, the address the plane's scalar base + a 32-bit lane offset -- left
    //  to the compiler it was a 64-bit vector multiply-add and a 32-bit vector multiply per row, both quarter rate)
    const int gy = reflect101(y0 + r - 3, h);
    const uint32_t roff = (uint32_t)__builtin_amdgcn_readfirstlane(gy * stride);
    return *(const uint32_t*)(P + (roff + (uint32_t)x4c));
  };
  // One input row: m is this lane's word of it, u = r % 7 its slot in the window (a compile-time constant at every call).
  auto row_step = [&](const int r, const int u, uint32_t m) __attribute__((always_inline)) {
    if (fix_l) {  // wave-uniform
      const uint32_t n1 = wave_shl1(m);
      m = lane_l ? __builtin_amdgcn_perm(n1, n1, 0x01020300u) : m;
    }
    if (fix_r) {  // wave-uniform
      const uint32_t a = wave_shr1(m), b = wave_shr1(a), c = wave_shr1(b);
      const uint32_t hi = lane_w ? m : (q_hi ? a : b), lo = lane_w ? a : (q_hi ? b : c);
      m = lane_r ? __builtin_amdgcn_perm(hi, lo, sel_r) : m;
    }
    const uint32_t lw = wave_shr1(m), rw = wave_shl1(m);
    // bytes B[0..11] = lw|m|rw; output j needs px[j-3..j+3] = B[1+j .. 7+j]: two unaligned 4-byte windows per
    // output (v_alignbyte) fed to two v_dot4_u32_u8 against the packed taps {t0..t3} and {t4..t6,0}
#if BLUR_DOT2
    uint32_t hh[4];
#else
    uint32_t* hh = win[u];  // window slot (r % 7) == u because r0 is a multiple of 7
#endif
    hh[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(m, lw, 1), T03, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(rw, m, 1), T46, 0u, false), false);
    hh[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(m, lw, 2), T03, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(rw, m, 2), T46, 0u, false), false);
    hh[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(m, lw, 3), T03, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(rw, m, 3), T46, 0u, false), false);
    hh[3] = __builtin_amdgcn_udot4(m, T03, __builtin_amdgcn_udot4(rw, T46, 0u, false), false);
    if (!no_sat) {  // wave-uniform
#pragma unroll
      for (int j = 0; j < 4; ++j) hh[j] = min(hh[j], 65535u);  // ufixedpoint16 saturation (only reachable with variant-1 taps)
    }
#if BLUR_DOT2
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      win[u][j] = hprev[j] | (hh[j] << 16);
      hprev[j] = hh[j];
    }
#endif
    if (r >= 6) {
      // rows r-6 .. r live in slots (u+1)%7 .. u ; tap k multiplies row r-6+k
      uint32_t acc4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#if BLUR_DOT2
        // pairs (r-6, r-5), (r-4, r-3), (r-2, r-1) and (r-1, r) with tap 6 on its upper half
        uint32_t acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, win[(u + 2) % 7][j]), T01, 0x8000u, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, win[(u + 4) % 7][j]), T23, acc, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, win[(u + 6) % 7][j]), T45, acc, false);
        acc4[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, win[u][j]), T6, acc, false);
#else
        // 8-bit tap x 16-bit row sum: v_mad_u32_u24 is exact here (hipcc would pick the quarter-rate v_mul_lo_u32)
        uint32_t acc = mad24(t0, win[(u + 1) % 7][j], 0x8000u);
        acc = mad24(t1, win[(u + 2) % 7][j], acc);
        acc = mad24(t2, win[(u + 3) % 7][j], acc);
        acc = mad24(t3, win[(u + 4) % 7][j], acc);
        acc = mad24(t4, win[(u + 5) % 7][j], acc);
        acc = mad24(t5, win[(u + 6) % 7][j], acc);
        acc4[j] = mad24(t6, win[u][j], acc);
#endif
      }
      uint32_t o;
      if (no_sat) {  // wave-uniform: byte 2 of each accumulator is the pixel, three v_perm_b32 pack them
        const uint32_t lo = __builtin_amdgcn_perm(acc4[1], acc4[0], 0x0c0c0602u);  // (0, 0, acc1.b2, acc0.b2)
        const uint32_t hi = __builtin_amdgcn_perm(acc4[3], acc4[2], 0x06020c0cu);  // (acc3.b2, acc2.b2, 0, 0)
        o = lo | hi;
      } else {
        o = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) o |= min(acc4[j] >> 16, 255u) << (8 * j);
      }
      if (writer) *(uint32_t*)(D + out_off) = o;
      out_off += (uint32_t)stride;
    }
  };
  // Seven row loads in flight per lane: the register of a row is refilled with the row seven further down before the row is worked on.
  // A wave with all BLUR_ROWS output rows (most of them) runs the rows as ONE straight-line block -- every row index, window slot and
  // prefetch decision a compile-time constant: with a branch per row the compiler's wait counts and register copies at the joins changed
  // with every edit (a full-window copy per row, or a wait for every outstanding load after each row); the rest take the loop.
  uint32_t mrow[7];
#pragma unroll
  for (int u = 0; u < 7; ++u) mrow[u] = load_row(u);  // (n_in >= 7)
  if (n_out == BLUR_ROWS) {  // wave-uniform
#pragma unroll
    for (int r = 0; r < BLUR_ROWS + 6; ++r) {
      const uint32_t m = mrow[r % 7];
      if (r + 7 < BLUR_ROWS + 6) mrow[r % 7] = load_row(r + 7);
      row_step(r, r % 7, m);
    }
    return;
  }
  for (int r0 = 0; r0 < n_in; r0 += 7) {
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int r = r0 + u;
      if (r < n_in) {  // wave-uniform
        const uint32_t m = mrow[u];
        if (r + 7 < n_in) mrow[u] = load_row(r + 7);  // wave-uniform
        row_step(r, u, m);
      }
    }
  }
}

}  // namespace orbfe
