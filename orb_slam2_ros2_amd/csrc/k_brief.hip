// k_brief.hip -- intensity-centroid orientation + rotated BRIEF-256 + keypoint assembly: three kernels
// (moments: one wave per keypoint; orientation: one lane per keypoint; descriptor: one wave per keypoint).
//
// Replaces ORBExtractor::computeBRIEF (src/ORB_SLAM2/src/ORBExtractor.cc:397-456), getGrayCentroid
// (:465-487), rotateTemplate (:534-540) and the level concatenation of ORBExtractor::extract (:499-508).
//
//  * IC moments: the 749-pixel disc (umax table of initMaxU, :217-236) is summed in int32 across the 64
//    lanes (two patch rows per step) on the UN-blurred plane, then wave-reduced with DPP.
//  * theta = atan2(m01, m10), cos/sin: fp64, evaluated with the shared deterministic routines of
//    orb_math.h (bit-identical on host and device; <= 1 ulp from libm) -- ONE LANE per keypoint (a block
//    owns 16 keypoints), not one wave per keypoint: the ~400 fp64 instructions are the expensive part.
//  * 256 tests: lane l evaluates pairs l, l+64, l+128, l+192; the rotated offsets are rounded exactly as
//    the reference does (double product -> float, float add, round-half-even); one __ballot per group of
//    64 tests yields 8 descriptor bytes already in the reference's LSB-first order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#include "orb_math.h"
#include "orbfe_internal.h"
#ifdef BRIEF_STAMPS
namespace orbfe { extern __device__ unsigned long long g_rt_t[4][8]; }
#define RT_STAMP(k) { __syncthreads(); if (tid == 0 && blockIdx.y == 0) orbfe::g_rt_t[k][0] = __builtin_amdgcn_s_memrealtime(); }
#define RTP_STAMP(k) { __syncthreads(); if (tid == 0 && blockIdx.y == 0) orbfe::g_rt_t[k][r] = __builtin_amdgcn_s_memrealtime(); }
#endif
#include "rowtable_body.h"
#include "wave_ops.h"

namespace orbfe {

// umax table (initMaxU, ORBExtractor.cc:217-236) packed 4 bits per row: half-width of the IC disc at |dy|
typedef unsigned long long UmaxPacked;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
// full-wave integer sum (all 64 lanes active), result in every lane
__device__ __forceinline__ int wave_sum_i(int v) {
  v += dpp_i32<0x111, 0xf>(v);  // row_shr:1
  v += dpp_i32<0x112, 0xf>(v);  // row_shr:2
  v += dpp_i32<0x114, 0xf>(v);  // row_shr:4
  v += dpp_i32<0x118, 0xf>(v);  // row_shr:8
  v += dpp_i32<0x142, 0xa>(v);  // row_bcast:15
  v += dpp_i32<0x143, 0xc>(v);  // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

// output index k of an image -> (level, index inside the level): level-major concatenation (ORBExtractor.cc:501-506)
__device__ __forceinline__ int locate_level(const int32_t* __restrict__ sc, int n_levels, int k, int* j_out, int* total_out) {
  int level = -1, acc = 0, j = 0;
  for (int l = 0; l < n_levels; ++l) {
    const int c = sc[l];
    if (level < 0 && k < acc + c) {
      level = l;
      j = k - acc;
    }
    acc += c;
  }
  *j_out = j;
  *total_out = acc;
  return level;
}

// ---------------------------------------------------------------------------------------------
// 0. level-major keypoint list: one lane per output index.  Every later kernel starts from ONE load of this list
//    instead of the count-prefix / level-table / selection-record chain of dependent loads.
//    entry = {x | y << 16 (level coordinates), level | response << 8}; x == 0xFFFF marks "no keypoint".
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kplist(const LevelDev* __restrict__ lv, int n_levels, const uint32_t* __restrict__ sel,
                                                const int32_t* __restrict__ sel_count, int n_features, uint4* __restrict__ kpl) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int img = blockIdx.y;
  if (k >= n_features) return;
  int j, total;
  const int level = locate_level(sel_count + (size_t)img * n_levels, n_levels, k, &j, &total);
  uint4 e = make_uint4(0xFFFFu, 0u, 0u, 0u);
  if (level >= 0) {
    const LevelDev& L = lv[level];
    const uint32_t rec = sel[(size_t)img * n_features + L.quota_off + j];
    e.x = (ORBFE_REC_X(rec) + ORBFE_EDGE) | ((ORBFE_REC_Y(rec) + ORBFE_EDGE) << 16);
    e.y = (uint32_t)level | (ORBFE_REC_R(rec) << 8);
    e.z = L.plane_off;  // the consumers address the plane straight from the entry: no dependent level-table lookup
    e.w = (uint32_t)L.stride;
  }
  kpl[(size_t)img * n_features + k] = e;
}

// ---------------------------------------------------------------------------------------------
// 1. intensity-centroid moments: FOUR keypoints per wave, one DPP row of 16 lanes each.  The 31x31 disc of the UN-blurred plane is
//    read as 32 rows x 8 words that start AT x - 15 (byte-aligned dword loads: gfx950 takes them), so what a lane does with its
//    16 words no longer depends on the keypoint: lane `sub` of a row owns word sub & 7 of the rows (sub >> 3) + 2 it, it = 0..15
//    -- one address, advanced by two image rows per load -- and its column offset dx = 4 (sub & 7) - 15 is a constant.  The disc
//    mask of a word depends on (it, sub) only: the 256 masks are computed once per workgroup (one per thread) and read back from LDS.
//    Per word that leaves one AND and three dot products that accumulate in place:
//        S += sum of the bytes,  W += sum of byte position x byte,  T += it x sum of the bytes
//    m10 = sum dx v = dx S + W;   m01 = sum dy v = (sub >> 3 - 15) S + 2 T       (then summed over the 16 lanes)
//    (Before: aligned words, 9 per row, and ~35 instructions per word to rebuild row, column, disc width and mask from the flat
//    index for every keypoint: 669 vector instructions per wave, now ~170.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ic_moments(const uint8_t* __restrict__ pyr, size_t img_pitch, const uint4* __restrict__ kpl,
                                                    int n_features, UmaxPacked umax, int2* __restrict__ moments) {
  __shared__ uint32_t s_mask[256];
  {
    const int it = threadIdx.x >> 4, sb = threadIdx.x & 15;
    const int dy = (sb >> 3) + 2 * it - 15;
    const int ady = dy < 0 ? -dy : dy;
    uint32_t mask = 0u;
    if (ady <= 15) {
      const int d = (int)((umax >> (4 * ady)) & 15ull);
      const int dx0 = 4 * (sb & 7) - 15;
      // bytes lo..hi of the word lie inside the disc row |dx| <= d
      const int lo = max(0, -d - dx0), hi = min(3, d - dx0);
      mask = (hi >= lo) ? ((0xFFFFFFFFu >> (8 * (3 - hi))) & (0xFFFFFFFFu << (8 * lo))) : 0u;
    }
    s_mask[threadIdx.x] = mask;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int k = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const int img = blockIdx.y;
  const bool in_range = k < n_features;
  const uint4 e = in_range ? kpl[(size_t)img * n_features + k] : make_uint4(0xFFFFu, 0u, 0u, 0u);
  const bool valid = (e.x & 0xFFFFu) != 0xFFFFu;
  const int x = valid ? (int)(e.x & 0xFFFFu) : 16, y = valid ? (int)(e.x >> 16) : 16;  // level coordinates (a safe spot if unused)
  const uint8_t* I = pyr + (size_t)img * img_pitch;  // wave-uniform base; the rest of the address is a 32-bit offset
  const uint32_t plane = valid ? e.z : 0u;
  const int stride = valid ? (int)e.w : 64;
  constexpr int NIT = 16;
  const uint32_t a0 = plane + (uint32_t)mad24u(y - 15 + (sub >> 3), stride, x - 15 + 4 * (sub & 7));  // rows and strides < 2^13
  const uint32_t step = 2u * (uint32_t)stride;
  uint32_t wv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {  // (the last load of the upper half-row is row y + 16: masked out, inside the image -- the border is 19)
    uint32_t w;
    __builtin_memcpy(&w, I + (a0 + (uint32_t)it * step), 4);
    wv[it] = w;
  }
  uint32_t S = 0u, Wp = 0u, T = 0u;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const uint32_t w = wv[it] & s_mask[it * 16 + sub];
    S = __builtin_amdgcn_udot4(w, 0x01010101u, S, false);
    Wp = __builtin_amdgcn_udot4(w, 0x03020100u, Wp, false);
    T = __builtin_amdgcn_udot4(w, 0x01010101u * (uint32_t)it, T, false);
  }
  // (S < 2^18, |dx|, |dy| <= 15: plain products, the compiler sees the ranges and takes the 24-bit multiplier; its own instructions
  //  also get the wait states a dot result needs before another instruction may read it -- an asm statement would not: wave_ops.h)
  const int S18 = (int)(S & 0x3FFFFu);
  int m10 = (4 * (sub & 7) - 15) * S18 + (int)Wp;
  int m01 = ((sub >> 3) - 15) * S18 + 2 * (int)T;
  // sum over the 16 lanes of the row: lane 15 of each row ends up with the total
#pragma unroll
  for (int sh = 0; sh < 4; ++sh) {
    m10 += (sh == 0) ? dpp_i32<0x111, 0xf>(m10) : (sh == 1) ? dpp_i32<0x112, 0xf>(m10) : (sh == 2) ? dpp_i32<0x114, 0xf>(m10) : dpp_i32<0x118, 0xf>(m10);
    m01 += (sh == 0) ? dpp_i32<0x111, 0xf>(m01) : (sh == 1) ? dpp_i32<0x112, 0xf>(m01) : (sh == 2) ? dpp_i32<0x114, 0xf>(m01) : dpp_i32<0x118, 0xf>(m01);
  }
  if (sub == 15 && valid) moments[(size_t)img * n_features + k] = make_int2(m10, m01);
}

// the stereo matcher's view of a keypoint (KpX): getPitch's patch centre is computed HERE, once, with the reference's own float operations
__device__ __forceinline__ KpX kpx_of(const orbfe_keypoint& kp, float sf) {
#pragma clang fp contract(off)
  const float fxq = kp.x / sf, fyq = kp.y / sf;
  int qx = (int)fxq, qy = (int)fyq;  // cvFloor
  qx -= (qx > fxq);
  qy -= (qy > fyq);
  KpX r;
  r.x = kp.x;
  r.q = ORBFE_KPX_Q(kp.octave, qx, qy);
  return r;
}

// ---------------------------------------------------------------------------------------------
// 2. orientation: ONE LANE per keypoint.  theta = atan2(m01, m10) and cos/sin in fp64 with the shared
//    deterministic routines; assembles the cv::KeyPoint record and the stereo row band.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_orient(const LevelDev* __restrict__ lv, const uint4* __restrict__ kpl, int n_features,
                                                const int2* __restrict__ moments, double2* __restrict__ sincos,
                                                orbfe_keypoint* __restrict__ kps, KpAux* __restrict__ aux, KpX* __restrict__ kx,
                                                double* __restrict__ theta_out, int rows0, const int32_t* __restrict__ sel_count, int n_levels,
                                                int32_t* __restrict__ n_kp, orbfe_keypoint* __restrict__ kps_host, int32_t* __restrict__ n_kp_host) {
  // kps_host / n_kp_host (nullable): page-locked HOST memory, same [image][n_features] layout -- the host-pointer path of a frame or two
  // lets the kernels deliver the results themselves (posted writes over PCIe) instead of queueing three copies behind the last kernel
#pragma clang fp contract(off)
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int img = blockIdx.y;
  if (k == 0) {  // the image's keypoint count is published HERE, with the arrays it counts: the pipelined stereo match of the previous
                 // batch reads count, keypoints and descriptors of the same slots, and only this kernel and k_brief wait for it
    int total = 0;
    for (int l = 0; l < n_levels; ++l) total += sel_count[(size_t)img * n_levels + l];
    n_kp[img] = total;
    if (n_kp_host) n_kp_host[img] = total;
  }
  const uint4 e = (k < n_features) ? kpl[(size_t)img * n_features + k] : make_uint4(0xFFFFu, 0u, 0u, 0u);
  if ((e.x & 0xFFFFu) != 0xFFFFu) {
    const int level = (int)(e.y & 0xFFu);
    const LevelDev& L = lv[level];
    const int x = (int)(e.x & 0xFFFFu), y = (int)(e.x >> 16);
    const size_t o = (size_t)img * n_features + k;
    const int2 m = moments[o];
    const double theta = orbmath::det_atan2((double)m.y, (double)m.x);
    double sn, cs;
    orbmath::det_sincos(theta, &sn, &cs);
    sincos[o] = make_double2(sn, cs);
    orbfe_keypoint kp;
    kp.x = (float)x * L.sf;  // keypoint.pt *= scale[octave] (ORBExtractor.cc:408-409)
    kp.y = (float)y * L.sf;
    kp.size = 7.0f;
    kp.angle = (float)(theta / 3.14159265358979323846 * 180);  // ORBExtractor.cc:407
    kp.response = (float)((e.y >> 8) & 0xFFu);
    kp.octave = level;
    kp.class_id = -1;
    kps[o] = kp;
  #ifndef EXP_NO_KPS_HOST
  if (kps_host) kps_host[o] = kp;
#endif
    kx[o] = kpx_of(kp, L.sf);
    // createRowIndexDB band (ORBMatcher.cc:924-927), stored with the keypoint for the stereo matcher
    const float r = (float)(2.0 * (double)L.sf);
    const unsigned row = (unsigned)__float2int_rn(kp.y);
    KpAux a;
    a.row_max = (int16_t)min(rows0, __float2int_rn((float)row + r + 1.0f));
    a.row_min = (int16_t)max(0, __float2int_rn((float)row - r));
    aux[o] = a;
    if (theta_out) theta_out[o] = theta;
  }
}

// ---------------------------------------------------------------------------------------------
// 0 + 1 + 2 in ONE launch, for a frame or two (the host-pointer / slot paths): the three kernels above take ~5 us each there -- the
//    floor of a dependent launch, not their work -- so the 16 lanes that sum a keypoint's moments first build its list entry themselves
//    (the same loads, redundantly) and lane 15 of the row, which ends up with the sums, goes on to the orientation.  Same arithmetic,
//    same results; a batch keeps the separate kernels (one lane per keypoint is the better shape for the fp64 trigonometry at scale).
// ---------------------------------------------------------------------------------------------
template <bool WITH_ORIENT>
__global__ __launch_bounds__(256) void k_list_moments_orient(const LevelDev* __restrict__ lv, int n_levels, const uint32_t* __restrict__ sel,
                                                             const int32_t* __restrict__ sel_count, int n_features,
                                                             const uint8_t* __restrict__ pyr, size_t img_pitch, UmaxPacked umax,
                                                             uint4* __restrict__ kpl, int2* __restrict__ moments, double2* __restrict__ sincos,
                                                             orbfe_keypoint* __restrict__ kps, KpAux* __restrict__ aux, KpX* __restrict__ kx,
                                                             double* __restrict__ theta_out, int rows0, int32_t* __restrict__ n_kp,
                                                             orbfe_keypoint* __restrict__ kps_host, int32_t* __restrict__ n_kp_host,
                                                             int32_t* __restrict__ rt_flags) {
#pragma clang fp contract(off)
  __shared__ uint32_t s_mask[256];
  // (rt_flags, nullable: the eight part totals of the row table the descriptor launch behind this one builds -- rowtable_build_part)
  if (rt_flags && blockIdx.x == 0 && threadIdx.x < 8) rt_flags[blockIdx.y * 8 + threadIdx.x] = 0;
  {
    const int it = threadIdx.x >> 4, sb = threadIdx.x & 15;
    const int dy = (sb >> 3) + 2 * it - 15;
    const int ady = dy < 0 ? -dy : dy;
    uint32_t mask = 0u;
    if (ady <= 15) {
      const int d = (int)((umax >> (4 * ady)) & 15ull);
      const int dx0 = 4 * (sb & 7) - 15;
      const int lo = max(0, -d - dx0), hi = min(3, d - dx0);
      mask = (hi >= lo) ? ((0xFFFFFFFFu >> (8 * (3 - hi))) & (0xFFFFFFFFu << (8 * lo))) : 0u;
    }
    s_mask[threadIdx.x] = mask;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int k = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const int img = blockIdx.y;
  const bool in_range = k < n_features;
  // the list entry (k_kplist)
  uint4 e = make_uint4(0xFFFFu, 0u, 0u, 0u);
  int level = -1, total = 0;
  {
    int j;
    level = locate_level(sel_count + (size_t)img * n_levels, n_levels, in_range ? k : 0x7FFFFFFF, &j, &total);
    if (level >= 0) {
      const LevelDev& L = lv[level];
      const uint32_t rec = sel[(size_t)img * n_features + L.quota_off + j];
      e.x = (ORBFE_REC_X(rec) + ORBFE_EDGE) | ((ORBFE_REC_Y(rec) + ORBFE_EDGE) << 16);
      e.y = (uint32_t)level | (ORBFE_REC_R(rec) << 8);
      e.z = L.plane_off;
      e.w = (uint32_t)L.stride;
    }
  }
  if (WITH_ORIENT && k == 0 && sub == 15) {
    n_kp[img] = total;
    if (n_kp_host) n_kp_host[img] = total;
  }
  // the moments (k_ic_moments)
  const bool valid = (e.x & 0xFFFFu) != 0xFFFFu;
  const int x = valid ? (int)(e.x & 0xFFFFu) : 16, y = valid ? (int)(e.x >> 16) : 16;
  const uint8_t* I = pyr + (size_t)img * img_pitch;
  const uint32_t plane = valid ? e.z : 0u;
  const int stride = valid ? (int)e.w : 64;
  constexpr int NIT = 16;
  const uint32_t a0 = plane + (uint32_t)mad24u(y - 15 + (sub >> 3), stride, x - 15 + 4 * (sub & 7));
  const uint32_t step = 2u * (uint32_t)stride;
  uint32_t wv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    uint32_t w;
    __builtin_memcpy(&w, I + (a0 + (uint32_t)it * step), 4);
    wv[it] = w;
  }
  uint32_t S = 0u, Wp = 0u, T = 0u;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const uint32_t w = wv[it] & s_mask[it * 16 + sub];
    S = __builtin_amdgcn_udot4(w, 0x01010101u, S, false);
    Wp = __builtin_amdgcn_udot4(w, 0x03020100u, Wp, false);
    T = __builtin_amdgcn_udot4(w, 0x01010101u * (uint32_t)it, T, false);
  }
  const int S18 = (int)(S & 0x3FFFFu);
  int m10 = (4 * (sub & 7) - 15) * S18 + (int)Wp;
  int m01 = ((sub >> 3) - 15) * S18 + 2 * (int)T;
#pragma unroll
  for (int sh = 0; sh < 4; ++sh) {
    m10 += (sh == 0) ? dpp_i32<0x111, 0xf>(m10) : (sh == 1) ? dpp_i32<0x112, 0xf>(m10) : (sh == 2) ? dpp_i32<0x114, 0xf>(m10) : dpp_i32<0x118, 0xf>(m10);
    m01 += (sh == 0) ? dpp_i32<0x111, 0xf>(m01) : (sh == 1) ? dpp_i32<0x112, 0xf>(m01) : (sh == 2) ? dpp_i32<0x114, 0xf>(m01) : dpp_i32<0x118, 0xf>(m01);
  }
  if (sub != 15 || !in_range) return;
  const size_t o = (size_t)img * n_features + k;
  kpl[o] = e;
  if (!valid) return;
  moments[o] = make_int2(m10, m01);
  if (!WITH_ORIENT) return;  // (a batch: k_orient follows, one lane per keypoint)
  // the orientation (k_orient)
  const LevelDev& L = lv[level];
  const double theta = orbmath::det_atan2((double)m01, (double)m10);
  double sn, cs;
  orbmath::det_sincos(theta, &sn, &cs);
  sincos[o] = make_double2(sn, cs);
  orbfe_keypoint kp;
  kp.x = (float)x * L.sf;
  kp.y = (float)y * L.sf;
  kp.size = 7.0f;
  kp.angle = (float)(theta / 3.14159265358979323846 * 180);
  kp.response = (float)((e.y >> 8) & 0xFFu);
  kp.octave = level;
  kp.class_id = -1;
  kps[o] = kp;
#ifndef EXP_NO_KPS_HOST
  if (kps_host) kps_host[o] = kp;
#endif
  kx[o] = kpx_of(kp, L.sf);
  const float r = (float)(2.0 * (double)L.sf);
  const unsigned row = (unsigned)__float2int_rn(kp.y);
  KpAux a;
  a.row_max = (int16_t)min(rows0, __float2int_rn((float)row + r + 1.0f));
  a.row_min = (int16_t)max(0, __float2int_rn((float)row - r));
  aux[o] = a;
  if (theta_out) theta_out[o] = theta;
}

// ---------------------------------------------------------------------------------------------
// 3. rotated BRIEF, one wave per keypoint.  The 37x37 window of the BLURRED plane that the rotated template
//    can reach (|offset| <= 18) is staged in LDS with coalesced word loads; the 512 data-dependent byte reads
//    then hit LDS instead of issuing 8 global gathers with ~64 distinct cache lines each.
// ---------------------------------------------------------------------------------------------
#define BRIEF_R 18
#define BRIEF_ROWS (2 * BRIEF_R + 1)
#define BRIEF_WORDS 11  // (3 + 37 + 3) / 4 rounded up

#ifndef BRIEF_WAVES
#define BRIEF_WAVES 4
#endif
#ifndef BRIEF_KPW
#define BRIEF_KPW 2  // consecutive keypoints per wave (measured: 2 and 3 equal, 4 and 8 slower -- fewer, longer waves balance worse)
#endif
#ifdef BRIEF_STAMPS  // diagnostic build only (tools/exp/brief_stamps.sh): start / end of every descriptor wave and of the row-table workgroups, 100 MHz ticks
__device__ unsigned long long g_rt_t[4][8];
__device__ unsigned long long g_bs_rec[8192][6];  // [wave]: start, first window parked, end, kind (1 descriptor wave, 2 row-table workgroup), list entry arrived, window arrived
#define BS_DECL const unsigned bs_w = ((blockIdx.y * 520u + blockIdx.x) * BRIEF_WAVES + (threadIdx.x >> 6)) & 8191u; const unsigned long long bs_t0 = __builtin_amdgcn_s_memrealtime(); unsigned long long bs_t1 = 0, bs_ta = 0, bs_tb = 0;
#define BS_MID if (!bs_t1) bs_t1 = __builtin_amdgcn_s_memrealtime();
#define BS_END(kind) if ((threadIdx.x & 63) == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g_bs_rec[bs_w][0] = bs_t0, g_bs_rec[bs_w][1] = bs_t1, g_bs_rec[bs_w][2] = __builtin_amdgcn_s_memrealtime(), g_bs_rec[bs_w][3] = kind, g_bs_rec[bs_w][4] = bs_ta, g_bs_rec[bs_w][5] = bs_tb; }
#else
#define BS_DECL
#define BS_MID
#define BS_END(kind)
#endif
struct BriefRowTable {  // rowoff == nullptr: descriptor blocks only
  const KpAux* aux;      // of the launch's first image
  const int32_t* n_kp;   // of the launch's first image
  uint32_t* rowoff;      // per-slot tables of the context (orbfe_api.hip: spec tables)
  uint16_t* rowlist;
  int32_t* n_match;      // per pair
  int rows, list_cap, slot0, n_brief_blocks;
  int stage_cap;  // entries of dynamic LDS the row-table workgroup may assemble its list in (0: none)
  int32_t* flags;  // != nullptr: the table by all eight spare workgroups (rowtable_build_part): their totals, [image][8], zeroed by the launch before
};
__global__ __launch_bounds__(64 * BRIEF_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_brief(const uint8_t* __restrict__ blur, size_t img_pitch,
                                              const uint4* __restrict__ kpl, int n_features, const int8_t* __restrict__ pattern,
                                              const double2* __restrict__ sincos, uint8_t* __restrict__ desc, uint8_t* __restrict__ desc_host,
                                              BriefRowTable rt) {
#pragma clang fp contract(off)
  // per wave (the waves never synchronise): the window, and v cos / v sin for every template coordinate v in [-18, 18] as two
  // arrays of doubles.  The kernel is bound by LDS cycles, most of them the table look-ups: as 16-byte (cos, sin) entries read by
  // ds_read_b128 a 256-byte bank sweep holds 16 entries, so the lanes' 27 distinct coordinates collided two to three deep on every
  // read; 8-byte entries read by ds_read_b64 put 32 to a sweep -- the coordinates -13 ... 13 of the standard template never collide
  // (equal addresses broadcast).  The window lies BETWEEN the two arrays so that no ds_read2_b64 can reach from one to the other:
  // that instruction takes twice the cycles of two single reads and banks 16 entries to the sweep again.
  struct __attribute__((aligned(16))) Lds {
    double rc[BRIEF_WAVES][40];
    uint32_t win[BRIEF_WAVES][BRIEF_ROWS * BRIEF_WORDS + 1];
    double rs[BRIEF_WAVES][40];
  };
  static_assert(sizeof(double) * 40 * BRIEF_WAVES + sizeof(uint32_t) * (BRIEF_ROWS * BRIEF_WORDS + 1) * BRIEF_WAVES > 255 * 8 &&
                    (sizeof(double) * 40 * BRIEF_WAVES + sizeof(uint32_t) * (BRIEF_ROWS * BRIEF_WORDS + 1) * BRIEF_WAVES) % 512 != 0,
                "rc[w] and rs[w] must not be reachable by one ds_read2(st64)_b64");
  __shared__ Lds lds;
  BS_DECL
  if (rt.rowoff && (int)blockIdx.x >= rt.n_brief_blocks) {
    // A frame or two: the LAST eight workgroups of a row are not descriptor blocks; the first of them builds the row table of the image
    // (ORBMatcher::createRowIndexDB, ORBMatcher.cc:915-932: it needs the row bands k_orient left, nothing of this kernel) into the
    // slot's own table and zeroes the match counter of the slot's pair, so that an orbfe_stereo_match that follows launches k_stereo
    // alone -- 14 us of table building run beside the descriptors instead of in front of the match
    // (r6) ... all EIGHT of them now, an eighth of the rows each (rowtable_build_part): one workgroup's 14 us were the launch's long pole.
    static_assert(sizeof(Lds) >= 9000, "the table's row counters borrow the descriptor kernel's LDS (launch_orient_brief checks the row count against 9000 bytes)");
    uint32_t* cnt = (uint32_t*)&lds;
    const int slot = rt.slot0 + (int)blockIdx.y;
    const int r = (int)blockIdx.x - rt.n_brief_blocks;
    if (rt.flags) {
      if (r == 0 && threadIdx.x == 0) rt.n_match[slot >> 1] = 0;
      const int rpw = (rt.rows + 7) >> 3;
      extern __shared__ __attribute__((aligned(16))) uint2 brief_bands[];  // (dynamic LDS of the small launches: rt.stage_cap / 4 band records)
      rowtable_build_part(rt.aux + (size_t)blockIdx.y * n_features, min(rt.n_kp[blockIdx.y], n_features), rt.rows, rt.list_cap,
                          rt.rowoff + (size_t)slot * (rt.rows + 1), rt.rowlist + (size_t)slot * rt.list_cap, cnt, cnt + rpw + 12, (int)threadIdx.x, r,
                          rt.flags + (size_t)blockIdx.y * 8, rt.stage_cap > 0 ? brief_bands : nullptr, rt.stage_cap / 4);
      BS_END(2)
      return;
    }
    if (r != 0) return;
    if (threadIdx.x == 0) rt.n_match[slot >> 1] = 0;
    extern __shared__ __attribute__((aligned(16))) uint16_t brief_stage[];  // (dynamic LDS of the small launches: rt.stage_cap entries)
    rowtable_build(rt.aux + (size_t)blockIdx.y * n_features, min(rt.n_kp[blockIdx.y], n_features), rt.rows, rt.list_cap,
                   rt.rowoff + (size_t)slot * (rt.rows + 1), rt.rowlist + (size_t)slot * rt.list_cap, cnt, cnt + rt.rows, (int)threadIdx.x,
                   rt.stage_cap > 0 ? brief_stage : nullptr, rt.stage_cap);
    BS_END(2)
    return;
  }
  const int lane = threadIdx.x & 63;
  uint32_t* win = lds.win[threadIdx.x >> 6];
  double* rc = lds.rc[threadIdx.x >> 6];
  double* rs = lds.rs[threadIdx.x >> 6];
  // XCD-aware block order: workgroups go round-robin to the 8 XCDs (own L2 each) and consecutive keypoints are spatial neighbours
  // (candidate order) whose 37x37 windows overlap; block b of the grid takes keypoint block (b % 8) * (grid / 8) + b / 8, so one XCD
  // works through one contiguous eighth of the list (gridDim.x is a multiple of 8).  Fetched bytes 2.69 -> 0.85 GB per 1024 images.
  // (r6: the number of descriptor blocks is a kernel ARGUMENT in both forms.  It used to be gridDim.x for the batches -- and the compiler
  //  derived that from the dispatch packet, which lives in the queue's ring buffer in HOST memory: every wave of the launch began with a
  //  scalar load across PCIe that no cache holds.  A pair's 2000 waves queued for it -- first data after 13.6 us on average, 26.5 at worst,
  //  whatever they asked for first (stamps build, tools/exp/brief_stamps.sh) -- and the launch took 18 - 31 us.)
  const int per_xcd = rt.n_brief_blocks >> 3;
  const int kb = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  const int k0 = (kb * BRIEF_WAVES + (threadIdx.x >> 6)) * BRIEF_KPW;
  const int img = blockIdx.y;
  if (k0 >= n_features) return;
  // A wave with one keypoint spent two thirds of its life parked at s_waitcnt: list entry -> window -> tests are dependent round
  // trips, and eight waves per SIMD do not cover them.  A wave now works through BRIEF_KPW consecutive keypoints as a three-stage
  // pipeline: while keypoint i is tested, the window and sin / cos of keypoint i + 1 are in flight (parked in registers) and so is
  // the list entry of keypoint i + 2; the lane's template pairs are fetched once.
  const uint4* list = kpl + (size_t)img * n_features;
  const double2* sc = sincos + (size_t)img * n_features;
  const uint8_t* W = blur + (size_t)img * img_pitch;  // wave-uniform base; the rest of the address is a 32-bit offset
  const int k_last = min(k0 + BRIEF_KPW, n_features) - 1;
  const uint32_t tp0 = *(const uint32_t*)(pattern + (0 * 64 + lane) * 4), tp1 = *(const uint32_t*)(pattern + (1 * 64 + lane) * 4);
  const uint32_t tp2 = *(const uint32_t*)(pattern + (2 * 64 + lane) * 4), tp3 = *(const uint32_t*)(pattern + (3 * 64 + lane) * 4);
#ifdef BRIEF_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(tp0), "v"(tp1), "v"(tp2), "v"(tp3) : "memory");
  bs_t1 = __builtin_amdgcn_s_memrealtime();  // (pattern arrived; the "first window parked" stamp is not taken in this build)
#endif
  // window load: 5 rows x 11 words per pass over lanes 0..54 (lane = 11 r0 + c0), 8 passes; a pass adds a wave-uniform 5 * stride to
  // one address and parks its words at lane + 55 * pass -- one v_add per load where the flat index -> (row, word) split cost five
  constexpr int NIT = (BRIEF_ROWS + 4) / 5;
  const int ll = min(lane, 54);
  const int r0 = (ll * 373) >> 12, c0 = ll - r0 * BRIEF_WORDS;  // ll / 11 for ll < 64
  const int r_last = min(r0 + 5 * (NIT - 1), BRIEF_ROWS - 1);    // last pass: rows 35, 36 exist, the lanes of rows 37..39 re-read row 36
  // (r6: the eight window words as two VECTORS, named component by component.  As an array `uint32_t wv[NIT]` the stage records that
  //  rotate through the keypoint loop were private arrays the compiler moved to LDS, 32 bytes per thread addressed by the FLAT thread
  //  number -- for which it reads the workgroup's y and z sizes from the dispatch packet, and that packet lives in the queue's ring
  //  buffer in HOST memory: every wave of every launch of this kernel began with a scalar load across PCIe that no cache holds.  A pair's
  //  2000 waves queued for theirs: first data after 13.6 us on average, 26.5 at worst, whatever they had asked for (stamps build,
  //  tools/exp/brief_stamps.sh), a launch of 18 - 31 us.  No kernel of this library may touch blockDim / gridDim or keep a private array.)
  static_assert(NIT == 8, "Stage holds eight window words");
  struct Stage {
    uint4 wa, wb;
    double2 scv;
  };
  // requests the window and sin / cos of the keypoint in entry e (an unused slot -- x = 0xFFFF -- reads row 0 of plane 0: harmless)
  auto request = [&](const uint4 e, int k, Stage& st) __attribute__((always_inline)) {
    const bool used = (e.x & 0xFFFFu) != 0xFFFFu;
    const int x = used ? (int)(e.x & 0xFFFFu) : BRIEF_R, y = used ? (int)(e.x >> 16) : BRIEF_R;
    const uint32_t plane = used ? e.z : 0u;
    const int stride = used ? (int)e.w : 0;
    const int xa = (x - BRIEF_R) & ~3;
    const uint32_t step = 5u * (uint32_t)__builtin_amdgcn_readfirstlane(stride);
    const uint32_t a0 = plane + (uint32_t)mad24u(y - BRIEF_R + r0, stride, xa + 4 * c0);
#define BRIEF_WLD(it) (*(const uint32_t*)(W + (a0 + (uint32_t)(it) * step)))
    st.wa.x = BRIEF_WLD(0), st.wa.y = BRIEF_WLD(1), st.wa.z = BRIEF_WLD(2), st.wa.w = BRIEF_WLD(3);
    st.wb.x = BRIEF_WLD(4), st.wb.y = BRIEF_WLD(5), st.wb.z = BRIEF_WLD(6);
#undef BRIEF_WLD
    st.wb.w = *(const uint32_t*)(W + (plane + (uint32_t)mad24u(y - BRIEF_R + r_last, stride, xa + 4 * c0)));
    st.scv = sc[k];
  };
  uint4 e_cur = list[k0];
  uint4 e_nxt = list[min(k0 + 1, k_last)];
#ifdef BRIEF_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bs_ta = __builtin_amdgcn_s_memrealtime();
#endif
  Stage st_cur;
  request(e_cur, k0, st_cur);
#ifdef BRIEF_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bs_tb = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
  for (int i = 0; i < BRIEF_KPW; ++i) {
    const int k = k0 + i;
    if (k > k_last) break;  // wave-uniform
    // stage 1 / 2 for the keypoints behind this one
    const uint4 e_nn = list[min(k + 2, k_last)];
    Stage st_nxt;
    if (i + 1 < BRIEF_KPW) request(e_nxt, min(k + 1, k_last), st_nxt);
    if ((e_cur.x & 0xFFFFu) != 0xFFFFu) {  // wave-uniform
      const int x = (int)(e_cur.x & 0xFFFFu), y = (int)(e_cur.x >> 16);
      const int xa = (x - BRIEF_R) & ~3;
      if (lane < 55) {
        win[lane] = st_cur.wa.x, win[lane + 55] = st_cur.wa.y, win[lane + 110] = st_cur.wa.z, win[lane + 165] = st_cur.wa.w;
        win[lane + 220] = st_cur.wb.x, win[lane + 275] = st_cur.wb.y, win[lane + 330] = st_cur.wb.z;
        if (r0 + 5 * (NIT - 1) < BRIEF_ROWS) win[lane + 385] = st_cur.wb.w;
      }
      const double sn = st_cur.scv.x, cs = st_cur.scv.y;
      // The rotation needs x cos, x sin, y cos, y sin in fp64 for 512 template points, but the coordinates are small integers:
      // 37 lanes form the products once ((double)v * cs is exactly what the per-point expression computes), every point then
      // takes two LDS reads and one fp64 add per coordinate instead of two fp64 multiplies and a conversion.
      if (lane <= 2 * BRIEF_R) {
        const double v = (double)(float)(lane - BRIEF_R);
        rc[lane] = v * cs;
        rs[lane] = v * sn;
      }
      // the LDS accesses of one wave execute in order, so the window and the table written above are visible to every lane of this
      // wave; the fence only pins the compiler
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      BS_MID
      const float px = (float)x, py = (float)y;
      // cvRound(float sum) and the window address without a subtraction.  Adding 1.5 * 2^23 rounds a float sum to an integer with ties to
      // even -- the unit in the last place of the result is 1 -- and leaves that integer in the low mantissa bits (0x4B400000 + n).  An
      // EVEN integer folded into the constant (exact in a float below 2^24) moves n without touching the rounding, ties included (the
      // parity of the candidates is unchanged): the row constant carries minus the window's first row rounded down to even, the column
      // constant minus the window's first column (a multiple of 4) plus the LDS byte address of the wave's window, less one window row
      // where the first row is odd.  v_mad_u32_u24 then reads the row in the low 24 bits of its first operand (0x400000 + row) and adds
      // the whole second float: 44 row + column + address + 0x56400000, whose low 16 bits are the address.  Exact for |sum| < 2^21;
      // the float additions are neither contracted nor re-associated (fp contract off, no fast-math).
      typedef const __attribute__((address_space(3))) uint8_t* lds_bytes_t;
      const int y0w = y - BRIEF_R;
      const float magic_y = (float)(12582912 - (y0w & ~1));
      const float magic_x = (float)(12582912 - xa + (int)(uint32_t)(uintptr_t)(lds_bytes_t)(const uint8_t*)win - (y0w & 1) * (BRIEF_WORDS * 4));
      auto tests = [&](const uint32_t tpg) __attribute__((always_inline)) -> unsigned long long {
        const int x1 = (int)(int8_t)(tpg & 255u), y1 = (int)(int8_t)((tpg >> 8) & 255u);
        const int x2 = (int)(int8_t)((tpg >> 16) & 255u), y2 = (int)(int8_t)(tpg >> 24);
        // float * double -> double, one rounding to float (rotateTemplate, ORBExtractor.cc:537-538)
        const float p1x = (float)(rc[x1 + BRIEF_R] - rs[y1 + BRIEF_R]);  // x1 cos - y1 sin
        const float p1y = (float)(rs[x1 + BRIEF_R] + rc[y1 + BRIEF_R]);  // x1 sin + y1 cos
        const float p2x = (float)(rc[x2 + BRIEF_R] - rs[y2 + BRIEF_R]);
        const float p2y = (float)(rs[x2 + BRIEF_R] + rc[y2 + BRIEF_R]);
        const uint32_t a1 = (uint32_t)mad24u(__float_as_int((py + p1y) + magic_y), BRIEF_WORDS * 4, __float_as_int((px + p1x) + magic_x)) & 0xFFFFu;
        const uint32_t a2 = (uint32_t)mad24u(__float_as_int((py + p2y) + magic_y), BRIEF_WORDS * 4, __float_as_int((px + p2x) + magic_x)) & 0xFFFFu;
        const int v1 = *(lds_bytes_t)(uintptr_t)a1;
        const int v2 = *(lds_bytes_t)(uintptr_t)a2;
        return __ballot(v1 < v2);
      };
      // (four named words, not an array: see Stage above)
      const unsigned long long bits0 = tests(tp0), bits1 = tests(tp1), bits2 = tests(tp2), bits3 = tests(tp3);
      if (lane < 4) {
        unsigned long long* d64 = (unsigned long long*)(desc + ((size_t)img * n_features + k) * 32);
        unsigned long long b = bits0;
        if (lane == 1) b = bits1;
        if (lane == 2) b = bits2;
        if (lane == 3) b = bits3;
        d64[lane] = b;
        if (desc_host) ((unsigned long long*)(desc_host + ((size_t)img * n_features + k) * 32))[lane] = b;
      }
      // every lane has read what it needs of this window and table before the next keypoint's are parked over them
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (i + 1 < BRIEF_KPW) {
      e_cur = e_nxt;
      e_nxt = e_nn;
      st_cur = st_nxt;
    }
  }
  BS_END(1)
}
#ifdef BRIEF_STAMPS
}  // namespace orbfe
extern "C" void orbfe_debug_brief_stamps() {
  static unsigned long long rec[8192][6];
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(rec, HIP_SYMBOL(orbfe::g_bs_rec), sizeof(rec));
  unsigned long long first = ~0ull, last = 0, n = 0, sum = 0, mx = 0, mid = 0, rt0 = 0, rt1 = 0, dlast = 0, sa = 0, sb = 0, ma = 0, mb = 0, smax = 0;
  for (int w = 0; w < 8192; ++w) {
    if (!rec[w][3]) continue;
    first = first < rec[w][0] ? first : rec[w][0], last = last > rec[w][2] ? last : rec[w][2];
    if (rec[w][3] == 1) {
      ++n, sum += rec[w][2] - rec[w][0], mx = mx > rec[w][2] - rec[w][0] ? mx : rec[w][2] - rec[w][0], mid += rec[w][1] ? rec[w][1] - rec[w][0] : 0;
      dlast = dlast > rec[w][2] ? dlast : rec[w][2];
      sa += rec[w][4] - rec[w][0], sb += rec[w][5] - rec[w][4];
      ma = ma > rec[w][4] - rec[w][0] ? ma : rec[w][4] - rec[w][0], mb = mb > rec[w][5] - rec[w][4] ? mb : rec[w][5] - rec[w][4];
    }
  }
  printf("BRIEF stamps (last launch): %llu descriptor waves, launch span %.2f us, last descriptor wave ends at %.2f us; wave mean %.2f max %.2f us, pattern words arrive after %.2f us (mean)\n",
         n, (last - first) * 0.01, (dlast - first) * 0.01, n ? sum * 0.01 / n : 0.0, mx * 0.01, n ? mid * 0.01 / n : 0.0);
  for (int w = 0; w < 8192; ++w)
    if (rec[w][3] == 1) smax = smax > rec[w][0] - first ? smax : rec[w][0] - first;
  printf("  pattern + list entries arrive after %.2f us (mean, max %.2f); window + sin/cos %.2f us later (mean, max %.2f); last descriptor wave starts at %.2f us\n",
         n ? sa * 0.01 / n : 0.0, ma * 0.01, n ? sb * 0.01 / n : 0.0, mb * 0.01, smax * 0.01);
  for (int w = 0; w < 8192; ++w)
    if (rec[w][3] == 2 && (w & 3) == 0) printf("  row-table workgroup: starts %.2f us, ends %.2f us after the launch's first wave\n", (rec[w][0] - first) * 0.01, (rec[w][2] - first) * 0.01);
  {
    unsigned long long rt[4][8];
    (void)hipMemcpyFromSymbol(rt, HIP_SYMBOL(orbfe::g_rt_t), sizeof(rt));
    for (int r = 0; r < 8; ++r)
      printf("  row table part %d (image 0): zeroed %.2f | counted %.2f | lower parts known %.2f | scattered %.2f us after the launch's first wave\n", r,
             (rt[0][r] - first) * 0.01, (rt[1][r] - first) * 0.01, (rt[2][r] - first) * 0.01, (rt[3][r] - first) * 0.01);
  }
  memset(rec, 0, sizeof(rec));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(orbfe::g_bs_rec), rec, sizeof(rec));
  (void)rt0, (void)rt1;
}
namespace orbfe {
#endif

void launch_orient_brief(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint8_t* d_pyr, const uint8_t* d_blur,
                         size_t img_pitch, const uint32_t* d_sel, const int32_t* d_sel_count, int n_features,
                         const int8_t* d_pattern, const int umax[16], orbfe_keypoint* d_kps, uint8_t* d_desc, KpAux* d_aux,
                         int32_t* d_n_kp, double* d_theta, int2* d_moments, double2* d_sincos, KpX* d_kx,
                         uint4* d_kpl, int rows0, int n_img, hipEvent_t before_brief, hipEvent_t before_lists, orbfe_keypoint* h_kps,
                         uint8_t* h_desc, int32_t* h_n_kp, bool fuse_small, uint32_t* d_rowoff_slot, uint16_t* d_rowlist_slot, int32_t* d_n_match,
                         int rt_rows, int rt_list_cap, int rt_slot0, int32_t* d_rt_flags) {
  // d_rowoff_slot != nullptr (a frame or two): the launch of the descriptors also builds the row table of every image into the slot's table
  if (n_img <= 0 || n_features <= 0) return;
  UmaxPacked u = 0;
  for (int i = 0; i < 16; ++i) u |= (unsigned long long)(umax[i] & 15) << (4 * i);
  if (fuse_small && n_img <= 2) {  // list + moments + orientation in one launch (a frame or two: three launch floors of ~5 us become one)
    if (before_lists) (void)hipStreamWaitEvent(s, before_lists, 0);
    hipLaunchKernelGGL(k_list_moments_orient<true>, dim3((n_features + 15) / 16, n_img), dim3(256), 0, s, d_lv, n_levels, d_sel, d_sel_count, n_features,
                       d_pyr, img_pitch, u, d_kpl, d_moments, d_sincos, d_kps, d_aux, d_kx, d_theta, rows0, d_n_kp, h_kps, h_n_kp,
                       (d_rowoff_slot && d_rt_flags) ? d_rt_flags + (size_t)rt_slot0 * 8 : (int32_t*)nullptr);
    if (before_brief) (void)hipStreamWaitEvent(s, before_brief, 0);
    const int nb = (((n_features + BRIEF_WAVES * BRIEF_KPW - 1) / (BRIEF_WAVES * BRIEF_KPW)) + 7) & ~7;
    BriefRowTable rt{};
    rt.n_brief_blocks = nb;
    if (d_rowoff_slot && (size_t)(rt_rows + 4) * 4 <= 9000)
      rt = BriefRowTable{d_aux, d_n_kp, d_rowoff_slot, d_rowlist_slot, d_n_match, rt_rows, rt_list_cap, rt_slot0, nb, 0,
                         d_rt_flags ? d_rt_flags + (size_t)rt_slot0 * 8 : (int32_t*)nullptr};
    // the row-table workgroup's list staging: 40 KB of dynamic LDS (20 k entries: ten rows a keypoint at 2000 features) where the list's
    // alignment allows 16-byte copies; two descriptor workgroups per CU still fit beside it
    if (rt.rowoff && (rt.flags || rt_list_cap % 8 == 0)) rt.stage_cap = rt.flags ? 4 * std::min(n_features, 4096) : std::min(rt_list_cap, 20480);  // (parts: a band record, 8 bytes, per keypoint)
    hipLaunchKernelGGL(k_brief, dim3(nb + (rt.rowoff ? 8 : 0), n_img), dim3(64 * BRIEF_WAVES), (size_t)rt.stage_cap * 2, s, d_blur, img_pitch, d_kpl, n_features,
                       d_pattern, d_sincos, d_desc, h_desc, rt);
#ifdef EXP_BRIEF_TWICE
    hipLaunchKernelGGL(k_brief, dim3(nb + (rt.rowoff ? 8 : 0), n_img), dim3(64 * BRIEF_WAVES), 0, s, d_blur, img_pitch, d_kpl, n_features, d_pattern,
                       d_sincos, d_desc, h_desc, rt);
#endif
    return;
  }
  // (measured and dropped: two or three groups of four keypoints per wave in k_ic_moments, all their loads in flight together -- the stage
  //  1.33 / 1.39 ms with one, 1.31 / 1.38 with two, 1.33 / 1.38 with three: the kernel is not waiting on its own round trips;)
  // (measured and dropped: k_brief's XCD-aware block order for k_ic_moments -- 5.61 / 5.66 ms per 512 pairs without, 5.69 / 5.66 with)
  // (measured for batches and dropped: the list entry built inside the moments kernel, k_orient kept -- 5.550 / 5.543 ms per 512 pairs
  //  without, 5.553 / 5.557 with: the list kernel's 0.12 ms hide nothing the moments do not already wait for)
  hipLaunchKernelGGL(k_kplist, dim3((n_features + 255) / 256, n_img), dim3(256), 0, s, d_lv, n_levels, d_sel, d_sel_count, n_features,
                     d_kpl);
  hipLaunchKernelGGL(k_ic_moments, dim3((n_features + 15) / 16, n_img), dim3(256), 0, s, d_pyr, img_pitch, d_kpl, n_features, u,
                     d_moments);
  // the previous batch's stereo match still reads counts, keypoints, row bands and descriptors: the list and the moments (private to
  // this batch) do not wait for it, only the kernels that rewrite those arrays do -- the match has the whole front of this batch to finish
  if (before_lists) (void)hipStreamWaitEvent(s, before_lists, 0);
  hipLaunchKernelGGL(k_orient, dim3((n_features + 255) / 256, n_img), dim3(256), 0, s, d_lv, d_kpl, n_features, d_moments, d_sincos,
                     d_kps, d_aux, d_kx, d_theta, rows0, d_sel_count, n_levels, d_n_kp, h_kps, h_n_kp);
  if (before_brief) (void)hipStreamWaitEvent(s, before_brief, 0);  // the blurred planes come from another stream
  BriefRowTable rt_none{};
  rt_none.n_brief_blocks = (((n_features + BRIEF_WAVES * BRIEF_KPW - 1) / (BRIEF_WAVES * BRIEF_KPW)) + 7) & ~7;  // (= the grid: no row-table blocks)
  hipLaunchKernelGGL(k_brief, dim3(rt_none.n_brief_blocks, n_img), dim3(64 * BRIEF_WAVES), 0, s, d_blur, img_pitch, d_kpl, n_features, d_pattern, d_sincos, d_desc,
                     h_desc, rt_none);
}

}  // namespace orbfe
