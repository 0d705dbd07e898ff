// k_pyramid.hip -- image pyramid for gfx950: fixed-point bilinear resize (levels 1..n-1, each from level 0)
// and the 7x7 sigma=2 fixed-point Gaussian blur of every level.
//
// Replaces ORBExtractor::initPyramid (src/ORB_SLAM2/src/ORBExtractor.cc:278-320):
//   cv::resize(image /*level 0*/, level_i, INTER_LINEAR)            (:316)
//   cv::GaussianBlur(level_i, brief_i, 7x7, 2, 2, BORDER_REFLECT_101)   (:319)
// Both are integer arithmetic and bit-exact by construction (see DESIGN.md for the formulas).
// HBM-bound streaming stencils: one pass reads level 0 (L2/Infinity-Cache resident after the first
// level) and writes 0.68x its size; the blur reads and writes every plane once, staging a 70x22 tile
// (64x16 outputs + 3-px halo) in LDS so each input byte is fetched once per tile.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"

namespace orbfe {

// ---------------------------------------------------------------------------------------------
// resize: 64x4 output pixels per 256-thread block; grid.x = tiles of all levels >= 1, grid.y = image
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(const LevelDev* __restrict__ lv, int n_levels,
                                                const ResizeTap* __restrict__ taps, uint8_t* __restrict__ pyr,
                                                size_t img_pitch) {
  const int img = blockIdx.y;
  const int tile = blockIdx.x;
  int l = 1;
  while (l + 1 < n_levels && tile >= lv[l + 1].rs_tile_base) ++l;
  const LevelDev& L = lv[l];
  const int t = tile - L.rs_tile_base;
  const int tx = t % L.rs_tiles_x, ty = t / L.rs_tiles_x;
  const int dx = tx * 64 + (threadIdx.x & 63);
  const int dy = ty * 4 + (threadIdx.x >> 6);
  if (dx >= L.w || dy >= L.h) return;
  const int sw = lv[0].w, sh = lv[0].h, sstride = lv[0].stride;
  uint8_t* base = pyr + (size_t)img * img_pitch;
  const uint8_t* S = base + lv[0].plane_off;
  const ResizeTap ax = taps[L.xtab_off + dx];
  const ResizeTap ay = taps[L.ytab_off + dy];
  const int sy0 = min(max(ay.ofs, 0), sh - 1);
  const int sy1 = min(max(ay.ofs + 1, 0), sh - 1);
  const int sx0 = ax.ofs;
  const int sx1 = min(ax.ofs + 1, sw - 1);  // tap c1 is 0 wherever the reference does not read S[sx+1]
  const uint8_t* r0 = S + (size_t)sy0 * sstride;
  const uint8_t* r1 = S + (size_t)sy1 * sstride;
  const int h0 = r0[sx0] * ax.c0 + r0[sx1] * ax.c1;
  const int h1 = r1[sx0] * ax.c0 + r1[sx1] * ax.c1;
  int v = ((((int)ay.c0 * (h0 >> 4)) >> 16) + (((int)ay.c1 * (h1 >> 4)) >> 16) + 2) >> 2;
  v = min(255, max(0, v));
  base[L.plane_off + (size_t)dy * L.stride + dx] = (uint8_t)v;
}

// ---------------------------------------------------------------------------------------------
// blur: 64x16 outputs per block, separable 7-tap, 8.8 fixed point, BORDER_REFLECT_101
// ---------------------------------------------------------------------------------------------
struct BlurTaps {
  int t[7];
};

__device__ __forceinline__ int reflect101(int p, int n) {
  // n >= 38 and |p| < n + 80 here, so at most two reflections
  while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
  return p;
}

__global__ __launch_bounds__(256) void k_blur(const LevelDev* __restrict__ lv, int n_levels, const uint8_t* __restrict__ pyr,
                                              uint8_t* __restrict__ blur, size_t img_pitch, BlurTaps taps) {
  __shared__ uint8_t tin[22][72];
  __shared__ uint16_t tmid[22][64];
  const int img = blockIdx.y;
  const int tile = blockIdx.x;
  int l = 0;
  while (l + 1 < n_levels && tile >= lv[l + 1].bl_tile_base) ++l;
  const LevelDev& L = lv[l];
  const int t = tile - L.bl_tile_base;
  const int x0 = (t % L.bl_tiles_x) * 64, y0 = (t / L.bl_tiles_x) * 16;
  const uint8_t* P = pyr + (size_t)img * img_pitch + L.plane_off;
  for (int i = threadIdx.x; i < 22 * 70; i += 256) {
    const int r = i / 70, c = i - r * 70;
    const int gy = reflect101(y0 + r - 3, L.h);
    const int gx = reflect101(x0 + c - 3, L.w);
    tin[r][c] = P[(size_t)gy * L.stride + gx];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 22 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc += (uint32_t)taps.t[k] * tin[r][c + k];
    tmid[r][c] = (uint16_t)min(acc, 65535u);  // ufixedpoint16 saturation (never hit when the taps sum to 256)
  }
  __syncthreads();
  uint8_t* D = blur + (size_t)img * img_pitch + L.plane_off;
  for (int i = threadIdx.x; i < 16 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    const int gx = x0 + c, gy = y0 + r;
    if (gx >= L.w || gy >= L.h) continue;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc += (uint32_t)taps.t[k] * tmid[r + k][c];
    const uint32_t v = (acc + 0x8000u) >> 16;
    D[(size_t)gy * L.stride + gx] = (uint8_t)min(v, 255u);
  }
}

// ---------------------------------------------------------------------------------------------
// level 0 <- caller's device images (ORBExtractor.cc:304 image.copyTo(mvPyramids[0])): slot
// slot0 + i*slot_step takes image i.  4 pixels per thread, one aligned 32-bit store.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_load_level0(const uint8_t* __restrict__ src, size_t src_stride, size_t src_pitch,
                                                     uint8_t* __restrict__ pyr, size_t img_pitch, uint32_t plane_off, int dst_stride,
                                                     int w, int h, int slot0, int slot_step) {
  const int i = blockIdx.z;
  const int y = blockIdx.y;
  const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x4 >= w || y >= h) return;
  const uint8_t* s = src + (size_t)i * src_pitch + (size_t)y * src_stride + x4;
  uint8_t* d = pyr + (size_t)(slot0 + i * slot_step) * img_pitch + plane_off + (size_t)y * dst_stride + x4;
  if (x4 + 3 < w) {
    const uint32_t v = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
    *(uint32_t*)d = v;
  } else {
    for (int k = 0; x4 + k < w; ++k) d[k] = s[k];
  }
}

void launch_load_level0(hipStream_t st, const uint8_t* d_src, size_t src_stride, size_t src_pitch, uint8_t* d_pyr, size_t img_pitch,
                        uint32_t plane_off, int dst_stride, int w, int h, int slot0, int slot_step, int n_img) {
  if (n_img <= 0) return;
  hipLaunchKernelGGL(k_load_level0, dim3((w + 1023) / 1024, h, n_img), dim3(256), 0, st, d_src, src_stride, src_pitch, d_pyr,
                     img_pitch, plane_off, dst_stride, w, h, slot0, slot_step);
}

// ---------------------------------------------------------------------------------------------
// launchers (called from the C-ABI layer)
// ---------------------------------------------------------------------------------------------
void launch_resize(hipStream_t s, const LevelDev* d_lv, int n_levels, int total_tiles, const ResizeTap* d_taps, uint8_t* d_pyr,
                   size_t img_pitch, int n_img) {
  if (total_tiles <= 0 || n_img <= 0) return;
  hipLaunchKernelGGL(k_resize, dim3(total_tiles, n_img), dim3(256), 0, s, d_lv, n_levels, d_taps, d_pyr, img_pitch);
}

void launch_blur(hipStream_t s, const LevelDev* d_lv, int n_levels, int total_tiles, const uint8_t* d_pyr, uint8_t* d_blur,
                 size_t img_pitch, const int taps[7], int n_img) {
  if (total_tiles <= 0 || n_img <= 0) return;
  BlurTaps bt;
  for (int i = 0; i < 7; ++i) bt.t[i] = taps[i];
  hipLaunchKernelGGL(k_blur, dim3(total_tiles, n_img), dim3(256), 0, s, d_lv, n_levels, d_pyr, d_blur, img_pitch, bt);
}

}  // namespace orbfe
