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
#include "blur_body.h"
#include "wave_ops.h"

namespace orbfe {

// ---------------------------------------------------------------------------------------------
// resize: 64 x TH output pixels per 256-thread block; grid.x = tiles of all levels >= 1 (host-built descriptors: no
// level search, no division, no dependent tap loads before the footprint is known), grid.y = image.
// The level-0 footprint of the tile (<= 60 rows x 240 bytes at scale 3.58) and the tile's 64 + TH taps are staged in
// LDS in ONE round trip; every thread then produces 4 horizontally adjacent outputs per row from LDS bytes and stores
// them as one word.
// ---------------------------------------------------------------------------------------------
// TH = 64 / 32 / 16 rows per tile: every level takes the tallest tile whose level-0 footprint fits RS_LDS_BYTES (four / two / one
// rows per thread: more work per memory round trip of a wave); the LDS of a launch is sized to its largest footprint.
template <int TH>
__global__ __launch_bounds__(256) void k_resize(const LevelDev* __restrict__ lv, const RsTile* __restrict__ tiles,
                                                const ResizeTap* __restrict__ taps, uint8_t* __restrict__ pyr,
                                                size_t img_pitch, int tile_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint32_t rs_lds[];
  uint32_t* tile = rs_lds;
  ResizeTap* xs = (ResizeTap*)(rs_lds + tile_bytes / 4);
  ResizeTap* ys = xs + RS_TW;
  const int img = blockIdx.y;
  const RsTile T = tiles[blockIdx.x];
  const LevelDev& L = lv[T.level];
  const int x0 = T.x0, y0 = T.y0;
  const int sw = lv[0].w, sh = lv[0].h, sstride = lv[0].stride;
  uint8_t* base = pyr + (size_t)img * img_pitch;
  const uint8_t* S = base + lv[0].plane_off;
  const int sx_lo = T.sx_lo, sy_lo = T.sy_lo, nw = T.nw, nr = T.nr;
  const bool fits = nw > 0;  // always true for pyramid scales; otherwise read global memory directly
  {
    const ResizeTap* xt = taps + L.xtab_off;
    const ResizeTap* yt = taps + L.ytab_off;
    const int k = threadIdx.x;
    if (k < RS_TW)
      xs[k] = xt[min(x0 + k, L.w - 1)];
    else if (k < RS_TW + TH)
      ys[k - RS_TW] = yt[min(y0 + k - RS_TW, L.h - 1)];
  }
  if (fits) {  // footprint rows start at a multiple of 16 bytes and are a whole number of 16-byte quads
    const int nq = nw >> 2;
    const uint32_t inv = ((1u << 20) + nq - 1) / nq;
    const uint8_t* src = S + (size_t)sy_lo * sstride + sx_lo;
    uint4* tile4 = (uint4*)tile;
    for (int k = threadIdx.x; k < nq * nr; k += 256) {
      // (k < 2^14, inv <= 2^20, rows / strides < 2^13: full-rate 24-bit products; the plain ones compile to the quarter-rate v_mul_lo_u32)
      const int r = (int)((uint32_t)mul24u(k, (int)inv) >> 20), c = k - mul24u(r, nq);
      tile4[k] = *(const uint4*)(src + (uint32_t)mad24u(r, sstride, 16 * c));
    }
  }
  __syncthreads();
  const int tx = (threadIdx.x & 15) * 4;
  const int cx = x0 + tx;
  if (cx >= L.w) return;
  const int pitch = nw * 4;
  // explicit address spaces keep the staged path on ds_read (a merged pointer would turn every byte read into a flat load)
  typedef const __attribute__((address_space(3))) uint8_t* lds_bytes_t;
  lds_bytes_t tb = (lds_bytes_t)tile;
  ResizeTap ax[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ax[j] = xs[tx + j];
  if (fits) {
    // Instruction diet (the kernel is VALU-bound: ~24 instructions per output pixel before, ~13 now):
    //  * H << 4 = S[sx] (a0 << 4) + S[sx + 1] (a1 << 4) is ONE v_dot2_u32_u16 on the byte pair spread to 16-bit halves and the two
    //    taps packed in one register (a << 4 <= 2^15 fits the unsigned half; H << 4 < 2^24);
    //  * ((b (H >> 4)) >> 16) is ONE v_mul_hi_u32_u24 of (b << 8) and ((H >> 4) << 8) = (H << 4) & ~0xFF: both below 2^24, and the
    //    high half of the 48-bit product is exactly the reference's floor;
    //  * S[sx + 1] is read at offset 1 of the same address (where the reference clamps sx + 1 its tap is 0, and the byte read instead
    //    lies inside the staged tile's 16-byte row padding), no min(), no second address;
    //  * saturate_cast<uchar> cannot fire: b0 + b1 <= 2049 and H >> 4 <= 32640 give (..) + 2 >> 2 <= 255.
    typedef unsigned short __attribute__((ext_vector_type(2))) us2;
    us2 taps2[4];
    int sxo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      taps2[j] = __builtin_bit_cast(us2, ((uint32_t)(uint16_t)ax[j].c0 << 4) | ((uint32_t)(uint16_t)ax[j].c1 << 20));
      sxo[j] = ax[j].ofs - sx_lo;
    }
#pragma unroll
    for (int rr = 0; rr < TH / 16; ++rr) {
      const int ty = (threadIdx.x >> 4) + 16 * rr;
      const int cy = y0 + ty;
      if (cy >= L.h) break;
      const ResizeTap ay = ys[ty];
      const int sy0 = min(max(ay.ofs, 0), sh - 1), sy1 = min(max(ay.ofs + 1, 0), sh - 1);
      const int o0 = mul24u(sy0 - sy_lo, pitch), o1 = mul24u(sy1 - sy_lo, pitch);
      const uint32_t b0 = (uint32_t)(uint16_t)ay.c0 << 8, b1 = (uint32_t)(uint16_t)ay.c1 << 8;
      uint32_t v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lds_bytes_t r0 = tb + (o0 + sxo[j]), r1 = tb + (o1 + sxo[j]);
        const uint32_t q0 = (uint32_t)r0[0] | ((uint32_t)r0[1] << 16), q1 = (uint32_t)r1[0] | ((uint32_t)r1[1] << 16);
        const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, q0), taps2[j], 0u, false) & 0xFFFFFF00u;
        const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, q1), taps2[j], 0u, false) & 0xFFFFFF00u;
        uint32_t m0, m1;  // (their inputs come from the compiler's own v_and, not straight from the dot: see wave_ops.h)
        asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(m0) : "v"(b0), "v"(h0));
        asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(m1) : "v"(b1), "v"(h1));
        v[j] = (m0 + m1 + 2u) >> 2;
      }
      *(uint32_t*)(base + (L.plane_off + (uint32_t)mad24u(cy, L.stride, cx))) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    }
    return;
  }
  // footprint too large for the LDS (never for pyramid scales): the reference formula straight from global memory
#pragma unroll
  for (int rr = 0; rr < TH / 16; ++rr) {
    const int ty = (threadIdx.x >> 4) + 16 * rr;
    const int cy = y0 + ty;
    if (cy >= L.h) break;
    const ResizeTap ay = ys[ty];
    const int sy0 = min(max(ay.ofs, 0), sh - 1), sy1 = min(max(ay.ofs + 1, 0), sh - 1);
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sx0 = ax[j].ofs;
      const int sx1 = min(ax[j].ofs + 1, sw - 1);  // tap c1 is 0 wherever the reference does not read S[sx+1]
      const uint32_t g0 = (uint32_t)mul24u(sy0, sstride), g1 = (uint32_t)mul24u(sy1, sstride);  // 32-bit offsets from the uniform base
      const int p00 = S[g0 + sx0], p01 = S[g0 + sx1], p10 = S[g1 + sx0], p11 = S[g1 + sx1];
      // (pixels < 2^8, taps <= 2^11, h >> 4 < 2^15: every product fits the full-rate 24-bit multiplier)
      const int h0 = mad24u(p00, ax[j].c0, mul24u(p01, ax[j].c1));
      const int h1 = mad24u(p10, ax[j].c0, mul24u(p11, ax[j].c1));
      int v = ((mul24u((int)ay.c0, h0 >> 4) >> 16) + (mul24u((int)ay.c1, h1 >> 4) >> 16) + 2) >> 2;
      v = min(255, max(0, v));
      out |= (uint32_t)v << (8 * j);
    }
    *(uint32_t*)(base + (L.plane_off + (uint32_t)mad24u(cy, L.stride, cx))) = out;  // stride is a multiple of 16: the padding absorbs the tail
  }
}

// ---------------------------------------------------------------------------------------------
// resize, region-driven (r2): one workgroup stages ONE block of level 0 (about 200 x 50 pixels, chosen per geometry, + the halo its taps reach) and writes
// the output words of ALL levels whose first source pixel lies in the block -- the reference resizes every level from level 0
// (quirk Q1), and with one launch per tile class above each level staged its own copy of the plane: 3.9 GB through the L2 per
// 1024 images, 0.72 ms of the 1.11 ms the three launches took with the arithmetic removed.  Here level 0 is staged once (plus halo).
// The arithmetic is k_resize's; the taps of the region's words are staged in LDS already in the form the inner loop wants
// (x: byte offset inside the staged tile and the two pre-shifted coefficients packed for v_dot2; y: the two row offsets and the two
// coefficients << 8), so a word costs one 16-byte and two 16-byte LDS reads of taps, not their unpacking.
// ---------------------------------------------------------------------------------------------
// PQ: the tile's LDS row pitch in 16-byte units when every region of the context has the same (0: read it per region)
template <int PQ>
__global__ __launch_bounds__(256) void k_resize_regions(const LevelDev* __restrict__ lv, int n_levels, const RsRegion* __restrict__ regions,
                                                        const RgXTap* __restrict__ xtaps, const RgYTap* __restrict__ ytaps,
                                                        uint8_t* __restrict__ pyr, size_t img_pitch, int tile_bytes, int xt_bytes,
                                                        const uint8_t* __restrict__ src_a, const uint8_t* __restrict__ src_b, size_t src_pitch,
                                                        int sstride, uint32_t src_bytes, int copy_l0, int32_t* __restrict__ d_zero, int n_zero,
                                                        int32_t* __restrict__ d_zero2, int n_zero2) {
  // d_zero[0 .. n_zero), d_zero2[0 .. n_zero2): cleared by the first block -- the candidate counters of the FAST launches that follow this
  // kernel and (r6) the per-image level counters of the batch quadtree, whose own memset was a 20 us launch on the critical path
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    for (int i = threadIdx.x; i < n_zero; i += 256) d_zero[i] = 0;
    for (int i = threadIdx.x; i < n_zero2; i += 256) d_zero2[i] = 0;
  }
  // level 0 is read from (src_a, src_b, src_pitch, sstride): the pyramid's own level-0 planes (src_b null: image i at src_a + i src_pitch),
  // or -- device batches -- the CALLER's left / right images (image i = eye i & 1 of pair i >> 1), so that the resize does not wait
  // for the copy-in but runs beside it.  src_bytes: size of one source image; a 16-byte unit that would end past it (the last unit
  // of the last row when the rows are not padded) is put together from single bytes.
  extern __shared__ __attribute__((aligned(16))) uint32_t rs_lds[];
  uint32_t* tile = rs_lds;
  RgXTap* xs = (RgXTap*)(rs_lds + tile_bytes / 4);
  RgYTap* ys = (RgYTap*)(rs_lds + (tile_bytes + xt_bytes) / 4);
  const int img = blockIdx.y;
  const RsRegion& R = regions[blockIdx.x];
  uint8_t* base = pyr + (size_t)img * img_pitch;
  const uint8_t* S = src_b ? ((img & 1) ? src_b : src_a) + (size_t)(img >> 1) * src_pitch : src_a + (size_t)img * src_pitch;
  const int nq = R.nq, nr = R.nr, pq = R.pq;  // units loaded per row, rows, LDS row pitch in units (odd: see RsRegion)
  // the level descriptors of the region, lane = level: requested with everything else, handed out by v_readlane (as scalar
  // loads at the top of each level's loop they were seven dependent memory round trips per workgroup)
  const uint4 gq = *(const uint4*)&R.lev[min((int)(threadIdx.x & 63), ORBFE_MAX_LEVELS - 2)];
  {
    // ONE memory round trip for everything the workgroup reads: the tile's 16-byte units and the two tap tables (already in their LDS
    // layout) are all requested before the first is parked -- as loops of load-then-store the compiler serialised them
    const uint32_t inv = R.inv_nq;
    const uint8_t* src = S + (size_t)R.sy0 * sstride + R.sx0;
    uint4* tile4 = (uint4*)tile;
    const uint4* xsrc = (const uint4*)(xtaps + R.xt_off);
    const uint4* ysrc = (const uint4*)(ytaps + R.yt_off);
    const int n_tile = nq * nr, n_x = R.n_xt >> 1, n_y = R.n_yt;
    constexpr int U = 3;  // units of one kind per thread and sweep (a 1241x376 region: 588 tile units, ~290 + ~155 of taps)
    for (int b0 = 0; b0 < max(n_tile, max(n_x, n_y)); b0 += 256 * U) {
      uint4 tq[U], xq[U], yq[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = b0 + u * 256 + threadIdx.x;
        const int kt = min(k, n_tile - 1);
        const int r = (int)((uint32_t)mul24u(kt, (int)inv) >> 20), c = kt - mul24u(r, nq);
        const uint32_t off = (uint32_t)mad24u(r, sstride, 16 * c);
        const uint32_t end = (uint32_t)mad24u(R.sy0, sstride, R.sx0) + off + 16u;  // offset past the unit, from the start of the image
        if (end <= src_bytes) {
          __builtin_memcpy(&tq[u], src + off, 16);  // (byte-aligned 16-byte load: the caller's rows are not padded)
        } else {
          uint32_t w4[4] = {0u, 0u, 0u, 0u};
          for (uint32_t b = 0; b < 16u && end - 16u + b < src_bytes; ++b) w4[b >> 2] |= (uint32_t)src[off + b] << (8 * (b & 3));
          tq[u] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
        xq[u] = xsrc[min(k, max(n_x - 1, 0))];
        yq[u] = ysrc[min(k, max(n_y - 1, 0))];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = b0 + u * 256 + threadIdx.x;
        const int r = (int)((uint32_t)mul24u(min(k, n_tile - 1), (int)inv) >> 20);
        if (k < n_tile) tile4[k + mul24u(r, pq - nq)] = tq[u];  // row r of the tile starts at unit r * pq
        if (k < n_x) ((uint4*)xs)[k] = xq[u];
        if (k < n_y) ((uint4*)ys)[k] = yq[u];
      }
    }
  }
  __syncthreads();
  if (copy_l0) {
    // level 0 of the pyramid <- the block of the source this region owns, straight from the staged tile (device batches: this IS the
    // copy-in; the columns start at sx0, a multiple of 16, and the plane rows are 16-byte aligned: aligned 16-byte stores)
    const int cq = R.cq, n_u = cq * R.ch;
    const uint32_t inv = ((1u << 20) + cq - 1) / cq;
    const uint4* t4 = (const uint4*)tile;
    uint8_t* dst = base + lv[0].plane_off + (size_t)R.cy0 * lv[0].stride + R.sx0;
    const int r_off = (R.cy0 - R.sy0) * pq;
    for (int k = threadIdx.x; k < n_u; k += 256) {
      const int r = (int)((uint32_t)mul24u(k, (int)inv) >> 20), cu = k - mul24u(r, cq);
      *(uint4*)(dst + (uint32_t)mad24u(r, lv[0].stride, 16 * cu)) = t4[r_off + mul24u(r, pq) + cu];
    }
  }
  typedef const __attribute__((address_space(3))) uint8_t* lds_bytes_t;
  typedef unsigned short __attribute__((ext_vector_type(2))) us2;
  const uint4* lds4 = (const uint4*)rs_lds;
  const uint32_t tile_addr = (uint32_t)(uintptr_t)(lds_bytes_t)(const uint8_t*)tile;  // the tile's LDS byte address, folded into the x offsets below
  // r6: a thread keeps its word COLUMN and walks down the level's rows (rows_per_pass = 256 / nwx rows per step of the workgroup).  The x
  // taps of the column -- two 16-byte LDS reads and the unpacking -- are then fetched once per level instead of once per word, the flat
  // index -> (row, column) split (two 24-bit products, shifts, subtractions) happens once per level, and the row's y tap and the output
  // address advance by constants.  With the tile's row pitch a compile-time constant (PQ: every region of a context has the same, see
  // orbfe_create) the second source row of a word is the first one's address + pitch in the instruction's offset field: the taps of a
  // downscale never clamp vertically (sy1 = sy0 + 1 always; orbfe_create checks it).  66 -> 51 vector and 19 -> 17 LDS instructions per word.
  for (int l = 1; l < n_levels; ++l) {
    struct {
      int wx0, nwx, oy0, noy, xt_lds, yt_lds;
      uint32_t inv_nwx;
    } G;
    {
      const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)gq.x, l - 1), g1 = (uint32_t)__builtin_amdgcn_readlane((int)gq.y, l - 1);
      const uint32_t g2 = (uint32_t)__builtin_amdgcn_readlane((int)gq.z, l - 1);
      G.wx0 = (int)(int16_t)(g0 & 0xFFFFu), G.nwx = (int)(g0 >> 16), G.oy0 = (int)(int16_t)(g1 & 0xFFFFu), G.noy = (int)(g1 >> 16);
      G.xt_lds = (int)(g2 & 0xFFFFu), G.yt_lds = (int)(g2 >> 16);
      G.inv_nwx = (uint32_t)__builtin_amdgcn_readlane((int)gq.w, l - 1);
    }
    const LevelDev& L = lv[l];
    const int nwx = G.nwx, noy = G.noy;
    if (nwx == 0 || noy == 0) continue;  // wave-uniform
    // (indexed in 16-byte units from the 16-byte aligned base: what the compiler needs to see to emit ds_read_b128)
    const uint4* xl = lds4 + ((tile_bytes >> 4) + (G.xt_lds >> 1));
    const uint4* yl = lds4 + (((tile_bytes + xt_bytes) >> 4) + G.yt_lds);
    const int rpp = (int)((256u * G.inv_nwx) >> 20);  // rows per pass = 256 / nwx (inv_nwx = ceil(2^20 / nwx), nwx <= 256: exact)
    const int rr = (int)((uint32_t)mul24u((int)threadIdx.x, (int)G.inv_nwx) >> 20), c = (int)threadIdx.x - mul24u(rr, nwx);
    if (rr >= rpp) continue;  // (the threads past rows_per_pass x nwx sit this level out)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 xa, xb;
    {
      // (the taps of word c's pixels 0, 1 and 2, 3 sit in two arrays of nwx 16-byte units each: a 16-byte lane stride, no bank conflict)
      const uint32_t xaddr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint4*)(xl + c);
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(xa), "=&v"(xb) : "v"(xaddr), "v"(xaddr + 16u * (uint32_t)nwx) : "memory");
    }
    const uint32_t sxa[4] = {xa.x + tile_addr, xa.z + tile_addr, xb.x + tile_addr, xb.z + tile_addr};
    const uint32_t t2[4] = {xa.y, xa.w, xb.y, xb.w};
    uint32_t out_off = L.plane_off + (uint32_t)mad24u(G.oy0 + rr, L.stride, 4 * (G.wx0 + c));
    const uint32_t out_step = (uint32_t)mul24u(rpp, L.stride);
    for (int r = rr; r < noy; r += rpp, out_off += out_step) {
      const uint4 ayq = yl[r];  // o0, o1 (byte offsets of the two source rows in the tile), b0, b1 (coefficients << 8)
      uint32_t v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // (k_resize's inner loop: see the comments there)
        lds_bytes_t r0 = (lds_bytes_t)(uintptr_t)(sxa[j] + ayq.x);
        lds_bytes_t r1 = PQ > 0 ? r0 + PQ * 16 : (lds_bytes_t)(uintptr_t)(sxa[j] + ayq.y);
        const uint32_t q0 = (uint32_t)r0[0] | ((uint32_t)r0[1] << 16), q1 = (uint32_t)r1[0] | ((uint32_t)r1[1] << 16);
        const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, q0), __builtin_bit_cast(us2, t2[j]), 0u, false) & 0xFFFFFF00u;
        const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, q1), __builtin_bit_cast(us2, t2[j]), 0u, false) & 0xFFFFFF00u;
        uint32_t m0, m1;
        asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(m0) : "v"(ayq.z), "v"(h0));
        asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(m1) : "v"(ayq.w), "v"(h1));
        v[j] = (m0 + m1 + 2u) >> 2;
      }
      *(uint32_t*)(base + out_off) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// blur: separable 7-tap, 8.8 fixed point, BORDER_REFLECT_101 -- register sliding window, no LDS.
//
// One wave owns a column strip of 62 words (248 px) and BLUR_ROWS output rows.  Per input row a lane
// loads ONE aligned 32-bit word (4 px; a wave reads 256 contiguous bytes), takes the two neighbouring
// words from the adjacent lanes with a DPP wave shift (lanes 0 and 63 only carry the halo), forms the
// four horizontal 8.8 sums and pushes them into a 7-row register window; once the window is full every
// new row yields one output row (vertical 16.16 sum, +0x8000 >> 16) stored as one 32-bit word.
// Integer arithmetic only => bit-exact with the two-pass definition.
// ---------------------------------------------------------------------------------------------
template <bool SAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blur(const LevelDev* __restrict__ lv, int n_levels, const uint8_t* __restrict__ pyr,
                                              uint8_t* __restrict__ blur, size_t img_pitch, BlurTaps taps, int tile_first) {
  blur_tile<SAT>(lv, n_levels, pyr, blur, img_pitch, taps, (int)blockIdx.x + tile_first, (int)blockIdx.y, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// level 0 <- caller's device images (ORBExtractor.cc:304 image.copyTo(mvPyramids[0])): slot
// slot0 + i*slot_step takes image i.  16 pixels per thread: one (unaligned) 16-byte load from the caller's rows, one aligned
// 16-byte store into the padded plane (4 bytes per thread ran at 2 TB/s).
// ---------------------------------------------------------------------------------------------
// One image row = ceil(w / 16) units of 16 bytes; the units of an image are numbered row-major and a thread takes two of them
// (rows are not a multiple of the block width: a 2-D block over (x, y) left 40 % of its lanes idle at w = 1241).  src_b (nullable):
// a second source whose image i goes to slot slot0 + i * slot_step + 1 -- left and right eyes of a stereo batch in one launch.
__global__ __launch_bounds__(256) void k_load_level0(const uint8_t* __restrict__ src_a, const uint8_t* __restrict__ src_b, size_t src_stride,
                                                     size_t src_pitch, uint8_t* __restrict__ pyr, size_t img_pitch, uint32_t plane_off,
                                                     int dst_stride, int w, int h, int slot0, int slot_step, int units_per_row,
                                                     uint32_t inv_upr) {
  const int z = blockIdx.y;
  const int i = src_b ? (z >> 1) : z;
  const uint8_t* src = (src_b && (z & 1)) ? src_b : src_a;
  const int slot = slot0 + i * slot_step + ((src_b && (z & 1)) ? 1 : 0);
  const uint8_t* sbase = src + (size_t)i * src_pitch;
  uint8_t* dbase = pyr + (size_t)slot * img_pitch + plane_off;
  const int n_units = units_per_row * h;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int u = (blockIdx.x * 2 + k) * 256 + threadIdx.x;
    if (u >= n_units) return;
    const int y = (int)(((unsigned long long)(uint32_t)u * inv_upr) >> 32), x16 = (u - y * units_per_row) * 16;
    const uint8_t* s = sbase + (size_t)y * src_stride + x16;
    uint8_t* d = dbase + (size_t)y * dst_stride + x16;
    if (x16 + 15 < w) {
      uint4 v;
      __builtin_memcpy(&v, s, 16);
      *(uint4*)d = v;
    } else {
      for (int b = 0; x16 + b < w; ++b) d[b] = s[b];
    }
  }
}

void launch_load_level0(hipStream_t st, const uint8_t* d_src, const uint8_t* d_src_b, size_t src_stride, size_t src_pitch, uint8_t* d_pyr,
                        size_t img_pitch, uint32_t plane_off, int dst_stride, int w, int h, int slot0, int slot_step, int n_img) {
  if (n_img <= 0) return;
  const int upr = (w + 15) / 16;
  const uint32_t inv = (uint32_t)(((1ull << 32) + upr - 1) / upr);  // exact floor(u / upr) for u < 2^20 (u * (inv * upr - 2^32) < 2^32)
  const int n_units = upr * h;
  hipLaunchKernelGGL(k_load_level0, dim3((n_units + 511) / 512, d_src_b ? 2 * n_img : n_img), dim3(256), 0, st, d_src, d_src_b, src_stride,
                     src_pitch, d_pyr, img_pitch, plane_off, dst_stride, w, h, slot0, slot_step, upr, inv);
}

// ---------------------------------------------------------------------------------------------
// launchers (called from the C-ABI layer)
// ---------------------------------------------------------------------------------------------
// n_tiles / lds_bytes: the 64x64, 64x32 and 64x16 tile classes, stored in this order in d_tiles
void launch_resize(hipStream_t s, const LevelDev* d_lv, const RsTile* d_tiles, const int* n_tiles, const int* lds_bytes,
                   const ResizeTap* d_taps, uint8_t* d_pyr, size_t img_pitch, int n_img) {
  if (n_img <= 0) return;
  const size_t extra = (RS_TW + 64) * sizeof(ResizeTap);
  const RsTile* t = d_tiles;
  if (n_tiles[0] > 0)
    hipLaunchKernelGGL((k_resize<64>), dim3(n_tiles[0], n_img), dim3(256), (size_t)lds_bytes[0] + extra, s, d_lv, t, d_taps, d_pyr, img_pitch,
                       lds_bytes[0]);
  t += n_tiles[0];
  if (n_tiles[1] > 0)
    hipLaunchKernelGGL((k_resize<32>), dim3(n_tiles[1], n_img), dim3(256), (size_t)lds_bytes[1] + extra, s, d_lv, t, d_taps, d_pyr, img_pitch,
                       lds_bytes[1]);
  t += n_tiles[1];
  if (n_tiles[2] > 0)
    hipLaunchKernelGGL((k_resize<16>), dim3(n_tiles[2], n_img), dim3(256), (size_t)lds_bytes[2] + extra, s, d_lv, t, d_taps, d_pyr, img_pitch,
                       lds_bytes[2]);
}

void launch_resize_regions(hipStream_t s, const LevelDev* d_lv, int n_levels, const RsRegion* d_regions, int n_regions, int tile_bytes,
                           int xt_bytes, int yt_bytes, const RgXTap* d_xtaps, const RgYTap* d_ytaps, uint8_t* d_pyr, size_t img_pitch, int n_img,
                           const uint8_t* src_a, const uint8_t* src_b, size_t src_pitch, int src_stride, uint32_t src_bytes, int copy_l0,
                           int32_t* d_zero, int n_zero, int pq, int32_t* d_zero2, int n_zero2) {
  // pq: the regions' common LDS row pitch (16-byte units; orbfe_create makes it uniform) -- a compile-time constant of the kernel for the
  // pitches region widths of 128 .. 256 pixels give, the run-time form (0) otherwise
  if (n_img <= 0 || n_regions <= 0) return;
#define RS_GO(PQ)                                                                                                                                   \
  hipLaunchKernelGGL(k_resize_regions<PQ>, dim3(n_regions, n_img), dim3(256), (size_t)(tile_bytes + xt_bytes + yt_bytes), s, d_lv, n_levels, d_regions, \
                     d_xtaps, d_ytaps, d_pyr, img_pitch, tile_bytes, xt_bytes, src_a, src_b, src_pitch, src_stride, src_bytes, copy_l0, d_zero, n_zero, d_zero2, n_zero2)
  switch (pq) {
    case 9: RS_GO(9); break;
    case 11: RS_GO(11); break;
    case 13: RS_GO(13); break;
    case 15: RS_GO(15); break;
    case 17: RS_GO(17); break;
    case 19: RS_GO(19); break;
    default: RS_GO(0); break;
  }
#undef RS_GO
}

// tiles [tile_first, tile_first + n_tiles) of the per-image tile list (level-major: a range of tiles is a range of levels)
void launch_blur(hipStream_t s, const LevelDev* d_lv, int n_levels, int tile_first, int n_tiles, const uint8_t* d_pyr, uint8_t* d_blur,
                 size_t img_pitch, const int taps[7], int n_img) {
  if (n_tiles <= 0 || n_img <= 0) return;
  BlurTaps bt;
  for (int i = 0; i < 7; ++i) bt.t[i] = taps[i];
  if (blur_taps_saturate(taps))
    hipLaunchKernelGGL(k_blur<true>, dim3(n_tiles, n_img), dim3(256), 0, s, d_lv, n_levels, d_pyr, d_blur, img_pitch, bt, tile_first);
  else
    hipLaunchKernelGGL(k_blur<false>, dim3(n_tiles, n_img), dim3(256), 0, s, d_lv, n_levels, d_pyr, d_blur, img_pitch, bt, tile_first);
}

}  // namespace orbfe
