// orbfe_internal.h -- structures shared by the host side of the C-ABI and the gfx950 kernels.
#pragma once
#include <stdint.h>

#include "../../include/orbfe.h"

#define ORBFE_MAX_STRIPS 16  // Quadtree::initSplit root strips, round(w/h)  (ORBExtractor.cc:85)
#define ORBFE_BORDER 19      // ORBExtractor::mnBorderSize (ORBExtractor.cc:523)
#define ORBFE_EDGE 16        // mnBorderSize - 3: origin of the FAST region (ORBExtractor.cc:336-337)
#define ORBFE_CENTROID_R 15  // ORBExtractor::mnCentroidR (ORBExtractor.cc:518)
#define ORBFE_MAX_CELL 72    // largest FAST patch side (cell < 60, +6 overlap), LDS tile bound

// fixed-point bilinear tap: dst(dx) = S[ofs]*c0 + S[ofs+1]*c1   (cv::resize INTER_LINEAR, 11-bit)
struct ResizeTap {
  int32_t ofs;
  int16_t c0, c1;
};

// k_blur: output rows per wave (a block = 4 waves stacked vertically); the host sizes the tile grid with it.  A wave reads six rows more than
// it writes: 64 rows per wave (70 / 64 = 9 % of halo work, 19 % at 32) -- step 5.198 -> 5.163 ms, same box, alternating, three rounds; 48: no gain
#ifndef BLUR_ROWS
#define BLUR_ROWS 64
#endif

// k_resize work item: RS_TW x RS_TH output pixels of one level and the level-0 footprint they read
#define RS_TW 64
#define RS_LDS_BYTES 16384  // largest footprint a tile may stage
struct RsTile {
  int16_t level, x0, y0;  // output tile origin
  int16_t sx_lo, nw;      // footprint: first level-0 column (multiple of 16), 32-bit words per row (multiple of 4); nw == 0: does not fit the LDS
  int16_t sy_lo, nr;      // first level-0 row, rows
  int16_t pad;
};

// k_resize_regions work item: a block of level 0 staged ONCE and the output words of EVERY level whose first source pixel lies in it.
// rg_w x rg_h level-0 pixels per region (plus the halo the bilinear taps of its last words reach into), chosen per geometry: the block
// width (a multiple of 16 from 128 to 256) and height (47 or 63) that tile the image with the least waste -- a last column of a few
// pixels is a workgroup in eight that stages a tile and walks seven levels for almost nothing (1241 px: 176 -> 0.79 ms, 208 or 256 -> 0.66).
struct RsRegionLevel {
  int16_t wx0, nwx;        // output words (4 px) of the level: first, count
  int16_t oy0, noy;        // output rows: first, count
  uint16_t xt_lds, yt_lds; // first x tap (8-byte units) / y tap (16-byte units) of the level in the workgroup's LDS tables
  uint32_t inv_nwx;        // ceil(2^20 / nwx): word index -> row without a division
};
// taps as the inner loop wants them, laid out per region exactly as they sit in LDS (the workgroup copies them, 16 bytes at a time)
struct RgXTap {
  int32_t sxo;     // byte offset of the tap's first source pixel inside a staged row
  uint32_t taps2;  // (c0 << 4) | (c1 << 20): the two coefficients, pre-shifted, as the halves v_dot2_u32_u16 multiplies
};
struct RgYTap {
  int32_t o0, o1;   // byte offsets of the two source rows inside the staged tile
  uint32_t b0, b1;  // the two coefficients << 8
};
struct RsRegion {
  int16_t sx0, sy0;        // first staged level-0 column (multiple of 16) / row
  int16_t nq, nr;          // staged 16-byte quads per row, rows
  uint32_t inv_nq;         // ceil(2^20 / nq)
  uint16_t n_xt, n_yt;     // taps of all levels (n_xt is a multiple of 4)
  int16_t cy0, ch, cq;     // the block of level 0 the region OWNS: rows [cy0, cy0 + ch), cq 16-byte units from column sx0 (always staged: the
  int16_t pq;              // region kernel can write level 0 of the pyramid from its tile, which replaces the copy-in of device batches)
                           // pq: LDS row pitch of the tile in 16-byte units, ODD (a pitch of 128 / 192 / 256 bytes puts the rows a wave reads on the
                           // same banks: region widths 112 / 176 / 240 ran 15 % slower than 96 / 128 / 144 / 160 until the pitch was padded)
  uint32_t xt_off, yt_off; // first entry of the region in the context's RgXTap / RgYTap arrays
  RsRegionLevel lev[ORBFE_MAX_LEVELS - 1];
};

// k_quadtree: the levels whose trees one wave works through, as bit masks (one per blockIdx.x)
struct QtGroups {
  uint32_t mask[ORBFE_MAX_LEVELS];
  // n_order > 0: the waves of an image do not take fixed masks but PULL levels, in this order (most expensive first), from a per-image
  // counter: list scheduling with the trees' actual durations (a level's candidate count depends on the image)
  uint8_t order[ORBFE_MAX_LEVELS];
  int32_t n_order;
};

// Per pyramid level, resident in device memory (one table per context).
// A frame or two: k_fast appends a level's candidates to ORBFE_FAST_SHARDS lists (cell index mod shards) instead of one -- see k_fast.hip
#define ORBFE_FAST_SHARDS 16
struct LevelDev {
  int32_t w, h, stride;  // plane size, row pitch in bytes (multiple of 16)
  uint32_t plane_off;    // byte offset of the plane inside one image's pyramid buffer
  float sf;              // mvfScaledFactors[level]
  int32_t quota;         // mvnFeatures[level]
  int32_t quota_off;     // sum of quotas of the lower levels
  // FAST cell grid (ORBExtractor.cc:334-343)
  int32_t reg_w, reg_h;  // region [16, w-16) x [16, h-16)
  int32_t n_cols, n_rows, w_cell, h_cell;
  uint32_t inv_w_cell, inv_h_cell;  // ceil(2^20 / cell size): exact floor-division of a 12-bit coordinate by the cell size
  int32_t cell_base;     // first flattened cell id of this level
  int32_t n_cells;
  int32_t cell_cap;      // upper bound on 3x3-strict local maxima in one cell interior
  // candidate list / quadtree
  uint32_t cand_base;    // first record (uint32 units) of this level inside one image's candidate buffer
  uint32_t cand_cap;     // = n_cells * cell_cap
  uint32_t shard_cap;    // contexts of a frame or two (ORBFE_FAST_SHARDS): records of ONE shard of the level's region, ceil(n_cells / shards) * cell_cap; else 0
  int32_t n_ini;         // root strips
  // a level whose quota does not fit the node table one CU's LDS can hold keeps its table (and sort buffer) in global memory:
  // qt_big_cap > 0 nodes at byte offset qt_big_off of the image's block of the context's d_qt_big buffer, qt_big_sort keys to sort
  uint32_t qt_big_off;
  int32_t qt_big_cap, qt_big_sort;
  uint32_t qt_tab_off;   // this level's coordinate -> code tables in the context's d_qt_tabs (uint16 units, even)
  double strips[ORBFE_MAX_STRIPS + 1];
  // resize tap tables (levels >= 1): offsets into the context's tap array
  uint32_t xtab_off, ytab_off;
  // launch geometry
  int32_t rs_tiles_x, rs_tiles_y, rs_tile_base;  // resize: 64x4 output tiles
  int32_t bl_tiles_x, bl_tiles_y, bl_tile_base;  // blur:   64x16 output tiles
};

// k_blur_mfma.hip: launch geometry and band tables of the batches' blur on the integer matrix cores
struct MbLevel {
  int32_t wg_base, strips;  // first workgroup of the level (four 48-column strips each), strips per row
  int32_t tx_off, ty_off;   // first band (512-byte units) of the level's column blocks / row blocks
};
struct MbGeom {
  MbLevel lv[ORBFE_MAX_LEVELS];
  int32_t n_wg;
  uint32_t spare_off;  // byte offset, inside an image's block of the pyramid buffers, of 256 bytes no plane uses (lanes with nothing to store write there)
};

// One FAST cell (flattened over levels).  Patch = what the reference hands to cv::FAST (ORBExtractor.cc:363).
struct CellDev {
  int16_t level;
  int16_t x0, y0;      // patch origin in level coordinates (iniX, iniY)
  int16_t pw, ph;      // patch size (maxX-iniX, maxY-iniY)
  int16_t offx, offy;  // jdx*wCell, idx*hCell  (ORBExtractor.cc:370-371)
  int16_t pad;
};

// candidate / selected-keypoint record: x (12 bit) | y (12 bit) | response (8 bit); x,y in region coordinates
#define ORBFE_PACK_XYR(x, y, r) ((((uint32_t)(x)) << 20) | (((uint32_t)(y)) << 8) | ((uint32_t)(r)))
#define ORBFE_REC_X(p) (((p) >> 20) & 0xFFFu)
#define ORBFE_REC_Y(p) (((p) >> 8) & 0xFFFu)
#define ORBFE_REC_R(p) ((p)&0xFFu)

// auxiliary per-keypoint data the stereo matcher reads (createRowIndexDB row band, ORBMatcher.cc:924-927)
struct KpAux {
  int16_t row_min, row_max;  // [row_min, row_max)
};

// what the stereo matcher reads of a RIGHT keypoint per candidate (8 bytes next to each other): x in level-0 coordinates (the disparity range
// test, ORBMatcher.cc:44-48) and what pixelSADMatch needs of the winner -- octave and the patch centre getPitch computes,
// cvFloor(pt / scale[octave]) (ORBMatcher.cc:1004-1006) -- so that the best candidate's keypoint record is not a memory round trip of its own
struct KpX {
  float x;
  uint32_t q;  // octave (4 bits: ORBFE_MAX_LEVELS = 16) | cvFloor(pt.x / sf) << 4 (14 bits) | cvFloor(pt.y / sf) << 18 (14 bits; orbfe_create bounds the image at 4096 x 4096)
};
#define ORBFE_KPX_Q(oct, qx, qy) ((uint32_t)(oct) | ((uint32_t)(qx) << 4) | ((uint32_t)(qy) << 18))
#define ORBFE_KPX_OCT(q) ((int)((q)&15u))
#define ORBFE_KPX_QX(q) ((int)(((q) >> 4) & 0x3FFFu))
#define ORBFE_KPX_QY(q) ((int)((q) >> 18))

// device buffers of the batches' row-parallel stereo matcher (k_match.hip), per pair
struct StereoRowsBuf {
  uint32_t* lrow_off;
  uint16_t* lrow_list;
  uint4* work;
  int32_t* work_n;
};

// local BA: non-fixed keyframes the dense reduced solver takes (its panel, right-hand side and diagonal blocks live in 64 KB of LDS)
#define LBA_MAX_FREE 100

struct BaParamsDev {
  double fx, fy, cx, cy, bf;
};

// ---- device-side Levenberg-Marquardt control of the local BA (k_lm.hip) -------------------------------------------------------------
#define LM_CHOL_MAX_NB 42  // free keyframes the register-resident Cholesky takes: 42 * 41 / 2 = 861 off-diagonal blocks, two per thread of 448
#define LM_BIG_MAX_NB 1000 // free keyframes the blocked multi-workgroup Cholesky takes (k_lmbig.hip: the back substitution keeps y in 48 KB of LDS)

// The state record one control lane advances between the trials (device memory; the host reads it once per call).
struct LmState {
  double lambda, ni, current_chi, rho, maxdiag;
  double chi_of[2];     // robust chi2 of the estimate in buffer 0 / 1
  int32_t iters[2];     // iteration budgets of the two optimize() calls (Optimizer.cc:336, :361)
  int32_t done[2];      // iterations started (what optimize() returns)
  int32_t round;        // 0 / 1: which optimize() call; 2: both over
  int32_t it;           // iteration index inside the round
  int32_t qmax;         // trials of the current iteration
  int32_t phase;        // 0: start an iteration | 1: a trial ran, decide it | 3: run another trial of this iteration | 2: round over
  int32_t cur;          // which of the two estimate / system buffers is current
  int32_t run_step, run_switch, run_final;  // gates the kernels of a step / the switch group / the final group read
  int32_t switched, finalized, stopped;
  int32_t ok;           // cleared by the solver kernels of a trial when a point block or a pivot is singular
  int32_t need_chi;     // the current system was (re)built outside a trial: take its chi2 from the partial sums
  int32_t pad;
};

struct LmLaunch {
  int NK, NP, E, nf;
  double *poses[2], *points[2], *terms[2], *Hpl[2], *Hpp[2], *bp[2], *Hll[2], *bl[2], *chi_part[2];
  LmState* state;
  LmState* state_out;  // copy of the state after the last control step of a pass, inside the result block (one download)
  const int32_t *edge_pose, *edge_point, *pt_off, *pt_edges, *ps_off, *ps_edges, *free_pose, *pose_slot;
  int2* pairs;          // [nf (nf + 1) / 2][pair_cap], built on the device (launch_lm_pairs)
  int32_t* pair_cnt;    // [nf (nf + 1) / 2]
  int32_t* pair_table;  // [nf][NP]: edge of free pose j observing the point, -1 if none
  int pair_cap;         // longest edge list of a free pose
  const double *meas, *info;
  const uint8_t *is_stereo, *fixed;
  double *info_eff, *delta_eff, *chi2_last;
  uint8_t* level;
  double *Dinv, *W, *Sblk, *rhs, *x, *scale_part;
  double* M;            // nf > LM_CHOL_MAX_NB: the dense reduced system of the blocked solver (k_lmbig.hip), (ld + 48) x ld, else nullptr
  int ld;
  int32_t* lmb_flags;   // [2 ld / 48 + 2]: [KT] bad pivot, [KT + 1 ...] the back substitution's column flags
  double* lmb_inv;      // [ld / 48][3][256]: inverses of the 16 x 16 diagonal blocks of the factorised diagonal tiles
  double *chi2_out, *poses_out, *points_out;
  uint8_t *bad, *level_out;
  const volatile uint8_t* abort_flag;  // device address of a host-mapped byte the caller's stop flag is mirrored into
  BaParamsDev prm;
};
