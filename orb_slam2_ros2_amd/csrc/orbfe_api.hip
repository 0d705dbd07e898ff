// orbfe_api.hip -- host side of the C-ABI declared in include/orbfe.h: geometry tables, device buffers,
// stream/event plumbing and the launch sequence.  No OpenCV, no torch, no CPU fallback.
#include "orbfe_ctx.h"

static const int8_t kEmbeddedPattern[256][4] = {
#include "brief_pattern.inc"
};

// The text of the last failed call is kept PER THREAD (like dlerror): slot calls of one context run on several threads at once.
static thread_local std::string g_last_error;

orbfe_status fail(orbfe_ctx* c, orbfe_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  (void)c;
  g_last_error = buf;
  return st;
}


// cv::resize INTER_LINEAR coefficient tables for one axis (imgproc/resize.cpp, 8-bit fixed-point path)
static void build_resize_axis(int s, int d, std::vector<ResizeTap>& out) {
  const double scale = 1.0 / ((double)d / s);
  for (int i = 0; i < d; ++i) {
    float f = (float)((i + 0.5) * scale - 0.5);
    int si = cv_floor_f(f);
    f -= si;
    ResizeTap t;
    t.ofs = si;
    t.c0 = sat_short_f((1.f - f) * 2048);
    t.c1 = sat_short_f(f * 2048);
    out.push_back(t);
  }
}

// Geometry exactly as the reference derives it: ORBExtractor.cc:283-317 (scales, quotas, level sizes),
// :334-343 (cell grid), :81-96 (root strips), :217-236 (umax).
static orbfe_status build_geometry(orbfe_ctx* c) {
  const orbfe_config& cfg = c->cfg;
  const int nl = cfg.n_levels;
  c->lv.assign(nl, LevelDev());
  std::vector<float> sf(nl);
  for (int l = 0; l < nl; ++l) sf[l] = (float)std::pow((double)cfg.scale_factor, (double)l);
  std::vector<int> quota(nl, 0);
  {
    const float scale = 1.0f / cfg.scale_factor;
    int sum = 0;
    int nfeats = cv_round_d((double)(cfg.n_features * (1 - scale)) / (1 - std::pow((double)scale, (double)nl)));
    for (int l = 0; l < nl - 1; ++l) {
      quota[l] = nfeats;
      sum += nfeats;
      nfeats = cv_round_f(nfeats * scale);
    }
    quota[nl - 1] = std::max(0, cfg.n_features - sum);
    // The reference's quotas are rounded per level and only the LAST level absorbs the difference (ORBExtractor.cc:292-300): for
    // small nFeatures (< 60 at 8 levels x 1.2) the first levels alone already exceed it and an image yields MORE than nFeatures
    // keypoints.  Every per-image array is therefore sized by the capacity max(nFeatures, sum of quotas), which the caller can
    // query (orbfe_get_capacity); it equals nFeatures for every configuration the reference ships.
    c->kp_cap = std::max(cfg.n_features, sum + quota[nl - 1]);
  }
  size_t plane_off = 0;
  uint32_t cand_base = 0;
  {
    const char* she = getenv("ORBFE_FAST_SHARDS");  // 1: one list per level everywhere (r5's layout)
    c->fast_shards = (cfg.max_images <= 16 && !(she && atoi(she) <= 1)) ? ORBFE_FAST_SHARDS : 1;
  }
  int cell_base = 0, quota_off = 0, rs_tiles = 0, bl_tiles = 0, max_quota = 0, max_ini = 4;
  c->cells.clear();
  c->taps.clear();
  for (int l = 0; l < nl; ++l) {
    LevelDev& L = c->lv[l];
    L.w = (l == 0) ? cfg.width : cv_round_f(cfg.width / sf[l]);
    L.h = (l == 0) ? cfg.height : cv_round_f(cfg.height / sf[l]);
    if (l > 0 && (L.w < 2 * ORBFE_BORDER || L.h < 2 * ORBFE_BORDER))
      return fail(c, ORBFE_EBADSIZE, "ImageSizeError: level %d would be %dx%d (< %d px)", l, L.w, L.h, 2 * ORBFE_BORDER);
    L.stride = (int)align_up((size_t)L.w, 16);
    L.plane_off = (uint32_t)plane_off;
    plane_off += align_up((size_t)L.stride * L.h, 256);
    L.sf = sf[l];
    L.quota = quota[l];
    L.quota_off = quota_off;
    quota_off += quota[l];
    max_quota = std::max(max_quota, quota[l]);
    // FAST grid
    L.reg_w = L.w - 2 * ORBFE_EDGE;
    L.reg_h = L.h - 2 * ORBFE_EDGE;
    L.n_cols = L.reg_w / 30;
    L.n_rows = L.reg_h / 30;
    if (L.n_cols <= 0 || L.n_rows <= 0)
      return fail(c, ORBFE_EBADSIZE, "level %d region %dx%d is smaller than one 30-px FAST cell", l, L.reg_w, L.reg_h);
    L.w_cell = L.reg_w / L.n_cols;  // ceil() of an integer division is a no-op (quirk Q2)
    L.h_cell = L.reg_h / L.n_rows;
    L.inv_w_cell = ((1u << 20) + L.w_cell - 1) / L.w_cell;
    L.inv_h_cell = ((1u << 20) + L.h_cell - 1) / L.h_cell;
    if (L.w_cell + 6 > ORBFE_MAX_CELL || L.h_cell + 6 > ORBFE_MAX_CELL)
      return fail(c, ORBFE_EBADSIZE, "level %d cell %dx%d exceeds the LDS tile bound", l, L.w_cell, L.h_cell);
    if (L.reg_w > 4095 || L.reg_h > 4095) return fail(c, ORBFE_EBADSIZE, "level %d region exceeds 4095 px", l);
    L.cell_base = cell_base;
    L.cell_cap = ((L.w_cell + 1) / 2) * ((L.h_cell + 1) / 2);
    const int max_bx = L.w - ORBFE_EDGE, max_by = L.h - ORBFE_EDGE;
    int n_cells = 0;
    for (int idx = 0; idx < L.n_rows; ++idx) {
      int ini_y = ORBFE_EDGE + idx * L.h_cell, max_y = ini_y + L.h_cell + 6;
      if (ini_y >= max_by - 6) continue;
      if (max_y > max_by) max_y = max_by;
      for (int jdx = 0; jdx < L.n_cols; ++jdx) {
        int ini_x = ORBFE_EDGE + jdx * L.w_cell, max_x = ini_x + L.w_cell + 6;
        if (ini_x >= max_bx - 6) continue;
        if (max_x > max_bx) max_x = max_bx;
        CellDev cd;
        cd.level = (int16_t)l;
        cd.x0 = (int16_t)ini_x;
        cd.y0 = (int16_t)ini_y;
        cd.pw = (int16_t)(max_x - ini_x);
        cd.ph = (int16_t)(max_y - ini_y);
        cd.offx = (int16_t)(jdx * L.w_cell);
        cd.offy = (int16_t)(idx * L.h_cell);
        cd.pad = 0;
        c->cells.push_back(cd);
        c->lvl_max_pw[l] = std::max(c->lvl_max_pw[l], (int)cd.pw);
        c->lvl_max_ph[l] = std::max(c->lvl_max_ph[l], (int)cd.ph);
        ++n_cells;
      }
    }
    // XCD-aware order: the workgroups of a launch go round-robin to the 8 XCDs, each with its own L2, and neighbouring patches
    // share 6 of their ~36 columns / rows -- in row-major order every neighbour pair sits on two different L2s and the shared lines
    // are fetched twice (or more).  Cells are regrouped into 8 vertical strips, strip x on table positions = x (mod 8), so that one
    // XCD walks one strip of the image top to bottom.  (The candidate list of a level is a set: the cell order is free.)
    if (n_cells >= 16 && !getenv("ORBFE_NO_XCD_ORDER")) {
      std::vector<CellDev> strip[8];
      const size_t first = c->cells.size() - (size_t)n_cells;
      for (size_t i = first; i < c->cells.size(); ++i) {
        const int jdx = c->cells[i].offx / std::max((int)L.w_cell, 1);
        strip[std::min(7, jdx * 8 / std::max((int)L.n_cols, 1))].push_back(c->cells[i]);
      }
      size_t w = first;
      for (size_t i = 0; w < c->cells.size(); ++i)
        for (int x = 0; x < 8; ++x)
          if (i < strip[x].size()) c->cells[w++] = strip[x][i];
    }
    L.n_cells = n_cells;
    cell_base += n_cells;
    L.cand_cap = (uint32_t)n_cells * (uint32_t)L.cell_cap;
    L.cand_base = cand_base;
    // (a frame or two: the level's region as fast_shards equal parts, each large enough for the cells it can receive)
    L.shard_cap = c->fast_shards > 1 ? (uint32_t)((n_cells + c->fast_shards - 1) / c->fast_shards) * (uint32_t)L.cell_cap : 0u;
    cand_base += (uint32_t)align_up(std::max<uint32_t>(L.cand_cap, (uint32_t)c->fast_shards * L.shard_cap), 32);
    // root strips (Quadtree::initSplit)
    {
      const double w = (double)L.reg_w, h = (double)L.reg_h;
      const int n_ini = (int)std::round(w / h);
      if (n_ini > ORBFE_MAX_STRIPS) return fail(c, ORBFE_EBADSIZE, "aspect ratio %d:1 exceeds %d root strips", n_ini, ORBFE_MAX_STRIPS);
      L.n_ini = n_ini;
      max_ini = std::max(max_ini, n_ini);
      std::memset(L.strips, 0, sizeof L.strips);
      if (n_ini > 0) {
        const float hx = (float)(w / n_ini);
        L.strips[0] = 0.0;
        for (int i = 1; i < n_ini; ++i) L.strips[i] = (double)((float)i * hx);
        L.strips[n_ini] = w;
      }
    }
    // resize tables and launch tiles
    L.rs_tile_base = rs_tiles;
    L.rs_tiles_x = L.rs_tiles_y = 0;
    L.xtab_off = L.ytab_off = 0;
    if (l > 0) {
      L.xtab_off = (uint32_t)c->taps.size();
      build_resize_axis(cfg.width, L.w, c->taps);
      L.ytab_off = (uint32_t)c->taps.size();
      build_resize_axis(cfg.height, L.h, c->taps);
      L.rs_tiles_x = (L.w + RS_TW - 1) / RS_TW;  // k_resize output tiles (rows: decided below, 16 or 32 per tile)
      L.rs_tiles_y = 0;
    }
    L.bl_tile_base = bl_tiles;
    L.bl_tiles_x = (L.w + 247) / 248;  // k_blur: 62 words (248 px) per wave, 4 waves x BLUR_ROWS rows per block
    L.bl_tiles_y = (L.h + 4 * BLUR_ROWS - 1) / (4 * BLUR_ROWS);
    bl_tiles += L.bl_tiles_x * L.bl_tiles_y;
  }
  // horizontal taps: clamp exactly as resizeGeneric_ does (sx<0 -> 0/fx=0; sx>=sw-1 -> sw-1/fx=0)
  for (int l = 1; l < nl; ++l) {
    LevelDev& L = c->lv[l];
    for (int i = 0; i < L.w; ++i) {
      ResizeTap& t = c->taps[L.xtab_off + i];
      if (t.ofs < 0) {
        t.ofs = 0;
        t.c0 = 2048;
        t.c1 = 0;
      }
      if (t.ofs >= cfg.width - 1) {
        t.ofs = cfg.width - 1;
        t.c0 = 2048;
        t.c1 = 0;
      }
    }
  }
  // resize work items: output tile + the level-0 footprint it reads (taps are monotone in the output coordinate).  Levels whose
  // 64x32 tiles all stage <= RS_TALL_LDS_BYTES use those ("tall", listed first), the coarser levels 64x16 tiles.
  c->rs_tile_tab.clear();
  std::vector<RsTile> cls[3];
  for (int k = 0; k < 3; ++k) c->rs_n[k] = c->rs_bytes[k] = 0;
  for (int l = 1; l < nl; ++l) {
    LevelDev& L = c->lv[l];
    const ResizeTap* xt = c->taps.data() + L.xtab_off;
    const ResizeTap* yt = c->taps.data() + L.ytab_off;
    auto make_tiles = [&](int th, std::vector<RsTile>& out, int& max_bytes) {
      max_bytes = 0;
      const int tys = (L.h + th - 1) / th;
      for (int ty = 0; ty < tys; ++ty)
        for (int tx = 0; tx < L.rs_tiles_x; ++tx) {
          const int x0 = tx * RS_TW, y0 = ty * th;
          const int x1 = std::min(x0 + RS_TW, (int)L.w) - 1, y1 = std::min(y0 + th, (int)L.h) - 1;
          const int sx_lo = xt[x0].ofs & ~15, sx_hi = std::min(xt[x1].ofs + 1, cfg.width - 1);
          const int sy_lo = std::min(std::max(yt[y0].ofs, 0), cfg.height - 1), sy_hi = std::min(std::max(yt[y1].ofs + 1, 0), cfg.height - 1);
          const int nw = (((sx_hi - sx_lo) >> 4) + 1) * 4, nr = sy_hi - sy_lo + 1;  // whole 16-byte quads (rows are padded to 16 B)
          RsTile t;
          t.level = (int16_t)l, t.x0 = (int16_t)x0, t.y0 = (int16_t)y0, t.sx_lo = (int16_t)sx_lo, t.sy_lo = (int16_t)sy_lo;
          t.nw = (int16_t)((nw * nr * 4 <= RS_LDS_BYTES) ? nw : 0);
          t.nr = (int16_t)nr, t.pad = 0;
          max_bytes = std::max(max_bytes, nw * nr * 4);
          out.push_back(t);
        }
      return tys;
    };
    for (int k = 0; k < 3; ++k) {  // the tallest tile whose footprint still fits
      const int th = 64 >> k;
      std::vector<RsTile> cand;
      int bytes = 0;
      const int tys = make_tiles(th, cand, bytes);
      if (bytes <= RS_LDS_BYTES || k == 2) {
        cls[k].insert(cls[k].end(), cand.begin(), cand.end());
        c->rs_bytes[k] = std::max(c->rs_bytes[k], std::min(bytes, RS_LDS_BYTES));
        L.rs_tiles_y = tys;
        break;
      }
    }
  }
  // XCD-aware order inside a class (one launch): workgroups go round-robin to the 8 XCDs, so the tiles are regrouped into 8 vertical
  // strips of level 0 -- strip x on positions = x (mod 8), top to bottom, the levels of the class interleaved by source row -- and one
  // XCD's L2 then serves its strip of level 0 to every level instead of each L2 fetching the whole plane for each level.
  if (!getenv("ORBFE_NO_XCD_ORDER"))
    for (int k = 0; k < 3; ++k) {
      if (cls[k].size() < 16) continue;
      std::vector<RsTile> strip[8];
      for (const RsTile& t : cls[k]) strip[std::min(7, std::max(0, (int)t.sx_lo * 8 / std::max(cfg.width, 1)))].push_back(t);
      for (auto& v : strip)
        std::stable_sort(v.begin(), v.end(), [](const RsTile& a, const RsTile& b) { return a.sy_lo < b.sy_lo; });
      size_t w = 0;
      for (size_t i = 0; w < cls[k].size(); ++i)
        for (int x = 0; x < 8; ++x)
          if (i < strip[x].size()) cls[k][w++] = strip[x][i];
    }
  for (int k = 0; k < 3; ++k) {
    c->rs_n[k] = (int)cls[k].size();
    c->rs_bytes[k] = (int)align_up((size_t)std::max(c->rs_bytes[k], 16), 16);
    c->rs_tile_tab.insert(c->rs_tile_tab.end(), cls[k].begin(), cls[k].end());
  }
  // region-driven resize: per block of level 0 and level, the rectangle of output words (4 px) x rows whose first source pixel /
  // source row lies in the block (taps are monotone, so these are ranges), the level-0 rectangle they read, and where the
  // level's taps sit in the workgroup's LDS tables
  c->rs_regions.clear();
  c->rg_xtaps.clear();
  c->rg_ytaps.clear();
  c->rg_tile_bytes = c->rg_xt_bytes = c->rg_yt_bytes = 0;
  if (nl > 1) {
    int RG_W = 176, RG_H = 47;
    {
      double best = -1.0;
      for (int hh : {47, 63})
        for (int ww = 128; ww <= 256; ww += 16) {
          const int nx = (cfg.width + ww - 1) / ww, ny = (cfg.height + hh - 1) / hh;
          const double fill = ((double)cfg.width / (nx * ww)) * ((double)cfg.height / (ny * hh));
          if (fill > best + 1e-9 || (fill > best - 1e-9 && ww * hh > RG_W * RG_H)) best = std::max(best, fill), RG_W = ww, RG_H = hh;
        }
    }
    const int nrx = (cfg.width + RG_W - 1) / RG_W, nry = (cfg.height + RG_H - 1) / RG_H;
    bool ok = true;
    for (int j = 0; j < nry && ok; ++j)
      for (int i = 0; i < nrx && ok; ++i) {
        RsRegion R;
        std::memset(&R, 0, sizeof R);
        int x_hi = -1, y_lo = 1 << 30, y_hi = -1, n_xt = 0, n_yt = 0;
        for (int l = 1; l < nl; ++l) {
          const LevelDev& L = c->lv[l];
          const ResizeTap* xt = c->taps.data() + L.xtab_off;
          const ResizeTap* yt = c->taps.data() + L.ytab_off;
          auto reg_of = [](int v, int step, int n) { return std::min(v / step, n - 1); };
          const int nw = (L.w + 3) / 4;
          int wx0 = -1, wx1 = -1, oy0 = -1, oy1 = -1;
          for (int wx = 0; wx < nw; ++wx)
            if (reg_of(xt[4 * wx].ofs, RG_W, nrx) == i) {
              if (wx0 < 0) wx0 = wx;
              wx1 = wx + 1;
            }
          for (int oy = 0; oy < L.h; ++oy)
            if (reg_of(std::min(std::max(yt[oy].ofs, 0), cfg.height - 1), RG_H, nry) == j) {
              if (oy0 < 0) oy0 = oy;
              oy1 = oy + 1;
            }
          RsRegionLevel& G = R.lev[l - 1];
          if (wx0 < 0 || oy0 < 0) continue;  // nothing of this level starts in the block
          G.wx0 = (int16_t)wx0, G.nwx = (int16_t)(wx1 - wx0), G.oy0 = (int16_t)oy0, G.noy = (int16_t)(oy1 - oy0);
          G.inv_nwx = ((1u << 20) + G.nwx - 1) / G.nwx;
          G.xt_lds = (uint16_t)n_xt, G.yt_lds = (uint16_t)n_yt;
          n_xt += 4 * G.nwx, n_yt += G.noy;
          for (int px = 4 * wx0; px < 4 * wx1; ++px) x_hi = std::max(x_hi, xt[std::min(px, (int)L.w - 1)].ofs + 1);
          for (int oy = oy0; oy < oy1; ++oy) {
            y_lo = std::min(y_lo, std::min(std::max(yt[oy].ofs, 0), cfg.height - 1));
            y_hi = std::max(y_hi, std::min(std::max(yt[oy].ofs + 1, 0), cfg.height - 1));
          }
          if ((long long)G.nwx * G.noy >= (1 << 20) / 256 * 256) ok = false;  // (never: a region holds a few thousand words)
        }
        // the block of level 0 this region owns is staged whether a level needs all of it or not (see RsRegion::cy0)
        {
          const int bx1 = std::min((i + 1) * RG_W, cfg.width) - 1, by0 = j * RG_H, by1 = std::min((j + 1) * RG_H, cfg.height) - 1;
          x_hi = std::max(x_hi, bx1), y_lo = std::min(y_lo, by0), y_hi = std::max(y_hi, by1);
          R.cy0 = (int16_t)by0, R.ch = (int16_t)(by1 - by0 + 1), R.cq = (int16_t)((bx1 - i * RG_W) / 16 + 1), R.pq = 0;
        }
        R.sx0 = (int16_t)(i * RG_W), R.sy0 = (int16_t)y_lo;
        R.nq = (int16_t)(((x_hi - R.sx0) >> 4) + 1), R.nr = (int16_t)(y_hi - y_lo + 1);
        R.inv_nq = ((1u << 20) + R.nq - 1) / R.nq;
        R.pq = (int16_t)(R.nq | 1);
        R.n_xt = (uint16_t)n_xt, R.n_yt = (uint16_t)n_yt;
        if (R.nq * R.nr >= (1 << 20) / 512 || n_xt > 65535 || n_yt > 65535) ok = false;
        // the region's taps in their LDS layout
        R.xt_off = (uint32_t)c->rg_xtaps.size(), R.yt_off = (uint32_t)c->rg_ytaps.size();
        for (int l = 1; l < nl; ++l) {
          const LevelDev& L = c->lv[l];
          const RsRegionLevel& G = R.lev[l - 1];
          const ResizeTap* xt = c->taps.data() + L.xtab_off;
          const ResizeTap* yt = c->taps.data() + L.ytab_off;
          for (int kk = 0; kk < 4 * G.nwx; ++kk) {
            // LDS order: the taps of pixels 0, 1 of every word, then those of pixels 2, 3 (two arrays of 16-byte units)
            const int half = kk / (2 * G.nwx), cw = (kk % (2 * G.nwx)) / 2, k = 4 * cw + 2 * half + (kk & 1);
            const ResizeTap t = xt[std::min(4 * G.wx0 + k, (int)L.w - 1)];
            RgXTap o;
            o.sxo = t.ofs - R.sx0;
            o.taps2 = ((uint32_t)(uint16_t)t.c0 << 4) | ((uint32_t)(uint16_t)t.c1 << 20);
            c->rg_xtaps.push_back(o);
          }
          for (int k = 0; k < G.noy; ++k) {
            const ResizeTap t = yt[G.oy0 + k];
            const int r0 = std::min(std::max(t.ofs, 0), cfg.height - 1), r1 = std::min(std::max(t.ofs + 1, 0), cfg.height - 1);
            RgYTap o;
            o.o0 = (r0 - R.sy0) * R.pq * 16;
            o.o1 = (r1 - R.sy0) * R.pq * 16;
            o.b0 = (uint32_t)(uint16_t)t.c0 << 8;
            o.b1 = (uint32_t)(uint16_t)t.c1 << 8;
            c->rg_ytaps.push_back(o);
          }
        }
        c->rg_tile_bytes = std::max(c->rg_tile_bytes, R.pq * 16 * R.nr);
        c->rg_xt_bytes = std::max(c->rg_xt_bytes, n_xt * 8);
        c->rg_yt_bytes = std::max(c->rg_yt_bytes, n_yt * 16);
        c->rs_regions.push_back(R);
      }
    // One LDS row pitch for every region of the context (the widest region's, odd): k_resize_regions then has it as a compile-time
    // constant and reaches a word's second source row through the instruction's offset field -- valid because a downscale's vertical taps
    // never clamp (sy1 = sy0 + 1 for every output row: checked here, rg_pq = 0 -- the run-time form -- otherwise)
    c->rg_pq = 0;
    if (ok && !c->rs_regions.empty()) {
      int pq_all = 0;
      for (const RsRegion& R : c->rs_regions) pq_all = std::max(pq_all, (int)R.pq);
      bool no_clamp = true;
      c->rg_tile_bytes = 0;
      for (RsRegion& R : c->rs_regions) {
        for (int k = 0; k < (int)R.n_yt; ++k) {
          RgYTap& o = c->rg_ytaps[R.yt_off + k];
          o.o0 = o.o0 / (R.pq * 16) * (pq_all * 16);
          o.o1 = o.o1 / (R.pq * 16) * (pq_all * 16);
          if (o.o1 != o.o0 + pq_all * 16) no_clamp = false;
        }
        R.pq = (int16_t)pq_all;
        c->rg_tile_bytes = std::max(c->rg_tile_bytes, pq_all * 16 * R.nr);
      }
      if (no_clamp) c->rg_pq = pq_all;
    }
    c->rg_xt_bytes = (int)align_up((size_t)c->rg_xt_bytes, 16);
    if (!ok || c->rg_tile_bytes + c->rg_xt_bytes + c->rg_yt_bytes > 60 * 1024) c->rs_regions.clear();  // fall back to the tile classes
  }
  rs_tiles = (int)c->rs_tile_tab.size();
  c->n_cells_total = cell_base;
  c->rs_tiles = rs_tiles;
  c->bl_tiles = bl_tiles;
  c->pyr_spare_off = (uint32_t)plane_off;           // 256 bytes behind the last plane that no plane uses (k_blur_mfma's idle lanes store there)
  c->img_pitch = align_up(plane_off + 256, 4096);
  c->scratch_pitch = align_up(cand_base, 64);
  // The node table of a level lives in one CU's LDS as far as that goes (~2700 nodes in 120 KB); the reference has no limit on
  // nFeatures (ORBExtractor.cc:291-301), so the levels with larger quotas keep theirs in global memory (k_quadtree, NODES_LDS = false)
  int lds_nodes = max_quota + max_ini + 8;
  auto sort_cap_of = [](int q) {
    int sc = 2;
    while (sc < q) sc <<= 1;
    return sc;
  };
  while (quadtree_lds_bytes(lds_nodes, 0, sort_cap_of(lds_nodes)) > 120 * 1024) lds_nodes -= 8;
  if (const char* env = getenv("ORBFE_QT_LDS_NODES")) lds_nodes = std::max(64, std::min(lds_nodes, atoi(env)));  // (tests: force the global path)
  int max_lds_quota = 0;
  size_t big_off = 0;
  for (int l = 0; l < nl; ++l) {
    LevelDev& L = c->lv[l];
    L.qt_big_off = 0, L.qt_big_cap = 0, L.qt_big_sort = 0;
    const int cap_l = quota[l] + max_ini + 8;
    if (cap_l <= lds_nodes) {
      max_lds_quota = std::max(max_lds_quota, quota[l]);
      continue;
    }
    int sl = 2;
    while (sl < quota[l]) sl <<= 1;
    L.qt_big_off = (uint32_t)big_off, L.qt_big_cap = cap_l, L.qt_big_sort = sl;
    big_off += align_up((size_t)cap_l * 16 + (size_t)sl * 8, 256);
  }
  c->qt_big_pitch = big_off;
  // the pre-partition's coordinate -> code tables, per level (k_quadtree.hip, quadtree_build_tables)
  c->qt_tabs.clear();
  int pp_nodes = 0;  // node-table entries the pre-partition's scratch needs (it borrows the table: k_quadtree.hip, pp_ok)
  for (int l = 0; l < nl; ++l) {
    std::vector<uint16_t> t;
    c->lv[l].qt_tab_off = 0;
    if (quadtree_build_tables(c->lv[l], t)) {
      if (c->qt_tabs.size() & 1) c->qt_tabs.push_back(0);  // (4-byte aligned tables)
      c->lv[l].qt_tab_off = (uint32_t)c->qt_tabs.size();
      c->qt_tabs.insert(c->qt_tabs.end(), t.begin(), t.end());
      const int ns = c->lv[l].n_ini;
      // bytes of the pre-partition's scratch (k_quadtree.hip, pp_ok): packed group counters + the two tables + the parked totals, in 16-byte nodes
      const size_t pp_bytes = (size_t)std::min(ns * 341, 704) * 4 + ((t.size() + 1) / 2) * 4 + (size_t)((ns * 84 + 3) & ~3) * 2;
      const int need = (int)((pp_bytes + 15) / 16) + 4;
      if (c->lv[l].qt_big_cap == 0) pp_nodes = std::max(pp_nodes, need);
    }
  }
  if (c->qt_tabs.size() < 2) c->qt_tabs.resize(2, 0);
  c->node_cap = std::min(lds_nodes, max_lds_quota + max_ini + 8);
  // The node table is at least as large as the pre-partition's scratch, up to 1024 entries (16 KB): with nFeatures = 800 .. 1600 on
  // a KITTI-sized image the largest quota is below the ~410 entries the scratch of level 0 takes, the pre-partition switched
  // itself off and the trees split their thousands of candidates 64 records a step -- 1.45 ms per 1024 images at nFeatures = 800
  // against 0.40 ms at 2000 (tools/exp/qt_occ.py).
  if (!getenv("ORBFE_QT_LDS_NODES")) c->node_cap = std::max(c->node_cap, std::min(pp_nodes, 1024));  // (1920 x 1080: ~600 entries, 9.5 KB per tree)
  c->node_cap = std::max(c->node_cap, 192);
  c->sort_cap = sort_cap_of(max_lds_quota);
  {
    uint32_t max_cand = 0;
    for (int l = 0; l < nl; ++l) max_cand = std::max(max_cand, c->lv[l].cand_cap);
    const size_t budget = 150 * 1024 - quadtree_lds_bytes(c->node_cap, 0, c->sort_cap);
    c->rec_cap = (int)std::min<size_t>(std::min<size_t>(max_cand, 8192), budget / 4);
    if (const char* env = getenv("ORBFE_QT_REC_CAP")) c->rec_cap = std::max(0, std::min(c->rec_cap, atoi(env)));
    {
      // levels -> waves: longest-processing-time first on the quotas (a tree's work grows with its quota and candidate count).
      // Tables for 1, 2 and 4 waves per image: a launch takes the one that makes ~8 tree waves per CU -- as many as are resident at once
      // (LDS and registers) -- so that the launch is ONE round of waves: same-box, per step of 128 / 256 / 512 pairs, one wave per
      // level | 4 | 2 waves per image: 0.154 | 0.182 | 0.300, 0.300 | 0.210 | 0.349, 0.61 | 0.388 | 0.366 ms (r3, at eight per CU).
      std::memset(&c->qt_single, 0, sizeof c->qt_single);
      for (int l = 0; l < nl; ++l) c->qt_single.mask[l] = 1u << l;
      std::vector<int> order(nl);
      for (int l = 0; l < nl; ++l) order[l] = l;
      // A tree's cost: half of it follows the level's CANDIDATES (the two passes of the pre-partition, the winner scan: ~ the level's
      // area), half its QUOTA (pops, the sort) -- r5 stamps of a level-0 tree, tools/exp/qt_stamps.sh.  Dealt by quota alone (r3 - r4) the
      // level-0 tree shared its wave with level 7 and that wave was the launch's critical path: {0,7} {1,6} {2,5} {3,4} -> {0} {1,6} {2,5}
      // {3,4,7} for 1241 x 376, the longest wave 15 % shorter.
      std::vector<double> cost(nl);
      for (int l = 0; l < nl; ++l)
        cost[l] = (double)c->lv[l].quota / std::max(1, c->lv[0].quota) +
                  ((double)c->lv[l].reg_w * c->lv[l].reg_h) / std::max(1.0, (double)c->lv[0].reg_w * c->lv[0].reg_h);
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
      auto deal = [&](int ng, QtGroups* out) {
        std::memset(out, 0, sizeof *out);
        std::vector<double> load(ng, 0.0);
        for (int l : order) {
          int g = 0;
          for (int k = 1; k < ng; ++k)
            if (load[k] < load[g]) g = k;
          load[g] += cost[l] + 1e-3;
          out->mask[g] |= 1u << l;
        }
        // several waves per image: they pull the levels in cost order instead (k_quadtree.hip: quadtree_levels); the masks above remain
        // what a launch without the counter uses
        out->n_order = ng > 1 ? nl : 0;
        for (int k = 0; k < nl; ++k) out->order[k] = (uint8_t)order[k];
      };
      for (int k = 0; k < 3; ++k) deal(std::min(nl, 1 << k), &c->qt_groups_of[k]);
    }
  }
  // umax (ORBExtractor::initMaxU)
  {
    const int R = ORBFE_CENTROID_R;
    int v, v0, vmax = cv_floor_f(R * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil_f(R * std::sqrt(2.f) / 2);
    const double hp2 = R * R;
    for (v = 0; v <= R; ++v) c->umax[v] = 0;
    for (v = 0; v <= vmax; ++v) c->umax[v] = cv_round_d(std::sqrt(hp2 - v * v));
    for (v = R, v0 = 0; v >= vmin; --v) {
      while (c->umax[v0] == c->umax[v0 + 1]) ++v0;
      c->umax[v] = v0;
      ++v0;
    }
  }
  for (int i = 0; i < 7; ++i) c->blur_taps[i] = kGaussTaps[cfg.blur_variant ? 1 : 0][i];
  return ORBFE_OK;
}


orbfe_status ensure_tmp(orbfe_ctx* c, size_t bytes) {
  if (bytes <= c->tmp_bytes) return ORBFE_OK;
  if (c->d_tmp) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(c->d_tmp));
    c->d_tmp = nullptr;
    c->tmp_bytes = 0;
  }
  bytes = align_up(bytes, 1 << 20);
  HIP_TRY(c, hipMalloc(&c->d_tmp, bytes));
  c->tmp_bytes = bytes;
  return ORBFE_OK;
}

// Pinned staging for the host-pointer API, grown on demand.  Layout per image: level-0 plane with the device row pitch
// (one contiguous DMA instead of a pageable 2-D copy), then the full keypoint and descriptor arrays.
orbfe_status ensure_stage(orbfe_ctx* c, orbfe_ctx::Lane& ln, size_t bytes) {
  if (bytes <= ln.h_stage_bytes) return ORBFE_OK;
  if (ln.h_stage) {
    HIP_TRY(c, hipStreamSynchronize(ln.stream));
    HIP_TRY(c, hipHostFree(ln.h_stage));
    ln.h_stage = nullptr;
    ln.h_stage_bytes = 0;
  }
  bytes = align_up(bytes, 1 << 20);
  HIP_TRY(c, hipHostMalloc((void**)&ln.h_stage, bytes, hipHostMallocDefault));
  ln.h_stage_bytes = bytes;
  return ORBFE_OK;
}
orbfe_status ensure_stage(orbfe_ctx* c, size_t bytes) { return ensure_stage(c, c->main, bytes); }


void drain_timers(orbfe_ctx* c) {
  for (auto& p : c->pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.second.second) == hipSuccess && hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
      c->stage_ms[p.first] += ms;
      c->stage_launches[p.first] += 1;
    }
    c->ev_pool.push_back(p.second.first);
    c->ev_pool.push_back(p.second.second);
  }
  c->pending.clear();
}

// make the context stream wait for a stereo match still running on the stereo stream (every entry point that reads or
// rewrites per-slot data calls this first; the pipelined batch call places the wait later, see run_extract)
orbfe_status join_stereo(orbfe_ctx* c) {
  if (c->stereo_pending) {
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_stereo_done, 0));
    c->stereo_pending = false;
  }
  return ORBFE_OK;
}

// ====================================================================================================
extern "C" {

int orbfe_abi_version(void) { return ORBFE_ABI_VERSION; }

const char* orbfe_last_error(const orbfe_ctx*) { return g_last_error.c_str(); }

const char* orbfe_stage_name(int32_t stage) {
  static const char* names[ORBFE_STAGE_COUNT] = {"resize", "blur", "fast", "quadtree", "orient_brief", "stereo", "match", "ba"};
  return (stage >= 0 && stage < ORBFE_STAGE_COUNT) ? names[stage] : "?";
}

void orbfe_destroy(orbfe_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  // every stream that may still touch the buffers freed below
  if (c->hs.h2d) (void)hipStreamSynchronize(c->hs.h2d);
  for (auto& sl : c->slot_lane)
    if (sl && sl->stream) (void)hipStreamSynchronize(sl->stream);
  if (c->stereo_stream) (void)hipStreamSynchronize(c->stereo_stream);
  if (c->blur_stream) (void)hipStreamSynchronize(c->blur_stream);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->hs.d2h) (void)hipStreamSynchronize(c->hs.d2h);
  drain_timers(c);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  void* ptrs[] = {c->d_lv,   c->d_cells,     c->d_taps,   c->d_rs_tiles, c->d_pattern, c->d_pyr,     c->d_blur,
                  c->d_scr_a, c->d_scr_c,   c->d_scr_b,  c->d_sel,     c->d_sel_count, c->d_n_cand, c->d_n_cand_sh, c->d_n_kp,
                  c->d_kps,  c->d_desc,      c->d_aux,    c->d_theta, c->d_moments, c->d_sincos, c->d_kx, c->d_kpl,   c->d_right_u, c->d_depth, c->d_n_match,
                  c->d_best_right, c->d_best_dist, c->d_tmp, c->d_rowoff, c->d_rowlist, c->d_rowoff_slot, c->d_rowlist_slot, c->d_rt_flags, c->d_rs_regions, c->d_rg_xtaps, c->d_rg_ytaps, c->d_qt_big, c->d_qt_tabs, c->d_qt_next, c->d_mb_tx, c->d_mb_ty, c->st_rows.lrow_off, c->st_rows.lrow_list, c->st_rows.work, c->st_rows.work_n};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (c->h_counts) (void)hipHostFree(c->h_counts);
  if (c->h_abort) (void)hipHostFree((void*)c->h_abort);
  if (c->h_lm_state) (void)hipHostFree(c->h_lm_state);
  if (c->hs.h2d) (void)hipStreamSynchronize(c->hs.h2d);
  if (c->hs.d2h) (void)hipStreamSynchronize(c->hs.d2h);
  for (int b = 0; b < orbfe_ctx::HostStream::kDepth; ++b) {
    if (c->hs.d_in[b]) (void)hipFree(c->hs.d_in[b]);
    if (c->hs.d_out[b]) (void)hipFree(c->hs.d_out[b]);
    for (hipEvent_t e : {c->hs.ev_h2d[b], c->hs.ev_in_free[b], c->hs.ev_out_ready[b], c->hs.ev_done[b]})
      if (e) (void)hipEventDestroy(e);
  }
  if (c->hs.h2d) (void)hipStreamDestroy(c->hs.h2d);
  if (c->hs.d2h) (void)hipStreamDestroy(c->hs.d2h);
  for (auto& sl : c->slot_lane)
    if (sl) {
      if (sl->stream) (void)hipStreamSynchronize(sl->stream);
      for (auto& ge : sl->graphs) (void)hipGraphExecDestroy(ge.exec);
      if (sl->h_stage) (void)hipHostFree(sl->h_stage);
      if (sl->ev_main) (void)hipEventDestroy(sl->ev_main);
      if (sl->stream) (void)hipStreamDestroy(sl->stream);
    }
  c->slot_lane.clear();
  if (c->main.h_stage) (void)hipHostFree(c->main.h_stage);
  for (auto& ge : c->main.graphs) (void)hipGraphExecDestroy(ge.exec);
  c->main.graphs.clear();
  if (c->ev_blur_go) (void)hipEventDestroy(c->ev_blur_go);
  if (c->ev_blur_done) (void)hipEventDestroy(c->ev_blur_done);
  if (c->blur_stream) (void)hipStreamDestroy(c->blur_stream);
  if (c->ev_fast_go) (void)hipEventDestroy(c->ev_fast_go);
  if (c->ev_fast_side_done) (void)hipEventDestroy(c->ev_fast_side_done);
  if (c->fast_stream) (void)hipStreamDestroy(c->fast_stream);
  if (c->stereo_stream) (void)hipStreamDestroy(c->stereo_stream);
  if (c->ev_brief_done) (void)hipEventDestroy(c->ev_brief_done);
  if (c->ev_stereo_done) (void)hipEventDestroy(c->ev_stereo_done);
  if (c->d_pyr_alt) (void)hipFree(c->d_pyr_alt);
  if (c->d_grid_off) (void)hipFree(c->d_grid_off);
  if (c->d_grid_feat) (void)hipFree(c->d_grid_feat);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

orbfe_status orbfe_create(const orbfe_config* cfg, orbfe_ctx** out) {
  if (!cfg || !out) return fail(nullptr, ORBFE_EBADARG, "orbfe_create: NULL argument");
  *out = nullptr;
  // 4096: candidate records carry x and y in 12 bits each (ORBFE_REC_*), the matcher's KpX record the patch centre in 14
  if (cfg->width <= 0 || cfg->height <= 0 || cfg->width > 4096 || cfg->height > 4096 || cfg->n_features < 0 || cfg->n_features > 65535 || cfg->n_levels < 1 || cfg->n_levels > ORBFE_MAX_LEVELS ||
      !(cfg->scale_factor > 1.0f) || cfg->max_images < 1 || cfg->max_images > 65535)
    return fail(nullptr, ORBFE_EBADARG, "orbfe_create: bad config (w=%d h=%d nfeat=%d levels=%d scale=%g max_images=%d)", cfg->width,
                cfg->height, cfg->n_features, cfg->n_levels, (double)cfg->scale_factor, cfg->max_images);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, ORBFE_EDEVICE, "orbfe_create: no HIP device (this library has no CPU fallback)");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, ORBFE_EBADARG, "orbfe_create: device %d of %d", cfg->device_id, ndev);
  orbfe_ctx* c = new orbfe_ctx();
  c->cfg = *cfg;
  c->cfg.fast_hi = std::min(std::max(cfg->fast_hi, 0), 255);  // cv::FAST clamps the threshold
  c->cfg.fast_lo = std::min(std::max(cfg->fast_lo, 0), 255);
  c->device = cfg->device_id;
  std::memset(c->stage_ms, 0, sizeof c->stage_ms);
  std::memset(c->stage_launches, 0, sizeof c->stage_launches);
  auto bail = [&](orbfe_status st) {
    const std::string keep = g_last_error;
    orbfe_destroy(c);
    g_last_error = keep;
    return st;
  };
  orbfe_status st = build_geometry(c);
  if (st != ORBFE_OK) return bail(st);
  if (c->kp_cap > 65535) {  // the stereo matcher's row table and its (distance, index) keys hold a keypoint index in 16 bits
    fail(c, ORBFE_EBADSIZE, "orbfe_create: %d keypoints per image exceed 65535", c->kp_cap);
    return bail(ORBFE_EBADSIZE);
  }
  c->cfg.n_features = c->kp_cap;  // from here on n_features is the per-image array stride (the quotas keep the requested value)
  if (hipSetDevice(c->device) != hipSuccess) {
    fail(c, ORBFE_EDEVICE, "hipSetDevice(%d) failed", c->device);
    return bail(ORBFE_EDEVICE);
  }
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && ncu > 0) c->n_cu = ncu;
  }
  if (cfg->stream) {
    c->stream = (hipStream_t)cfg->stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      fail(c, ORBFE_EDEVICE, "hipStreamCreate failed");
      return bail(ORBFE_EDEVICE);
    }
    c->own_stream = true;
  }
  c->main.stream = c->stream;
  c->slot_lane.resize((size_t)cfg->max_images);
  {
    // (streams are not free: HIP multiplexes all of a process's streams onto a few hardware queues, and two busy streams on one queue
    //  serialise -- none is created that the schedule does not use.  Cutting a batch into chunks on streams of their own was measured
    //  in rounds 1-3 and never won against the one-chunk schedule below: profiles/NOTES_r1-r3.md)
    if (const char* gr = getenv("ORBFE_GRAPHS")) c->use_graphs = atoi(gr) != 0;
    {
      const char* ps = getenv("ORBFE_PIPELINE_STEREO");
      c->pipeline_stereo = !ps || atoi(ps) != 0;
      if (c->pipeline_stereo &&
          (hipStreamCreateWithFlags(&c->stereo_stream, hipStreamNonBlocking) != hipSuccess ||
           hipEventCreateWithFlags(&c->ev_brief_done, hipEventDisableTiming) != hipSuccess ||
           hipEventCreateWithFlags(&c->ev_stereo_done, hipEventDisableTiming) != hipSuccess)) {
        fail(c, ORBFE_EDEVICE, "cannot create the stereo stream");
        return bail(ORBFE_EDEVICE);
      }
    }
    if (const char* hl = getenv("ORBFE_LBA_HOST_LM")) c->lm_on_device = atoi(hl) == 0;
    if (const char* fc = getenv("ORBFE_FAST_CPW")) c->fast_cpw = std::max(0, std::min(64, atoi(fc)));
    if (const char* fm = getenv("ORBFE_FAST_SIDE_MASK")) c->fast_side_mask = (int64_t)(strtoul(fm, nullptr, 0) & 0xFFFFu);
    if (const char* fg = getenv("ORBFE_FAST_SIDE_MERGE")) c->fast_side_merge = atoi(fg) != 0;
    const char* ov = getenv("ORBFE_OVERLAP_BLUR");
    if (!ov || atoi(ov) != 0) {
      if (hipStreamCreateWithFlags(&c->blur_stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&c->ev_blur_go, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&c->ev_blur_done, hipEventDisableTiming) != hipSuccess) {
        fail(c, ORBFE_EDEVICE, "cannot create the blur stream");
        return bail(ORBFE_EDEVICE);
      }
      // (the side stream of FAST's small levels: only contexts that can hold a batch ever use it)
      if (c->fast_side_mask != 0 && cfg->max_images >= 32 &&
          (hipStreamCreateWithFlags(&c->fast_stream, hipStreamNonBlocking) != hipSuccess ||
           hipEventCreateWithFlags(&c->ev_fast_go, hipEventDisableTiming) != hipSuccess ||
           hipEventCreateWithFlags(&c->ev_fast_side_done, hipEventDisableTiming) != hipSuccess)) {
        fail(c, ORBFE_EDEVICE, "cannot create the FAST side stream");
        return bail(ORBFE_EDEVICE);
      }
    }
  }
  const size_t M = (size_t)cfg->max_images, NF = (size_t)std::max(c->cfg.n_features, 1), NL = (size_t)cfg->n_levels;
  const size_t NP = (M + 1) / 2;
#define ALLOC(ptr, count)                          \
  do {                                             \
    st = dev_alloc(c, &(ptr), (count));            \
    if (st != ORBFE_OK) return bail(st);           \
  } while (0)
  ALLOC(c->d_lv, NL);
  ALLOC(c->d_cells, c->cells.size());
  ALLOC(c->d_taps, c->taps.size());
  ALLOC(c->d_rs_tiles, c->rs_tile_tab.size());
  ALLOC(c->d_rs_regions, std::max<size_t>(c->rs_regions.size(), 1));
  ALLOC(c->d_rg_xtaps, std::max<size_t>(c->rg_xtaps.size(), 2));
  ALLOC(c->d_rg_ytaps, std::max<size_t>(c->rg_ytaps.size(), 1));
  ALLOC(c->d_pattern, 1024);
  ALLOC(c->d_pyr, M * c->img_pitch);
  ALLOC(c->d_blur, M * c->img_pitch);
  ALLOC(c->d_scr_a, M * c->scratch_pitch);
  ALLOC(c->d_scr_b, M * c->scratch_pitch);
  ALLOC(c->d_scr_c, M * c->scratch_pitch);
  if (c->qt_big_pitch) ALLOC(c->d_qt_big, M * c->qt_big_pitch);
  ALLOC(c->d_qt_tabs, c->qt_tabs.size());
  ALLOC(c->d_sel, M * NF);
  ALLOC(c->d_sel_count, M * NL);
  ALLOC(c->d_qt_next, M);
  {
    const char* mbe = getenv("ORBFE_BLUR_MFMA");
    c->mb_ok = (!mbe || atoi(mbe) != 0) && mb_build(c->lv.data(), (int)NL, c->blur_taps, c->pyr_spare_off, &c->mb, &c->mb_tx, &c->mb_ty);
    if (c->mb_ok) {
      ALLOC(c->d_mb_tx, c->mb_tx.size());
      ALLOC(c->d_mb_ty, c->mb_ty.size());
    }
  }
  ALLOC(c->d_n_cand, M * NL);
  if (c->fast_shards > 1) ALLOC(c->d_n_cand_sh, M * NL * (size_t)c->fast_shards);
  c->slot_sharded.assign((size_t)M, 0);
  ALLOC(c->d_n_kp, M);
  ALLOC(c->d_kps, M * NF);
  ALLOC(c->d_desc, M * NF * 32);
  ALLOC(c->d_aux, M * NF);
  ALLOC(c->d_theta, M * NF);
  ALLOC(c->d_moments, M * NF);
  ALLOC(c->d_kpl, M * NF);
  ALLOC(c->d_sincos, M * NF);
  ALLOC(c->d_kx, M * NF);
  {
    // widest band of createRowIndexDB: rows rn(y - r) .. rn(y + r + 1) - 1 with r = 2 * scale of the coarsest level
    float sf_max = 1.f;
    for (int l = 0; l < NL; ++l) sf_max = std::max(sf_max, c->lv[l].sf);
    c->row_list_cap = (int)NF * ((int)(4.0f * sf_max) + 4);
  }
  ALLOC(c->d_rowoff, NP * (size_t)(c->cfg.height + 1));
  ALLOC(c->st_rows.lrow_off, NP * (size_t)(c->cfg.height + 1));
  ALLOC(c->st_rows.lrow_list, NP * NF);
  ALLOC(c->st_rows.work, NP * NF);
  ALLOC(c->st_rows.work_n, NP);
  ALLOC(c->d_rowlist, NP * (size_t)c->row_list_cap);
  c->grid_key.reset(new std::atomic<uint64_t>[M]);  // (here, not on first use: slot calls on other threads clear entries without the API lock)
  for (size_t k = 0; k < M; ++k) c->grid_key[k] = 0;
  if (M <= 16 && ((size_t)c->cfg.height + 4) * 4 <= 9000) {  // (k_brief's table workgroup borrows 9000 bytes of the descriptor kernel's LDS)
    ALLOC(c->d_rowoff_slot, M * (size_t)(c->cfg.height + 1));
    if (!getenv("ORBFE_ROWTABLE_ONE_WG")) ALLOC(c->d_rt_flags, M * 8);  // (1: r5's one-workgroup table in k_brief's small launches)
    ALLOC(c->d_rowlist_slot, M * (size_t)c->row_list_cap);
    c->slot_table_ok.reset(new std::atomic<uint8_t>[M]);
    c->pair_count_zero.reset(new std::atomic<uint8_t>[M]);
    for (size_t k = 0; k < M; ++k) c->slot_table_ok[k] = 0, c->pair_count_zero[k] = 0;
  }
  ALLOC(c->d_right_u, NP * NF);
  ALLOC(c->d_depth, NP * NF);
  ALLOC(c->d_n_match, NP);
  ALLOC(c->d_best_right, NP * NF);
  ALLOC(c->d_best_dist, NP * NF);
#undef ALLOC
  hipError_t e = hipSuccess;
  const int8_t* pat = cfg->brief_pairs ? cfg->brief_pairs : &kEmbeddedPattern[0][0];
  for (int i = 0; i < 512; ++i) {
    const int px = pat[2 * i], py = pat[2 * i + 1];
    if (px * px + py * py > 338) {  // 13^2 + 13^2: the rotated point must stay within 18 px (image border 19, LDS window of k_brief)
      fail(c, ORBFE_EBADARG, "BRIEF template point (%d,%d) is farther than sqrt(338) px from the centre", px, py);
      return bail(ORBFE_EBADARG);
    }
  }
  c->cfg.brief_pairs = nullptr;  // not retained
  if (e == hipSuccess) e = hipMemcpy(c->d_lv, c->lv.data(), sizeof(LevelDev) * NL, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->d_cells, c->cells.data(), sizeof(CellDev) * c->cells.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->d_qt_tabs, c->qt_tabs.data(), sizeof(uint16_t) * c->qt_tabs.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && !c->taps.empty()) e = hipMemcpy(c->d_taps, c->taps.data(), sizeof(ResizeTap) * c->taps.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && !c->rs_tile_tab.empty())
    e = hipMemcpy(c->d_rs_tiles, c->rs_tile_tab.data(), sizeof(RsTile) * c->rs_tile_tab.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && !c->rs_regions.empty())
    e = hipMemcpy(c->d_rs_regions, c->rs_regions.data(), sizeof(RsRegion) * c->rs_regions.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && !c->rg_xtaps.empty())
    e = hipMemcpy(c->d_rg_xtaps, c->rg_xtaps.data(), sizeof(RgXTap) * c->rg_xtaps.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && !c->rg_ytaps.empty())
    e = hipMemcpy(c->d_rg_ytaps, c->rg_ytaps.data(), sizeof(RgYTap) * c->rg_ytaps.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && c->mb_ok) e = hipMemcpy(c->d_mb_tx, c->mb_tx.data(), c->mb_tx.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && c->mb_ok) e = hipMemcpy(c->d_mb_ty, c->mb_ty.data(), c->mb_ty.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->d_pattern, pat, 1024, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(c->d_n_kp, 0, sizeof(int32_t) * M);
  if (e == hipSuccess) e = hipMemset(c->d_sel_count, 0, sizeof(int32_t) * M * NL);
  if (e == hipSuccess) e = hipMemset(c->d_n_match, 0, sizeof(int32_t) * NP);
  if (e == hipSuccess) e = hipMemset(c->d_pyr, 0, M * c->img_pitch);
  if (e == hipSuccess) e = hipMemset(c->d_blur, 0, M * c->img_pitch);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_counts, sizeof(int32_t) * std::max<size_t>(M, 64), hipHostMallocDefault);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    fail(c, ORBFE_EDEVICE, "device initialisation failed: %s", hipGetErrorString(e));
    return bail(ORBFE_EDEVICE);
  }
  e = quadtree_configure(quadtree_lds_bytes(c->node_cap, c->rec_cap, c->sort_cap) + 2048 + 4096);  // (+ the rank buffer and the level-4 totals of the several-waves-per-tree launches)
  if (e != hipSuccess) {
    fail(c, ORBFE_EDEVICE, "cannot reserve %zu B of LDS for the quadtree kernel: %s", quadtree_lds_bytes(c->node_cap, c->rec_cap, c->sort_cap),
         hipGetErrorString(e));
    return bail(ORBFE_EDEVICE);
  }
  *out = c;
  return ORBFE_OK;
}

orbfe_status orbfe_get_level_info(const orbfe_ctx* c, int32_t level, orbfe_level_info* out) {
  if (!c || !out || level < 0 || level >= c->cfg.n_levels) return ORBFE_EBADARG;
  const LevelDev& L = c->lv[level];
  out->width = L.w;
  out->height = L.h;
  out->scale = L.sf;
  out->quota = L.quota;
  out->grid_cols = L.n_cols;
  out->grid_rows = L.n_rows;
  out->cell_w = L.w_cell;
  out->cell_h = L.h_cell;
  return ORBFE_OK;
}

orbfe_status orbfe_get_scale_factors(const orbfe_ctx* c, float* out, int32_t n) {
  if (!c || !out || n < c->cfg.n_levels) return ORBFE_EBADARG;
  for (int l = 0; l < c->cfg.n_levels; ++l) out[l] = c->lv[l].sf;
  return ORBFE_OK;
}

int32_t orbfe_get_capacity(const orbfe_ctx* c) { return c ? c->kp_cap : 0; }

orbfe_status orbfe_sync(orbfe_ctx* c) {
  ApiLock api_lk(c);
  if (!c) return ORBFE_EBADARG;
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  return ORBFE_OK;
}

orbfe_status orbfe_fetch_features(orbfe_ctx* c, int32_t slot, orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out) {
  ApiLock api_lk(c);
  if (!c || slot < 0 || slot >= c->cfg.max_images) return fail(c, ORBFE_EBADARG, "fetch_features: slot %d", slot);
  TRY(slots_idle(c, slot, 1, "fetch_features"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t NF = (size_t)c->cfg.n_features;
  // count, keypoints and descriptors of the whole slot into the page-locked staging buffer behind ONE synchronisation (the count first and
  // then exactly n entries into pageable memory were two round trips)
  const size_t h_k = 256, h_d = h_k + align_up(NF * sizeof(orbfe_keypoint), 256), h_total = h_d + align_up(NF * 32, 256);
  TRY(ensure_stage(c, h_total));
  uint8_t* hs = c->main.h_stage;
  HIP_TRY(c, hipMemcpyAsync(hs, c->d_n_kp + slot, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (kps && NF) HIP_TRY(c, hipMemcpyAsync(hs + h_k, c->d_kps + (size_t)slot * NF, sizeof(orbfe_keypoint) * NF, hipMemcpyDeviceToHost, c->stream));
  if (desc && NF) HIP_TRY(c, hipMemcpyAsync(hs + h_d, c->d_desc + (size_t)slot * NF * 32, (size_t)32 * NF, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  int32_t n = 0;
  std::memcpy(&n, hs, 4);
  if (n < 0 || (size_t)n > NF) return fail(c, ORBFE_EDEVICE, "fetch_features: corrupt count %d", n);
  if (kps && n) std::memcpy(kps, hs + h_k, sizeof(orbfe_keypoint) * (size_t)n);
  if (desc && n) std::memcpy(desc, hs + h_d, (size_t)32 * n);
  if (n_out) *n_out = n;
  return ORBFE_OK;
}

orbfe_status orbfe_fetch_stereo(orbfe_ctx* c, int32_t pair, double* right_u, double* depth, int32_t* n_matches, int32_t* best_right,
                                int32_t* best_dist) {
  ApiLock api_lk(c);
  if (!c || pair < 0 || pair >= (c->cfg.max_images + 1) / 2) return fail(c, ORBFE_EBADARG, "fetch_stereo: pair %d", pair);
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t NF = (size_t)c->cfg.n_features;
  const size_t o = (size_t)pair * NF;
  const size_t o_ru = 0, o_dp = align_up(NF * 8, 256), o_br = o_dp + align_up(NF * 8, 256), o_bd = o_br + align_up(NF * 4, 256),
               o_nm = o_bd + align_up(NF * 4, 256), total = o_nm + 256;
  TRY(ensure_stage(c, total));
  uint8_t* h = c->main.h_stage;
  if (right_u && NF) HIP_TRY(c, hipMemcpyAsync(h + o_ru, c->d_right_u + o, sizeof(double) * NF, hipMemcpyDeviceToHost, c->stream));
  if (depth && NF) HIP_TRY(c, hipMemcpyAsync(h + o_dp, c->d_depth + o, sizeof(double) * NF, hipMemcpyDeviceToHost, c->stream));
  if (best_right && NF) HIP_TRY(c, hipMemcpyAsync(h + o_br, c->d_best_right + o, sizeof(int32_t) * NF, hipMemcpyDeviceToHost, c->stream));
  if (best_dist && NF) HIP_TRY(c, hipMemcpyAsync(h + o_bd, c->d_best_dist + o, sizeof(int32_t) * NF, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(h + o_nm, c->d_n_match + pair, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  if (right_u && NF) std::memcpy(right_u, h + o_ru, sizeof(double) * NF);
  if (depth && NF) std::memcpy(depth, h + o_dp, sizeof(double) * NF);
  if (best_right && NF) std::memcpy(best_right, h + o_br, sizeof(int32_t) * NF);
  if (best_dist && NF) std::memcpy(best_dist, h + o_bd, sizeof(int32_t) * NF);
  if (n_matches) std::memcpy(n_matches, h + o_nm, sizeof(int32_t));
  return ORBFE_OK;
}

// Bulk fetches: the packed result arrays of a range of slots / pairs in one copy each (full [n_features] strides), one synchronisation.
orbfe_status orbfe_fetch_batch(orbfe_ctx* c, int32_t slot0, int32_t n_slots, orbfe_keypoint* kps, uint8_t* desc, int32_t* counts) {
  ApiLock api_lk(c);
  if (!c || slot0 < 0 || n_slots < 0 || slot0 + n_slots > c->cfg.max_images) return fail(c, ORBFE_EBADARG, "fetch_batch: slots [%d, %d)", slot0, slot0 + n_slots);
  if (n_slots == 0) return ORBFE_OK;
  TRY(slots_idle(c, slot0, n_slots, "fetch_batch"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1), s0 = (size_t)slot0, n = (size_t)n_slots;
  if (counts) HIP_TRY(c, hipMemcpyAsync(counts, c->d_n_kp + s0, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
  if (kps) HIP_TRY(c, hipMemcpyAsync(kps, c->d_kps + s0 * NF, sizeof(orbfe_keypoint) * n * NF, hipMemcpyDeviceToHost, c->stream));
  if (desc) HIP_TRY(c, hipMemcpyAsync(desc, c->d_desc + s0 * NF * 32, n * NF * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  return ORBFE_OK;
}

orbfe_status orbfe_fetch_stereo_batch(orbfe_ctx* c, int32_t pair0, int32_t n_pairs, double* right_u, double* depth, int32_t* n_matches) {
  ApiLock api_lk(c);
  if (!c || pair0 < 0 || n_pairs < 0 || pair0 + n_pairs > (c->cfg.max_images + 1) / 2)
    return fail(c, ORBFE_EBADARG, "fetch_stereo_batch: pairs [%d, %d)", pair0, pair0 + n_pairs);
  if (n_pairs == 0) return ORBFE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1), p0 = (size_t)pair0, n = (size_t)n_pairs;
  if (right_u) HIP_TRY(c, hipMemcpyAsync(right_u, c->d_right_u + p0 * NF, sizeof(double) * n * NF, hipMemcpyDeviceToHost, c->stream));
  if (depth) HIP_TRY(c, hipMemcpyAsync(depth, c->d_depth + p0 * NF, sizeof(double) * n * NF, hipMemcpyDeviceToHost, c->stream));
  if (n_matches) HIP_TRY(c, hipMemcpyAsync(n_matches, c->d_n_match + p0, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  return ORBFE_OK;
}

orbfe_status orbfe_device_results(orbfe_ctx* c, const void** d_kps, const void** d_desc, const void** d_counts, const void** d_right_u,
                                  const void** d_depth, const void** d_nmatch) {
  ApiLock api_lk(c);
  if (!c) return ORBFE_EBADARG;
  if (d_kps) *d_kps = c->d_kps;
  if (d_desc) *d_desc = c->d_desc;
  if (d_counts) *d_counts = c->d_n_kp;
  if (d_right_u) *d_right_u = c->d_right_u;
  if (d_depth) *d_depth = c->d_depth;
  if (d_nmatch) *d_nmatch = c->d_n_match;
  return ORBFE_OK;
}
orbfe_status orbfe_profile_enable(orbfe_ctx* c, int32_t on) {
  ApiLock api_lk(c);
  if (!c) return ORBFE_EBADARG;
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  c->prof = on < 0 ? 0 : on;
  return ORBFE_OK;
}

orbfe_status orbfe_profile_read(orbfe_ctx* c, double* ms, int64_t* launches, int32_t reset) {
  ApiLock api_lk(c);
  if (!c) return ORBFE_EBADARG;
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  for (int i = 0; i < ORBFE_STAGE_COUNT; ++i) {
    if (ms) ms[i] = c->stage_ms[i];
    if (launches) launches[i] = c->stage_launches[i];
    if (reset) {
      c->stage_ms[i] = 0;
      c->stage_launches[i] = 0;
    }
  }
  return ORBFE_OK;
}

orbfe_status orbfe_debug_candidates(orbfe_ctx* c, int32_t slot, int32_t level, float* xyr, int32_t cap, int32_t* n_out) {
  ApiLock api_lk(c);
  if (!c || slot < 0 || slot >= c->cfg.max_images || level < 0 || level >= c->cfg.n_levels || !n_out)
    return fail(c, ORBFE_EBADARG, "debug_candidates: bad argument");
  TRY(slots_idle(c, slot, 1, "debug_candidates"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const LevelDev& L = c->lv[level];
  int32_t n = 0;
  std::vector<uint32_t> rec;
  if (c->fast_shards > 1 && c->slot_sharded[(size_t)slot]) {  // the level's list in shards (a frame or two): gathered here
    const int ns = c->fast_shards;
    std::vector<int32_t> cnt((size_t)ns, 0);
    HIP_TRY(c, hipMemcpyAsync(cnt.data(), c->d_n_cand_sh + ((size_t)slot * c->cfg.n_levels + level) * ns, sizeof(int32_t) * ns, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < ns; ++s) {
      const int32_t k = std::min<int32_t>(std::max(cnt[(size_t)s], 0), (int32_t)L.shard_cap);
      if (!k) continue;
      rec.resize((size_t)n + k);
      HIP_TRY(c, hipMemcpyAsync(rec.data() + n, c->d_scr_a + (size_t)slot * c->scratch_pitch + L.cand_base + (size_t)s * L.shard_cap, sizeof(uint32_t) * k,
                                hipMemcpyDeviceToHost, c->stream));
      n += k;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (rec.empty()) rec.resize(1);
  } else {
    HIP_TRY(c, hipMemcpyAsync(&n, c->d_n_cand + (size_t)slot * c->cfg.n_levels + level, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    n = std::min<int32_t>(std::max(n, 0), (int32_t)L.cand_cap);
    rec.resize((size_t)std::max(n, 1));
    if (n) HIP_TRY(c, hipMemcpyAsync(rec.data(), c->d_scr_a + (size_t)slot * c->scratch_pitch + L.cand_base, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  // reference order = (cell row, cell col, y, x)
  std::vector<std::pair<uint64_t, uint32_t>> keyed((size_t)n);
  for (int i = 0; i < n; ++i) {
    const uint32_t x = ORBFE_REC_X(rec[i]), y = ORBFE_REC_Y(rec[i]);
    const uint32_t jdx = std::min<uint32_t>((x - 3) / L.w_cell, L.n_cols - 1), idx = std::min<uint32_t>((y - 3) / L.h_cell, L.n_rows - 1);
    keyed[i] = {((uint64_t)(idx * L.n_cols + jdx) << 24) | ((uint64_t)y << 12) | x, rec[i]};
  }
  std::sort(keyed.begin(), keyed.end());
  for (int i = 0; i < n && xyr && i < cap; ++i) {
    xyr[3 * i] = (float)ORBFE_REC_X(keyed[i].second);
    xyr[3 * i + 1] = (float)ORBFE_REC_Y(keyed[i].second);
    xyr[3 * i + 2] = (float)ORBFE_REC_R(keyed[i].second);
  }
  *n_out = n;
  return ORBFE_OK;
}

}  // extern "C"

