// orbfe_guided.hip -- the matchers beside the stereo match: brute-force getBestMatch loops, the grid-guided searches (findFeaturesInArea +
// getBestMatch, ORBMatcher.cc:265-347, 561-612), MapPoint::isInVision / predictLevel, and the fused tracking chains
// (orbfe_track_local_map / orbfe_track_motion_model).  (Split from orbfe_api.hip in r5, no change of behaviour.)
#include "orbfe_ctx.h"
extern "C" {


orbfe_status orbfe_match_bruteforce(orbfe_ctx* c, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, const uint32_t* cand_offsets,
                                    const uint32_t* cand_idx, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist) {
  ApiLock api_lk(c);
  if (!c || nq < 0 || nt < 0 || (nq && !q) || (nt && !t) || !best_idx || !best_dist || !second_dist)
    return fail(c, ORBFE_EBADARG, "match_bruteforce: NULL argument");
  if (cand_offsets && !cand_idx && cand_offsets[nq] > 0) return fail(c, ORBFE_EBADARG, "match_bruteforce: cand_idx is NULL");
  if (nq == 0) return ORBFE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t n_cand = cand_offsets ? cand_offsets[nq] : 0;
  if (cand_offsets)
    for (size_t i = 0; i < n_cand; ++i)
      if (cand_idx[i] >= (uint32_t)nt) return fail(c, ORBFE_EBADARG, "match_bruteforce: candidate %u >= nt %d", cand_idx[i], nt);
  size_t o_q = 0, o_t = align_up((size_t)nq * 32, 256), o_off = o_t + align_up((size_t)std::max(nt, 1) * 32, 256);
  size_t o_cand = o_off + align_up(((size_t)nq + 1) * 4, 256), o_bi = o_cand + align_up(std::max<size_t>(n_cand, 1) * 4, 256);
  size_t o_bd = o_bi + align_up((size_t)nq * 4, 256), o_sd = o_bd + align_up((size_t)nq * 4, 256), total = o_sd + align_up((size_t)nq * 4, 256);
  TRY(ensure_tmp(c, total));
  uint8_t* base = (uint8_t*)c->d_tmp;
  // up to 8 MB: one upload and one download through the page-locked staging buffer (seven copies from / to pageable memory otherwise)
  const bool staged = total <= ((size_t)8 << 20);
  uint8_t* hs = nullptr;
  if (staged) {
    TRY(ensure_stage(c, total));
    hs = c->main.h_stage;
    std::memcpy(hs + o_q, q, (size_t)nq * 32);
    if (nt) std::memcpy(hs + o_t, t, (size_t)nt * 32);
    if (cand_offsets) {
      std::memcpy(hs + o_off, cand_offsets, ((size_t)nq + 1) * 4);
      if (n_cand) std::memcpy(hs + o_cand, cand_idx, n_cand * 4);
    }
    HIP_TRY(c, hipMemcpyAsync(base, hs, o_bi, hipMemcpyHostToDevice, c->stream));
  } else {
    HIP_TRY(c, hipMemcpyAsync(base + o_q, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    if (nt) HIP_TRY(c, hipMemcpyAsync(base + o_t, t, (size_t)nt * 32, hipMemcpyHostToDevice, c->stream));
    if (cand_offsets) {
      HIP_TRY(c, hipMemcpyAsync(base + o_off, cand_offsets, ((size_t)nq + 1) * 4, hipMemcpyHostToDevice, c->stream));
      if (n_cand) HIP_TRY(c, hipMemcpyAsync(base + o_cand, cand_idx, n_cand * 4, hipMemcpyHostToDevice, c->stream));
    }
  }
  {
    StageTimer tm(c, ORBFE_STAGE_MATCH, c->stream);
    launch_match_bruteforce(c->stream, base + o_q, nq, base + o_t, nt, cand_offsets ? (const uint32_t*)(base + o_off) : nullptr,
                            (const uint32_t*)(base + o_cand), (int32_t*)(base + o_bi), (int32_t*)(base + o_bd), (int32_t*)(base + o_sd));
  }
  HIP_TRY(c, hipGetLastError());
  if (staged) {
    HIP_TRY(c, hipMemcpyAsync(hs, base + o_bi, total - o_bi, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drain_timers(c);
    std::memcpy(best_idx, hs, (size_t)nq * 4);
    std::memcpy(best_dist, hs + (o_bd - o_bi), (size_t)nq * 4);
    std::memcpy(second_dist, hs + (o_sd - o_bi), (size_t)nq * 4);
    return ORBFE_OK;
  }
  HIP_TRY(c, hipMemcpyAsync(best_idx, base + o_bi, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(best_dist, base + o_bd, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(second_dist, base + o_sd, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  return ORBFE_OK;
}

// The grid-guided search against a feature set on the device: the features of an image slot (orbfe_search_in_area) or a set the caller
// uploaded (orbfe_search_in_area_features: a KeyFrame's keypoints and descriptors -- keyframes are not resident in a slot).
// grid of a frame: VirtualFrame::initGrid (Frame.cc:55-56) sizes it from the undistorted bounds, findFeaturesInArea clips the box at
// (int)mfMaxU / (int)mfMaxV (:291-293).  bounds = {min_u, max_u, min_v, max_v}; NULL: the image itself (no distortion: 0, width, 0, height)
struct AreaGrid {
  int rows, cols, clip_w, clip_h;
};
static bool area_grid(const orbfe_ctx* c, const float* bounds, AreaGrid* g) {
  if (!bounds) {
    *g = {(c->cfg.height + 47) / 48, (c->cfg.width + 63) / 64, c->cfg.width, c->cfg.height};
    return true;
  }
  if (!(std::isfinite(bounds[0]) && std::isfinite(bounds[1]) && std::isfinite(bounds[2]) && std::isfinite(bounds[3])) ||
      !(bounds[1] > bounds[0]) || !(bounds[3] > bounds[2]) || bounds[1] > 65536.f || bounds[3] > 65536.f || bounds[1] < 1.f || bounds[3] < 1.f)
    return false;
  *g = {cv_ceil_f((float)(bounds[3] - bounds[2]) / 48), cv_ceil_f((float)(bounds[1] - bounds[0]) / 64), (int)bounds[1], (int)bounds[3]};
  return g->rows >= 1 && g->cols >= 1;
}

// the grid of `slot` for the geometry ag on stream st: the one kept from the last search if the slot's keypoints are still the same
static orbfe_status slot_grid(orbfe_ctx* c, hipStream_t st, int slot, const AreaGrid& ag, const int32_t** d_off, const int32_t** d_feat) {
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1), M = (size_t)c->cfg.max_images, ncells = (size_t)ag.rows * ag.cols;
  if (ncells + 1 > c->grid_cells) {  // first use, or a larger grid than any before: (re)allocate, nothing cached survives
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->d_grid_off) (void)hipFree(c->d_grid_off);
    if (!c->d_grid_feat) HIP_TRY(c, hipMalloc((void**)&c->d_grid_feat, M * NF * sizeof(int32_t)));
    c->d_grid_off = nullptr;
    HIP_TRY(c, hipMalloc((void**)&c->d_grid_off, M * (ncells + 1) * sizeof(int32_t)));
    c->grid_cells = ncells + 1;
    for (size_t k = 0; k < M; ++k) grid_invalidate(c, (int)k);
  }
  int32_t* off = c->d_grid_off + (size_t)slot * c->grid_cells;
  int32_t* feat = c->d_grid_feat + (size_t)slot * NF;
  const uint32_t key = ((uint32_t)ag.rows << 16) | (uint32_t)ag.cols;
  uint64_t seen = c->grid_key[(size_t)slot].load();
  if ((uint32_t)seen != key) {
    launch_grid_build(st, c->d_kps + (size_t)slot * NF, c->d_n_kp + slot, (int)NF, ag.rows, ag.cols, off, feat);
    // kept for the next search only if no extraction touched the slot since `seen` (its generation is part of the compared value); on
    // failure nothing is cached: this call still uses what it built, the next one builds again
    (void)c->grid_key[(size_t)slot].compare_exchange_strong(seen, (seen & 0xFFFFFFFF00000000ull) | key);
  }
  *d_off = off, *d_feat = feat;
  return ORBFE_OK;
}

static orbfe_status search_area_core(orbfe_ctx* c, const char* who, const orbfe_keypoint* d_kps, const int32_t* d_n_kp, const uint4* d_kpl,
                                     const uint8_t* d_desc, size_t n_target, size_t tmp_used, int32_t nq, const float* qxy,
                                     const float* radius, const int8_t* min_level, const int8_t* max_level, const uint8_t* q_desc,
                                     const uint8_t* exclude, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist, int32_t* n_cand,
                                     const float* bounds = nullptr, int32_t* excluded_hits = nullptr, bool staged_prefix = false,
                                     int cache_slot = -1) {
  // staged_prefix: the caller has written the first tmp_used bytes of the scratch into the staging buffer (same offsets): they go up
  // with the queries.  cache_slot >= 0: the target is that slot -- its grid is kept between searches (slot_grid)
  AreaGrid ag;
  if (!area_grid(c, bounds, &ag)) return fail(c, ORBFE_EBADARG, "%s: bad frame bounds", who);
  const int rows = ag.rows, cols = ag.cols;
  const size_t ncells = (size_t)rows * cols;
  if ((2 * ncells + 1) * 4 > 60 * 1024) return fail(c, ORBFE_EBADSIZE, "%s: %zu grid cells exceed the LDS counters", who, ncells);
  const size_t NT = std::max<size_t>(n_target, 1);
  size_t off = tmp_used;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  // queries first (they continue the caller's uploaded block, if any, so that everything goes up as ONE copy through the page-locked
  // staging buffer), then the grid, then the results (one download): ten copies from / to pageable memory were most of a 0.2 ms call
  const size_t o_q = take((size_t)nq * 8), o_r = take((size_t)nq * 4), o_lo = take((size_t)nq), o_hi = take((size_t)nq),
               o_d = take((size_t)nq * 32), o_ex = take(NT), o_in_end = take(8), o_co = take((ncells + 1) * 4), o_cf = take(NT * 4),
               o_bi = take((size_t)nq * 4), o_bd = take((size_t)nq * 4), o_sd = take((size_t)nq * 4), o_nc = take((size_t)nq * 4),
               o_eh = take(NT * 4), o_out_end = take(8);
  if (off > c->tmp_bytes) return fail(c, ORBFE_ENOMEM, "%s: scratch not reserved", who);  // (the callers reserve before they upload)
  uint8_t* b = (uint8_t*)c->d_tmp;
  const bool hits = exclude && excluded_hits;
  const size_t out_bytes = (hits ? o_out_end : o_eh) - o_bi;
  TRY(ensure_stage(c, std::max(o_in_end, out_bytes)));  // (a caller with a staged prefix has reserved at least this much already)
  uint8_t* hs = c->main.h_stage;
  std::memcpy(hs + o_q, qxy, (size_t)nq * 8);
  std::memcpy(hs + o_r, radius, (size_t)nq * 4);
  std::memcpy(hs + o_lo, min_level, (size_t)nq);
  std::memcpy(hs + o_hi, max_level, (size_t)nq);
  std::memcpy(hs + o_d, q_desc, (size_t)nq * 32);
  if (exclude) std::memcpy(hs + o_ex, exclude, n_target);
  const size_t up0 = staged_prefix ? 0 : o_q;
  HIP_TRY(c, hipMemcpyAsync(b + up0, hs + up0, o_in_end - up0, hipMemcpyHostToDevice, c->stream));
  if (hits) HIP_TRY(c, hipMemsetAsync(b + o_eh, 0, NT * 4, c->stream));
  {
    StageTimer tm(c, ORBFE_STAGE_MATCH, c->stream);
    const int32_t *g_off = (const int32_t*)(b + o_co), *g_feat = (const int32_t*)(b + o_cf);
    if (cache_slot >= 0)
      TRY(slot_grid(c, c->stream, cache_slot, ag, &g_off, &g_feat));
    else
      launch_grid_build(c->stream, d_kps, d_n_kp, (int)NT, rows, cols, (int32_t*)(b + o_co), (int32_t*)(b + o_cf));
    launch_search_area(c->stream, d_kpl, d_desc, ag.clip_w, ag.clip_h, rows, cols, g_off,
                       g_feat, nq, (const float*)(b + o_q), (const float*)(b + o_r), (const int8_t*)(b + o_lo),
                       (const int8_t*)(b + o_hi), b + o_d, exclude ? b + o_ex : nullptr, (int32_t*)(b + o_bi), (int32_t*)(b + o_bd),
                       (int32_t*)(b + o_sd), (int32_t*)(b + o_nc), hits ? (int32_t*)(b + o_eh) : nullptr);
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(hs, b + o_bi, out_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  std::memcpy(best_idx, hs, (size_t)nq * 4);
  std::memcpy(best_dist, hs + (o_bd - o_bi), (size_t)nq * 4);
  std::memcpy(second_dist, hs + (o_sd - o_bi), (size_t)nq * 4);
  std::memcpy(n_cand, hs + (o_nc - o_bi), (size_t)nq * 4);
  if (hits) std::memcpy(excluded_hits, hs + (o_eh - o_bi), n_target * 4);
  else if (excluded_hits && n_target) std::memset(excluded_hits, 0, n_target * 4);
  return ORBFE_OK;
}
// scratch the core needs beyond `tmp_used`
static size_t search_area_scratch(const orbfe_ctx* c, size_t n_target, int32_t nq, const float* bounds = nullptr) {
  AreaGrid ag;
  if (!area_grid(c, bounds, &ag)) ag = {(c->cfg.height + 47) / 48, (c->cfg.width + 63) / 64, 0, 0};
  const size_t ncells = (size_t)ag.rows * ag.cols, NT = std::max<size_t>(n_target, 1);
  return ((ncells + 1) * 4 + NT * 9 + (size_t)nq * (8 + 4 + 1 + 1 + 32 + 16)) + 15 * 256 + 4096;
}

orbfe_status orbfe_search_in_area(orbfe_ctx* c, int32_t slot, int32_t nq, const float* qxy, const float* radius, const int8_t* min_level,
                                  const int8_t* max_level, const uint8_t* q_desc, const uint8_t* exclude, int32_t* best_idx,
                                  int32_t* best_dist, int32_t* second_dist, int32_t* n_cand) {
  ApiLock api_lk(c);
  if (!c || slot < 0 || slot >= c->cfg.max_images || nq < 0) return fail(c, ORBFE_EBADARG, "search_in_area: bad slot / count");
  if (nq && (!qxy || !radius || !min_level || !max_level || !q_desc || !best_idx || !best_dist || !second_dist || !n_cand))
    return fail(c, ORBFE_EBADARG, "search_in_area: NULL argument");
  if (nq == 0) return ORBFE_OK;
  TRY(slots_idle(c, slot, 1, "search_in_area"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  TRY(ensure_tmp(c, search_area_scratch(c, NF, nq)));
  return search_area_core(c, "search_in_area", c->d_kps + (size_t)slot * NF, c->d_n_kp + slot, c->d_kpl + (size_t)slot * NF,
                          c->d_desc + (size_t)slot * NF * 32, NF, 0, nq, qxy, radius, min_level, max_level, q_desc, exclude, best_idx,
                          best_dist, second_dist, n_cand, nullptr, nullptr, false, slot);
}

orbfe_status orbfe_search_in_area_features(orbfe_ctx* c, int32_t nt, const orbfe_keypoint* t_kps, const uint8_t* t_desc, int32_t nq,
                                           const float* qxy, const float* radius, const int8_t* min_level, const int8_t* max_level,
                                           const uint8_t* q_desc, const uint8_t* exclude, int32_t* best_idx, int32_t* best_dist,
                                           int32_t* second_dist, int32_t* n_cand) {
  return orbfe_search_in_area_features_ex(c, nt, t_kps, t_desc, nullptr, nq, qxy, radius, min_level, max_level, q_desc, exclude, best_idx,
                                          best_dist, second_dist, n_cand, nullptr);
}

orbfe_status orbfe_search_in_area_features_ex(orbfe_ctx* c, int32_t nt, const orbfe_keypoint* t_kps, const uint8_t* t_desc,
                                              const float* bounds, int32_t nq, const float* qxy, const float* radius,
                                              const int8_t* min_level, const int8_t* max_level, const uint8_t* q_desc, const uint8_t* exclude,
                                              int32_t* best_idx, int32_t* best_dist, int32_t* second_dist, int32_t* n_cand,
                                              int32_t* excluded_hits) {
  ApiLock api_lk(c);
  if (!c || nt < 0 || nq < 0 || (nt && (!t_kps || !t_desc))) return fail(c, ORBFE_EBADARG, "search_in_area_features: bad count / NULL features");
  if (nq && (!qxy || !radius || !min_level || !max_level || !q_desc || !best_idx || !best_dist || !second_dist || !n_cand))
    return fail(c, ORBFE_EBADARG, "search_in_area_features: NULL argument");
  if (excluded_hits && nt) std::memset(excluded_hits, 0, (size_t)nt * 4);
  if (nq == 0) return ORBFE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t NT = (size_t)std::max(nt, 1);
  // the uploaded feature set at the front of the scratch: keypoints | octave list in the layout of the slot arrays | descriptors | count
  const size_t o_k = 0, o_l = o_k + align_up(NT * sizeof(orbfe_keypoint), 256), o_d = o_l + align_up(NT * sizeof(uint4), 256),
               o_n = o_d + align_up(NT * 32, 256), used = o_n + 256;
  TRY(ensure_tmp(c, used + search_area_scratch(c, NT, nq, bounds)));
  TRY(ensure_stage(c, used + search_area_scratch(c, NT, nq, bounds)));
  uint8_t* b = (uint8_t*)c->d_tmp;
  uint8_t* hs = c->main.h_stage;
  uint4* kpl = (uint4*)(hs + o_l);  // (built in the staging buffer: it goes up with everything else)
  std::memset(kpl, 0, NT * sizeof(uint4));
  for (int i = 0; i < nt; ++i) {
    // caller-supplied features (a KeyFrame's undistorted mvFeatsLeft): coordinates may lie outside the image or be non-finite -- the grid
    // kernel clamps them into the border cells; the octave must be one a pyramid can have (it is compared as an unsigned byte)
    if (t_kps[i].octave < 0 || t_kps[i].octave >= ORBFE_MAX_LEVELS)
      return fail(c, ORBFE_EBADARG, "search_in_area_features: feature %d has octave %d (0..%d expected)", i, t_kps[i].octave, ORBFE_MAX_LEVELS - 1);
    kpl[(size_t)i].y = (uint32_t)t_kps[i].octave;  // the search reads the octave from here
  }
  if (nt) {
    std::memcpy(hs + o_k, t_kps, (size_t)nt * sizeof(orbfe_keypoint));
    std::memcpy(hs + o_d, t_desc, (size_t)nt * 32);
  }
  std::memcpy(hs + o_n, &nt, 4);
  return search_area_core(c, "search_in_area_features", (const orbfe_keypoint*)(b + o_k), (const int32_t*)(b + o_n), (const uint4*)(b + o_l),
                          b + o_d, (size_t)nt, used, nq, qxy, radius, min_level, max_level, q_desc, exclude, best_idx, best_dist,
                          second_dist, n_cand, bounds, excluded_hits, true);
}

orbfe_status orbfe_project_map_points(orbfe_ctx* c, int32_t n, const float* pos, const float* view_dir, const float* max_dist,
                                      const float* min_dist, const orbfe_frame_pose* pose, const orbfe_camera* cam, float* uv,
                                      float* distance, float* cos_theta, int8_t* level, uint8_t* visible) {
  ApiLock api_lk(c);
  if (!c || n < 0 || !pose || !cam) return fail(c, ORBFE_EBADARG, "project_map_points: NULL argument");
  if (n && (!pos || !view_dir || !max_dist || !min_dist || !uv || !distance || !cos_theta || !level || !visible))
    return fail(c, ORBFE_EBADARG, "project_map_points: NULL argument");
  if (n == 0) return ORBFE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t N = (size_t)n;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  const size_t o_p = take(N * 12), o_v = take(N * 12), o_mx = take(N * 4), o_mn = take(N * 4), o_in_end = take(8), o_uv = take(N * 8),
               o_d = take(N * 4), o_c = take(N * 4), o_l = take(N), o_s = take(N), o_out_end = take(8);
  TRY(ensure_tmp(c, off));
  TRY(ensure_stage(c, std::max(o_in_end, o_out_end - o_uv)));  // one copy up, one down, through the page-locked staging buffer
  uint8_t* b = (uint8_t*)c->d_tmp;
  uint8_t* hs = c->main.h_stage;
  std::memcpy(hs + o_p, pos, N * 12);
  std::memcpy(hs + o_v, view_dir, N * 12);
  std::memcpy(hs + o_mx, max_dist, N * 4);
  std::memcpy(hs + o_mn, min_dist, N * 4);
  HIP_TRY(c, hipMemcpyAsync(b, hs, o_in_end, hipMemcpyHostToDevice, c->stream));
  const float cam4[4] = {cam->fx, cam->fy, cam->cx, cam->cy};
  const float bounds4[4] = {pose->min_u, pose->max_u, pose->min_v, pose->max_v};
  {
    StageTimer tm(c, ORBFE_STAGE_MATCH, c->stream);
    // std::log(ORBExtractor::mfScaledFactor): float argument, float result
    launch_project_map_points(c->stream, n, (const float*)(b + o_p), (const float*)(b + o_v), (const float*)(b + o_mx),
                              (const float*)(b + o_mn), pose->Rcw, pose->tcw, cam4, bounds4, std::log(c->cfg.scale_factor), 7,
                              (float*)(b + o_uv), (float*)(b + o_d), (float*)(b + o_c), (int8_t*)(b + o_l), b + o_s);
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(hs, b + o_uv, o_out_end - o_uv, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  std::memcpy(uv, hs, N * 8);
  std::memcpy(distance, hs + (o_d - o_uv), N * 4);
  std::memcpy(cos_theta, hs + (o_c - o_uv), N * 4);
  std::memcpy(level, hs + (o_l - o_uv), N);
  std::memcpy(visible, hs + (o_s - o_uv), N);
  return ORBFE_OK;
}
// Tracking::trackLocalMap's device work as ONE call (src/Tracking.cc:641-675): isInVision / predictLevel per map point
// (MapPoint.cc:141-201), ORBMatcher::searchByProjection(frame, map points, th) (ORBMatcher.cc:561-612) against the features of `slot`,
// and Optimizer::OptimizePoseOnly (Optimizer.cc:33-178) on what the frame holds afterwards -- one upload, seven launches, one download;
// the projections, the windows, the candidate lists, the assignment and the edge list never leave the device.
orbfe_status orbfe_track_local_map(orbfe_ctx* c, int32_t slot, const orbfe_frame_pose* pose, const orbfe_camera* cam, const orbfe_track_input* in,
                                   const orbfe_track_output* out) {
  ApiLock api_lk(c);
  if (!c || !pose || !cam || !in || !out || slot < 0 || slot >= c->cfg.max_images) return fail(c, ORBFE_EBADARG, "track_local_map: NULL argument / bad slot");
  const int n = in->n_mp, nl = c->cfg.n_levels;
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  if (n < 0 || !in->pose_se3 || !in->level_sigma2 || !in->level_inv_sigma2 || !out->assigned || !out->n_matches || !out->n_edges || !out->n_good ||
      !out->pose_out || !out->inlier || (n && (!in->pos || !in->view_dir || !in->max_dist || !in->min_dist || !in->desc || !in->flags)))
    return fail(c, ORBFE_EBADARG, "track_local_map: NULL array");
  if (NF > 2048) return fail(c, ORBFE_EBADSIZE, "track_local_map: %zu features per frame (the fused pose kernel keeps up to 2048 edges in registers)", NF);
  if (in->held)
    for (size_t f = 0; f < NF; ++f)
      if (in->held[f] < -1 || in->held[f] >= n) return fail(c, ORBFE_EBADARG, "track_local_map: held[%zu] = %d out of range", f, in->held[f]);
  const float bounds[4] = {pose->min_u, pose->max_u, pose->min_v, pose->max_v};
  AreaGrid ag;
  if (!area_grid(c, bounds, &ag)) return fail(c, ORBFE_EBADARG, "track_local_map: bad frame bounds");
  const size_t ncells = (size_t)ag.rows * ag.cols;
  if ((2 * ncells + 1) * 4 > 60 * 1024) return fail(c, ORBFE_EBADSIZE, "track_local_map: %zu grid cells exceed the LDS counters", ncells);
  TRY(slots_idle(c, slot, 1, "track_local_map"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t N = (size_t)std::max(n, 1);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  // [ upload | claim (0x7F fill) | device-only | download ]
  const size_t o_pos = take(N * 12), o_vd = take(N * 12), o_mx = take(N * 4), o_mn = take(N * 4), o_desc = take(N * 32), o_fl = take(N),
               o_held = take(NF * 4), o_ru = take(NF * 8), o_s2 = take((size_t)nl * 4), o_is2 = take((size_t)nl * 4), o_p0 = take(56),
               o_up_end = take(8), o_claim = take(NF * 4), o_claim_end = take(8), o_uv = take(N * 8), o_dist = take(N * 4), o_cos = take(N * 4),
               o_lvl = take(N), o_vis = take(N), o_rad = take(N * 4), o_lo = take(N), o_hi = take(N),
               o_bi = take(N * 4), o_bd = take(N * 4), o_sd = take(N * 4), o_nc = take(N * 4), o_xw = take(NF * 24),
               o_ms = take(NF * 24), o_info = take(NF * 8), o_sig = take(NF * 4), o_err = take(NF * 24), o_l = take(NF), o_r = take(NF),
               o_dn = take(0), o_cnt = take(16), o_ng = take(8), o_po = take(56), o_asg = take(NF * 4), o_eo = take(NF * 4), o_in = take(NF),
               o_dn_end = take(8);
  (void)o_dn;
  TRY(ensure_tmp(c, off));
  TRY(ensure_stage(c, std::max(o_up_end, o_dn_end - o_cnt)));
  uint8_t* b = (uint8_t*)c->d_tmp;
  uint8_t* hs = c->main.h_stage;
  if (n) {
    std::memcpy(hs + o_pos, in->pos, (size_t)n * 12);
    std::memcpy(hs + o_vd, in->view_dir, (size_t)n * 12);
    std::memcpy(hs + o_mx, in->max_dist, (size_t)n * 4);
    std::memcpy(hs + o_mn, in->min_dist, (size_t)n * 4);
    std::memcpy(hs + o_desc, in->desc, (size_t)n * 32);
    std::memcpy(hs + o_fl, in->flags, (size_t)n);
  }
  if (in->held) std::memcpy(hs + o_held, in->held, NF * 4);
  else std::memset(hs + o_held, 0xFF, NF * 4);
  if (in->right_u) std::memcpy(hs + o_ru, in->right_u, NF * 8);
  else
    for (size_t f = 0; f < NF; ++f) ((double*)(hs + o_ru))[f] = -1.0;
  std::memcpy(hs + o_s2, in->level_sigma2, (size_t)nl * 4);
  std::memcpy(hs + o_is2, in->level_inv_sigma2, (size_t)nl * 4);
  std::memcpy(hs + o_p0, in->pose_se3, 56);
  hipStream_t st = c->stream;
  HIP_TRY(c, hipMemcpyAsync(b, hs, o_up_end, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemsetAsync(b + o_claim, 0x7F, o_claim_end - o_claim, st));
  const float cam4[4] = {cam->fx, cam->fy, cam->cx, cam->cy};
  const BaParamsDev prm = {(double)cam->fx, (double)cam->fy, (double)cam->cx, (double)cam->cy, (double)cam->bf};
  {
    StageTimer tm(c, ORBFE_STAGE_MATCH, st);
    launch_project_map_points(st, n, (const float*)(b + o_pos), (const float*)(b + o_vd), (const float*)(b + o_mx), (const float*)(b + o_mn),
                              pose->Rcw, pose->tcw, cam4, bounds, std::log(c->cfg.scale_factor), 7, (float*)(b + o_uv), (float*)(b + o_dist),
                              (float*)(b + o_cos), (int8_t*)(b + o_lvl), b + o_vis);
    launch_track_queries(st, n, b + o_fl, b + o_vis, (const float*)(b + o_cos), (const int8_t*)(b + o_lvl), in->th, (const float*)(b + o_s2), nl,
                         (float*)(b + o_rad), (int8_t*)(b + o_lo), (int8_t*)(b + o_hi));
    const int32_t *g_off = nullptr, *g_feat = nullptr;
    TRY(slot_grid(c, st, slot, ag, &g_off, &g_feat));
    launch_search_area(st, c->d_kpl + (size_t)slot * NF, c->d_desc + (size_t)slot * NF * 32, ag.clip_w, ag.clip_h, ag.rows, ag.cols,
                       g_off, g_feat, n, (const float*)(b + o_uv), (const float*)(b + o_rad),
                       (const int8_t*)(b + o_lo), (const int8_t*)(b + o_hi), b + o_desc, nullptr, (int32_t*)(b + o_bi), (int32_t*)(b + o_bd),
                       (int32_t*)(b + o_sd), (int32_t*)(b + o_nc), nullptr);
    launch_track_claim(st, n, (const int32_t*)(b + o_nc), (const int32_t*)(b + o_bi), (const int32_t*)(b + o_bd), (const int32_t*)(b + o_sd),
                       in->min_threshold, in->ratio, (int32_t*)(b + o_claim));
    launch_track_edges(st, c->d_kps + (size_t)slot * NF, c->d_n_kp + slot, (int)NF, (const int32_t*)(b + o_held), (const int32_t*)(b + o_claim),
                       b + o_fl, (const float*)(b + o_pos), (const double*)(b + o_ru), (const float*)(b + o_s2), (const float*)(b + o_is2),
                       in->min_matches, (int32_t*)(b + o_asg), (int32_t*)(b + o_eo), (double*)(b + o_xw), (double*)(b + o_ms),
                       (double*)(b + o_info), (float*)(b + o_sig), (int32_t*)(b + o_cnt));
  }
  {
    StageTimer tm(c, ORBFE_STAGE_BA, st);
    launch_pose_only(st, (int)NF, (const double*)(b + o_xw), (const double*)(b + o_ms), (const double*)(b + o_info), (const float*)(b + o_sig),
                     (const double*)(b + o_p0), prm, (double)(float)std::sqrt(5.991), (double)(float)std::sqrt(7.815), (double*)(b + o_err), b + o_l,
                     b + o_r, b + o_in, (double*)(b + o_po), (int32_t*)(b + o_ng), (const int32_t*)(b + o_cnt) + 1);
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(hs, b + o_cnt, o_dn_end - o_cnt, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  drain_timers(c);
  const int32_t* cnt = (const int32_t*)hs;
  *out->n_matches = cnt[0];
  *out->n_edges = cnt[1];
  std::memcpy(out->n_good, hs + (o_ng - o_cnt), 4);
  std::memcpy(out->pose_out, hs + (o_po - o_cnt), 56);
  std::memcpy(out->assigned, hs + (o_asg - o_cnt), NF * 4);
  const int32_t* eo = (const int32_t*)(hs + (o_eo - o_cnt));
  const uint8_t* ein = hs + (o_in - o_cnt);
  const bool optimised = cnt[1] >= 0;
  for (size_t f = 0; f < NF; ++f) out->inlier[f] = (optimised && eo[f] >= 0) ? ein[eo[f]] : 0;
  if (out->edge_of) std::memcpy(out->edge_of, eo, NF * 4);
  return ORBFE_OK;
}

// The middle of Tracking::trackMotionModel (src/Tracking.cc:382-396) as one call: ORBMatcher::searchByProjection(frame, lastFrame, matches, th)
// -- in this reference a search around the LAST frame's feature positions, no projection (src/ORBMatcher.cc:265-347) -- then, with fewer than
// min_matches matches, the same search again with th_second among the features still free, then Optimizer::OptimizePoseOnly(frame).  The
// second search is decided on the host (one more synchronisation in the rare frame that needs it).
orbfe_status orbfe_track_motion_model(orbfe_ctx* c, int32_t slot, const float* bounds4, const orbfe_camera* cam, const orbfe_motion_input* in,
                                      const orbfe_track_output* out, int32_t* excluded_hits, int32_t* query_matches, int32_t* passes) {
  ApiLock api_lk(c);
  if (!c || !bounds4 || !cam || !in || !out || slot < 0 || slot >= c->cfg.max_images) return fail(c, ORBFE_EBADARG, "track_motion_model: NULL argument / bad slot");
  const int n = in->n, nl = c->cfg.n_levels;
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  if (n < 0 || !in->pose_se3 || !in->level_sigma2 || !in->level_inv_sigma2 || !out->assigned || !out->n_matches || !out->n_edges || !out->n_good ||
      !out->pose_out || !out->inlier || (n && (!in->qxy || !in->q_octave || !in->q_min_level || !in->q_max_level || !in->desc || !in->pos)))
    return fail(c, ORBFE_EBADARG, "track_motion_model: NULL array");
  for (int i = 0; i < n; ++i)
    if (in->q_octave[i] < 0 || in->q_octave[i] >= nl) return fail(c, ORBFE_EBADARG, "track_motion_model: q_octave[%d] = %d", i, (int)in->q_octave[i]);
  if (NF > 2048) return fail(c, ORBFE_EBADSIZE, "track_motion_model: %zu features per frame (the fused pose kernel keeps up to 2048 edges in registers)", NF);
  if (in->held)
    for (size_t f = 0; f < NF; ++f)
      if (in->held[f] < -1 || in->held[f] >= n) return fail(c, ORBFE_EBADARG, "track_motion_model: held[%zu] = %d out of range", f, in->held[f]);
  AreaGrid ag;
  if (!area_grid(c, bounds4, &ag)) return fail(c, ORBFE_EBADARG, "track_motion_model: bad frame bounds");
  const size_t ncells = (size_t)ag.rows * ag.cols;
  if ((2 * ncells + 1) * 4 > 60 * 1024) return fail(c, ORBFE_EBADSIZE, "track_motion_model: %zu grid cells exceed the LDS counters", ncells);
  TRY(slots_idle(c, slot, 1, "track_motion_model"));
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t N = (size_t)std::max(n, 1);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  // [ upload once | upload per pass | claim (-1 fill), hits and counter (0 fill) | device-only | download ]
  const size_t o_qxy = take(N * 8), o_lo = take(N), o_hi = take(N), o_desc = take(N * 32), o_pos = take(N * 12), o_fl = take(N), o_ru = take(NF * 8),
               o_s2 = take((size_t)nl * 4), o_is2 = take((size_t)nl * 4), o_p0 = take(56), o_up1_end = take(8), o_rad = take(N * 4), o_held = take(NF * 4),
               o_ex = take(NF), o_up2_end = take(8), o_claim = take(NF * 4), o_claim_end = take(8), o_eh = take(NF * 4), o_acc = take(16),
               o_qa = take(N), o_zero_end = take(8), o_bi = take(N * 4), o_bd = take(N * 4), o_sd = take(N * 4),
               o_nc = take(N * 4), o_xw = take(NF * 24), o_ms = take(NF * 24), o_info = take(NF * 8), o_sig = take(NF * 4), o_err = take(NF * 24),
               o_l = take(NF), o_r = take(NF), o_cnt = take(16), o_ng = take(8), o_po = take(56), o_asg = take(NF * 4), o_eo = take(NF * 4), o_in = take(NF),
               o_ehd = take(NF * 4), o_qad = take(N), o_dn_end = take(8);
  (void)o_up1_end;
  TRY(ensure_tmp(c, off));
  TRY(ensure_stage(c, std::max(o_up2_end, o_dn_end - o_cnt)));
  uint8_t* b = (uint8_t*)c->d_tmp;
  uint8_t* hs = c->main.h_stage;
  if (n) {
    std::memcpy(hs + o_qxy, in->qxy, (size_t)n * 8);
    std::memcpy(hs + o_lo, in->q_min_level, (size_t)n);
    std::memcpy(hs + o_hi, in->q_max_level, (size_t)n);
    std::memcpy(hs + o_desc, in->desc, (size_t)n * 32);
    std::memcpy(hs + o_pos, in->pos, (size_t)n * 12);
    std::memset(hs + o_fl, 3, (size_t)n);  // every query is a good map point in the map (the caller's filter, ORBMatcher.cc:286-289)
  }
  if (in->right_u) std::memcpy(hs + o_ru, in->right_u, NF * 8);
  else
    for (size_t f = 0; f < NF; ++f) ((double*)(hs + o_ru))[f] = -1.0;
  std::memcpy(hs + o_s2, in->level_sigma2, (size_t)nl * 4);
  std::memcpy(hs + o_is2, in->level_inv_sigma2, (size_t)nl * 4);
  std::memcpy(hs + o_p0, in->pose_se3, 56);
  std::vector<int32_t> held(NF, -1);
  if (in->held) std::memcpy(held.data(), in->held, NF * 4);
  std::vector<int32_t> hits_total(excluded_hits ? NF : 0, 0), qm_total(query_matches ? N : 0, 0);
  hipStream_t st = c->stream;
  const BaParamsDev prm = {(double)cam->fx, (double)cam->fy, (double)cam->cx, (double)cam->cy, (double)cam->bf};
  int base_matches = 0, n_pass = 0;
  const int32_t* cnt = (const int32_t*)hs;
  for (int pass = 0; pass < 2; ++pass) {
    const float th = pass == 0 ? in->th : in->th_second;
    if (pass == 1 && !(th > 0)) break;
    // the per-pass upload: radius, what the features hold, and the candidates that are excluded (a feature that holds a map point: :322-331)
    for (int i = 0; i < n; ++i) ((float*)(hs + o_rad))[i] = th * in->level_sigma2[in->q_octave[i]];  // findFeaturesInArea: radius * getScaledFactor2(octave)
    std::memcpy(hs + o_held, held.data(), NF * 4);
    for (size_t f = 0; f < NF; ++f) hs[o_ex + f] = held[f] >= 0 ? 1 : 0;
    if (pass == 0)
      HIP_TRY(c, hipMemcpyAsync(b, hs, o_up2_end, hipMemcpyHostToDevice, st));
    else
      HIP_TRY(c, hipMemcpyAsync(b + o_rad, hs + o_rad, o_up2_end - o_rad, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(b + o_claim, 0xFF, o_claim_end - o_claim, st));
    HIP_TRY(c, hipMemsetAsync(b + o_eh, 0, o_zero_end - o_eh, st));
    {
      StageTimer tm(c, ORBFE_STAGE_MATCH, st);
      const int32_t *g_off = nullptr, *g_feat = nullptr;
      TRY(slot_grid(c, st, slot, ag, &g_off, &g_feat));
      launch_search_area(st, c->d_kpl + (size_t)slot * NF, c->d_desc + (size_t)slot * NF * 32, ag.clip_w, ag.clip_h, ag.rows, ag.cols,
                         g_off, g_feat, n, (const float*)(b + o_qxy), (const float*)(b + o_rad),
                         (const int8_t*)(b + o_lo), (const int8_t*)(b + o_hi), b + o_desc, b + o_ex, (int32_t*)(b + o_bi), (int32_t*)(b + o_bd),
                         (int32_t*)(b + o_sd), (int32_t*)(b + o_nc), (int32_t*)(b + o_eh));
      launch_track_claim(st, n, (const int32_t*)(b + o_nc), (const int32_t*)(b + o_bi), (const int32_t*)(b + o_bd), (const int32_t*)(b + o_sd),
                         in->min_threshold, in->ratio, (int32_t*)(b + o_claim), 1, (int32_t*)(b + o_acc), b + o_qa);
      launch_track_edges(st, c->d_kps + (size_t)slot * NF, c->d_n_kp + slot, (int)NF, (const int32_t*)(b + o_held), (const int32_t*)(b + o_claim),
                         b + o_fl, (const float*)(b + o_pos), (const double*)(b + o_ru), (const float*)(b + o_s2), (const float*)(b + o_is2),
                         in->min_matches, (int32_t*)(b + o_asg), (int32_t*)(b + o_eo), (double*)(b + o_xw), (double*)(b + o_ms),
                         (double*)(b + o_info), (float*)(b + o_sig), (int32_t*)(b + o_cnt), -1, (const int32_t*)(b + o_acc), base_matches);
    }
    {
      StageTimer tm(c, ORBFE_STAGE_BA, st);
      launch_pose_only(st, (int)NF, (const double*)(b + o_xw), (const double*)(b + o_ms), (const double*)(b + o_info), (const float*)(b + o_sig),
                       (const double*)(b + o_p0), prm, (double)(float)std::sqrt(5.991), (double)(float)std::sqrt(7.815), (double*)(b + o_err), b + o_l,
                       b + o_r, b + o_in, (double*)(b + o_po), (int32_t*)(b + o_ng), (const int32_t*)(b + o_cnt) + 1);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(b + o_ehd, b + o_eh, NF * 4, hipMemcpyDeviceToDevice, st));  // (the hits sit in front of the downloaded block)
    HIP_TRY(c, hipMemcpyAsync(b + o_qad, b + o_qa, N, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(hs, b + o_cnt, o_dn_end - o_cnt, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    ++n_pass;
    if (excluded_hits) {
      const int32_t* eh = (const int32_t*)(hs + (o_ehd - o_cnt));
      for (size_t f = 0; f < NF; ++f) hits_total[f] += eh[f];
    }
    if (query_matches)
      for (int i = 0; i < n; ++i) qm_total[(size_t)i] += hs[(o_qad - o_cnt) + (size_t)i];
    if (cnt[1] >= 0 || pass == 1 || !(in->th_second > 0)) break;
    // fewer than min_matches: the matches of this pass stay (setMapPoints, :344-345) and are excluded from the next one
    base_matches = cnt[0];
    std::memcpy(held.data(), hs + (o_asg - o_cnt), NF * 4);
  }
  drain_timers(c);
  *out->n_matches = cnt[0];
  *out->n_edges = cnt[1];
  std::memcpy(out->n_good, hs + (o_ng - o_cnt), 4);
  std::memcpy(out->pose_out, hs + (o_po - o_cnt), 56);
  std::memcpy(out->assigned, hs + (o_asg - o_cnt), NF * 4);
  const int32_t* eo = (const int32_t*)(hs + (o_eo - o_cnt));
  const uint8_t* ein = hs + (o_in - o_cnt);
  const bool optimised = cnt[1] >= 0;
  for (size_t f = 0; f < NF; ++f) out->inlier[f] = (optimised && eo[f] >= 0) ? ein[eo[f]] : 0;
  if (out->edge_of) std::memcpy(out->edge_of, eo, NF * 4);
  if (excluded_hits) std::memcpy(excluded_hits, hits_total.data(), NF * 4);
  if (query_matches && n) std::memcpy(query_matches, qm_total.data(), (size_t)n * 4);
  if (passes) *passes = n_pass;
  return ORBFE_OK;
}


}  // extern "C"
