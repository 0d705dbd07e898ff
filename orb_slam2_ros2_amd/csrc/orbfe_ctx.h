// orbfe_ctx.h -- what the translation units of the C-ABI's host side share (r5: orbfe_api.hip split by entry-point group, no change of
// behaviour): the kernel launchers' declarations, the context record, the error / allocation / timing helpers and the launch sequences
// more than one group uses.
//   orbfe_api.hip      geometry tables, create / destroy, the fetch calls, profiling, the shared helpers' definitions
//   orbfe_extract.hip  the launch sequence of an extraction and of the stereo match; orbfe_extract* / frame_stereo* / frame_rgbd* /
//                      stereo_match / stereo_batch_device
//   orbfe_stream.hip   page-locked allocation and the host-image stream (orbfe_stream_*)
//   orbfe_guided.hip   brute-force and grid-guided matching, map-point projection, the fused tracking chains
//   orbfe_ba.hip       g2o edge evaluation, normal equations, local BA, pose-only optimisation
//   orbfe_map.hip      map.pb / text maps
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <limits>
#include <memory>
#include <vector>

#include "orbfe_internal.h"

namespace orbfe {
// k_pyramid.hip
void launch_resize(hipStream_t s, const LevelDev* d_lv, const RsTile* d_tiles, const int* n_tiles, const int* lds_bytes,
                   const ResizeTap* d_taps, uint8_t* d_pyr,
                   size_t img_pitch, int n_img);
void launch_resize_regions(hipStream_t s, const LevelDev* d_lv, int n_levels, const RsRegion* d_regions, int n_regions, int tile_bytes,
                           int xt_bytes, int yt_bytes, const RgXTap* d_xtaps, const RgYTap* d_ytaps, uint8_t* d_pyr, size_t img_pitch, int n_img,
                           const uint8_t* src_a, const uint8_t* src_b, size_t src_pitch, int src_stride, uint32_t src_bytes, int copy_l0, int32_t* d_zero, int n_zero, int pq, int32_t* d_zero2 = nullptr, int n_zero2 = 0);
void launch_blur(hipStream_t s, const LevelDev* d_lv, int n_levels, int tile_first, int n_tiles, const uint8_t* d_pyr, uint8_t* d_blur,
                 size_t img_pitch, const int taps[7], int n_img);
// k_blur_mfma.hip: the batches' blur on the integer matrix cores (band tables built by mb_build; ok = false: keep k_blur)
bool mb_build(const LevelDev* lv, int n_levels, const int taps[7], uint32_t spare_off, MbGeom* g, std::vector<uint8_t>* tx, std::vector<uint8_t>* ty);
void launch_blur_mfma(hipStream_t s, const LevelDev* d_lv, int n_levels, const MbGeom& g, const uint8_t* d_pyr, uint8_t* d_blur, size_t img_pitch,
                      const uint8_t* d_tx, const uint8_t* d_ty, int n_img);
void launch_load_level0(hipStream_t st, const uint8_t* d_src, const uint8_t* d_src_b, size_t src_stride, size_t src_pitch, uint8_t* d_pyr,
                        size_t img_pitch, uint32_t plane_off, int dst_stride, int w, int h, int slot0, int slot_step, int n_img);
// k_fast.hip
void launch_fast(hipStream_t s, const LevelDev* d_lv, const CellDev* d_cells, const LevelDev* h_lv, const int* lvl_max_pw,
                 const int* lvl_max_ph, const uint8_t* d_pyr, size_t img_pitch, int t_hi, int t_lo, uint32_t* d_cand, size_t cand_pitch,
                 int32_t* d_n_cand, int n_levels, int n_img, int cpw_force, uint32_t level_mask = ~0u, bool merge_masked = false,
                 int32_t* d_n_cand_sh = nullptr, int n_shards = 1);
bool fast_single_launch(const LevelDev* h_lv, const int* lvl_max_pw, const int* lvl_max_ph, int n_levels, int n_img);  // launch_fast's one-launch rule
// k_quadtree.hip
size_t quadtree_lds_bytes(int node_cap, int rec_cap, int sort_cap);
hipError_t quadtree_configure(size_t lds_bytes);
void launch_quadtree(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint32_t* d_cand, uint32_t* d_scr_b, uint32_t* d_scr_c,
                     size_t scratch_pitch, uint32_t* d_sel, int32_t* d_sel_count, int n_features, const int32_t* d_n_cand, int node_cap, int sort_cap,
                     int rec_cap, int n_img, int batch, const QtGroups& groups, int n_groups, int waves_per_tree, uint8_t* d_big, size_t big_pitch,
                     const uint16_t* d_qt_tabs, const uint8_t* blur_pyr, uint8_t* blur_out, size_t img_pitch, const int* blur_taps, int blur_tiles,
                     int32_t* d_qt_next, bool next_zeroed = false, const int32_t* d_n_cand_sh = nullptr, int n_shards = 1);
bool quadtree_build_tables(const LevelDev& L, std::vector<uint16_t>& out);
// k_brief.hip
void launch_orient_brief(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint8_t* d_pyr, const uint8_t* d_blur,
                         size_t img_pitch, const uint32_t* d_sel, const int32_t* d_sel_count, int n_features,
                         const int8_t* d_pattern, const int umax[16], orbfe_keypoint* d_kps, uint8_t* d_desc, KpAux* d_aux,
                         int32_t* d_n_kp, double* d_theta, int2* d_moments, double2* d_sincos, KpX* d_kx,
                         uint4* d_kpl, int rows0, int n_img, hipEvent_t before_brief, hipEvent_t before_lists, orbfe_keypoint* h_kps, uint8_t* h_desc, int32_t* h_n_kp,
                         bool fuse_small, uint32_t* d_rowoff_slot = nullptr, uint16_t* d_rowlist_slot = nullptr,
                         int32_t* d_n_match = nullptr, int rt_rows = 0, int rt_list_cap = 0, int rt_slot0 = 0, int32_t* d_rt_flags = nullptr);
// k_match.hip
void launch_match_bruteforce(hipStream_t s, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, const uint32_t* d_off,
                             const uint32_t* d_cand, int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second);
void launch_stereo(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint8_t* d_pyr, size_t img_pitch, const orbfe_keypoint* d_kps,
                   const uint8_t* d_desc, const KpAux* d_aux, const KpX* d_kx, uint32_t* d_rowoff, uint16_t* d_rowlist, int rows, int list_cap,
                   const int32_t* d_n_kp, int n_features, float fx, float bf, int cols0, int mean_threshold, double* d_right_u, double* d_depth, int32_t* d_n_match, int32_t* d_best_right,
                   int32_t* d_best_dist, int slot_l0, int slot_r0, int slot_step, int pair0, int n_pairs, double* h_right_u, double* h_depth,
                   int32_t* h_best_right, int32_t* h_best_dist, bool table_ready = false, const StereoRowsBuf* rowsbuf = nullptr);
// k_glue.hip
void launch_cvt_gray(hipStream_t s, const uint8_t* d_src, size_t src_stride, uint8_t* d_dst, int dst_stride, int w, int h, int order,
                     int variant);
void launch_frame_rgbd(hipStream_t s, orbfe_keypoint* d_kps, const int32_t* d_n_kp, int n_features, const orbfe_camera& cam,
                       const uint8_t* d_depth, int depth_type, size_t depth_stride, float depth_scale, double* d_depth_out, double* d_right_u, orbfe_keypoint* h_kps = nullptr);
void launch_pack_records(hipStream_t s, const uint8_t* d_kps, const uint8_t* d_desc, const int32_t* d_counts, const uint8_t* d_ru,
                         const uint8_t* d_dp, const int32_t* d_nm, int nf, int n_pairs, void* d_out);
// k_lba.hip
void launch_lba_chi2_sum(hipStream_t s, int n_edges, const double* chi2, const double* rho, const uint8_t* level, double* chi2_last,
                         double* out);
void launch_lba_maxdiag(hipStream_t s, int n_poses, int n_points, const double* Hpp, const double* Hll, const uint8_t* fixed, double* out);
void launch_lba_solve(hipStream_t s, int n_poses, int n_points, int n_edges, int nf, const int32_t* free_pose, const int32_t* pose_slot,
                      const int32_t* pair_off, const int2* pairs, const int32_t* ps_off, const int32_t* ps_edges, const int32_t* pt_off,
                      const int32_t* pt_edges, const int32_t* edge_pose, const int32_t* edge_point, const uint8_t* fixed, const double* Hpp,
                      const double* bp, const double* Hll, const double* bl, const double* Hpl, const double* lambda_p, double* Dinv, double* W,
                      double* S, double* rhs, double* x, int* ok, double* poses, double* points, double* dxp, double* dxl, double* scale_out,
                      double* big_scratch);
// k_lmbig.hip
size_t lm_big_bytes(int nf);
size_t lm_big_inv_bytes(int nf);
int lm_big_ld(int nf);
void launch_lm_big_init(hipStream_t s, const LmLaunch& L);
void launch_lba_classify(hipStream_t s, int n_edges, const double* chi2_last, const uint8_t* depth_pos, const uint8_t* is_stereo,
                         uint8_t* level, double* info_eff, double* delta_eff);
void launch_lba_final(hipStream_t s, int n_edges, const double* chi2, const uint8_t* depth_pos, const uint8_t* is_stereo, uint8_t* bad);
// k_ba.hip
void launch_ba_edges(hipStream_t s, int n_edges, const double* d_poses, const double* d_points, const int32_t* d_edge_pose,
                     const int32_t* d_edge_point, const double* d_meas, const uint8_t* d_is_stereo, const double* d_info,
                     const double* d_delta, BaParamsDev prm, double* d_error, double* d_chi2, double* d_rho, double* d_jpoint,
                     double* d_jpose, uint8_t* d_depth_pos);
void launch_ba_system(hipStream_t s, int n_poses, int n_points, int n_edges, const double* poses, const double* points,
                      const int32_t* edge_pose, const int32_t* edge_point, const double* meas, const uint8_t* is_stereo,
                      const double* info, const double* delta, BaParamsDev prm, const uint8_t* pose_fixed, const int32_t* pt_off,
                      const int32_t* pt_edges, const int32_t* ps_off, const int32_t* ps_edges, double* Hpp, double* bp, double* Hll,
                      double* bl, double* Hpl);
void launch_project_map_points(hipStream_t s, int n, const float* d_pos, const float* d_vdir, const float* d_max, const float* d_min,
                               const float* R, const float* t, const float* cam4, const float* bounds4, float log_sf, int max_level,
                               float* d_uv, float* d_dist, float* d_cos, int8_t* d_level, uint8_t* d_vis);
void launch_grid_build(hipStream_t s, const orbfe_keypoint* d_kps, const int32_t* d_n_kp, int n_cap, int rows, int cols, int32_t* d_cell_off,
                       int32_t* d_cell_feat);
void launch_search_area(hipStream_t s, const uint4* d_kpl, const uint8_t* d_desc, int width, int height, int rows, int cols,
                        const int32_t* d_cell_off, const int32_t* d_cell_feat, int nq, const float* d_qxy, const float* d_radius,
                        const int8_t* d_min_level, const int8_t* d_max_level, const uint8_t* d_q_desc, const uint8_t* d_exclude,
                        int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second, int32_t* d_n_cand, int32_t* d_excluded_hits);
void launch_lm_build(hipStream_t s, const LmLaunch& L, int gate, int which, int write_last, bool with_poses);
void launch_lm_maxdiag(hipStream_t s, const LmLaunch& L, int gate);
void launch_lm_pairs(hipStream_t s, const LmLaunch& L);
void launch_lm_steps(hipStream_t s, const LmLaunch& L, int n);
void launch_lm_switch(hipStream_t s, const LmLaunch& L);
void launch_lm_final(hipStream_t s, const LmLaunch& L);
void launch_pose_only(hipStream_t s, int n, const double* Xw, const double* meas, const double* info, const float* sigma2,
                      const double* pose_in, BaParamsDev prm, double d_mono, double d_stereo, double* err, uint8_t* level,
                      uint8_t* robust, uint8_t* inlier, double* pose_out, int32_t* n_good, const int32_t* n_dev = nullptr);
void launch_track_queries(hipStream_t s, int n, const uint8_t* d_flags, const uint8_t* d_visible, const float* d_cos, const int8_t* d_level, float th,
                          const float* d_sigma2, int n_levels, float* d_radius, int8_t* d_min_level, int8_t* d_max_level);
void launch_track_claim(hipStream_t s, int n, const int32_t* d_n_cand, const int32_t* d_best_idx, const int32_t* d_best_dist, const int32_t* d_second,
                        int min_threshold, float ratio, int32_t* d_claim, int last_wins = 0, int32_t* d_n_accept = nullptr, uint8_t* d_accepted = nullptr);
void launch_track_edges(hipStream_t s, const orbfe_keypoint* d_kps, const int32_t* d_n_kp, int n_features, const int32_t* d_held, const int32_t* d_claim,
                        const uint8_t* d_mp_flags, const float* d_mp_pos, const double* d_right_u, const float* d_sigma2, const float* d_inv_sigma2,
                        int min_matches, int32_t* d_assigned, int32_t* d_edge_of, double* d_Xw, double* d_meas, double* d_info, float* d_sig,
                        int32_t* d_counts, int32_t unclaimed = 0x7F7F7F7F, const int32_t* d_n_accept = nullptr, int base_matches = 0);
}  // namespace orbfe

using namespace orbfe;

static const int kGaussTaps[2][7] = {{18, 34, 48, 56, 48, 34, 18}, {18, 34, 49, 55, 49, 34, 18}};
static const int kMeanThreshold = 75;  // ORBMatcher::mnMeanThreshold (ORBMatcher.cc:1088)

struct orbfe_ctx {
  // Every entry point except orbfe_extract_slot takes this lock: the context's stream, scratch buffer, staging and timers serve one
  // call at a time, whichever threads the calls come from (the reference's matchers run on three threads: Tracking, LocalMapping,
  // LoopClosing).  Slot calls touch only their own lane and may overlap with anything but a call that rewrites their slot.
  std::recursive_mutex api_mu;
  orbfe_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // the blur of a batch runs on its own stream under the (latency-bound, LDS-hungry, SIMD-idle) quadtree of the same batch
  // the host-pointer path for one or two images (the drop-in call shape) is launch-bound: its copy-in / kernels / copy-out
  // sequence is captured once into a hipGraph per (image count, outputs wanted) and replayed
  struct GraphEntry {
    int slot0, n_img;
    bool want_kps, want_desc;
    bool stereo;   // the stereo match of the two slots rides in the graph (orbfe_frame_stereo), with these camera constants
    float fx, bf;
    const uint8_t* stage;
    const uint8_t* pyr;  // the pyramid buffer baked into the captured kernels (the pipelined batch path swaps the context's two buffers)
    hipGraphExec_t exec;
    bool rgbd = false;     // the RGB-D tail rides in the graph (orbfe_frame_rgbd_image), with the request's constants (FrameRgbdKey)
    unsigned char rkey[72] = {0};
  };
  // One in-order host-pointer pipeline: a stream, its pinned staging buffer and the hipGraphs captured on it.  The context has a main
  // lane (its own stream) and, created on first use, one lane per image slot for orbfe_extract_slot: the reference extracts the left
  // and the right image on two threads (src/Frame.cc:100-105), so two slots of one context must be usable at the same time.
  struct Lane {
    std::mutex mu;  // serialises the calls on this lane
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_main = nullptr;  // slot lanes: "the context stream has got this far" (work queued by asynchronous batch calls)
    uint8_t* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    std::vector<GraphEntry> graphs;
    bool use_graphs = true;
    std::atomic<bool> pending{false};  // orbfe_extract_slot_begin has enqueued an extraction that orbfe_extract_slot_end has not collected yet
                                       // (written under `mu`, read by slots_idle() of every other entry point that names the slot)
  };
  // Host-image stream (orbfe_stream_submit / _wait): batch k+1 is uploaded and the packed results of batch k-1 are downloaded while
  // batch k is computed.  kDepth (three) input and result buffers on the device, one copy stream per direction.
  struct HostStream {
    // Ring depth 3: with two buffers the caller's "collect k-1, then submit k+1" makes the upload of k+1 wait for the DOWNLOAD of
    // k-1, and a step costs (compute + download + upload) / 2 instead of max(compute, upload): measured 11.6 ms against 8.2 ms of
    // compute and 8.7 ms of upload per 512 pairs.
    static const int kDepth = 3;
    bool init = false;
    hipStream_t h2d = nullptr, d2h = nullptr;
    uint8_t* d_in[kDepth] = {nullptr, nullptr, nullptr};   // [left images | right images] of one batch
    size_t in_bytes = 0;
    uint8_t* d_out[kDepth] = {nullptr, nullptr, nullptr};  // packed results of one batch: kps | desc | counts | right_u | depth | n_match
    size_t out_bytes = 0;
    hipEvent_t ev_h2d[kDepth] = {nullptr, nullptr, nullptr}, ev_in_free[kDepth] = {nullptr, nullptr, nullptr},
               ev_out_ready[kDepth] = {nullptr, nullptr, nullptr}, ev_done[kDepth] = {nullptr, nullptr, nullptr};
    int64_t next_ticket = 0;
    int32_t n_pairs_of[kDepth] = {0, 0, 0};  // pairs of the ticket that last used buffer set b: the packed layout depends on it
  } hs;
  Lane main;
  std::vector<std::unique_ptr<Lane>> slot_lane;  // [max_images], entries created lazily under slot_lane_mu
  std::mutex slot_lane_mu;
  bool use_graphs = true;
  // the stereo match of a device-resident batch runs on its own stream: it is latency-bound and reads only the keypoint /
  // descriptor arrays and the pyramid, so the NEXT batch's copy-in, resize and FAST (second pyramid buffer) run under it
  uint8_t* d_pyr_alt = nullptr;
  hipStream_t stereo_stream = nullptr;
  hipEvent_t ev_brief_done = nullptr, ev_stereo_done = nullptr;
  std::atomic<bool> stereo_pending{false};  // (read by slot calls on other threads)
  bool pipeline_stereo = true;
  hipStream_t blur_stream = nullptr;
  int fast_cpw = 0;  // ORBFE_FAST_CPW: cells per k_fast wave (0: one for small launches, four for large ones)
  hipEvent_t ev_blur_go = nullptr, ev_blur_done = nullptr;
  // r6: the k_fast launches of the SMALL levels (fewer than 131072 cell x images: they run one or four cells per wave and each is little more
  // than a ramp and a tail) go to fast_stream beside the launches of the large levels.  fast_side_mask: -1 = that rule (default), else the
  // levels as a bit mask (ORBFE_FAST_SIDE_MASK; 0: everything on the context stream, r5's schedule)
  int64_t fast_side_mask = -1;
  bool fast_side_merge = false;  // ORBFE_FAST_SIDE_MERGE=1 (experiment): the side levels as one launch
  hipStream_t fast_stream = nullptr;
  hipEvent_t ev_fast_go = nullptr, ev_fast_side_done = nullptr;

  // geometry (host copies)
  std::vector<LevelDev> lv;
  std::vector<CellDev> cells;
  std::vector<ResizeTap> taps;
  std::vector<RsTile> rs_tile_tab;
  RsTile* d_rs_tiles = nullptr;
  int umax[16];
  int blur_taps[7];
  int n_cells_total = 0, rs_tiles = 0, bl_tiles = 0;
  int kp_cap = 0;  // keypoints one image can yield = stride of every per-image array (>= n_features, see build_geometry)
  std::vector<RsRegion> rs_regions;  // region-driven resize (k_resize_regions): level 0 staged once for all levels
  RsRegion* d_rs_regions = nullptr;
  std::vector<RgXTap> rg_xtaps;
  std::vector<RgYTap> rg_ytaps;
  RgXTap* d_rg_xtaps = nullptr;
  RgYTap* d_rg_ytaps = nullptr;
  int rg_tile_bytes = 0, rg_xt_bytes = 0, rg_yt_bytes = 0;
  int rg_pq = 0;  // the regions' common LDS row pitch in 16-byte units when the kernel may take it as a constant (0: per region, at run time)
  int rs_n[3] = {0, 0, 0}, rs_bytes[3] = {0, 0, 0};  // resize tiles of 64x64 / 64x32 / 64x16 outputs (in this order) and their LDS
  size_t img_pitch = 0;      // bytes per image in pyr / blur
  size_t scratch_pitch = 0;  // uint32 records per image
  QtGroups qt_groups_of[3];  // the same for 1, 2 and 4 waves per image (picked by launch size)
  uint32_t pyr_spare_off = 0;    // offset of 256 spare bytes inside every image's block of d_pyr / d_blur (behind the last plane)
  MbGeom mb = {};                // k_blur_mfma's geometry; mb_ok: built for this tap set and pyramid (ORBFE_BLUR_MFMA=0 turns it off)
  bool mb_ok = false;
  uint8_t *d_mb_tx = nullptr, *d_mb_ty = nullptr;
  std::vector<uint8_t> mb_tx, mb_ty;
  int32_t* d_qt_next = nullptr;  // [max_images]: the per-image level counter of launches whose waves pull their levels (QtGroups::order)
  QtGroups qt_single;        // one level per wave: launches too small to fill the wave slots (a frame or two: the chain of several trees in one wave would only add latency)
  int rec_cap = 0;           // upper bound of candidate records one quadtree wave keeps in LDS (launch picks <= this)
  int n_cu = 256;            // compute units of the device
  int node_cap = 0, sort_cap = 0;
  int lvl_max_pw[ORBFE_MAX_LEVELS] = {0}, lvl_max_ph[ORBFE_MAX_LEVELS] = {0};  // largest FAST cell patch per level (sizes the LDS of k_fast)

  // device
  LevelDev* d_lv = nullptr;
  CellDev* d_cells = nullptr;
  ResizeTap* d_taps = nullptr;
  int8_t* d_pattern = nullptr;
  uint8_t *d_pyr = nullptr, *d_blur = nullptr;
  uint32_t *d_scr_a = nullptr, *d_scr_b = nullptr, *d_scr_c = nullptr;  // candidate lists | quadtree home / bounce buffers
  uint16_t* d_qt_tabs = nullptr;  // per level: the quadtree pre-partition's coordinate -> code tables (LevelDev::qt_tab_off)
  std::vector<uint16_t> qt_tabs;
  uint8_t* d_qt_big = nullptr;  // node tables + sort buffers of the levels whose quota does not fit one CU's LDS (qt_big_pitch bytes per image)
  size_t qt_big_pitch = 0;
  uint32_t* d_sel = nullptr;
  int32_t *d_sel_count = nullptr, *d_n_cand = nullptr, *d_n_kp = nullptr;
  // a frame or two (contexts of <= 16 slots): FAST's candidate lists in fast_shards shards per level (k_fast.hip: the reservations of a pair's
  // 2462 one-cell waves on sixteen counters were 20 of the launch's 32 us); [slot][level][shard] counters, which slots hold sharded lists
  int fast_shards = 1;
  int32_t* d_n_cand_sh = nullptr;
  std::vector<uint8_t> slot_sharded;
  orbfe_keypoint* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  KpAux* d_aux = nullptr;
  double* d_theta = nullptr;
  uint4* d_kpl = nullptr;        // level-major keypoint list {x | y<<16, level | response<<8, plane offset, row stride}
  int2* d_moments = nullptr;     // per keypoint (m10, m01)
  double2* d_sincos = nullptr;   // per keypoint (sin, cos) of the orientation
  KpX* d_kx = nullptr;           // per keypoint x (level-0 coordinates) + octave / patch centre: what the stereo match reads per candidate
  StereoRowsBuf st_rows = {nullptr, nullptr, nullptr, nullptr};  // the row-parallel matcher's buffers (batches)
  uint32_t* d_rowoff = nullptr;  // per pair: offsets[height + 1] of the right image's row table (createRowIndexDB)
  uint16_t* d_rowlist = nullptr; // per pair: the table's entries, row_list_cap = n_features x the widest band
  int row_list_cap = 0;
  // Contexts of a few slots (the one-frame-at-a-time call shapes): per-SLOT row tables, built by the descriptor launch of every
  // extraction of one or two images, so that orbfe_stereo_match launches k_stereo alone.  slot_table_ok[s]: slot s's table belongs to
  // its current features; pair_count_zero[p]: the match counter of pair p has not been counted into since an extraction zeroed it.
  uint32_t* d_rowoff_slot = nullptr;
  int32_t* d_rt_flags = nullptr;  // [slot][8]: part totals of a slot's row table while k_brief's eight spare workgroups build it (rowtable_build_part)
  uint16_t* d_rowlist_slot = nullptr;
  std::unique_ptr<std::atomic<uint8_t>[]> slot_table_ok, pair_count_zero;
  // The frame grid of a slot (VirtualFrame::initGrid) is kept from one guided search to the next: Tracking searches the same frame two to
  // four times.  grid_key[s] = generation << 32 | (rows << 16 | cols) of the grid held for slot s's current keypoints, low half 0: none.
  // A new extraction into the slot or an in-place undistortion bumps the generation and clears the key in ONE atomic step
  // (grid_invalidate) -- slot calls do that without the API lock -- and a search publishes the grid it built only by compare-exchange
  // from the state it saw before building: a slot rewritten in between leaves no stale grid marked valid.  Allocated on
  // first use, grid_cells entries per slot.
  int32_t *d_grid_off = nullptr, *d_grid_feat = nullptr;
  size_t grid_cells = 0;
  std::unique_ptr<std::atomic<uint64_t>[]> grid_key;
  double *d_right_u = nullptr, *d_depth = nullptr;
  int32_t *d_n_match = nullptr, *d_best_right = nullptr, *d_best_dist = nullptr;
  // generic staging for match / BA calls
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  // pinned host staging for small result reads
  int32_t* h_counts = nullptr;
  // local BA with the Levenberg-Marquardt control on the device (k_lm.hip): a host-mapped byte the control kernel polls -- the caller's
  // stop flag is mirrored into it while the call waits -- and the page-locked copy of the state record
  volatile uint8_t* h_abort = nullptr;
  LmState* h_lm_state = nullptr;
  bool lm_on_device = true;  // ORBFE_LBA_HOST_LM=1: round 2's host-driven loop (kept for A/B runs and for > LM_CHOL_MAX_NB free keyframes)

  // profiling
  int prof = 0;  // 0 off | 1 every stage timed alone (overlaps and graphs off) | 2..: only stage (prof - 2) timed, in the production schedule
  hipEvent_t ev[2 * ORBFE_STAGE_COUNT];
  bool ev_init = false;
  double stage_ms[ORBFE_STAGE_COUNT];
  int64_t stage_launches[ORBFE_STAGE_COUNT];
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
  std::vector<hipEvent_t> ev_pool;
};

struct ApiLock {
  std::unique_lock<std::recursive_mutex> lk;
  explicit ApiLock(orbfe_ctx* c) {
    if (c) lk = std::unique_lock<std::recursive_mutex>(c->api_mu);
  }
};

orbfe_status fail(orbfe_ctx* c, orbfe_status st, const char* fmt, ...);

#define HIP_TRY(c, expr)                                                                          \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail((c), ORBFE_EDEVICE, "%s -> %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// ---- OpenCV rounding (cvRound = round half to even, cvFloor, cvCeil) ------------------------------
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) {
  int i = (int)v;
  return i - (i > v);
}
static inline int cv_ceil_f(float v) {
  int i = (int)v;
  return i + (i < v);
}
static inline short sat_short_f(float v) { return (short)std::min(32767, std::max(-32768, cv_round_f(v))); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
template <typename T>
static orbfe_status dev_alloc(orbfe_ctx* c, T** p, size_t count) {
  HIP_TRY(c, hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
  return ORBFE_OK;
}
#define TRY(expr)                          \
  do {                                     \
    orbfe_status st_ = (expr);             \
    if (st_ != ORBFE_OK) return st_;       \
  } while (0)

// include/orbfe.h: between orbfe_extract_slot_begin and orbfe_extract_slot_end a slot accepts no other call.  Every entry point that reads or
// writes slots [slot0, slot0 + n) asks here first: the begun extraction runs on the slot's own lane, which the context stream does not wait for.
static inline orbfe_status slots_idle(orbfe_ctx* c, int slot0, int n, const char* who) {
  std::lock_guard<std::mutex> lk(c->slot_lane_mu);
  const int hi = std::min(slot0 + n, (int)c->slot_lane.size());
  for (int k = std::max(slot0, 0); k < hi; ++k)
    if (c->slot_lane[(size_t)k] && c->slot_lane[(size_t)k]->pending.load())
      return fail(c, ORBFE_EBADARG, "%s: slot %d has an outstanding orbfe_extract_slot_begin (call orbfe_extract_slot_end first)", who, k);
  return ORBFE_OK;
}

orbfe_status ensure_tmp(orbfe_ctx* c, size_t bytes);
orbfe_status ensure_stage(orbfe_ctx* c, orbfe_ctx::Lane& ln, size_t bytes);
orbfe_status ensure_stage(orbfe_ctx* c, size_t bytes);

// ---- stage timing ---------------------------------------------------------------------------------
static inline bool timed(const orbfe_ctx* c, int stage) { return c->prof == 1 || c->prof == stage + 2; }

struct StageTimer {
  orbfe_ctx* c;
  int stage;
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t stream;
  StageTimer(orbfe_ctx* ctx, int st, hipStream_t s, bool enabled = true) : c(ctx), stage(st), stream(s) {
    if (!enabled || !timed(c, st)) return;
    auto get = [&]() {
      hipEvent_t e = nullptr;
      if (!c->ev_pool.empty()) {
        e = c->ev_pool.back();
        c->ev_pool.pop_back();
      } else if (hipEventCreate(&e) != hipSuccess)
        e = nullptr;
      return e;
    };
    a = get();
    b = get();
    if (a) (void)hipEventRecord(a, stream);
  }
  ~StageTimer() {
    if (!a || !b) return;
    (void)hipEventRecord(b, stream);
    c->pending.push_back({stage, {a, b}});
  }
};

void drain_timers(orbfe_ctx* c);
orbfe_status join_stereo(orbfe_ctx* c);

// ---- the launch sequence for slots [0, n_img) ---------------------------------------------------------
// Slots [img0, img0 + n_img) on stream `st`.  Every per-image array is offset on the host, so the kernels index from 0.
// level 0 read straight from the caller's images by the resize (device batches): see k_resize_regions
// results delivered by the kernels themselves into page-locked host memory (the host-pointer path of a frame or two)
struct HostMirror {
  orbfe_keypoint* kps;  // [n_img][n_features], nullable
  uint8_t* desc;        // [n_img][n_features][32], nullable
  int32_t* n_kp;        // [n_img]
};
struct ExtLevel0 {
  const uint8_t *left, *right;  // image p of the batch at left / right + p * pitch
  size_t pitch;
  int stride;
  uint32_t bytes;               // size of one image
  hipEvent_t inputs_free;       // nullable: recorded once the resize (which also writes level 0 of the pyramid) is done with the caller's images
};
void grid_invalidate(orbfe_ctx* c, int slot);
void note_slots_written(orbfe_ctx* c, int s0, int n, bool small);
orbfe_status run_extract(orbfe_ctx* c, hipStream_t st, int img0, int n_img, hipEvent_t before_lists = nullptr, bool timing = true,
                         const ExtLevel0* ext = nullptr, const HostMirror* mirror = nullptr);


// Where the packed results of a batch go on the device (host-image stream), and the events around that copy.
struct PackDst {
  uint8_t* base;
  hipEvent_t wait_free;  // the buffer's previous contents have been downloaded
  hipEvent_t ready;      // recorded once the results are in the buffer
  hipEvent_t in_free;    // recorded once the input images have been consumed (level 0 of every pyramid written)
};
struct PackLayout {
  size_t o_kps, o_desc, o_cnt, o_ru, o_dp, o_nm, total;
};
static inline PackLayout pack_layout(const orbfe_ctx* c, int n_pairs) {
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1), n = (size_t)n_pairs;
  PackLayout l;
  l.o_kps = 0;
  l.o_desc = l.o_kps + align_up(2 * n * NF * sizeof(orbfe_keypoint), 256);
  l.o_cnt = l.o_desc + align_up(2 * n * NF * 32, 256);
  l.o_ru = l.o_cnt + align_up(2 * n * 4, 256);
  l.o_dp = l.o_ru + align_up(n * NF * 8, 256);
  l.o_nm = l.o_dp + align_up(n * NF * 8, 256);
  l.total = l.o_nm + align_up(n * 4, 256);
  return l;
}

// the batched stereo step on images resident in HBM (orbfe_stereo_batch_device; the host-image stream runs it per batch with a PackDst)
orbfe_status batch_device_core(orbfe_ctx* c, const uint8_t* d_left, const uint8_t* d_right, size_t stride, size_t image_pitch, int32_t n_pairs,
                               float fx, float bf, const PackDst* pack);
