// orbfe_api.hip -- host side of the C-ABI declared in include/orbfe.h: geometry tables, device buffers,
// stream/event plumbing and the launch sequence.  No OpenCV, no torch, no CPU fallback.
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
                           const uint8_t* src_a, const uint8_t* src_b, size_t src_pitch, int src_stride, uint32_t src_bytes, int copy_l0, int32_t* d_zero, int n_zero);
void launch_blur(hipStream_t s, const LevelDev* d_lv, int n_levels, int tile_first, int n_tiles, const uint8_t* d_pyr, uint8_t* d_blur,
                 size_t img_pitch, const int taps[7], int n_img);
void launch_load_level0(hipStream_t st, const uint8_t* d_src, const uint8_t* d_src_b, size_t src_stride, size_t src_pitch, uint8_t* d_pyr,
                        size_t img_pitch, uint32_t plane_off, int dst_stride, int w, int h, int slot0, int slot_step, int n_img);
// k_fast.hip
void launch_fast(hipStream_t s, const LevelDev* d_lv, const CellDev* d_cells, const LevelDev* h_lv, const int* lvl_max_pw,
                 const int* lvl_max_ph, const uint8_t* d_pyr, size_t img_pitch, int t_hi, int t_lo, uint32_t* d_cand, size_t cand_pitch,
                 int32_t* d_n_cand, int n_levels, int n_img, int cpw_force);
// k_quadtree.hip
size_t quadtree_lds_bytes(int node_cap, int rec_cap, int sort_cap);
hipError_t quadtree_configure(size_t lds_bytes);
void launch_quadtree(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint32_t* d_cand, uint32_t* d_scr_b, uint32_t* d_scr_c,
                     size_t scratch_pitch, uint32_t* d_sel, int32_t* d_sel_count, int n_features, const int32_t* d_n_cand, int node_cap, int sort_cap,
                     int rec_cap, int n_img, int batch, const QtGroups& groups, int n_groups, int waves_per_tree, uint8_t* d_big, size_t big_pitch,
                     const uint16_t* d_qt_tabs, const uint8_t* blur_pyr, uint8_t* blur_out, size_t img_pitch, const int* blur_taps, int blur_tiles);
bool quadtree_build_tables(const LevelDev& L, std::vector<uint16_t>& out);
// k_brief.hip
void launch_orient_brief(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint8_t* d_pyr, const uint8_t* d_blur,
                         size_t img_pitch, const uint32_t* d_sel, const int32_t* d_sel_count, int n_features,
                         const int8_t* d_pattern, const int umax[16], orbfe_keypoint* d_kps, uint8_t* d_desc, KpAux* d_aux,
                         int32_t* d_n_kp, double* d_theta, int2* d_moments, double2* d_sincos, float* d_kx,
                         uint4* d_kpl, int rows0, int n_img, hipEvent_t before_brief, hipEvent_t before_lists, orbfe_keypoint* h_kps, uint8_t* h_desc, int32_t* h_n_kp,
                         bool fuse_small, uint32_t* d_rowoff_slot = nullptr, uint16_t* d_rowlist_slot = nullptr,
                         int32_t* d_n_match = nullptr, int rt_rows = 0, int rt_list_cap = 0, int rt_slot0 = 0);
// k_match.hip
void launch_match_bruteforce(hipStream_t s, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, const uint32_t* d_off,
                             const uint32_t* d_cand, int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second);
void launch_stereo(hipStream_t s, const LevelDev* d_lv, int n_levels, const uint8_t* d_pyr, size_t img_pitch, const orbfe_keypoint* d_kps,
                   const uint8_t* d_desc, const KpAux* d_aux, const float* d_kx, uint32_t* d_rowoff, uint16_t* d_rowlist, int rows, int list_cap,
                   const int32_t* d_n_kp, int n_features, float fx, float bf, int cols0, int mean_threshold, double* d_right_u, double* d_depth, int32_t* d_n_match, int32_t* d_best_right,
                   int32_t* d_best_dist, int slot_l0, int slot_r0, int slot_step, int pair0, int n_pairs, double* h_right_u, double* h_depth,
                   int32_t* h_best_right, int32_t* h_best_dist, bool table_ready = false);
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

static const int8_t kEmbeddedPattern[256][4] = {
#include "brief_pattern.inc"
};
static const int kGaussTaps[2][7] = {{18, 34, 48, 56, 48, 34, 18}, {18, 34, 49, 55, 49, 34, 18}};
static const int kMeanThreshold = 75;  // ORBMatcher::mnMeanThreshold (ORBMatcher.cc:1088)

// The text of the last failed call is kept PER THREAD (like dlerror): slot calls of one context run on several threads at once.
static thread_local std::string g_last_error;

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
  int fast_side_from = 0;  // k_fast launches of levels >= this run on the blur stream beside the large levels (ORBFE_FAST_SIDE_FROM; 0: off -- the default
                           // since the level-0 blur occupies that stream until well into FAST: the small levels queued behind it, 3 / 5 / 0: 5.75 / 5.72 / 5.71 ms)

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
  int rs_n[3] = {0, 0, 0}, rs_bytes[3] = {0, 0, 0};  // resize tiles of 64x64 / 64x32 / 64x16 outputs (in this order) and their LDS
  size_t img_pitch = 0;      // bytes per image in pyr / blur
  size_t scratch_pitch = 0;  // uint32 records per image
  QtGroups qt_groups_of[3];  // the same for 1, 2 and 4 waves per image (picked by launch size)
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
  orbfe_keypoint* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  KpAux* d_aux = nullptr;
  double* d_theta = nullptr;
  uint4* d_kpl = nullptr;        // level-major keypoint list {x | y<<16, level | response<<8, plane offset, row stride}
  int2* d_moments = nullptr;     // per keypoint (m10, m01)
  double2* d_sincos = nullptr;   // per keypoint (sin, cos) of the orientation
  float* d_kx = nullptr;         // per keypoint x (level-0 coordinates), SoA copy for the stereo candidate scan
  uint32_t* d_rowoff = nullptr;  // per pair: offsets[height + 1] of the right image's row table (createRowIndexDB)
  uint16_t* d_rowlist = nullptr; // per pair: the table's entries, row_list_cap = n_features x the widest band
  int row_list_cap = 0;
  // Contexts of a few slots (the one-frame-at-a-time call shapes): per-SLOT row tables, built by the descriptor launch of every
  // extraction of one or two images, so that orbfe_stereo_match launches k_stereo alone.  slot_table_ok[s]: slot s's table belongs to
  // its current features; pair_count_zero[p]: the match counter of pair p has not been counted into since an extraction zeroed it.
  uint32_t* d_rowoff_slot = nullptr;
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

static orbfe_status fail(orbfe_ctx* c, orbfe_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  (void)c;
  g_last_error = buf;
  return st;
}

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
    cand_base += (uint32_t)align_up(L.cand_cap, 32);
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
    c->rg_xt_bytes = (int)align_up((size_t)c->rg_xt_bytes, 16);
    if (!ok || c->rg_tile_bytes + c->rg_xt_bytes + c->rg_yt_bytes > 60 * 1024) c->rs_regions.clear();  // fall back to the tile classes
  }
  rs_tiles = (int)c->rs_tile_tab.size();
  c->n_cells_total = cell_base;
  c->rs_tiles = rs_tiles;
  c->bl_tiles = bl_tiles;
  c->img_pitch = align_up(plane_off, 4096);
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
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return c->lv[a].quota > c->lv[b].quota; });
      auto deal = [&](int ng, QtGroups* out) {
        std::memset(out, 0, sizeof *out);
        std::vector<double> load(ng, 0.0);
        for (int l : order) {
          int g = 0;
          for (int k = 1; k < ng; ++k)
            if (load[k] < load[g]) g = k;
          load[g] += (double)c->lv[l].quota + 1.0;
          out->mask[g] |= 1u << l;
        }
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

static orbfe_status ensure_tmp(orbfe_ctx* c, size_t bytes) {
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
static orbfe_status ensure_stage(orbfe_ctx* c, orbfe_ctx::Lane& ln, size_t bytes) {
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
static orbfe_status ensure_stage(orbfe_ctx* c, size_t bytes) { return ensure_stage(c, c->main, bytes); }

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

static void drain_timers(orbfe_ctx* c) {
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
static orbfe_status join_stereo(orbfe_ctx* c) {
  if (c->stereo_pending) {
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_stereo_done, 0));
    c->stereo_pending = false;
  }
  return ORBFE_OK;
}

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
// slots [s0, s0 + n) have just been (or are about to be) rewritten by an extraction; small: one that also built their row tables
static void grid_invalidate(orbfe_ctx* c, int slot) {
  if (!c->grid_key) return;
  std::atomic<uint64_t>& a = c->grid_key[(size_t)slot];
  uint64_t v = a.load();
  while (!a.compare_exchange_weak(v, ((v >> 32) + 1) << 32)) {
  }
}
static void note_slots_written(orbfe_ctx* c, int s0, int n, bool small) {
  for (int s = s0; s < s0 + n && s < c->cfg.max_images; ++s) grid_invalidate(c, s);
  if (!c->slot_table_ok) return;
  for (int s = s0; s < s0 + n && s < c->cfg.max_images; ++s) {
    c->slot_table_ok[(size_t)s] = small ? 1 : 0;
    if (small) c->pair_count_zero[(size_t)(s >> 1)] = 1;
  }
}
static orbfe_status run_extract(orbfe_ctx* c, hipStream_t st, int img0, int n_img, hipEvent_t before_lists = nullptr,
                                bool timing = true, const ExtLevel0* ext = nullptr, const HostMirror* mirror = nullptr) {
  // timing = false: a slot lane (orbfe_extract_slot) -- several of them run at once, so nothing shared by the context is touched:
  // no stage timers (their event lists belong to the main lane), no second stream
  // before_lists: event the keypoint-list / orientation / descriptor kernels must wait for (the previous batch's stereo match still
  // reads the arrays they rewrite); callers that do not pipeline have joined the stereo stream already
  const int nl = c->cfg.n_levels;
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  const size_t i0 = (size_t)img0;
  note_slots_written(c, img0, n_img, n_img <= 2);
  uint8_t* pyr = c->d_pyr + i0 * c->img_pitch;
  uint8_t* blur = c->d_blur + i0 * c->img_pitch;
  int32_t* n_cand = c->d_n_cand + i0 * nl;
  const bool overlap_blur = timing && c->blur_stream && c->prof != 1 && n_img >= 32;  // a frame or two: nothing to hide, only event latency to add (measured r3: the blur of one pair on the second stream, inside the captured graph: extract_batch 0.306 -> 0.366 ms)
  // the blur of LEVEL 0 needs nothing but the copy-in: it starts beside the resize (a third of the blur's work out of the way of the
  // moments, which are as memory-bound as it is and take the sum of the two times when they meet)
  const int l0_tiles = (overlap_blur && nl > 1) ? c->lv[1].bl_tile_base : 0;
  bool blur_queued = false;
  if (l0_tiles > 0 && !ext) {
    HIP_TRY(c, hipEventRecord(c->ev_blur_go, st));
    HIP_TRY(c, hipStreamWaitEvent(c->blur_stream, c->ev_blur_go, 0));
    launch_blur(c->blur_stream, c->d_lv, nl, 0, l0_tiles, pyr, blur, c->img_pitch, c->blur_taps, n_img);
  }
  // FAST's candidate counters are zeroed by the resize kernel (block 0): a memset between the blur and FAST is one more launch in the
  // chain -- 4.6 us of a 0.2 ms frame
  const bool zeroed_by_resize = !c->rs_regions.empty();  // (empty: the geometry rules the region-driven resize out -- the per-class tile launches)
  {
    StageTimer t(c, ORBFE_STAGE_RESIZE, st, timing);
    if (zeroed_by_resize)
      launch_resize_regions(st, c->d_lv, nl, c->d_rs_regions, (int)c->rs_regions.size(), c->rg_tile_bytes, c->rg_xt_bytes, c->rg_yt_bytes,
                            c->d_rg_xtaps, c->d_rg_ytaps, pyr, c->img_pitch, n_img, ext ? ext->left : pyr + c->lv[0].plane_off,
                            ext ? ext->right : nullptr, ext ? ext->pitch : c->img_pitch, ext ? ext->stride : c->lv[0].stride,
                            ext ? ext->bytes : 0xFFFFFFFFu, ext ? 1 : 0, n_cand, n_img * nl);
    else
      launch_resize(st, c->d_lv, c->d_rs_tiles, c->rs_n, c->rs_bytes, c->d_taps, pyr, c->img_pitch, n_img);
  }
  if (ext) {  // the resize has also written level 0 of the pyramid (its blocks of the caller's images): the images are free, the level-0 blur may start
    if (ext->inputs_free) HIP_TRY(c, hipEventRecord(ext->inputs_free, st));
    if (l0_tiles > 0) {
      // ... and here, where the resize has just produced every level at once, the WHOLE blur goes to the second stream in one launch
      // (level 0 first): it runs beside FAST, mostly in the slots the eight launches leave at their tails, instead of the levels
      // above 0 waiting for FAST's last launch to drain (5.52 -> 5.50 ms per 512 pairs, four same-box rounds; two launches the same)
      HIP_TRY(c, hipEventRecord(c->ev_blur_go, st));
      HIP_TRY(c, hipStreamWaitEvent(c->blur_stream, c->ev_blur_go, 0));
      {
        StageTimer t(c, ORBFE_STAGE_BLUR, c->blur_stream);  // (events on the stream the kernel is launched on)
        launch_blur(c->blur_stream, c->d_lv, nl, 0, c->bl_tiles, pyr, blur, c->img_pitch, c->blur_taps, n_img);
      }
      HIP_TRY(c, hipEventRecord(c->ev_blur_done, c->blur_stream));
      blur_queued = true;
    }
  }
  // Only the descriptors read the blurred planes, so the blur need not sit between resize and FAST: the levels above 0 are issued on a second
  // stream once FAST is done (beside FAST, which saturates the vector units, they cost more than they hide: +2 %) and runs UNDER the quadtree, which keeps 8 waves per CU busy with dependent LDS steps and leaves
  // the SIMDs idle (the blur uses no LDS, the quadtree all of it).  With stage timing on, or when several chunks share the
  // context, the blur stays in line.  (Measured and dropped: starting each level's quadtree under FAST of the smaller levels on
  // a third stream -- the tree waves then share their SIMDs with a VALU-saturating kernel and the dependent chain stretches:
  // 3.04 -> 4.6 ms per 128 pairs.)
  // A frame or two without stage timing: the blur's tiles ride in the quadtree launch below as extra workgroups (the launch has sixteen
  // tree workgroups per image on 256 CUs; the blurred planes are read by the descriptors only) -- one launch and its ~15 us off the chain.
  // With stage timing on the blur keeps its own launch so that the stages are timed apart.
  const bool qt_small = nl > 0 && (long long)nl * n_img <= c->n_cu;  // (= launch_quadtree's four-waves-per-tree condition below)
  const bool blur_in_qt = !overlap_blur && qt_small && c->prof == 0;
  if (!overlap_blur && !blur_in_qt) {
    StageTimer t(c, ORBFE_STAGE_BLUR, st, timing);
    launch_blur(st, c->d_lv, nl, 0, c->bl_tiles, pyr, blur, c->img_pitch, c->blur_taps, n_img);
  }
  if (!zeroed_by_resize) HIP_TRY(c, hipMemsetAsync(n_cand, 0, sizeof(int32_t) * (size_t)n_img * nl, st));
  {
    StageTimer t(c, ORBFE_STAGE_FAST, st, timing);
    launch_fast(st, c->d_lv, c->d_cells, c->lv.data(), c->lvl_max_pw, c->lvl_max_ph, pyr, c->img_pitch, c->cfg.fast_hi, c->cfg.fast_lo,
                c->d_scr_a + i0 * c->scratch_pitch, c->scratch_pitch, n_cand, nl, n_img,
                c->fast_cpw);
  }
  if (overlap_blur && !blur_queued) {
    HIP_TRY(c, hipEventRecord(c->ev_blur_go, st));
    HIP_TRY(c, hipStreamWaitEvent(c->blur_stream, c->ev_blur_go, 0));
    {
      StageTimer t(c, ORBFE_STAGE_BLUR, c->blur_stream);  // (events on the stream the kernel is launched on)
      launch_blur(c->blur_stream, c->d_lv, nl, l0_tiles, c->bl_tiles - l0_tiles, pyr, blur, c->img_pitch, c->blur_taps, n_img);
    }
    HIP_TRY(c, hipEventRecord(c->ev_blur_done, c->blur_stream));
  }
  {
    StageTimer t(c, ORBFE_STAGE_QUADTREE, st, timing);
    // LDS residency of the candidate records is traded against concurrency: the kernel is latency-bound (one wave per
    // tree, 40-150 dependent steps), so what matters most is that EVERY tree of the launch is resident at once; the
    // records go to LDS only as far as that still holds (measured at 1024 trees: 4 trees/CU 0.59 ms, 3 trees/CU 0.96 ms).
    // several levels per wave only where one wave per level would overfill the chip: waves per image so that the launch has about
    // sixteen tree waves per CU (one round) -- 4 waves per image from 1024 images, 2 from 2048, 1 from 4096 on 256 CUs
    int gsel = -1;  // -1: one wave per level
    if (nl > 4) {
      const long long per8 = (long long)c->n_cu * 16;  // sixteen tree waves per CU: 16-byte nodes, 128 VGPRs (k_quadtree.hip)
      if ((long long)n_img * 1 >= per8) gsel = 0;
      else if ((long long)n_img * 2 >= per8) gsel = 1;
      else if ((long long)n_img * 4 >= per8) gsel = 2;
    }
    const bool grouped = gsel >= 0;
    const int n_groups = gsel >= 0 ? std::min(nl, 1 << gsel) : nl;
    const QtGroups& qt_tab = gsel >= 0 ? c->qt_groups_of[gsel] : c->qt_single;
    const int trees = n_groups * n_img;
    const int per_cu = (trees + c->n_cu - 1) / c->n_cu;
    const size_t lds_cu = 160 * 1024 - 2048;
    const size_t node_bytes = quadtree_lds_bytes(c->node_cap, 0, c->sort_cap);
    size_t budget = lds_cu / (size_t)std::max(per_cu, 1);
    budget -= budget % 512;
    const int rec_cap = budget > node_bytes ? (int)std::min<size_t>((budget - node_bytes) / 4, (size_t)c->rec_cap) : 0;
    launch_quadtree(st, c->d_lv, nl, c->d_scr_a + i0 * c->scratch_pitch, c->d_scr_b + i0 * c->scratch_pitch,
                    c->d_scr_c + i0 * c->scratch_pitch, c->scratch_pitch, c->d_sel + i0 * NF, c->d_sel_count + i0 * nl,
                    c->cfg.n_features, n_cand, c->node_cap, c->sort_cap, rec_cap, n_img, 1, qt_tab, n_groups,
                    // helper waves for the data-parallel phases of a tree where the launch leaves the chip empty (a frame or two)
                    (!grouped && trees * 4 <= c->n_cu * 4) ? 4 : 1, c->d_qt_big ? c->d_qt_big + i0 * c->qt_big_pitch : nullptr,
                    c->qt_big_pitch, c->d_qt_tabs, blur_in_qt ? pyr : nullptr, blur, c->img_pitch, c->blur_taps, c->bl_tiles);
  }
  {
    StageTimer t(c, ORBFE_STAGE_BRIEF, st, timing);
    launch_orient_brief(st, c->d_lv, nl, pyr, blur, c->img_pitch, c->d_sel + i0 * NF, c->d_sel_count + i0 * nl, c->cfg.n_features,
                        c->d_pattern, c->umax, c->d_kps + i0 * NF, c->d_desc + i0 * NF * 32, c->d_aux + i0 * NF, c->d_n_kp + i0,
                        c->d_theta + i0 * NF, c->d_moments + i0 * NF, c->d_sincos + i0 * NF, c->d_kx + i0 * NF,
                        c->d_kpl + i0 * NF, c->cfg.height, n_img,
                        overlap_blur ? c->ev_blur_done : nullptr, before_lists, mirror ? mirror->kps : nullptr, mirror ? mirror->desc : nullptr,
                        mirror ? mirror->n_kp : nullptr, true, n_img <= 2 ? c->d_rowoff_slot : nullptr, c->d_rowlist_slot, c->d_n_match, c->cfg.height,
                        c->row_list_cap, img0);
    // (the per-slot row tables: valid after an extraction of one or two images, stale after any other -- the flags are host state and
    //  this function also runs under graph CAPTURE, so the callers set them: extract_lane / note_slots_written)
  }
  HIP_TRY(c, hipGetLastError());
  return ORBFE_OK;
}

struct StereoHostOut {  // page-locked destinations for the results of one pair, written by k_stereo itself (nullable members)
  double *right_u, *depth;
  int32_t *best_right, *best_dist;
};
static orbfe_status run_stereo(orbfe_ctx* c, hipStream_t st, int slot_l0, int slot_r0, int slot_step, int pair0, int n_pairs, float fx,
                               float bf, const StereoHostOut* ho = nullptr, bool table_ready = false, bool timing = true) {
  // (c->d_pyr is read here, at launch time: a later swap of the pyramid buffers does not affect a launch already queued)
  // (the match counters are zeroed by k_rowtable)
  {
    StageTimer t(c, ORBFE_STAGE_STEREO, st, timing);  // (timing = false: a slot lane, which touches nothing the context shares)
    launch_stereo(st, c->d_lv, c->cfg.n_levels, c->d_pyr, c->img_pitch, c->d_kps, c->d_desc, c->d_aux, c->d_kx,
                  table_ready ? c->d_rowoff_slot : c->d_rowoff, table_ready ? c->d_rowlist_slot : c->d_rowlist,
                  c->cfg.height, c->row_list_cap, c->d_n_kp,
                  c->cfg.n_features, fx, bf,
                  c->cfg.width, kMeanThreshold, c->d_right_u, c->d_depth, c->d_n_match, c->d_best_right, c->d_best_dist, slot_l0,
                  slot_r0, slot_step, pair0, n_pairs, ho ? ho->right_u : nullptr, ho ? ho->depth : nullptr, ho ? ho->best_right : nullptr,
                  ho ? ho->best_dist : nullptr, table_ready);
  }
  HIP_TRY(c, hipGetLastError());
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
                  c->d_scr_a, c->d_scr_c,   c->d_scr_b,  c->d_sel,     c->d_sel_count, c->d_n_cand, c->d_n_kp,
                  c->d_kps,  c->d_desc,      c->d_aux,    c->d_theta, c->d_moments, c->d_sincos, c->d_kx, c->d_kpl,   c->d_right_u, c->d_depth, c->d_n_match,
                  c->d_best_right, c->d_best_dist, c->d_tmp, c->d_rowoff, c->d_rowlist, c->d_rowoff_slot, c->d_rowlist_slot, c->d_rs_regions, c->d_rg_xtaps, c->d_rg_ytaps, c->d_qt_big, c->d_qt_tabs};
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
  if (cfg->width <= 0 || cfg->height <= 0 || cfg->n_features < 0 || cfg->n_features > 65535 || cfg->n_levels < 1 || cfg->n_levels > ORBFE_MAX_LEVELS ||
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
    const char* ov = getenv("ORBFE_OVERLAP_BLUR");
    if (!ov || atoi(ov) != 0) {
      if (hipStreamCreateWithFlags(&c->blur_stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&c->ev_blur_go, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&c->ev_blur_done, hipEventDisableTiming) != hipSuccess) {
        fail(c, ORBFE_EDEVICE, "cannot create the blur stream");
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
  ALLOC(c->d_n_cand, M * NL);
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
  ALLOC(c->d_rowlist, NP * (size_t)c->row_list_cap);
  c->grid_key.reset(new std::atomic<uint64_t>[M]);  // (here, not on first use: slot calls on other threads clear entries without the API lock)
  for (size_t k = 0; k < M; ++k) c->grid_key[k] = 0;
  if (M <= 16 && ((size_t)c->cfg.height + 4) * 4 <= 9000) {  // (k_brief's table workgroup borrows 9000 bytes of the descriptor kernel's LDS)
    ALLOC(c->d_rowoff_slot, M * (size_t)(c->cfg.height + 1));
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

// results of slots slot0..slot0+n_img-1 to the host through the lane's pinned staging buffer: one batch of D2H copies (full arrays:
// the counts are not known on the host yet), ONE synchronisation
static orbfe_status enqueue_fetch(orbfe_ctx* c, orbfe_ctx::Lane& ln, int slot0, int n_img, size_t o_kps, size_t o_desc, size_t o_cnt,
                                  bool want_kps, bool want_desc) {
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1), s0 = (size_t)slot0;
  HIP_TRY(c, hipMemcpyAsync(ln.h_stage + o_cnt, c->d_n_kp + s0, sizeof(int32_t) * n_img, hipMemcpyDeviceToHost, ln.stream));
  if (want_kps)
    HIP_TRY(c, hipMemcpyAsync(ln.h_stage + o_kps, c->d_kps + s0 * NF, (size_t)n_img * NF * sizeof(orbfe_keypoint), hipMemcpyDeviceToHost,
                              ln.stream));
  if (want_desc)
    HIP_TRY(c, hipMemcpyAsync(ln.h_stage + o_desc, c->d_desc + s0 * NF * 32, (size_t)n_img * NF * 32, hipMemcpyDeviceToHost, ln.stream));
  return ORBFE_OK;
}
static orbfe_status finish_fetch(orbfe_ctx* c, orbfe_ctx::Lane& ln, int n_img, size_t o_kps, size_t o_desc, size_t o_cnt,
                                 orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out, bool timing) {
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  HIP_TRY(c, hipStreamSynchronize(ln.stream));
  if (timing) drain_timers(c);
  const int32_t* cnt = (const int32_t*)(ln.h_stage + o_cnt);
  for (int i = 0; i < n_img; ++i) {
    const int32_t n = cnt[i];
    if (n < 0 || (size_t)n > NF) return fail(c, ORBFE_EDEVICE, "extract: corrupt count %d for image %d", n, i);
    if (kps) std::memcpy(kps + (size_t)i * NF, ln.h_stage + o_kps + (size_t)i * NF * sizeof(orbfe_keypoint), sizeof(orbfe_keypoint) * n);
    if (desc) std::memcpy(desc + (size_t)i * NF * 32, ln.h_stage + o_desc + (size_t)i * NF * 32, (size_t)32 * n);
    if (n_out) n_out[i] = n;
  }
  return ORBFE_OK;
}
static orbfe_status fetch_extract_results(orbfe_ctx* c, int n_img, size_t o_kps, size_t o_desc, size_t o_cnt, orbfe_keypoint* kps,
                                          uint8_t* desc, int32_t* n_out) {
  TRY(enqueue_fetch(c, c->main, 0, n_img, o_kps, o_desc, o_cnt, kps != nullptr, desc != nullptr));
  return finish_fetch(c, c->main, n_img, o_kps, o_desc, o_cnt, kps, desc, n_out, true);
}

// Host images -> slots [slot0, slot0 + n_img) on lane `ln`: copy-in, the launch sequence, results back, one synchronisation.  One or
// two images (the drop-in call shape) are launch-bound: the whole sequence is captured once per lane into a hipGraph and replayed.
// fs (orbfe_frame_stereo; two images): the stereo match of (slot0, slot0 + 1) follows the extraction in the same launch sequence, its
// results come back through the staging buffer as the features do
struct FrameStereoReq {
  float fx, bf;
  double *right_u, *depth;  // [n_features], caller's
  int32_t* n_matches;
};
// fr (orbfe_frame_rgbd_image; one image): the image may be a 3-channel one (converted to gray on the way into level 0), and the RGB-D tail of
// the Frame constructor -- undistortion, depth / rightU lookup -- follows the extraction in the same launch sequence; the depth image is
// read by that kernel straight from the staging buffer (one 2- or 4-byte read per keypoint: it is never uploaded)
struct FrameRgbdReq {
  int32_t color_order;  // 0: the image is gray | 1: RGB | 2: BGR
  orbfe_camera cam;
  const void* depth;    // nullable: undistortion only
  int32_t depth_type;
  size_t depth_stride;
  float depth_scale;
  double *depth_out, *right_u_out;  // [n_features], caller's, nullable
  bool no_tail;                     // orbfe_extract_color: the conversion and the extraction only (the keypoints stay as extracted)
};
struct FrameRgbdKey {  // what of a request is baked into a captured launch sequence
  int32_t color_order, has_depth /* 2: no tail at all */, depth_type;
  size_t depth_stride;
  float depth_scale;
  orbfe_camera cam;
};
static orbfe_status extract_lane(orbfe_ctx* c, orbfe_ctx::Lane& ln, int slot0, int n_img, const uint8_t* const* imgs, size_t stride,
                                 orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out, bool timing, const FrameStereoReq* fs = nullptr,
                                 const FrameRgbdReq* fr = nullptr) {
  const LevelDev& L0 = c->lv[0];
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  const bool color = fr && fr->color_order != 0;
  const size_t crow = align_up((size_t)c->cfg.width * 3, 16) + 16;  // staged colour rows: 4-aligned, with room for the last 12-byte group
  const size_t plane = align_up(color ? crow * (size_t)L0.h : (size_t)L0.stride * L0.h, 256);
  const size_t o_kps = (size_t)n_img * plane, o_desc = o_kps + align_up((size_t)n_img * NF * sizeof(orbfe_keypoint), 256);
  const size_t o_cnt = o_desc + align_up((size_t)n_img * NF * 32, 256), o_ru = o_cnt + align_up((size_t)n_img * 4, 256);
  const bool extra = fs || fr;
  const size_t o_dp = o_ru + (extra ? align_up(NF * 8, 256) : 0), o_dimg = o_dp + (extra ? align_up(NF * 8, 256) : 0);
  const size_t d_bytes = (fr && fr->depth) ? fr->depth_stride * (size_t)c->cfg.height : 0;
  const size_t total = o_dimg + align_up(d_bytes, 256);
  TRY(ensure_stage(c, ln, total));
  for (int i = 0; i < n_img; ++i) {
    if (!imgs[i]) return fail(c, ORBFE_EBADARG, "extract: image %d is NULL", i);
    uint8_t* dst = ln.h_stage + (size_t)i * plane;
    if (color)
      for (int y = 0; y < L0.h; ++y) std::memcpy(dst + (size_t)y * crow, imgs[i] + (size_t)y * stride, (size_t)c->cfg.width * 3);
    else
      for (int y = 0; y < L0.h; ++y) std::memcpy(dst + (size_t)y * L0.stride, imgs[i] + (size_t)y * stride, (size_t)c->cfg.width);
  }
  if (d_bytes) std::memcpy(ln.h_stage + o_dimg, fr->depth, d_bytes);
  FrameRgbdKey rkey;
  static_assert(sizeof(FrameRgbdKey) <= sizeof(orbfe_ctx::GraphEntry::rkey), "GraphEntry::rkey");
  std::memset(&rkey, 0, sizeof rkey);
  if (fr) {
    rkey.color_order = fr->color_order, rkey.has_depth = fr->no_tail ? 2 : (fr->depth ? 1 : 0), rkey.depth_type = fr->depth_type;
    rkey.depth_stride = fr->depth_stride, rkey.depth_scale = fr->depth_scale, rkey.cam = fr->cam;
  }
  uint8_t* const pyr_now = c->d_pyr;
  note_slots_written(c, slot0, n_img, n_img <= 2);  // (also when a captured graph is replayed: run_extract does not run then)
  // The results come back through the staging buffer too: the orientation and the descriptor kernels write keypoints, counts and
  // descriptors there themselves (posted PCIe writes, ~120 KB per image) beside the device arrays the stereo match reads -- three
  // device-to-host copies queued behind the last kernel cost ~17 us of a ~0.3 ms call.  More than two images: the copies.
  const bool mirror_on = n_img <= 2;
  const bool rgbd_tail = fr && !fr->no_tail;
  HostMirror mir = {(kps && !rgbd_tail) ? (orbfe_keypoint*)(ln.h_stage + o_kps) : nullptr, desc ? ln.h_stage + o_desc : nullptr, (int32_t*)(ln.h_stage + o_cnt)};
  auto enqueue_all = [&]() -> orbfe_status {
    // (more images: both in ONE copy -- rows = images: the staging planes are `plane` bytes apart, the pyramid slots img_pitch)
    // One or two images: level 0 is read from the page-locked staging planes by a copy KERNEL (16 bytes per load over PCIe, every byte
    // once) -- 6 us less per pair than the copy engine's 27 us transfer and its hand-over to the compute queue (same box, alternating:
    // extraction 0.278 -> 0.271 ms).  (The resize reading the staged planes itself was measured in r3 and dropped: it reads a pixel more than once.)
    if (color)  // (one image) cv::cvtColor of Tracking::grabFrame on the way in: the kernel reads the staged rows itself
      launch_cvt_gray(ln.stream, ln.h_stage, crow, pyr_now + (size_t)slot0 * c->img_pitch + L0.plane_off, L0.stride, c->cfg.width, c->cfg.height,
                      fr->color_order, c->cfg.gray_variant ? 1 : 0);
    else if (n_img <= 2)
      launch_load_level0(ln.stream, ln.h_stage, nullptr, (size_t)L0.stride, plane, pyr_now, c->img_pitch, (uint32_t)L0.plane_off, L0.stride, c->cfg.width,
                         L0.h, slot0, 1, n_img);
    else
      HIP_TRY(c, hipMemcpy2DAsync(pyr_now + (size_t)slot0 * c->img_pitch + L0.plane_off, c->img_pitch, ln.h_stage, plane, (size_t)L0.stride * L0.h,
                                (size_t)n_img, hipMemcpyHostToDevice, ln.stream));
    TRY(run_extract(c, ln.stream, slot0, n_img, nullptr, timing, nullptr, mirror_on ? &mir : nullptr));
    if (fs) {
      // the right image's row table and the zeroed pair counter come out of the extraction above when this context builds them there
      const StereoHostOut ho = {(double*)(ln.h_stage + o_ru), (double*)(ln.h_stage + o_dp), nullptr, nullptr};
      const bool table_ready = c->slot_table_ok && c->slot_table_ok[(size_t)slot0 + 1] != 0;
      TRY(run_stereo(c, ln.stream, slot0, slot0 + 1, 0, slot0 / 2, 1, fs->fx, fs->bf, &ho, table_ready, timing));
    }
    if (rgbd_tail) {
      launch_frame_rgbd(ln.stream, c->d_kps + (size_t)slot0 * NF, c->d_n_kp + slot0, (int)NF, fr->cam, d_bytes ? ln.h_stage + o_dimg : nullptr,
                        fr->depth_type, fr->depth_stride, fr->depth_scale, (double*)(ln.h_stage + o_dp), (double*)(ln.h_stage + o_ru),
                        (orbfe_keypoint*)(ln.h_stage + o_kps));
      HIP_TRY(c, hipGetLastError());
    }
    if (mirror_on) return ORBFE_OK;
    return enqueue_fetch(c, ln, slot0, n_img, o_kps, o_desc, o_cnt, kps != nullptr, desc != nullptr);
  };
  // the match has counted into the pair's counter: a later orbfe_stereo_match on these slots clears it first
  auto finish = [&]() -> orbfe_status {
    TRY(finish_fetch(c, ln, n_img, o_kps, o_desc, o_cnt, kps, desc, n_out, timing));
    if (fs) {
      if (c->pair_count_zero) c->pair_count_zero[(size_t)(slot0 / 2)] = 0;
      const double* ru = (const double*)(ln.h_stage + o_ru);
      const double* dp = (const double*)(ln.h_stage + o_dp);
      const size_t n = (size_t)c->cfg.n_features;
      int32_t nm = 0;  // k_stereo counts exactly the features it gives a right coordinate (>= 0; -1 otherwise)
      for (size_t i = 0; i < n; ++i) nm += ru[i] >= 0.0 ? 1 : 0;
      if (fs->right_u && n) std::memcpy(fs->right_u, ru, sizeof(double) * n);
      if (fs->depth && n) std::memcpy(fs->depth, dp, sizeof(double) * n);
      if (fs->n_matches) *fs->n_matches = nm;
    }
    if (rgbd_tail) {
      const size_t n = (size_t)c->cfg.n_features;
      if (fr->depth_out && n) std::memcpy(fr->depth_out, ln.h_stage + o_dp, sizeof(double) * n);
      if (fr->right_u_out && n) std::memcpy(fr->right_u_out, ln.h_stage + o_ru, sizeof(double) * n);
    }
    return ORBFE_OK;
  };
  if (c->use_graphs && ln.use_graphs && c->prof == 0 && n_img <= 2) {
    hipGraphExec_t exec = nullptr;
    for (auto it = ln.graphs.begin(); it != ln.graphs.end();) {
      if (it->stage != ln.h_stage) {  // the staging buffer was re-allocated: the captured addresses are stale
        (void)hipGraphExecDestroy(it->exec);
        it = ln.graphs.erase(it);
        continue;
      }
      if (it->slot0 == slot0 && it->n_img == n_img && it->want_kps == (kps != nullptr) && it->want_desc == (desc != nullptr) && it->pyr == pyr_now &&
          it->stereo == (fs != nullptr) && (!fs || (it->fx == fs->fx && it->bf == fs->bf)) && it->rgbd == (fr != nullptr) &&
          (!fr || std::memcmp(&it->rkey, &rkey, sizeof rkey) == 0))
        exec = it->exec;
      ++it;
    }
    if (!exec) {
      hipGraph_t g = nullptr;
      bool ok = hipStreamBeginCapture(ln.stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
      const orbfe_status st = ok ? enqueue_all() : ORBFE_EDEVICE;
      if (ok) ok = hipStreamEndCapture(ln.stream, &g) == hipSuccess && st == ORBFE_OK && g;
      if (ok) ok = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) == hipSuccess;
      if (g) (void)hipGraphDestroy(g);
      if (ok) {
        orbfe_ctx::GraphEntry ge{slot0, n_img, kps != nullptr, desc != nullptr, fs != nullptr, fs ? fs->fx : 0.f, fs ? fs->bf : 0.f, ln.h_stage, pyr_now, exec};
        ge.rgbd = fr != nullptr;
        std::memcpy(ge.rkey, &rkey, sizeof rkey);
        ln.graphs.push_back(ge);
      } else {
        (void)hipGetLastError();
        exec = nullptr;
        ln.use_graphs = false;  // this runtime cannot capture the sequence: plain launches on this lane from now on
      }
    }
    if (exec) {
      HIP_TRY(c, hipGraphLaunch(exec, ln.stream));
      return finish();
    }
  }
  TRY(enqueue_all());
  return finish();
}

orbfe_status orbfe_extract_batch(orbfe_ctx* c, int32_t n_img, const uint8_t* const* imgs, size_t stride, orbfe_keypoint* kps,
                                 uint8_t* desc, int32_t* n_out) {
  ApiLock api_lk(c);
  if (!c || !imgs || n_img < 0) return fail(c, ORBFE_EBADARG, "extract_batch: NULL argument");
  if (n_img > c->cfg.max_images) return fail(c, ORBFE_ECAPACITY, "extract_batch: %d images > max_images %d", n_img, c->cfg.max_images);
  if (stride < (size_t)c->cfg.width) return fail(c, ORBFE_EBADARG, "extract_batch: stride %zu < width %d", stride, c->cfg.width);
  if (n_img == 0) return ORBFE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  return extract_lane(c, c->main, 0, n_img, imgs, stride, kps, desc, n_out, true);
}

// The device work of Frame::createStereo (include/ORB_SLAM2/Frame.h:313-323: the constructor's two extractions, src/Frame.cc:100-105,
// then ORBMatcher::searchByStereo) as ONE call: both images up, the extraction of slots 0 and 1 and their stereo match as one launch sequence (one graph replay), every
// result back through the staging buffer, one synchronisation.  Same results as orbfe_extract_batch([left, right]) followed by
// orbfe_stereo_match(0, 1) -- the same kernels in the same order -- without the second call's launch, copy and wake-up.
orbfe_status orbfe_frame_stereo(orbfe_ctx* c, const uint8_t* left, const uint8_t* right, size_t stride, float fx, float bf,
                                orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out, double* right_u, double* depth, int32_t* n_matches) {
  ApiLock api_lk(c);
  if (!c || !left || !right) return fail(c, ORBFE_EBADARG, "frame_stereo: NULL argument");
  if (c->cfg.max_images < 2) return fail(c, ORBFE_ECAPACITY, "frame_stereo: the context holds %d image(s), a stereo frame needs 2", c->cfg.max_images);
  if (stride < (size_t)c->cfg.width) return fail(c, ORBFE_EBADARG, "frame_stereo: stride %zu < width %d", stride, c->cfg.width);
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const uint8_t* imgs[2] = {left, right};
  const FrameStereoReq fs = {fx, bf, right_u, depth, n_matches};
  return extract_lane(c, c->main, 0, 2, imgs, stride, kps, desc, n_out, true, &fs);
}

// One image -> slot `slot` on that slot's own lane.  Calls on DIFFERENT slots may run at the same time on different threads.
orbfe_status orbfe_extract_slot(orbfe_ctx* c, int32_t slot, const uint8_t* img, size_t stride, orbfe_keypoint* kps, uint8_t* desc,
                                int32_t* n_out) {
  const uint8_t* one[1] = {img};
  return orbfe_extract_slots(c, slot, 1, one, stride, kps, desc, n_out);
}

// n_img images -> slots [slot0, slot0 + n_img) on slot0's lane: what a caller does who holds BOTH images of a stereo frame when the
// first extract() is reached (host/orbfe_shim.hpp keeps one orbfe_extract_slot per extract() thread: pairing the threads was measured and dropped)
static orbfe_status extract_slots_impl(orbfe_ctx* c, int32_t slot, int32_t n_img, const uint8_t* const* imgs, size_t stride, orbfe_keypoint* kps,
                                       uint8_t* desc, int32_t* n_out, const FrameStereoReq* fs, const FrameRgbdReq* fr = nullptr) {
  if (!c || !imgs || n_img < 1) return fail(c, ORBFE_EBADARG, "extract_slots: NULL argument / no image");
  const uint8_t* img = imgs[0];
  if (!img) return fail(c, ORBFE_EBADARG, "extract_slot: NULL argument");
  if (slot < 0 || slot + n_img > c->cfg.max_images) return fail(c, ORBFE_EBADARG, "extract_slot: slots %d..%d of %d", slot, slot + n_img - 1, c->cfg.max_images);
  if (stride < (size_t)c->cfg.width * ((fr && fr->color_order) ? 3 : 1)) return fail(c, ORBFE_EBADARG, "extract_slot: stride %zu < the row's %d bytes", stride, c->cfg.width * ((fr && fr->color_order) ? 3 : 1));
  HIP_TRY(c, hipSetDevice(c->device));
  // the lanes of EVERY slot the call writes, created on first use and locked in index order (a concurrent slot call on any of them waits;
  // two multi-slot calls cannot deadlock); the work runs on the first slot's lane
  std::vector<orbfe_ctx::Lane*> lanes((size_t)n_img, nullptr);
  {
    std::lock_guard<std::mutex> lk(c->slot_lane_mu);
    for (int k = 0; k < n_img; ++k) {
      if (!c->slot_lane[(size_t)(slot + k)]) {
        std::unique_ptr<orbfe_ctx::Lane> fresh(new orbfe_ctx::Lane());
        HIP_TRY(c, hipStreamCreateWithFlags(&fresh->stream, hipStreamNonBlocking));
        fresh->own_stream = true;
        if (hipEventCreateWithFlags(&fresh->ev_main, hipEventDisableTiming) != hipSuccess) {
          (void)hipStreamDestroy(fresh->stream);
          return fail(c, ORBFE_EDEVICE, "extract_slot: cannot create the lane event");
        }
        c->slot_lane[(size_t)(slot + k)] = std::move(fresh);
      }
      lanes[(size_t)k] = c->slot_lane[(size_t)(slot + k)].get();
    }
  }
  std::vector<std::unique_lock<std::mutex>> held;
  held.reserve(lanes.size());
  for (orbfe_ctx::Lane* l : lanes) held.emplace_back(l->mu);
  orbfe_ctx::Lane* ln = lanes[0];
  // a stereo match of an earlier device batch may still be reading the slot arrays (the flag is only read here: the calls that
  // change it must not overlap with slot calls)
  if (c->stereo_pending) HIP_TRY(c, hipStreamWaitEvent(ln->stream, c->ev_stereo_done, 0));
  // ... and an asynchronous batch call may have left work on the context stream that still writes this slot.  The marker is recorded
  // under the API lock: another thread may be inside hipStreamBeginCapture on the context stream (the first orbfe_extract /
  // orbfe_extract_batch of a shape), and an event recorded into that capture would pull this lane's stream into it -- both graphs fail
  {
    ApiLock api_lk(c);
    HIP_TRY(c, hipEventRecord(ln->ev_main, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(ln->stream, ln->ev_main, 0));
  }
  return extract_lane(c, *ln, slot, n_img, imgs, stride, kps, desc, n_out, false, fs, fr);
}
orbfe_status orbfe_extract_slots(orbfe_ctx* c, int32_t slot, int32_t n_img, const uint8_t* const* imgs, size_t stride, orbfe_keypoint* kps,
                                 uint8_t* desc, int32_t* n_out) {
  return extract_slots_impl(c, slot, n_img, imgs, stride, kps, desc, n_out, nullptr);
}
// The device work of Frame::createRGBD (include/ORB_SLAM2/Frame.h:326-331) for one image as ONE call: Tracking::grabFrame's cvtColor when the
// image has three channels (src/Tracking.cc:55-68), the extraction (the RGB-D Frame constructor, src/Frame.cc:125-135), then
// Camera::undistortPoints and the depth / rightU lookup (:136-158) -- what orbfe_extract_color / orbfe_extract_slot followed by
// orbfe_frame_rgbd do in two calls.  On the slot's own lane (as orbfe_extract_slot).  The depth image is never uploaded: the last kernel
// reads one value per keypoint from the page-locked staging copy.
orbfe_status orbfe_frame_rgbd_image(orbfe_ctx* c, int32_t slot, const uint8_t* img, size_t stride, int32_t color_order, const orbfe_camera* cam,
                                    const void* depth, int32_t depth_type, size_t depth_stride, float depth_scale, orbfe_keypoint* kps_undistorted,
                                    uint8_t* desc, int32_t* n_out, double* depth_out, double* right_u_out) {
  if (!c || !img || !cam) return fail(c, ORBFE_EBADARG, "frame_rgbd_image: NULL argument");
  if (color_order < 0 || color_order > 2) return fail(c, ORBFE_EBADARG, "frame_rgbd_image: color_order %d (0 = gray, 1 = RGB, 2 = BGR)", color_order);
  const size_t px = depth_type == 0 ? 2 : 4;
  if (depth && (depth_type < 0 || depth_type > 1 || depth_stride < (size_t)c->cfg.width * px || !(depth_scale > 0)))
    return fail(c, ORBFE_EBADARG, "frame_rgbd_image: depth type %d stride %zu scale %g", depth_type, depth_stride, (double)depth_scale);
  const uint8_t* one[1] = {img};
  const FrameRgbdReq fr = {color_order, *cam, depth, depth_type, depth_stride, depth_scale, depth_out, right_u_out, false};
  return extract_slots_impl(c, slot, 1, one, stride, kps_undistorted, desc, n_out, nullptr, &fr);
}
// orbfe_frame_stereo into the slot pair (slot_left, slot_left + 1), slot_left even, on slot_left's lane: what the drop-in's frame-level
// adapter calls (the extractor objects rotate over the context's slots; a Frame's device-side features live as long as its slots do)
orbfe_status orbfe_frame_stereo_slots(orbfe_ctx* c, int32_t slot_left, const uint8_t* left, const uint8_t* right, size_t stride, float fx,
                                      float bf, orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out, double* right_u, double* depth,
                                      int32_t* n_matches) {
  if (!c || !left || !right) return fail(c, ORBFE_EBADARG, "frame_stereo_slots: NULL argument");
  if (slot_left < 0 || (slot_left & 1) || slot_left + 2 > c->cfg.max_images)
    return fail(c, ORBFE_EBADARG, "frame_stereo_slots: slot %d (even, and slot + 1 < max_images %d)", slot_left, c->cfg.max_images);
  const uint8_t* imgs[2] = {left, right};
  const FrameStereoReq fs = {fx, bf, right_u, depth, n_matches};
  return extract_slots_impl(c, slot_left, 2, imgs, stride, kps, desc, n_out, &fs);
}

orbfe_status orbfe_extract(orbfe_ctx* c, const uint8_t* img, size_t stride, orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out) {
  const uint8_t* one[1] = {img};
  return orbfe_extract_batch(c, 1, one, stride, kps, desc, n_out);
}

orbfe_status orbfe_extract_color(orbfe_ctx* c, const uint8_t* img, size_t stride, int32_t color_order, orbfe_keypoint* kps, uint8_t* desc,
                                 int32_t* n_out) {
  // (through the one-frame launch sequence since late r4: the conversion kernel reads the staged colour rows itself, the sequence is
  //  replayed from a captured graph and the results come back through the staging buffer -- the same path as orbfe_extract_batch)
  ApiLock api_lk(c);
  if (!c || !img) return fail(c, ORBFE_EBADARG, "extract_color: NULL argument");
  if (color_order != 1 && color_order != 2) return fail(c, ORBFE_EBADARG, "extract_color: color_order %d (1 = RGB, 2 = BGR)", color_order);
  if (stride < (size_t)c->cfg.width * 3) return fail(c, ORBFE_EBADARG, "extract_color: stride %zu < 3 * width", stride);
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const uint8_t* one[1] = {img};
  FrameRgbdReq fr;
  std::memset(&fr, 0, sizeof fr);
  fr.color_order = color_order, fr.no_tail = true;
  return extract_lane(c, c->main, 0, 1, one, stride, kps, desc, n_out, true, nullptr, &fr);
}

orbfe_status orbfe_frame_rgbd(orbfe_ctx* c, int32_t slot, const orbfe_camera* cam, const void* depth, int32_t depth_type,
                              size_t depth_stride, float depth_scale, orbfe_keypoint* kps_out, double* depth_out, double* right_u_out) {
  ApiLock api_lk(c);
  if (!c || !cam || slot < 0 || slot >= c->cfg.max_images) return fail(c, ORBFE_EBADARG, "frame_rgbd: bad slot / NULL camera");
  const size_t px = depth_type == 0 ? 2 : 4;
  if (depth && (depth_type < 0 || depth_type > 1 || depth_stride < (size_t)c->cfg.width * px || !(depth_scale > 0)))
    return fail(c, ORBFE_EBADARG, "frame_rgbd: depth type %d stride %zu scale %g", depth_type, depth_stride, (double)depth_scale);
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  const size_t d_bytes = depth ? depth_stride * (size_t)c->cfg.height : 0;
  // The depth image is staged in page-locked memory and READ FROM THERE by the kernel (one 2- or 4-byte value per keypoint: uploading
  // 614 KB of a 640 x 480 16-bit image for 1000 reads was a third of the call); depth, rightU and the undistorted keypoints are written
  // to the staging buffer by the kernel as well: one 4-byte copy (the count), one synchronisation.
  const size_t h_res = align_up(d_bytes, 256), h_ru = h_res + align_up(NF * 8, 256), h_n = h_ru + align_up(NF * 8, 256), h_k = h_n + 256,
               h_total = h_k + align_up(NF * sizeof(orbfe_keypoint), 256);
  TRY(ensure_stage(c, h_total));
  uint8_t* hs = c->main.h_stage;
  if (depth) std::memcpy(hs, depth, d_bytes);
  grid_invalidate(c, slot);  // (the keypoints move: a grid kept for the slot is stale)
  launch_frame_rgbd(c->stream, c->d_kps + (size_t)slot * NF, c->d_n_kp + slot, (int)NF, *cam, depth ? hs : nullptr, depth_type, depth_stride,
                    depth_scale, (double*)(hs + h_res), (double*)(hs + h_ru), kps_out ? (orbfe_keypoint*)(hs + h_k) : nullptr);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(hs + h_n, c->d_n_kp + slot, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int32_t n = 0;
  std::memcpy(&n, hs + h_n, 4);
  if (depth_out) std::memcpy(depth_out, hs + h_res, NF * 8);
  if (right_u_out) std::memcpy(right_u_out, hs + h_ru, NF * 8);
  if (kps_out && n > 0) std::memcpy(kps_out, hs + h_k, sizeof(orbfe_keypoint) * (size_t)std::min<int64_t>(n, (int64_t)NF));
  return ORBFE_OK;
}

orbfe_status orbfe_get_pyramid(orbfe_ctx* c, int32_t slot, int32_t level, int32_t blurred, uint8_t* dst) {
  ApiLock api_lk(c);
  if (!c || !dst || slot < 0 || slot >= c->cfg.max_images || level < 0 || level >= c->cfg.n_levels)
    return fail(c, ORBFE_EBADARG, "get_pyramid: slot %d level %d", slot, level);
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const LevelDev& L = c->lv[level];
  const uint8_t* src = (blurred ? c->d_blur : c->d_pyr) + (size_t)slot * c->img_pitch + L.plane_off;
  HIP_TRY(c, hipMemcpy2DAsync(dst, L.w, src, L.stride, L.w, L.h, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ORBFE_OK;
}

orbfe_status orbfe_stereo_match(orbfe_ctx* c, int32_t slot_left, int32_t slot_right, float fx, float bf, double* right_u,
                                double* depth, int32_t* n_matches, int32_t* best_right, int32_t* best_dist) {
  ApiLock api_lk(c);
  if (!c || slot_left < 0 || slot_right < 0 || slot_left >= c->cfg.max_images || slot_right >= c->cfg.max_images)
    return fail(c, ORBFE_EBADARG, "stereo_match: slots %d/%d", slot_left, slot_right);
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const int pair = slot_left / 2;
  // the kernel writes the requested arrays into the page-locked staging buffer itself; only the match count is copied (4 bytes)
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1);
  const size_t o_ru = 0, o_dp = align_up(NF * 8, 256), o_br = o_dp + align_up(NF * 8, 256), o_bd = o_br + align_up(NF * 4, 256),
               o_nm = o_bd + align_up(NF * 4, 256), total = o_nm + 256;
  TRY(ensure_stage(c, total));
  uint8_t* h = c->main.h_stage;
  const StereoHostOut ho = {right_u ? (double*)(h + o_ru) : nullptr, depth ? (double*)(h + o_dp) : nullptr,
                            best_right ? (int32_t*)(h + o_br) : nullptr, best_dist ? (int32_t*)(h + o_bd) : nullptr};
  // The right image's row table: built by its extraction when that was a one- or two-image call of this (small) context -- the match
  // is then k_stereo alone; the pair's counter was zeroed there too unless an earlier match has counted into it since.
  const bool table_ready = c->slot_table_ok && c->slot_table_ok[(size_t)slot_right] != 0;
  if (table_ready && !c->pair_count_zero[(size_t)pair].exchange(0)) HIP_TRY(c, hipMemsetAsync(c->d_n_match + pair, 0, sizeof(int32_t), c->stream));
  TRY(run_stereo(c, c->stream, slot_left, slot_right, 0, pair, 1, fx, bf, &ho, table_ready));
  // (the count: with right_u in the staging buffer it is counted there -- k_stereo counts exactly the features it gives a right coordinate --
  //  and the 4-byte copy, a transfer of its own behind the kernel, is left out)
  const bool count_on_host = right_u != nullptr;
  if (n_matches && !count_on_host) HIP_TRY(c, hipMemcpyAsync(h + o_nm, c->d_n_match + pair, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  const size_t n = (size_t)c->cfg.n_features;
  if (right_u && n) std::memcpy(right_u, h + o_ru, sizeof(double) * n);
  if (depth && n) std::memcpy(depth, h + o_dp, sizeof(double) * n);
  if (best_right && n) std::memcpy(best_right, h + o_br, sizeof(int32_t) * n);
  if (best_dist && n) std::memcpy(best_dist, h + o_bd, sizeof(int32_t) * n);
  if (n_matches) {
    if (count_on_host) {
      int32_t nm = 0;
      const double* ru = (const double*)(h + o_ru);
      for (size_t i = 0; i < n; ++i) nm += ru[i] >= 0.0 ? 1 : 0;
      *n_matches = nm;
    } else
      std::memcpy(n_matches, h + o_nm, sizeof(int32_t));
  }
  return ORBFE_OK;
}

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
static PackLayout pack_layout(const orbfe_ctx* c, int n_pairs) {
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

static orbfe_status batch_device_core(orbfe_ctx* c, const uint8_t* d_left, const uint8_t* d_right, size_t stride, size_t image_pitch,
                                      int32_t n_pairs, float fx, float bf, const PackDst* pack) {
  const LevelDev& L0 = c->lv[0];
  // The stereo match goes to its own stream and this call returns with it still queued; the next call starts its copy-in / resize / FAST
  // on the context stream right away, into the OTHER pyramid buffer, and only its keypoint-list kernels wait for the match (they rewrite
  // what it reads).  Every other entry point joins the stereo stream first.  With stage timing on (prof == 1) everything runs in line.
  const bool pipe = c->pipeline_stereo && c->stereo_stream && c->prof != 1 && n_pairs >= 16;
  if (pipe) {
    if (!c->d_pyr_alt) {
      // (cleared ON THE CONTEXT STREAM: a null-stream hipMemset returns before the device has run it and is not ordered with this
      //  non-blocking stream -- it was seen zeroing rows of the first batch's level 0 after k_load_level0 had written them)
      if (hipMalloc((void**)&c->d_pyr_alt, (size_t)c->cfg.max_images * c->img_pitch) != hipSuccess ||
          hipMemsetAsync(c->d_pyr_alt, 0, (size_t)c->cfg.max_images * c->img_pitch, c->stream) != hipSuccess) {
        (void)hipGetLastError();
        c->d_pyr_alt = nullptr;
        c->pipeline_stereo = false;  // no room for the second pyramid: plain in-order execution
      }
    }
  }
  const bool piped = pipe && c->d_pyr_alt;
  if (piped)
    std::swap(c->d_pyr, c->d_pyr_alt);
  else
    TRY(join_stereo(c));
  {
    hipStream_t st = c->stream;
    // level 0 of slot 2p / 2p+1 <- left / right image p.  >= 32 images: the resize reads the caller's images itself and every workgroup
    // writes its block of level 0 into the pyramid from the tile it has staged anyway -- no copy-in kernel, half its traffic
    const bool ext0 = c->blur_stream && c->prof != 1 && !c->rs_regions.empty() && 2 * n_pairs >= 32 && (size_t)stride * c->cfg.height <= 0xFFFFFFF0u;
    ExtLevel0 ext;
    if (ext0) {
      ext.left = d_left, ext.right = d_right;
      ext.pitch = image_pitch, ext.stride = (int)stride, ext.bytes = (uint32_t)((size_t)stride * (c->cfg.height - 1) + c->cfg.width);
      ext.inputs_free = (pack && pack->in_free) ? pack->in_free : nullptr;
    } else {
      launch_load_level0(st, d_left, d_right, stride, image_pitch, c->d_pyr, c->img_pitch, L0.plane_off, L0.stride, c->cfg.width, c->cfg.height, 0, 2,
                         n_pairs);
      if (pack && pack->in_free) HIP_TRY(c, hipEventRecord(pack->in_free, st));  // the images may be overwritten
    }
    TRY(run_extract(c, st, 0, 2 * n_pairs, (piped && c->stereo_pending) ? c->ev_stereo_done : nullptr, true, ext0 ? &ext : nullptr));
    if (piped) {
      HIP_TRY(c, hipEventRecord(c->ev_brief_done, st));
      HIP_TRY(c, hipStreamWaitEvent(c->stereo_stream, c->ev_brief_done, 0));
      TRY(run_stereo(c, c->stereo_stream, 0, 1, 2, 0, n_pairs, fx, bf));
    } else {
      TRY(run_stereo(c, st, 0, 1, 2, 0, n_pairs, fx, bf));
    }
  }
  if (pack) {
    // the packed results of this batch -> the stream's result buffer (device to device: ~0.1 ms for 512 pairs), on the stream the
    // match ran on, BEFORE the next batch may rewrite the per-slot arrays; the download then runs beside the next batch
    hipStream_t ps = piped ? c->stereo_stream : c->stream;
    const PackLayout l = pack_layout(c, n_pairs);
    const size_t NF = (size_t)std::max(c->cfg.n_features, 1), n = (size_t)n_pairs;
    HIP_TRY(c, hipStreamWaitEvent(ps, pack->wait_free, 0));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_kps, c->d_kps, 2 * n * NF * sizeof(orbfe_keypoint), hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_desc, c->d_desc, 2 * n * NF * 32, hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_cnt, c->d_n_kp, 2 * n * 4, hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_ru, c->d_right_u, n * NF * 8, hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_dp, c->d_depth, n * NF * 8, hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipMemcpyAsync(pack->base + l.o_nm, c->d_n_match, n * 4, hipMemcpyDeviceToDevice, ps));
    HIP_TRY(c, hipEventRecord(pack->ready, ps));
  }
  if (piped) {
    HIP_TRY(c, hipEventRecord(c->ev_stereo_done, c->stereo_stream));
    c->stereo_pending = true;
  }
  return ORBFE_OK;
}

orbfe_status orbfe_stereo_batch_device(orbfe_ctx* c, const uint8_t* d_left, const uint8_t* d_right, size_t stride, size_t image_pitch,
                                       int32_t n_pairs, float fx, float bf) {
  ApiLock api_lk(c);
  if (!c || !d_left || !d_right || n_pairs < 0) return fail(c, ORBFE_EBADARG, "stereo_batch_device: NULL argument");
  if (2 * n_pairs > c->cfg.max_images) return fail(c, ORBFE_ECAPACITY, "stereo_batch_device: %d pairs need %d slots > %d", n_pairs, 2 * n_pairs, c->cfg.max_images);
  if (stride < (size_t)c->cfg.width || image_pitch < stride * (size_t)c->cfg.height)
    return fail(c, ORBFE_EBADARG, "stereo_batch_device: stride/pitch too small");
  if (n_pairs == 0) return ORBFE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  return batch_device_core(c, d_left, d_right, stride, image_pitch, n_pairs, fx, bf, nullptr);
}

// ---- host-image stream ---------------------------------------------------------------------------------------------------------
// CPUs of the NUMA node the current HIP device hangs off (sysfs local_cpulist of its PCI function); empty set if unknown.
static bool device_local_cpus(int dev, cpu_set_t* set) {
  char bus[64] = {0};
  if ((dev < 0 && hipGetDevice(&dev) != hipSuccess) || hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  for (char* q = bus; *q; ++q) *q = (char)tolower(*q);
  char path[160];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
  FILE* f = fopen(path, "r");
  if (!f) return false;
  char line[1024] = {0};
  const bool got = fgets(line, sizeof line, f) != nullptr;
  fclose(f);
  if (!got) return false;
  CPU_ZERO(set);
  int n = 0;
  char* save = nullptr;  // (strtok_r: allocations may come from several threads at once)
  for (char* tok = strtok_r(line, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {
    int a = 0, b = 0;
    const int k = sscanf(tok, "%d-%d", &a, &b);
    if (k == 1) b = a;
    if (k < 1) continue;
    for (int c = a; c <= b && c < CPU_SETSIZE; ++c) {
      CPU_SET(c, set);
      ++n;
    }
  }
  return n > 0;
}

// Page-locked host memory ON THE NUMA NODE OF THE DEVICE: the pages are placed where the allocating thread runs, and a buffer on the
// other socket is read by the DMA engines across the inter-socket link (measured on a two-socket MI355X host: 41 GB/s instead of
// 57 GB/s host to device).  The calling thread is moved to the device's local CPUs for the allocation and the first touch, then back.
void* orbfe_host_alloc(size_t bytes) { return orbfe_host_alloc_on(-1, bytes); }

// device_id < 0: the calling thread's current HIP device; a multi-rank job passes its own device so that ranks that never called
// hipSetDevice do not all pin to GPU 0's node.
void* orbfe_host_alloc_on(int32_t device_id, size_t bytes) {
  cpu_set_t old_set, local;
  const bool have_old = sched_getaffinity(0, sizeof old_set, &old_set) == 0;
  bool moved = false;
  if (have_old && device_local_cpus(device_id, &local)) {
    cpu_set_t both;
    CPU_AND(&both, &local, &old_set);  // stay inside what this process is allowed to use
    if (CPU_COUNT(&both) > 0) moved = sched_setaffinity(0, sizeof both, &both) == 0;
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    p = nullptr;
  } else if (moved) {
    for (size_t o = 0; o < bytes; o += 4096) ((volatile uint8_t*)p)[o] = 0;  // first touch, should the driver place lazily
  }
  if (moved && sched_setaffinity(0, sizeof old_set, &old_set) != 0 && sched_setaffinity(0, sizeof old_set, &old_set) != 0)
    g_last_error = "orbfe_host_alloc: the calling thread's CPU affinity could not be restored (it stays on the device's NUMA node)";
  return p;
}
void orbfe_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

// The streaming entry points keep up to eight HIP streams busy at once (compute, stereo match, blur, upload, download, slot lanes, the
// caller's and RCCL's own), and the HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues -- 4 by
// default.  Two streams that share a queue run one after the other: with 4 queues the upload of batch k + 1 queues behind the kernels
// of batch k and the 4541-pair sequence takes 0.127 s, with 16 it takes 0.087 s (profiles/r3_hw_queues.txt).  The runtime reads the
// variable at its first HIP call, so a library cannot set it: the process that streams exports it (bench.py does; a process that builds
// one frame at a time should NOT -- 4 queues are ~50 us per frame faster there), and orbfe_stream_submit says so once if it is missing.
int32_t orbfe_recommended_hw_queues(void) { return 16; }

orbfe_status orbfe_stream_submit(orbfe_ctx* c, const uint8_t* left, const uint8_t* right, size_t stride, size_t image_pitch, int32_t n_pairs,
                                 float fx, float bf, const orbfe_batch_results* out, int64_t* ticket) {
  {
    static std::once_flag warned;
    std::call_once(warned, [] {
      const char* q = getenv("GPU_MAX_HW_QUEUES");
      if (!q || atoi(q) < 8)
        fprintf(stderr,
                "[orbfe] orbfe_stream_submit: GPU_MAX_HW_QUEUES is %s; the streaming path overlaps upload, compute and download on streams of their "
                "own and runs ~30 %% slower when they share hardware queues -- export GPU_MAX_HW_QUEUES=%d before the process's first HIP call "
                "(orbfe_recommended_hw_queues())\n",
                q ? q : "unset (4)", orbfe_recommended_hw_queues());
    });
  }
  ApiLock api_lk(c);
  if (!c || !left || !right || !out || !ticket || n_pairs <= 0) return fail(c, ORBFE_EBADARG, "stream_submit: NULL argument / no pairs");
  if (2 * n_pairs > c->cfg.max_images) return fail(c, ORBFE_ECAPACITY, "stream_submit: %d pairs need %d slots > %d", n_pairs, 2 * n_pairs, c->cfg.max_images);
  if (stride < (size_t)c->cfg.width || image_pitch < stride * (size_t)c->cfg.height)
    return fail(c, ORBFE_EBADARG, "stream_submit: stride/pitch too small");
  HIP_TRY(c, hipSetDevice(c->device));
  orbfe_ctx::HostStream& hs = c->hs;
  if (!hs.init) {
    HIP_TRY(c, hipStreamCreateWithFlags(&hs.h2d, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&hs.d2h, hipStreamNonBlocking));
    for (int b = 0; b < orbfe_ctx::HostStream::kDepth; ++b) {
      HIP_TRY(c, hipEventCreateWithFlags(&hs.ev_h2d[b], hipEventDisableTiming));
      HIP_TRY(c, hipEventCreateWithFlags(&hs.ev_in_free[b], hipEventDisableTiming));
      HIP_TRY(c, hipEventCreateWithFlags(&hs.ev_out_ready[b], hipEventDisableTiming));
      HIP_TRY(c, hipEventCreateWithFlags(&hs.ev_done[b], hipEventDisableTiming));
    }
    hs.init = true;
  }
  const size_t eye = image_pitch * (size_t)n_pairs;
  const PackLayout l = pack_layout(c, n_pairs);
  if (2 * eye > hs.in_bytes || l.total > hs.out_bytes) {  // (re)size the device buffers: quiesce everything first
    TRY(join_stereo(c));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(hs.h2d));
    HIP_TRY(c, hipStreamSynchronize(hs.d2h));
    const size_t in_bytes = std::max(hs.in_bytes, align_up(2 * eye, 1 << 20));
    const PackLayout lmax = pack_layout(c, c->cfg.max_images / 2);
    const size_t out_bytes = std::max(hs.out_bytes, lmax.total);
    for (int b = 0; b < orbfe_ctx::HostStream::kDepth; ++b) {
      if (in_bytes != hs.in_bytes) {
        if (hs.d_in[b]) HIP_TRY(c, hipFree(hs.d_in[b]));
        hs.d_in[b] = nullptr;
        HIP_TRY(c, hipMalloc((void**)&hs.d_in[b], in_bytes));
      }
      if (out_bytes != hs.out_bytes) {
        if (hs.d_out[b]) HIP_TRY(c, hipFree(hs.d_out[b]));
        hs.d_out[b] = nullptr;
        HIP_TRY(c, hipMalloc((void**)&hs.d_out[b], out_bytes));
      }
    }
    hs.in_bytes = in_bytes;
    hs.out_bytes = out_bytes;
  }
  const int D = orbfe_ctx::HostStream::kDepth;
  const int b = (int)(hs.next_ticket % D);
  // the ticket that last used this set of buffers must be complete ("at most three outstanding" is what makes three sets enough)
  if (hs.next_ticket >= D && hipEventQuery(hs.ev_done[b]) != hipSuccess) {
    (void)hipGetLastError();
    HIP_TRY(c, hipEventSynchronize(hs.ev_done[b]));
  }
  // upload: after the batch that last read this input buffer has consumed it
  HIP_TRY(c, hipStreamWaitEvent(hs.h2d, hs.ev_in_free[b], 0));
  HIP_TRY(c, hipMemcpyAsync(hs.d_in[b], left, eye, hipMemcpyHostToDevice, hs.h2d));
  HIP_TRY(c, hipMemcpyAsync(hs.d_in[b] + eye, right, eye, hipMemcpyHostToDevice, hs.h2d));
  HIP_TRY(c, hipEventRecord(hs.ev_h2d[b], hs.h2d));
  // compute: the device-batch schedule, results packed into this ticket's result buffer
  HIP_TRY(c, hipStreamWaitEvent(c->stream, hs.ev_h2d[b], 0));
  const PackDst pack = {hs.d_out[b], hs.ev_done[b], hs.ev_out_ready[b], hs.ev_in_free[b]};
  TRY(batch_device_core(c, hs.d_in[b], hs.d_in[b] + eye, stride, image_pitch, n_pairs, fx, bf, &pack));
  // download
  const size_t NF = (size_t)std::max(c->cfg.n_features, 1), n = (size_t)n_pairs;
  // (measured and dropped: writing the results into the page-locked arrays with a copy KERNEL through their device mapping instead of
  //  the DMA engine -- 16 to 1024 workgroups, four 16-byte loads in flight per lane: the step takes 11.7 ms against 9.1 ms.)
  // The download goes onto the stream the pack ran on.  In the pipelined schedule that is the stereo stream, which has nothing else
  // to do until the next batch's match ~8 ms later -- a stream of its own would be one more hardware queue, and HIP multiplexes all
  // streams of a process onto 4 of them (GPU_MAX_HW_QUEUES): a download that shares its queue with the uploads or with the compute
  // stream serialises with them (measured: 9.1 -> 11.8 ms per 512-pair step, depending on what the process had created before).
  hipStream_t ds = c->stereo_pending ? c->stereo_stream : hs.d2h;
  if (ds == hs.d2h) HIP_TRY(c, hipStreamWaitEvent(hs.d2h, hs.ev_out_ready[b], 0));
  const uint8_t* src = hs.d_out[b];
  if (out->kps) HIP_TRY(c, hipMemcpyAsync(out->kps, src + l.o_kps, 2 * n * NF * sizeof(orbfe_keypoint), hipMemcpyDeviceToHost, ds));
  if (out->desc) HIP_TRY(c, hipMemcpyAsync(out->desc, src + l.o_desc, 2 * n * NF * 32, hipMemcpyDeviceToHost, ds));
  if (out->counts) HIP_TRY(c, hipMemcpyAsync(out->counts, src + l.o_cnt, 2 * n * 4, hipMemcpyDeviceToHost, ds));
  if (out->right_u) HIP_TRY(c, hipMemcpyAsync(out->right_u, src + l.o_ru, n * NF * 8, hipMemcpyDeviceToHost, ds));
  if (out->depth) HIP_TRY(c, hipMemcpyAsync(out->depth, src + l.o_dp, n * NF * 8, hipMemcpyDeviceToHost, ds));
  if (out->n_matches) HIP_TRY(c, hipMemcpyAsync(out->n_matches, src + l.o_nm, n * 4, hipMemcpyDeviceToHost, ds));
  HIP_TRY(c, hipEventRecord(hs.ev_done[b], ds));
  hs.n_pairs_of[b] = n_pairs;
  *ticket = hs.next_ticket++;
  return ORBFE_OK;
}

orbfe_status orbfe_stream_wait(orbfe_ctx* c, int64_t ticket) {
  ApiLock api_lk(c);
  if (!c) return ORBFE_EBADARG;
  orbfe_ctx::HostStream& hs = c->hs;
  if (!hs.init || ticket < 0 || ticket >= hs.next_ticket) return fail(c, ORBFE_EBADARG, "stream_wait: ticket %lld was never issued", (long long)ticket);
  const int D = orbfe_ctx::HostStream::kDepth;
  if (ticket + D < hs.next_ticket) return ORBFE_OK;  // ticket + D has been submitted since, and that submit waited for this one
  HIP_TRY(c, hipSetDevice(c->device));
  hipEvent_t done = hs.ev_done[ticket % D];
  api_lk.lk.unlock();  // the wait itself needs nothing of the context: another thread may submit meanwhile
  HIP_TRY(c, hipEventSynchronize(done));
  return ORBFE_OK;
}

orbfe_status orbfe_stream_device_results(orbfe_ctx* c, int64_t ticket, int32_t n_pairs, const void** d_kps, const void** d_desc,
                                         const void** d_counts, const void** d_right_u, const void** d_depth, const void** d_nmatch) {
  ApiLock api_lk(c);
  if (!c) return ORBFE_EBADARG;
  orbfe_ctx::HostStream& hs = c->hs;
  if (!hs.init || ticket < 0 || ticket >= hs.next_ticket || ticket + orbfe_ctx::HostStream::kDepth < hs.next_ticket || n_pairs <= 0 ||
      2 * n_pairs > c->cfg.max_images)
    return fail(c, ORBFE_EBADARG, "stream_device_results: ticket %lld is not live (next %lld) or bad pair count %d", (long long)ticket,
                (long long)hs.next_ticket, n_pairs);
  if (n_pairs != hs.n_pairs_of[ticket % orbfe_ctx::HostStream::kDepth])
    return fail(c, ORBFE_EBADARG, "stream_device_results: ticket %lld was submitted with %d pairs, not %d (the packed layout depends on it)",
                (long long)ticket, hs.n_pairs_of[ticket % orbfe_ctx::HostStream::kDepth], n_pairs);
  const PackLayout l = pack_layout(c, n_pairs);
  const uint8_t* b = hs.d_out[ticket % orbfe_ctx::HostStream::kDepth];
  if (d_kps) *d_kps = b + l.o_kps;
  if (d_desc) *d_desc = b + l.o_desc;
  if (d_counts) *d_counts = b + l.o_cnt;
  if (d_right_u) *d_right_u = b + l.o_ru;
  if (d_depth) *d_depth = b + l.o_dp;
  if (d_nmatch) *d_nmatch = b + l.o_nm;
  return ORBFE_OK;
}

// Frame records of a ticket (layout: k_glue.hip, k_pack_records) into caller-provided DEVICE memory, for the sequence-level gather.
size_t orbfe_record_bytes(const orbfe_ctx* c) { return c ? 16 + (size_t)std::max(c->cfg.n_features, 1) * (28 + 32 + 8 + 8) : 0; }

orbfe_status orbfe_stream_pack_records(orbfe_ctx* c, int64_t ticket, int32_t n_pairs, void* d_records) {
  ApiLock api_lk(c);
  if (!c || !d_records) return fail(c, ORBFE_EBADARG, "stream_pack_records: NULL argument");
  orbfe_ctx::HostStream& hs = c->hs;
  if (!hs.init || ticket < 0 || ticket >= hs.next_ticket || ticket + orbfe_ctx::HostStream::kDepth < hs.next_ticket || n_pairs <= 0 ||
      2 * n_pairs > c->cfg.max_images)
    return fail(c, ORBFE_EBADARG, "stream_pack_records: ticket %lld is not live (next %lld) or bad pair count %d", (long long)ticket,
                (long long)hs.next_ticket, n_pairs);
  if (n_pairs != hs.n_pairs_of[ticket % orbfe_ctx::HostStream::kDepth])
    return fail(c, ORBFE_EBADARG, "stream_pack_records: ticket %lld was submitted with %d pairs, not %d (the packed layout depends on it)",
                (long long)ticket, hs.n_pairs_of[ticket % orbfe_ctx::HostStream::kDepth], n_pairs);
  HIP_TRY(c, hipSetDevice(c->device));
  const int b = (int)(ticket % orbfe_ctx::HostStream::kDepth);
  const PackLayout l = pack_layout(c, n_pairs);
  const uint8_t* src = hs.d_out[b];
  // on the download stream, behind the ticket's own completion: ordered after the results are in the buffer and before the
  // buffer is handed to ticket + 3 (whose pack waits for the event recorded here)
  HIP_TRY(c, hipStreamWaitEvent(hs.d2h, hs.ev_done[b], 0));
  launch_pack_records(hs.d2h, src + l.o_kps, src + l.o_desc, (const int32_t*)(src + l.o_cnt), src + l.o_ru, src + l.o_dp,
                      (const int32_t*)(src + l.o_nm), std::max(c->cfg.n_features, 1), n_pairs, d_records);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(hs.ev_done[b], hs.d2h));
  hipEvent_t done = hs.ev_done[b];
  api_lk.lk.unlock();  // the wait lasts a batch's compute and needs nothing of the context: another thread may submit / fetch meanwhile
  HIP_TRY(c, hipEventSynchronize(done));
  return ORBFE_OK;
}

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

orbfe_status orbfe_ba_eval_edges(orbfe_ctx* c, const orbfe_ba_problem* p, const orbfe_ba_edge_out* o) {
  ApiLock api_lk(c);
  if (!c || !p || !o) return fail(c, ORBFE_EBADARG, "ba_eval_edges: NULL argument");
  const int E = p->n_edges;
  if (E < 0 || p->n_poses < 0 || p->n_points < 0) return fail(c, ORBFE_EBADARG, "ba_eval_edges: negative size");
  if (E == 0) return ORBFE_OK;
  if (!p->poses || !p->points || !p->edge_pose || !p->edge_point || !p->meas || !p->is_stereo || !p->info || !p->huber_delta || !o->error ||
      !o->chi2 || !o->rho)
    return fail(c, ORBFE_EBADARG, "ba_eval_edges: NULL array");
  for (int e = 0; e < E; ++e)
    if (p->edge_pose[e] < 0 || p->edge_pose[e] >= p->n_poses || p->edge_point[e] < 0 || p->edge_point[e] >= p->n_points)
      return fail(c, ORBFE_EBADARG, "ba_eval_edges: edge %d references vertex out of range", e);
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  const size_t o_pose = take((size_t)p->n_poses * 56), o_pt = take((size_t)p->n_points * 24), o_ep = take((size_t)E * 4),
               o_et = take((size_t)E * 4), o_meas = take((size_t)E * 24), o_st = take((size_t)E), o_info = take((size_t)E * 8),
               o_delta = take((size_t)E * 8), o_up_end = take(8), o_err = take((size_t)E * 24), o_chi = take((size_t)E * 8),
               o_rho = take((size_t)E * 16), o_dp = take((size_t)E), o_jpt = take((size_t)E * 72), o_jps = take((size_t)E * 144), o_out_end = take(8);
  TRY(ensure_tmp(c, off));
  uint8_t* b = (uint8_t*)c->d_tmp;
  // up to 16 MB in all: inputs as ONE upload through the page-locked staging buffer and the results as one download (eight copies from
  // and six to pageable memory otherwise -- each staged by the runtime on its own)
  const size_t out_last = o->j_pose ? o_out_end : (o->j_point ? o_jps : o_jpt);
  const bool staged = o_up_end + (out_last - o_err) <= ((size_t)16 << 20);
  uint8_t* hs = nullptr;
  if (staged) {
    TRY(ensure_stage(c, std::max(o_up_end, out_last - o_err)));
    hs = c->main.h_stage;
  }
  auto up = [&](size_t o2, const void* src, size_t bytes) -> hipError_t {
    if (!bytes) return hipSuccess;
    if (staged) {
      std::memcpy(hs + o2, src, bytes);
      return hipSuccess;
    }
    return hipMemcpyAsync(b + o2, src, bytes, hipMemcpyHostToDevice, c->stream);
  };
  HIP_TRY(c, up(o_pose, p->poses, (size_t)p->n_poses * 56));
  HIP_TRY(c, up(o_pt, p->points, (size_t)p->n_points * 24));
  HIP_TRY(c, up(o_ep, p->edge_pose, (size_t)E * 4));
  HIP_TRY(c, up(o_et, p->edge_point, (size_t)E * 4));
  HIP_TRY(c, up(o_meas, p->meas, (size_t)E * 24));
  HIP_TRY(c, up(o_st, p->is_stereo, (size_t)E));
  HIP_TRY(c, up(o_info, p->info, (size_t)E * 8));
  HIP_TRY(c, up(o_delta, p->huber_delta, (size_t)E * 8));
  if (staged) HIP_TRY(c, hipMemcpyAsync(b, hs, o_up_end, hipMemcpyHostToDevice, c->stream));
  BaParamsDev prm = {p->fx, p->fy, p->cx, p->cy, p->bf};
  {
    StageTimer tm(c, ORBFE_STAGE_BA, c->stream);
    launch_ba_edges(c->stream, E, (const double*)(b + o_pose), (const double*)(b + o_pt), (const int32_t*)(b + o_ep),
                    (const int32_t*)(b + o_et), (const double*)(b + o_meas), b + o_st, (const double*)(b + o_info),
                    (const double*)(b + o_delta), prm, (double*)(b + o_err), (double*)(b + o_chi), (double*)(b + o_rho),
                    o->j_point ? (double*)(b + o_jpt) : nullptr, o->j_pose ? (double*)(b + o_jps) : nullptr,
                    o->depth_positive ? b + o_dp : nullptr);
  }
  HIP_TRY(c, hipGetLastError());
  if (staged) {
    HIP_TRY(c, hipMemcpyAsync(hs, b + o_err, out_last - o_err, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drain_timers(c);
    std::memcpy(o->error, hs, (size_t)E * 24);
    std::memcpy(o->chi2, hs + (o_chi - o_err), (size_t)E * 8);
    std::memcpy(o->rho, hs + (o_rho - o_err), (size_t)E * 16);
    if (o->depth_positive) std::memcpy(o->depth_positive, hs + (o_dp - o_err), (size_t)E);
    if (o->j_point) std::memcpy(o->j_point, hs + (o_jpt - o_err), (size_t)E * 72);
    if (o->j_pose) std::memcpy(o->j_pose, hs + (o_jps - o_err), (size_t)E * 144);
    return ORBFE_OK;
  }
  HIP_TRY(c, hipMemcpyAsync(o->error, b + o_err, (size_t)E * 24, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(o->chi2, b + o_chi, (size_t)E * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(o->rho, b + o_rho, (size_t)E * 16, hipMemcpyDeviceToHost, c->stream));
  if (o->j_point) HIP_TRY(c, hipMemcpyAsync(o->j_point, b + o_jpt, (size_t)E * 72, hipMemcpyDeviceToHost, c->stream));
  if (o->j_pose) HIP_TRY(c, hipMemcpyAsync(o->j_pose, b + o_jps, (size_t)E * 144, hipMemcpyDeviceToHost, c->stream));
  if (o->depth_positive) HIP_TRY(c, hipMemcpyAsync(o->depth_positive, b + o_dp, (size_t)E, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  return ORBFE_OK;
}

orbfe_status orbfe_ba_build_system(orbfe_ctx* c, const orbfe_ba_problem* p, const uint8_t* pose_fixed, const orbfe_ba_system_out* o) {
  ApiLock api_lk(c);
  if (!c || !p || !o) return fail(c, ORBFE_EBADARG, "ba_build_system: NULL argument");
  const int E = p->n_edges, NK = p->n_poses, NP = p->n_points;
  if (E < 0 || NK < 0 || NP < 0) return fail(c, ORBFE_EBADARG, "ba_build_system: negative size");
  if (!o->Hpp || !o->bp || !o->Hll || !o->bl) return fail(c, ORBFE_EBADARG, "ba_build_system: NULL output");
  if (E && (!p->poses || !p->points || !p->edge_pose || !p->edge_point || !p->meas || !p->is_stereo || !p->info || !p->huber_delta))
    return fail(c, ORBFE_EBADARG, "ba_build_system: NULL array");
  for (int e = 0; e < E; ++e)
    if (p->edge_pose[e] < 0 || p->edge_pose[e] >= NK || p->edge_point[e] < 0 || p->edge_point[e] >= NP)
      return fail(c, ORBFE_EBADARG, "ba_build_system: edge %d references vertex out of range", e);
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  // vertex -> edges lists, edges in ascending index (counting sort): the summation order of the segmented reductions
  std::vector<int32_t> pt_off(NP + 1, 0), ps_off(NK + 1, 0), pt_edges(std::max(E, 1)), ps_edges(std::max(E, 1));
  for (int e = 0; e < E; ++e) {
    ++pt_off[p->edge_point[e] + 1];
    ++ps_off[p->edge_pose[e] + 1];
  }
  for (int i = 0; i < NP; ++i) pt_off[i + 1] += pt_off[i];
  for (int i = 0; i < NK; ++i) ps_off[i + 1] += ps_off[i];
  {
    std::vector<int32_t> pc(pt_off.begin(), pt_off.end() - 1), kc(ps_off.begin(), ps_off.end() - 1);
    for (int e = 0; e < E; ++e) {
      pt_edges[pc[p->edge_point[e]]++] = e;
      ps_edges[kc[p->edge_pose[e]]++] = e;
    }
  }
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  // r3: the system is built by the kernels of the device-side Levenberg-Marquardt path (k_lm.hip: every edge linearised once, eight lanes
  // per point, a workgroup per pose -- 22 us where round 1's three kernels, each recomputing every edge's Jacobians, took 171); inputs
  // and lists go up as ONE block through the page-locked staging buffer, the blocks come back as one
  const size_t o_pose = take((size_t)NK * 56), o_pt = take((size_t)NP * 24), o_ep = take((size_t)E * 4), o_et = take((size_t)E * 4),
               o_meas = take((size_t)E * 24), o_st = take((size_t)E), o_info = take((size_t)E * 8), o_delta = take((size_t)E * 8),
               o_fix = take((size_t)NK), o_pto = take((size_t)(NP + 1) * 4), o_pte = take((size_t)E * 4), o_pso = take((size_t)(NK + 1) * 4),
               o_pse = take((size_t)E * 4), o_state = take(sizeof(LmState)), o_level = take((size_t)E), o_up_end = take(8),
               o_hpp = take((size_t)NK * 288), o_bp = take((size_t)NK * 48), o_hll = take((size_t)NP * 72),
               o_bl = take((size_t)NP * 24), o_hpl = take((size_t)E * 144), o_out_end = take(8), o_terms = take((size_t)E * 256),
               o_chi = take((size_t)((NP + 31) / 32) * 8);
  TRY(ensure_tmp(c, off));
  TRY(ensure_stage(c, std::max(o_up_end, o_out_end - o_hpp)));
  uint8_t* b = (uint8_t*)c->d_tmp;
  uint8_t* hs = c->main.h_stage;
  auto up = [&](size_t o2, const void* src, size_t bytes) {
    if (bytes) std::memcpy(hs + o2, src, bytes);
  };
  up(o_pose, p->poses, (size_t)NK * 56);
  up(o_pt, p->points, (size_t)NP * 24);
  up(o_ep, p->edge_pose, (size_t)E * 4);
  up(o_et, p->edge_point, (size_t)E * 4);
  up(o_meas, p->meas, (size_t)E * 24);
  up(o_st, p->is_stereo, (size_t)E);
  up(o_info, p->info, (size_t)E * 8);
  up(o_delta, p->huber_delta, (size_t)E * 8);
  if (pose_fixed)
    up(o_fix, pose_fixed, (size_t)NK);
  else
    std::memset(hs + o_fix, 0, (size_t)std::max(NK, 1));
  up(o_pto, pt_off.data(), (size_t)(NP + 1) * 4);
  up(o_pte, pt_edges.data(), (size_t)E * 4);
  up(o_pso, ps_off.data(), (size_t)(NK + 1) * 4);
  up(o_pse, ps_edges.data(), (size_t)E * 4);
  std::memset(hs + o_state, 0, o_up_end - o_state);  // control state (buffer 0 current) and the edge levels (all active)
  HIP_TRY(c, hipMemcpyAsync(b, hs, o_up_end, hipMemcpyHostToDevice, c->stream));
  BaParamsDev prm = {p->fx, p->fy, p->cx, p->cy, p->bf};
  {
    LmLaunch L{};
    L.NK = NK, L.NP = NP, L.E = E, L.nf = 0;
    L.poses[0] = L.poses[1] = (double*)(b + o_pose), L.points[0] = L.points[1] = (double*)(b + o_pt);
    L.terms[0] = L.terms[1] = (double*)(b + o_terms), L.Hpl[0] = L.Hpl[1] = (double*)(b + o_hpl);
    L.Hpp[0] = L.Hpp[1] = (double*)(b + o_hpp), L.bp[0] = L.bp[1] = (double*)(b + o_bp);
    L.Hll[0] = L.Hll[1] = (double*)(b + o_hll), L.bl[0] = L.bl[1] = (double*)(b + o_bl);
    L.chi_part[0] = L.chi_part[1] = (double*)(b + o_chi);
    L.state = (LmState*)(b + o_state);
    L.edge_pose = (const int32_t*)(b + o_ep), L.edge_point = (const int32_t*)(b + o_et);
    L.pt_off = (const int32_t*)(b + o_pto), L.pt_edges = (const int32_t*)(b + o_pte);
    L.ps_off = (const int32_t*)(b + o_pso), L.ps_edges = (const int32_t*)(b + o_pse);
    L.meas = (const double*)(b + o_meas), L.info = (const double*)(b + o_info), L.is_stereo = b + o_st, L.fixed = b + o_fix;
    L.info_eff = (double*)(b + o_info), L.delta_eff = (double*)(b + o_delta), L.chi2_last = nullptr, L.level = b + o_level;
    L.prm = prm;
    StageTimer tm(c, ORBFE_STAGE_BA, c->stream);
    launch_lm_build(c->stream, L, 0, 0, 0, true);
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(hs, b + o_hpp, (o->Hpl ? o_out_end : o_hpl) - o_hpp, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  std::memcpy(o->Hpp, hs, (size_t)NK * 288);
  std::memcpy(o->bp, hs + (o_bp - o_hpp), (size_t)NK * 48);
  std::memcpy(o->Hll, hs + (o_hll - o_hpp), (size_t)NP * 72);
  std::memcpy(o->bl, hs + (o_bl - o_hpp), (size_t)NP * 24);
  if (o->Hpl) std::memcpy(o->Hpl, hs + (o_hpl - o_hpp), (size_t)E * 144);
  return ORBFE_OK;
}

// Optimizer::OptimizeLocalMap's two optimize() calls (Optimizer.cc:336-362) with g2o's Levenberg-Marquardt control on the host
// (a handful of scalars per trial) and every vertex / edge / block operation on the device.
orbfe_status orbfe_ba_local_optimize(orbfe_ctx* c, const orbfe_ba_problem* p, const uint8_t* pose_fixed, int32_t iters_first,
                                     int32_t iters_second, const volatile uint8_t* stop_flag, const orbfe_ba_optimize_out* o) {
  ApiLock api_lk(c);
  static const bool trace_host = getenv("ORBFE_LBA_TRACE") != nullptr;  // diagnostic: host phases of this call on stderr
  auto t_prev = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!trace_host) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[orbfe lba] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_prev).count());
    t_prev = now;
  };
  if (!c || !p || !o) return fail(c, ORBFE_EBADARG, "ba_local_optimize: NULL argument");
  const int E = p->n_edges, NK = p->n_poses, NP = p->n_points;
  if (E < 0 || NK < 0 || NP < 0 || iters_first < 0 || iters_second < 0) return fail(c, ORBFE_EBADARG, "ba_local_optimize: negative size");
  if (!o->poses || !o->points) return fail(c, ORBFE_EBADARG, "ba_local_optimize: NULL output");
  if ((NK && !p->poses) || (NP && !p->points) ||
      (E && (!p->edge_pose || !p->edge_point || !p->meas || !p->is_stereo || !p->info || !p->huber_delta)))
    return fail(c, ORBFE_EBADARG, "ba_local_optimize: NULL array");
  for (int e = 0; e < E; ++e)
    if (p->edge_pose[e] < 0 || p->edge_pose[e] >= NK || p->edge_point[e] < 0 || p->edge_point[e] >= NP)
      return fail(c, ORBFE_EBADARG, "ba_local_optimize: edge %d references vertex out of range", e);
  // free poses, vertex -> edges lists (ascending edge index), pose-pair lists of the Schur complement
  std::vector<int32_t> slot(std::max(NK, 1), -1), free_pose;
  for (int k = 0; k < NK; ++k)
    if (!(pose_fixed && pose_fixed[k])) {
      slot[k] = (int32_t)free_pose.size();
      free_pose.push_back(k);
    }
  const int nf = (int)free_pose.size();
  // (no bound on nf: up to LBA_MAX_FREE free keyframes the reduced system is factorised by one workgroup out of LDS, beyond that by the
  //  multi-workgroup path of k_lba.hip with its panel in global memory)
  std::vector<int32_t> pt_off(NP + 1, 0), ps_off(NK + 1, 0), pt_edges(std::max(E, 1)), ps_edges(std::max(E, 1));
  for (int e = 0; e < E; ++e) {
    ++pt_off[p->edge_point[e] + 1];
    ++ps_off[p->edge_pose[e] + 1];
  }
  for (int i = 0; i < NP; ++i) pt_off[i + 1] += pt_off[i];
  for (int i = 0; i < NK; ++i) ps_off[i + 1] += ps_off[i];
  {
    std::vector<int32_t> pc(pt_off.begin(), pt_off.end() - 1), kc(ps_off.begin(), ps_off.end() - 1);
    for (int e = 0; e < E; ++e) {
      pt_edges[pc[p->edge_point[e]]++] = e;
      ps_edges[kc[p->edge_pose[e]]++] = e;
    }
  }
  // The device-side Levenberg-Marquardt path (k_lm.hip) builds the pair lists of the reduced system itself, from a (pose, point) -> edge
  // table: that needs a pose to observe a point at most once (as every map of the reference does); anything else takes the host-driven path.
  bool single_obs = true;
  int pair_cap = 1;
  {
    std::vector<int32_t> seen(std::max(NK, 1), -1);
    for (int pt = 0; pt < NP && single_obs; ++pt)
      for (int a = pt_off[pt]; a < pt_off[pt + 1]; ++a) {
        const int k = p->edge_pose[pt_edges[a]];
        if (seen[k] == pt) {
          single_obs = false;
          break;
        }
        seen[k] = pt;
      }
    for (int k = 0; k < NK; ++k)
      if (slot[k] >= 0) pair_cap = std::max(pair_cap, ps_off[k + 1] - ps_off[k]);
  }
  mark("validate + vertex lists");
  std::vector<int32_t> pair_off(1, 0);
  std::vector<int2> pairs;
  const bool lower_only = c && c->lm_on_device && E > 0 && nf <= LM_BIG_MAX_NB && single_obs;  // (= dev_lm below)
  const bool big_solver = lower_only && nf > LM_CHOL_MAX_NB;  // the blocked multi-workgroup Cholesky of k_lmbig.hip
  if (!lower_only) {
    pair_off.assign((size_t)nf * nf + 1, 0);
    for (int pt = 0; pt < NP; ++pt)
      for (int a = pt_off[pt]; a < pt_off[pt + 1]; ++a) {
        const int i = slot[p->edge_pose[pt_edges[a]]];
        if (i < 0) continue;
        for (int b2 = pt_off[pt]; b2 < pt_off[pt + 1]; ++b2) {
          const int j = slot[p->edge_pose[pt_edges[b2]]];
          if (j >= 0) ++pair_off[(size_t)i * nf + j + 1];
        }
      }
    for (size_t q = 0; q < (size_t)nf * nf; ++q) pair_off[q + 1] += pair_off[q];
    pairs.resize(std::max<size_t>(pair_off.back(), 1));
    std::vector<int32_t> cur(pair_off.begin(), pair_off.end() - 1);
    for (int pt = 0; pt < NP; ++pt)
      for (int a = pt_off[pt]; a < pt_off[pt + 1]; ++a) {
        const int e1 = pt_edges[a], i = slot[p->edge_pose[e1]];
        if (i < 0) continue;
        for (int b2 = pt_off[pt]; b2 < pt_off[pt + 1]; ++b2) {
          const int e2 = pt_edges[b2], j = slot[p->edge_pose[e2]];
          if (j >= 0) pairs[cur[(size_t)i * nf + j]++] = make_int2(e1, e2);
        }
      }
  }
  mark("pair lists");
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  const size_t n = (size_t)6 * nf;
  const size_t o_pose = take((size_t)NK * 56), o_pt = take((size_t)NP * 24), o_pose_bk = take((size_t)NK * 56), o_pt_bk = take((size_t)NP * 24),
               o_ep = take((size_t)E * 4), o_et = take((size_t)E * 4), o_meas = take((size_t)E * 24), o_st = take((size_t)E),
               o_info = take((size_t)E * 8), o_info_eff = take((size_t)E * 8), o_delta = take((size_t)E * 8), o_fix = take((size_t)NK),
               o_pto = take((size_t)(NP + 1) * 4), o_pte = take((size_t)E * 4), o_pso = take((size_t)(NK + 1) * 4), o_pse = take((size_t)E * 4),
               o_free = take((size_t)nf * 4), o_slot = take((size_t)NK * 4), o_pairoff = take(pair_off.size() * 4),
               o_pairs = take(pairs.size() * 8), o_lmstate = take(sizeof(LmState)),  // (the initial control state rides in the one upload)
               o_hpp = take((size_t)NK * 288), o_bp = take((size_t)NK * 48), o_hll = take((size_t)NP * 72),
               o_bl = take((size_t)NP * 24), o_hpl = take((size_t)E * 144), o_w = take((size_t)E * 144),
               o_s = take(n * n * 8), o_rhs = take(n * 8), o_x = take((n + 48) * 8), o_dxp = take((size_t)NK * 48), o_dxl = take((size_t)NP * 24),
               o_err = take((size_t)E * 24), o_chi2 = take((size_t)E * 8), o_rho = take((size_t)E * 16),
               // one block that starts as zeros (ONE fill): edge levels | chi2 of the last linearisation | point inverses
               o_level = take((size_t)E), o_last = take((size_t)E * 8), o_dinv = take((size_t)NP * 72), o_zero_end = take(8),
               o_depth = take((size_t)E), o_bad = take((size_t)E), o_sc = take(64),
               o_big = take(nf > LBA_MAX_FREE ? ((n + 1) * 6 + (size_t)nf * 36 + n) * 8 : 8);
  // the device-side Levenberg-Marquardt path (k_lm.hip): second estimate / system buffers, per-edge terms, blocked reduced system
  const bool dev_lm = lower_only;
  const int chi_blocks = (NP + 31) / 32, scale_blocks = (NP + 31) / 32 + (NK + 255) / 256;  // (k_lm_linpoints: a partial sum per block of 32 points)
  size_t l_ptable = 0, l_pairs = 0, l_paircnt = 0, l_pose1 = 0, l_pt1 = 0, l_terms[2] = {0, 0}, l_hpl1 = 0, l_hpp1 = 0, l_bp1 = 0, l_hll1 = 0, l_bl1 = 0, l_chi[2] = {0, 0}, l_sblk = 0,
         l_scale = 0, l_big = 0, l_bigflags = 0, l_biginv = 0, l_pose_out = 0, l_pt_out = 0, l_chi2_out = 0, l_level_out = 0, l_bad_out = 0, l_state_out = 0, l_out_end = 0;
  if (dev_lm) {
    l_pose1 = take((size_t)NK * 56), l_pt1 = take((size_t)NP * 24);
    l_ptable = take((size_t)nf * NP * 4), l_pairs = take((size_t)nf * (nf + 1) / 2 * pair_cap * 8), l_paircnt = take((size_t)nf * (nf + 1) / 2 * 4);
    l_terms[0] = take((size_t)E * 256), l_terms[1] = take((size_t)E * 256);
    l_hpl1 = take((size_t)E * 144), l_hpp1 = take((size_t)NK * 288), l_bp1 = take((size_t)NK * 48), l_hll1 = take((size_t)NP * 72),
    l_bl1 = take((size_t)NP * 24);
    l_chi[0] = take((size_t)chi_blocks * 8), l_chi[1] = take((size_t)chi_blocks * 8);
    l_sblk = take(big_solver ? 8 : (size_t)nf * (nf + 1) / 2 * 288), l_scale = take((size_t)scale_blocks * 8);
    l_big = take(big_solver ? lm_big_bytes(nf) : 8), l_bigflags = take(big_solver ? (2 * ((size_t)lm_big_ld(nf) / 48) + 4) * 4 : 8),
    l_biginv = take(big_solver ? lm_big_inv_bytes(nf) : 8);
    // the results as ONE block (one download): poses | points | chi2 | level | bad
    l_pose_out = take((size_t)NK * 56), l_pt_out = take((size_t)NP * 24), l_chi2_out = take((size_t)E * 8), l_level_out = take((size_t)E),
    l_bad_out = take((size_t)E), l_state_out = take(sizeof(LmState)), l_out_end = take(8);  // (+ the control state as the last control step left it)
  }
  TRY(ensure_tmp(c, off));
  uint8_t* b = (uint8_t*)c->d_tmp;
  hipStream_t st = c->stream;
  // ONE upload: the inputs and the lists built above are laid out in page-locked staging memory exactly as in the device scratch
  // (they are its first o_hpp bytes) and go up as a single asynchronous copy -- eighteen copies from pageable memory were staged by the
  // runtime one by one, ~0.3 ms of a 4 ms call
  const size_t up_bytes = o_hpp;
  TRY(ensure_stage(c, std::max(up_bytes, dev_lm ? l_out_end - l_pose_out : (size_t)0)));  // (also the target of the one result download)
  uint8_t* hs = c->main.h_stage;
  auto up = [&](size_t o2, const void* src, size_t bytes) -> hipError_t {
    if (bytes) std::memcpy(hs + o2, src, bytes);
    return hipSuccess;
  };
  std::vector<uint8_t> fixed_h(std::max(NK, 1), 0);
  if (pose_fixed) std::memcpy(fixed_h.data(), pose_fixed, NK);
  HIP_TRY(c, up(o_pose, p->poses, (size_t)NK * 56));
  HIP_TRY(c, up(o_pt, p->points, (size_t)NP * 24));
  HIP_TRY(c, up(o_ep, p->edge_pose, (size_t)E * 4));
  HIP_TRY(c, up(o_et, p->edge_point, (size_t)E * 4));
  HIP_TRY(c, up(o_meas, p->meas, (size_t)E * 24));
  HIP_TRY(c, up(o_st, p->is_stereo, (size_t)E));
  HIP_TRY(c, up(o_info, p->info, (size_t)E * 8));
  HIP_TRY(c, up(o_info_eff, p->info, (size_t)E * 8));
  HIP_TRY(c, up(o_delta, p->huber_delta, (size_t)E * 8));
  HIP_TRY(c, up(o_fix, fixed_h.data(), (size_t)NK));
  HIP_TRY(c, up(o_pto, pt_off.data(), (size_t)(NP + 1) * 4));
  HIP_TRY(c, up(o_pte, pt_edges.data(), (size_t)E * 4));
  HIP_TRY(c, up(o_pso, ps_off.data(), (size_t)(NK + 1) * 4));
  HIP_TRY(c, up(o_pse, ps_edges.data(), (size_t)E * 4));
  HIP_TRY(c, up(o_free, free_pose.data(), (size_t)nf * 4));
  HIP_TRY(c, up(o_slot, slot.data(), (size_t)NK * 4));
  HIP_TRY(c, up(o_pairoff, pair_off.data(), pair_off.size() * 4));
  HIP_TRY(c, up(o_pairs, pairs.data(), pairs.size() * 8));
  {
    LmState init{};
    init.iters[0] = iters_first, init.iters[1] = iters_second, init.need_chi = 1, init.ok = 1;
    HIP_TRY(c, up(o_lmstate, &init, sizeof init));
  }
  mark("stage inputs");
  HIP_TRY(c, hipMemcpyAsync(b, hs, up_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemsetAsync(b + o_level, 0, o_zero_end - o_level, st));  // (every memset is a launch of 4.6 us: six of them preceded the first kernel)

  const BaParamsDev prm = {p->fx, p->fy, p->cx, p->cy, p->bf};
  double* d_poses = (double*)(b + o_pose);
  double* d_points = (double*)(b + o_pt);
  const int32_t* d_ek = (const int32_t*)(b + o_ep);
  const int32_t* d_ep = (const int32_t*)(b + o_et);
  double* d_sc = (double*)(b + o_sc);  // [0] robust chi2, [1] max diagonal, [2] lambda, [3] ok (int), [4] scale
  if (dev_lm) {
    // ---- Levenberg-Marquardt control on the device: enqueue the whole optimisation, synchronise once ------------------------------
    if (!c->h_abort) {
      HIP_TRY(c, hipHostMalloc((void**)&c->h_abort, 64, hipHostMallocMapped));
    }
    void* d_abort = nullptr;
    HIP_TRY(c, hipHostGetDevicePointer(&d_abort, (void*)c->h_abort, 0));
    *c->h_abort = (stop_flag && *stop_flag) ? 1 : 0;
    LmLaunch L{};
    L.NK = NK, L.NP = NP, L.E = E, L.nf = nf;
    L.poses[0] = d_poses, L.poses[1] = (double*)(b + l_pose1), L.points[0] = d_points, L.points[1] = (double*)(b + l_pt1);
    L.terms[0] = (double*)(b + l_terms[0]), L.terms[1] = (double*)(b + l_terms[1]);
    L.Hpl[0] = (double*)(b + o_hpl), L.Hpl[1] = (double*)(b + l_hpl1), L.Hpp[0] = (double*)(b + o_hpp), L.Hpp[1] = (double*)(b + l_hpp1);
    L.bp[0] = (double*)(b + o_bp), L.bp[1] = (double*)(b + l_bp1), L.Hll[0] = (double*)(b + o_hll), L.Hll[1] = (double*)(b + l_hll1);
    L.bl[0] = (double*)(b + o_bl), L.bl[1] = (double*)(b + l_bl1), L.chi_part[0] = (double*)(b + l_chi[0]), L.chi_part[1] = (double*)(b + l_chi[1]);
    L.state = (LmState*)(b + o_lmstate);
    L.edge_pose = d_ek, L.edge_point = d_ep, L.pt_off = (const int32_t*)(b + o_pto), L.pt_edges = (const int32_t*)(b + o_pte);
    L.ps_off = (const int32_t*)(b + o_pso), L.ps_edges = (const int32_t*)(b + o_pse), L.free_pose = (const int32_t*)(b + o_free);
    L.pose_slot = (const int32_t*)(b + o_slot), L.pairs = (int2*)(b + l_pairs), L.pair_cnt = (int32_t*)(b + l_paircnt);
    L.pair_table = (int32_t*)(b + l_ptable), L.pair_cap = pair_cap;
    L.meas = (const double*)(b + o_meas), L.info = (const double*)(b + o_info), L.is_stereo = b + o_st, L.fixed = b + o_fix;
    L.info_eff = (double*)(b + o_info_eff), L.delta_eff = (double*)(b + o_delta), L.chi2_last = (double*)(b + o_last), L.level = b + o_level;
    L.Dinv = (double*)(b + o_dinv), L.W = (double*)(b + o_w), L.Sblk = (double*)(b + l_sblk), L.rhs = (double*)(b + o_rhs), L.x = (double*)(b + o_x);
    L.scale_part = (double*)(b + l_scale), L.chi2_out = (double*)(b + l_chi2_out), L.poses_out = (double*)(b + l_pose_out);
    L.points_out = (double*)(b + l_pt_out), L.bad = b + l_bad_out, L.level_out = b + l_level_out;
    L.abort_flag = (const volatile uint8_t*)d_abort, L.prm = prm;
    // (the initial state went up with the inputs; the ticket and the point inverses -- read by a trial whose point block was singular --
    //  are part of the one zero fill)
    L.state_out = (LmState*)(b + l_state_out);
    L.M = big_solver ? (double*)(b + l_big) : nullptr, L.ld = big_solver ? lm_big_ld(nf) : 0, L.lmb_flags = (int32_t*)(b + l_bigflags), L.lmb_inv = (double*)(b + l_biginv);
    StageTimer tm(c, ORBFE_STAGE_BA, st);
    if (big_solver) {
      HIP_TRY(c, hipMemsetAsync(b + l_big, 0, (l_bigflags - l_big) + (2 * ((size_t)L.ld / 48) + 4) * 4, st));  // the matrix and the flags behind it
      launch_lm_big_init(st, L);
    }
    HIP_TRY(c, hipMemsetAsync(L.pair_table, 0xFF, (size_t)nf * NP * 4, st));
    launch_lm_pairs(st, L);
    launch_lm_build(st, L, 0, 0, iters_first > 0 ? 1 : 0, true);  // computeActiveErrors + buildSystem at the initial estimate
    launch_lm_maxdiag(st, L, 0);
    // trials provisioned per pass: every iteration needs at least one, a rejected trial costs one more; what is left over runs as no-ops
    // (a few microseconds each), what is missing is enqueued in the next pass, after the one synchronisation of this one
    // (measured: a provisioned trial that turns out not to be needed is six empty launches of 4.6 us; one spare
    // -- in round 0, where coming up one short would leave the ten trials of round 1 as no-ops in this pass; round 1 gets none: if a trial
    // of it is rejected, the second pass enqueues what is missing)
    int steps_a = std::min(iters_first + 1, 24), steps_b = std::min(iters_second, 24);
    LmState fin{};
    for (int pass = 0;; ++pass) {
      launch_lm_steps(st, L, steps_a);
      launch_lm_switch(st, L);
      launch_lm_steps(st, L, steps_b);
      launch_lm_final(st, L);
      HIP_TRY(c, hipGetLastError());
      const size_t out_bytes = l_out_end - l_pose_out;
      HIP_TRY(c, hipMemcpyAsync(hs, b + l_pose_out, out_bytes, hipMemcpyDeviceToHost, st));  // the upload from hs finished long ago (stream order)
      if (stop_flag) {
        // the device polls the mapped byte between the trials; the caller's flag (LocalMapping::mbAbortBA, written by the Tracking
        // thread) is mirrored into it while this thread waits
        hipEvent_t ev = nullptr;
        HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t er = hipEventRecord(ev, st);
        while (er == hipSuccess) {
          if (*stop_flag) *c->h_abort = 1;
          er = hipEventQuery(ev);
          if (er == hipErrorNotReady) {
            (void)hipGetLastError();
            er = hipSuccess;
            sched_yield();
            continue;
          }
          break;
        }
        (void)hipEventDestroy(ev);
        HIP_TRY(c, er);
      }
      mark("enqueue");
      HIP_TRY(c, hipStreamSynchronize(st));
      mark("device (wait)");
      std::memcpy(&fin, hs + (l_state_out - l_pose_out), sizeof fin);
      if (fin.finalized) {
        std::memcpy(o->poses, hs, (size_t)NK * 56);
        std::memcpy(o->points, hs + (l_pt_out - l_pose_out), (size_t)NP * 24);
        if (o->chi2) std::memcpy(o->chi2, hs + (l_chi2_out - l_pose_out), (size_t)E * 8);
        if (o->level) std::memcpy(o->level, hs + (l_level_out - l_pose_out), (size_t)E);
        if (o->bad) std::memcpy(o->bad, hs + (l_bad_out - l_pose_out), (size_t)E);
        break;
      }
      if (pass >= 4096) return fail(c, ORBFE_EDEVICE, "ba_local_optimize: the device-side Levenberg-Marquardt loop did not finish (round %d, phase %d)", fin.round, fin.phase);
      // more trials were needed than provisioned: continue where the state stands
      steps_a = fin.switched || fin.round == 2 ? 0 : std::min(std::max(iters_first - fin.it, 0) + 2, 24);
      steps_b = std::min((fin.switched ? std::max(iters_second - fin.it, 0) : iters_second) + 2, 24);
    }
    if (o->iterations) {
      o->iterations[0] = fin.done[0];
      o->iterations[1] = fin.done[1];
    }
    drain_timers(c);
    mark("results out");
    return ORBFE_OK;
  }
  struct HostScalars {
    double chi, maxdiag, lambda;
    int32_t ok, pad;
    double scale;
  };
  auto evaluate = [&](const double* d_info) {  // computeActiveErrors + activeRobustChi2
    launch_ba_edges(st, E, d_poses, d_points, d_ek, d_ep, (const double*)(b + o_meas), b + o_st, d_info, (const double*)(b + o_delta), prm,
                    (double*)(b + o_err), (double*)(b + o_chi2), (double*)(b + o_rho), nullptr, nullptr, b + o_depth);
    launch_lba_chi2_sum(st, E, (const double*)(b + o_chi2), (const double*)(b + o_rho), b + o_level, (double*)(b + o_last), d_sc);
  };
  auto read_scalars = [&](HostScalars& h) -> hipError_t {
    hipError_t e = hipMemcpyAsync(&h, d_sc, sizeof h, hipMemcpyDeviceToHost, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
  };
  auto stopped = [&]() { return stop_flag && *stop_flag; };
  StageTimer tm(c, ORBFE_STAGE_BA, st);
  auto optimize = [&](int iterations, int32_t& done) -> orbfe_status {  // SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve
    done = 0;
    if (E == 0) return ORBFE_OK;
    double lambda = 0, ni = 2;
    for (int it = 0; it < iterations; ++it) {
      if (stopped()) break;
      ++done;
      evaluate((const double*)(b + o_info_eff));
      launch_ba_system(st, NK, NP, E, d_poses, d_points, d_ek, d_ep, (const double*)(b + o_meas), b + o_st, (const double*)(b + o_info_eff),
                       (const double*)(b + o_delta), prm, b + o_fix, (const int32_t*)(b + o_pto), (const int32_t*)(b + o_pte),
                       (const int32_t*)(b + o_pso), (const int32_t*)(b + o_pse), (double*)(b + o_hpp), (double*)(b + o_bp),
                       (double*)(b + o_hll), (double*)(b + o_bl), (double*)(b + o_hpl));
      if (it == 0) launch_lba_maxdiag(st, NK, NP, (const double*)(b + o_hpp), (const double*)(b + o_hll), b + o_fix, d_sc + 1);
      HostScalars h;
      HIP_TRY(c, read_scalars(h));
      double current_chi = h.chi;
      if (it == 0) {
        lambda = 1e-5 * h.maxdiag;  // computeLambdaInit, tau = 1e-5
        ni = 2;
      }
      double rho = 0;
      int qmax = 0;
      do {
        HIP_TRY(c, hipMemcpyAsync(b + o_pose_bk, d_poses, (size_t)NK * 56, hipMemcpyDeviceToDevice, st));  // push()
        HIP_TRY(c, hipMemcpyAsync(b + o_pt_bk, d_points, (size_t)NP * 24, hipMemcpyDeviceToDevice, st));
        struct {
          double lambda;
          int32_t ok, pad;
        } upv = {lambda, 1, 0};
        HIP_TRY(c, hipMemcpyAsync(d_sc + 2, &upv, sizeof upv, hipMemcpyHostToDevice, st));
        launch_lba_solve(st, NK, NP, E, nf, (const int32_t*)(b + o_free), (const int32_t*)(b + o_slot), (const int32_t*)(b + o_pairoff),
                         (const int2*)(b + o_pairs), (const int32_t*)(b + o_pso), (const int32_t*)(b + o_pse), (const int32_t*)(b + o_pto),
                         (const int32_t*)(b + o_pte), d_ek, d_ep, b + o_fix, (const double*)(b + o_hpp), (const double*)(b + o_bp),
                         (const double*)(b + o_hll), (const double*)(b + o_bl), (const double*)(b + o_hpl), d_sc + 2, (double*)(b + o_dinv),
                         (double*)(b + o_w), (double*)(b + o_s), (double*)(b + o_rhs), (double*)(b + o_x), (int*)(d_sc + 3), d_poses, d_points,
                         (double*)(b + o_dxp), (double*)(b + o_dxl), d_sc + 4, (double*)(b + o_big));
        evaluate((const double*)(b + o_info_eff));
        HIP_TRY(c, read_scalars(h));
        const bool ok2 = h.ok != 0;
        const double temp_chi = ok2 ? h.chi : std::numeric_limits<double>::max();
        rho = (current_chi - temp_chi) / (h.scale + 1e-3);
        if (!ok2) rho = -1.0;  // the linear solver failed: the trial is rejected whatever its step looked like
        if (rho > 0 && std::isfinite(temp_chi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          current_chi = temp_chi;
        } else {
          lambda *= ni;
          ni *= 2;
          HIP_TRY(c, hipMemcpyAsync(d_poses, b + o_pose_bk, (size_t)NK * 56, hipMemcpyDeviceToDevice, st));  // pop()
          HIP_TRY(c, hipMemcpyAsync(d_points, b + o_pt_bk, (size_t)NP * 24, hipMemcpyDeviceToDevice, st));
          if (!std::isfinite(lambda)) break;
        }
        ++qmax;
      } while (rho < 0 && qmax < 10 && !stopped());
      if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) break;  // OptimizationAlgorithm::Terminate
    }
    return ORBFE_OK;
  };
  int32_t it1 = 0, it2 = 0;
  TRY(optimize(iters_first, it1));
  if (!stopped()) {
    // edge->chi2() is the chi2 of the last evaluated trial; isDepthPositive() reads the current estimates (Optimizer.cc:338-359)
    launch_ba_edges(st, E, d_poses, d_points, d_ek, d_ep, (const double*)(b + o_meas), b + o_st, (const double*)(b + o_info),
                    (const double*)(b + o_delta), prm, (double*)(b + o_err), (double*)(b + o_chi2), (double*)(b + o_rho), nullptr, nullptr,
                    b + o_depth);
    launch_lba_classify(st, E, (const double*)(b + o_last), b + o_depth, b + o_st, b + o_level, (double*)(b + o_info_eff),
                        (double*)(b + o_delta));
    TRY(optimize(iters_second, it2));
  }
  // final computeError() on every edge with the final estimates (Optimizer.cc:364-391)
  launch_ba_edges(st, E, d_poses, d_points, d_ek, d_ep, (const double*)(b + o_meas), b + o_st, (const double*)(b + o_info),
                  (const double*)(b + o_delta), prm, (double*)(b + o_err), (double*)(b + o_chi2), (double*)(b + o_rho), nullptr, nullptr,
                  b + o_depth);
  launch_lba_final(st, E, (const double*)(b + o_chi2), b + o_depth, b + o_st, b + o_bad);
  HIP_TRY(c, hipGetLastError());
  auto down = [&](void* dst, size_t o2, size_t bytes) -> hipError_t {
    return (bytes && dst) ? hipMemcpyAsync(dst, b + o2, bytes, hipMemcpyDeviceToHost, st) : hipSuccess;
  };
  HIP_TRY(c, down(o->poses, o_pose, (size_t)NK * 56));
  HIP_TRY(c, down(o->points, o_pt, (size_t)NP * 24));
  HIP_TRY(c, down(o->level, o_level, (size_t)E));
  HIP_TRY(c, down(o->chi2, o_chi2, (size_t)E * 8));
  HIP_TRY(c, down(o->bad, o_bad, (size_t)E));
  HIP_TRY(c, hipStreamSynchronize(st));
  if (o->iterations) {
    o->iterations[0] = it1;
    o->iterations[1] = it2;
  }
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

orbfe_status orbfe_pose_only_optimize(orbfe_ctx* c, int32_t n, const double* xw, const double* meas, const double* info, const float* sigma2,
                                      const double* pose_in, double fx, double fy, double cx, double cy, double bf, double* pose_out,
                                      uint8_t* inlier_out, int32_t* n_good) {
  ApiLock api_lk(c);
  if (!c || n < 0 || !pose_in || !pose_out || !n_good || (n && (!xw || !meas || !info || !sigma2)))
    return fail(c, ORBFE_EBADARG, "pose_only_optimize: NULL argument");
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const size_t N = (size_t)std::max(n, 1);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o2 = off;
    off += align_up(std::max<size_t>(bytes, 8), 256);
    return o2;
  };
  // inputs as ONE upload through the page-locked staging buffer, results as one download (five copies from pageable memory up and three down
  // were a fifth of the call)
  const size_t o_x = take(N * 24), o_m = take(N * 24), o_i = take(N * 8), o_s = take(N * 4), o_p = take(56), o_up_end = take(8),
               o_po = take(56), o_ng = take(8), o_in = take(N), o_out_end = take(8), o_e = take(N * 24), o_l = take(N), o_r = take(N);
  TRY(ensure_tmp(c, off));
  TRY(ensure_stage(c, std::max(o_up_end, o_out_end - o_po)));
  uint8_t* b = (uint8_t*)c->d_tmp;
  uint8_t* hs = c->main.h_stage;
  if (n) {
    std::memcpy(hs + o_x, xw, (size_t)n * 24);
    std::memcpy(hs + o_m, meas, (size_t)n * 24);
    std::memcpy(hs + o_i, info, (size_t)n * 8);
    std::memcpy(hs + o_s, sigma2, (size_t)n * 4);
  }
  std::memcpy(hs + o_p, pose_in, 56);
  HIP_TRY(c, hipMemcpyAsync(b, hs, o_up_end, hipMemcpyHostToDevice, c->stream));
  BaParamsDev prm = {fx, fy, cx, cy, bf};
  {
    StageTimer tm(c, ORBFE_STAGE_BA, c->stream);
    launch_pose_only(c->stream, n, (const double*)(b + o_x), (const double*)(b + o_m), (const double*)(b + o_i), (const float*)(b + o_s),
                     (const double*)(b + o_p), prm, (double)(float)std::sqrt(5.991), (double)(float)std::sqrt(7.815), (double*)(b + o_e),
                     b + o_l, b + o_r, b + o_in, (double*)(b + o_po), (int32_t*)(b + o_ng));
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(hs, b + o_po, (inlier_out && n ? o_in + (size_t)n : o_in) - o_po, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drain_timers(c);
  std::memcpy(pose_out, hs, 56);
  std::memcpy(n_good, hs + (o_ng - o_po), 4);
  if (inlier_out && n) std::memcpy(inlier_out, hs + (o_in - o_po), (size_t)n);
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
  HIP_TRY(c, hipSetDevice(c->device));
  TRY(join_stereo(c));
  const LevelDev& L = c->lv[level];
  int32_t n = 0;
  HIP_TRY(c, hipMemcpyAsync(&n, c->d_n_cand + (size_t)slot * c->cfg.n_levels + level, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  n = std::min<int32_t>(std::max(n, 0), (int32_t)L.cand_cap);
  std::vector<uint32_t> rec((size_t)std::max(n, 1));
  if (n) HIP_TRY(c, hipMemcpyAsync(rec.data(), c->d_scr_a + (size_t)slot * c->scratch_pitch + L.cand_base, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
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
