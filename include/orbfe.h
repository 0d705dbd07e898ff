/* orbfe.h -- C-ABI of the MI355X (gfx950) ORB front end + local-BA edge kernels.
 *
 * Drop-in boundary for ONE hot path of sunshanlu/ORB_SLAM2_ROS2 (all citations relative to
 * src/ORB_SLAM2/ in that repository):
 *
 *   orbfe_extract*            replaces  ORBExtractor::ORBExtractor + ::extract
 *                                       (include/ORB_SLAM2/ORBExtractor.h:107-116, src/ORBExtractor.cc:205-214, 499-508),
 *                                       i.e. initPyramid (:278-320), extractFast (:331-387), Quadtree (:19-192),
 *                                       getGrayCentroid (:465-487) and computeBRIEF (:397-456)
 *   orbfe_get_pyramid         replaces  ORBExtractor::getPyramid()            (ORBExtractor.h:113)
 *   orbfe_get_scale_factors   replaces  ORBExtractor::getScaledFactors()      (ORBExtractor.h:116)
 *   orbfe_stereo_match        replaces  ORBMatcher::searchByStereo            (ORBMatcher.h:38, src/ORBMatcher.cc:18-81)
 *                                       incl. createRowIndexDB (:915-932), getBestMatch (:967-990),
 *                                       pixelSADMatch/SAD/getPitch (:841-905, :1002-1011)
 *   orbfe_match_bruteforce    replaces  ORBMatcher::getBestMatch / descDistance loops (src/ORBMatcher.cc:941-990)
 *   orbfe_stereo_batch_device           the same extract(L) + extract(R) + searchByStereo for a batch of stereo
 *                                       pairs whose images are already resident in device memory
 *                                       (Frame::Frame stereo ctor + Frame::createStereo, src/Frame.cc:85-111,
 *                                       include/ORB_SLAM2/Frame.h:313-322)
 *   orbfe_ba_eval_edges       replaces  the g2o edge evaluation that Optimizer::OptimizeLocalMap /
 *                                       OptimizePoseOnly drive (src/Optimizer.cc:225-442, :33-203):
 *                                       EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ computeError + linearizeOplus
 *                                       + RobustKernelHuber
 *
 * Conventions: plain C, caller owns every host buffer, the context owns every device buffer.  No call
 * throws or aborts; every call returns an orbfe_status and orbfe_last_error() gives the text of the calling
 * THREAD's last failed call.  Threading: calls on distinct contexts are thread-safe.  Calls on ONE context are
 * serialised by the library (a per-context lock every entry point takes -- the reference's matchers are called
 * from the Tracking, LocalMapping and LoopClosing threads), with one exception made for the reference's call
 * pattern -- Frame::Frame runs the left and the right ORBExtractor::extract on two std::threads
 * (src/Frame.cc:100-105): orbfe_extract_slot calls on DIFFERENT slots of one context run concurrently with each
 * other and with the other entry points (each slot has its own stream, pinned staging and launch graph); the
 * caller only has to keep a slot call away from calls that read or rewrite THAT slot (a stereo match on it, a
 * batch call over it).  A context used from several threads serves them one at a time: give every thread role
 * its own context where they should not wait for each other (host/orbfe_shim.hpp: matcherContext, solverContext).
 *
 * There is NO CPU fallback behind this interface: if no HIP device is usable, orbfe_create fails.
 */
#ifndef ORBFE_H_
#define ORBFE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_ABI_VERSION 4
#define ORBFE_MAX_LEVELS 16
#define ORBFE_DESC_BYTES 32

typedef enum orbfe_status {
  ORBFE_OK = 0,
  ORBFE_EBADARG = 1,   /* NULL pointer, index out of range, unsupported parameter              */
  ORBFE_EBADSIZE = 2,  /* reference: ImageSizeError (src/ORBExtractor.cc:310-314) and friends   */
  ORBFE_EDEVICE = 3,   /* HIP runtime error / no gfx950 device                                  */
  ORBFE_ECAPACITY = 4, /* batch larger than max_images, output buffer too small                 */
  ORBFE_ENOMEM = 5
} orbfe_status;

/* Same memory layout as cv::KeyPoint {Point2f pt; float size, angle, response; int octave, class_id} (28 B). */
typedef struct orbfe_keypoint {
  float x, y;      /* level-0 coordinates: level coordinate * scale[octave] (src/ORBExtractor.cc:408-409) */
  float size;      /* 7.0 (cv::FAST)                                                                     */
  float angle;     /* signed degrees in (-180,180] (src/ORBExtractor.cc:407)                             */
  float response;  /* FAST score                                                                         */
  int32_t octave;
  int32_t class_id; /* -1 */
} orbfe_keypoint;

typedef struct orbfe_config {
  int32_t width, height;       /* level-0 image size (e.g. 1241x376 KITTI, 640x480 TUM); each <= 4096    */
  int32_t n_features;          /* ORBExtractor.nFeatures.  No limit besides 65535 (indices of the stereo row table are 16 bit):
                                  a level whose quota exceeds ~2700 keypoints (nFeatures > ~12000 at 8 levels x 1.2) keeps its
                                  quadtree node table in global memory instead of one CU's LDS -- exact, slower.
                                  (One more deviation, harmless: the quadtree's split loop is capped at 80 N + 1024 steps for N
                                  candidates; the reference's degenerate case -- fewer candidates than the quota, quirk Q3 -- ends
                                  by itself after ~50 halvings per point, far below the cap.)                              */
  int32_t n_levels;            /* ORBExtractor.nLevels (<= ORBFE_MAX_LEVELS)                            */
  float scale_factor;          /* ORBExtractor.scaleFactor                                              */
  int32_t fast_hi, fast_lo;    /* ORBExtractor.iniThFAST / minThFAST                                    */
  const int8_t* brief_pairs;   /* 256 x {x1,y1,x2,y2}; NULL = the embedded table (config/brief_template.txt) */
  int32_t blur_variant;        /* cv::GaussianBlur 7x7 sigma 2, 8.8 fixed-point taps (un-vendored OpenCV):
                                  0: {18,34,48,56,48,34,18} (OpenCV >= 4.3, default)  1: {18,34,49,55,49,34,18}          */
  int32_t gray_variant;        /* cv::cvtColor RGB/BGR -> gray fixed-point coefficients (orbfe_extract_color only):
                                  0: 14-bit R 4899 G 9617 B 1868, (sum + 2^13) >> 14 (default)
                                  1: 15-bit R 9798 G 19235 B 3735, (sum + 2^14) >> 15 (newer OpenCV 4.x builds)           */
  int32_t device_id;           /* HIP device ordinal                                                    */
  int32_t max_images;          /* image slots held on the device (a stereo pair uses two)               */
  void* stream;                /* optional hipStream_t to run on; NULL = context-owned stream           */
} orbfe_config;

typedef struct orbfe_ctx orbfe_ctx;

/* Per-level geometry as the reference derives it (src/ORBExtractor.cc:283-317, 334-343). */
typedef struct orbfe_level_info {
  int32_t width, height;
  float scale;                 /* mvfScaledFactors[level]                                               */
  int32_t quota;               /* mvnFeatures[level]                                                    */
  int32_t grid_cols, grid_rows, cell_w, cell_h; /* FAST cell grid of extractFast                        */
} orbfe_level_info;

int orbfe_abi_version(void);
orbfe_status orbfe_create(const orbfe_config* cfg, orbfe_ctx** out);
void orbfe_destroy(orbfe_ctx* ctx);
const char* orbfe_last_error(const orbfe_ctx* ctx); /* ctx may be NULL: error of the last failed orbfe_create */

orbfe_status orbfe_get_level_info(const orbfe_ctx* ctx, int32_t level, orbfe_level_info* out);
orbfe_status orbfe_get_scale_factors(const orbfe_ctx* ctx, float* out, int32_t n);
/* Keypoints one image can yield = the stride of every per-image array of this API ("[n_features]" below means this number).
 * It equals n_features for every configuration the reference ships; it is larger only where the reference itself returns more
 * than nFeatures keypoints: its per-level quotas are rounded level by level and only the last level absorbs the difference
 * (src/ORBExtractor.cc:292-300), so e.g. nFeatures = 7 at 8 levels x 1.2 asks for 2 keypoints on each of seven levels.          */
int32_t orbfe_get_capacity(const orbfe_ctx* ctx);

/* ---- extraction (host buffers) ---------------------------------------------------------------
 * One image -> slot 0.  kps has room for n_features entries, desc for n_features*32 bytes.          */
orbfe_status orbfe_extract(orbfe_ctx* ctx, const uint8_t* img, size_t stride_bytes, orbfe_keypoint* kps, uint8_t* desc,
                           int32_t* n_out);
/* n_img images -> slots 0..n_img-1; kps/desc are [n_img][n_features] arrays, n_out[n_img].           */
orbfe_status orbfe_extract_batch(orbfe_ctx* ctx, int32_t n_img, const uint8_t* const* imgs, size_t stride_bytes,
                                 orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out);
/* One image -> slot `slot` (0 <= slot < max_images), otherwise like orbfe_extract.  The drop-in for ORBExtractor::extract as
 * Frame::Frame calls it (src/Frame.cc:100-105): two extractor objects, two threads -- give each object its own slot; the results
 * stay resident in the slot for orbfe_stereo_match(ctx, slot_left, slot_right, ..) / orbfe_get_pyramid / orbfe_search_in_area.   */
orbfe_status orbfe_extract_slot(orbfe_ctx* ctx, int32_t slot, const uint8_t* img, size_t stride_bytes, orbfe_keypoint* kps,
                                uint8_t* desc, int32_t* n_out);
/* orbfe_extract_slot in two halves (ABI 4).  _begin stages the image and enqueues the whole extraction on the slot's lane, then RETURNS;
 * _end waits for it and delivers what orbfe_extract_slot delivers (kps / desc may be NULL: only the count; all NULL: just drain).  For the
 * reference's own construction order: `ORBExtractor::ORBExtractor(image, ...)` runs on the constructing thread BEFORE Frame::Frame starts
 * its two extract() threads (src/ORB_SLAM2/src/Frame.cc:91-105) and already does the pyramid there (src/ORBExtractor.cc:205-214) -- a
 * constructor that calls _begin lets the device work while the std::threads are being created (30 - 50 us each, p99 80 - 130 us on the
 * GPU boxes of the pool), and extract() is then only _end.  The image must stay readable until _begin returns; between _begin and _end
 * the slot accepts no other call (ORBFE_EBADARG), and every _begin must be followed by an _end, from any thread.                       */
orbfe_status orbfe_extract_slot_begin(orbfe_ctx* ctx, int32_t slot, const uint8_t* img, size_t stride_bytes);
orbfe_status orbfe_extract_slot_end(orbfe_ctx* ctx, int32_t slot, orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out);
/* The same for n_img images into the consecutive slots slot .. slot + n_img - 1 as ONE launch sequence on slot's lane (arrays as
 * orbfe_extract_batch): both eyes of a stereo frame when one caller holds both images, about half the time of two slot calls.  (The C++
 * mirror keeps one orbfe_extract_slot per extract() thread, src/Frame.cc:100-105: pairing the two threads into one such call was
 * measured and dropped)  The call holds the lanes of all the slots it writes; a slot call on any of them waits.
 * Slot calls run concurrently with each other; they order themselves behind work the locked entry points left on the context
 * stream, which takes the context's API lock for a moment -- a long call that holds it (orbfe_ba_local_optimize holds it for the whole
 * optimisation) delays them: give the back end its own context.
 * LIFETIME of a slot's device-resident results: until the next extraction INTO THAT SLOT.  The drop-in extractor rotates each eye over
 * four slots (host/orbfe_dropin.hpp, ContextPool::kSlots), so a Frame's device-side features stay valid for the four extractions that
 * follow it -- orbfe_stereo_match / orbfe_search_in_area / orbfe_get_pyramid on an older Frame read a newer frame's data.            */
orbfe_status orbfe_extract_slots(orbfe_ctx* ctx, int32_t slot, int32_t n_img, const uint8_t* const* imgs, size_t stride_bytes,
                                 orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out);
/* Copy one pyramid level of a slot to the host (tight rows).  blurred=0: the planes getPyramid() returns;
 * blurred=1: the Gaussian-blurred planes BRIEF samples (mvBriefMat).  dst needs width*height bytes.   */
orbfe_status orbfe_get_pyramid(orbfe_ctx* ctx, int32_t slot, int32_t level, int32_t blurred, uint8_t* dst);

/* ---- stereo matching over the device-resident results of the last extract --------------------
 * right_u / depth: n_features doubles each, -1 where unmatched (mvFeatsRightU / mvDepths).
 * best_right / best_dist (nullable, n_features int32): getBestMatch result before the thresholds
 * (-1 where the candidate list was empty).                                                           */
orbfe_status orbfe_stereo_match(orbfe_ctx* ctx, int32_t slot_left, int32_t slot_right, float fx, float bf, double* right_u,
                                double* depth, int32_t* n_matches, int32_t* best_right, int32_t* best_dist);

/* ---- one stereo frame, host images in, everything out: ONE call -------------------------------
 * The device work of Frame::createStereo (include/ORB_SLAM2/Frame.h:313-323: the Frame constructor, src/ORB_SLAM2/src/Frame.cc:85-110,
 * with ORBExtractor::extract on the left and the right image, then ORBMatcher::searchByStereo, src/ORB_SLAM2/src/ORBMatcher.cc:18):
 * left -> slot 0, right -> slot 1, their stereo match behind them in the same launch
 * sequence, one synchronisation.  kps / desc / n_out as orbfe_extract_batch with two images ([2][n_features]), right_u / depth as
 * orbfe_stereo_match.  Results are those of orbfe_extract_batch([left, right]) + orbfe_stereo_match(0, 1): the same kernels in the
 * same order; what the call saves is the second call's launch, copy and wake-up (tests/test_gpu_frame_stereo.py compares them).   */
orbfe_status orbfe_frame_stereo(orbfe_ctx* ctx, const uint8_t* left, const uint8_t* right, size_t stride_bytes, float fx, float bf,
                                orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out, double* right_u, double* depth,
                                int32_t* n_matches);
/* The same into the slot pair (slot_left, slot_left + 1), slot_left even, on slot_left's lane -- concurrent with slot calls on other
 * slots, lifetime of the device-resident results as for orbfe_extract_slot (the drop-in's createStereo adapter rotates over the pairs
 * of its context: host/orbfe_dropin.hpp).                                                                                          */
orbfe_status orbfe_frame_stereo_slots(orbfe_ctx* ctx, int32_t slot_left, const uint8_t* left, const uint8_t* right, size_t stride_bytes,
                                      float fx, float bf, orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out, double* right_u,
                                      double* depth, int32_t* n_matches);

/* ---- whole stereo pairs, images already in device memory --------------------------------------
 * d_left/d_right: device pointers to n_pairs images each, image p at base + p*image_pitch_bytes,
 * rows stride_bytes apart.  Pair p uses slots 2p (left) and 2p+1 (right).  Asynchronous on the
 * context stream; results stay on the device until fetched.  The images must be complete when the call
 * is made and stay unchanged until the call's first kernels have run (the resize reads them directly and
 * writes level 0 of the pyramid from them): do not overwrite them before orbfe_sync or a fetch of this
 * batch's results has returned (the host-image stream entry points manage their own ring of buffers). */
orbfe_status orbfe_stereo_batch_device(orbfe_ctx* ctx, const uint8_t* d_left, const uint8_t* d_right, size_t stride_bytes,
                                       size_t image_pitch_bytes, int32_t n_pairs, float fx, float bf);
orbfe_status orbfe_sync(orbfe_ctx* ctx);

/* ---- whole stereo pairs from HOST memory, copies overlapped with compute ------------------------
 * The loop of example/Stereo/KittiStereo.cc:28-37 (imread x2 -> Frame::createStereo) as a three-stage pipeline over batches: while
 * batch k is computed, batch k+1 is uploaded and the packed results of batch k-1 are downloaded (three input and three result buffers
 * on the device, one copy stream per direction).  left / right: n_pairs images each, image p at base + p * image_pitch_bytes, rows
 * stride_bytes apart.  orbfe_stream_submit returns at once with a ticket; the images must stay untouched, and `out` is not
 * valid, until orbfe_stream_wait(ticket) has returned.  At most THREE tickets may be outstanding (a fourth submit waits for the oldest).  Host memory from
 * orbfe_host_alloc (page-locked) makes both copies asynchronous DMA; ordinary memory works but the runtime stages it and the
 * overlap is lost.  The slots of the context hold the results of the newest submitted batch (pair p in slots 2p / 2p+1).
 * `out` arrays: kps [2 n_pairs][n_features], desc [2 n_pairs][n_features][32], counts [2 n_pairs] (slot order); right_u / depth
 * [n_pairs][n_features], n_matches [n_pairs]; entries past a slot's count are stale; any pointer may be NULL.                      */
typedef struct orbfe_batch_results {
  orbfe_keypoint* kps;
  uint8_t* desc;
  int32_t* counts;
  double* right_u;
  double* depth;
  int32_t* n_matches;
} orbfe_batch_results;
void* orbfe_host_alloc(size_t bytes); /* page-locked host memory (NULL on failure); release with orbfe_host_free */
/* ... placed on the NUMA node of HIP device `device_id` (< 0: the calling thread's current device, = orbfe_host_alloc): what a
 * multi-rank job calls with its own device, so that ranks which never called hipSetDevice do not all pin to GPU 0's node          */
void* orbfe_host_alloc_on(int32_t device_id, size_t bytes);
void orbfe_host_free(void* p);
/* Hardware queues the streaming entry points want (GPU_MAX_HW_QUEUES, read by the HIP runtime at the process's FIRST HIP call: the
 * process exports it, a library cannot).  orbfe_stream_submit prints one line on stderr when the variable is unset or below 8; the
 * one-frame-at-a-time entry points are faster with the runtime's default and never ask for it.                                   */
int32_t orbfe_recommended_hw_queues(void);
orbfe_status orbfe_stream_submit(orbfe_ctx* ctx, const uint8_t* left, const uint8_t* right, size_t stride_bytes,
                                 size_t image_pitch_bytes, int32_t n_pairs, float fx, float bf, const orbfe_batch_results* out,
                                 int64_t* ticket);
orbfe_status orbfe_stream_wait(orbfe_ctx* ctx, int64_t ticket);
/* Device pointers of the packed results of `ticket` (same arrays and strides as orbfe_batch_results), for consumers that stay on
 * the device (a gather over RCCL): complete once orbfe_stream_wait(ticket) has returned, valid until ticket + 3 is submitted.      */
orbfe_status orbfe_stream_device_results(orbfe_ctx* ctx, int64_t ticket, int32_t n_pairs, const void** d_kps, const void** d_desc,
                                         const void** d_counts, const void** d_right_u, const void** d_depth, const void** d_nmatch);
/* Fetch the results of slot (keypoints/descriptors) and, for a left slot, of its pair. Any pointer may be NULL. */
orbfe_status orbfe_fetch_features(orbfe_ctx* ctx, int32_t slot, orbfe_keypoint* kps, uint8_t* desc, int32_t* n_out);
orbfe_status orbfe_fetch_stereo(orbfe_ctx* ctx, int32_t pair, double* right_u, double* depth, int32_t* n_matches,
                                int32_t* best_right, int32_t* best_dist);
/* The packed results of slots [slot0, slot0 + n_slots) / pairs [pair0, pair0 + n_pairs) in one copy per array, full strides:
 * kps [n_slots][n_features], desc [n_slots][n_features][32], counts [n_slots]; right_u / depth [n_pairs][n_features], n_matches
 * [n_pairs].  Entries past a slot's count are stale.  Any pointer may be NULL.                                                */
orbfe_status orbfe_fetch_batch(orbfe_ctx* ctx, int32_t slot0, int32_t n_slots, orbfe_keypoint* kps, uint8_t* desc, int32_t* counts);
orbfe_status orbfe_fetch_stereo_batch(orbfe_ctx* ctx, int32_t pair0, int32_t n_pairs, double* right_u, double* depth,
                                      int32_t* n_matches);
/* Frame records of a ticket for the sequence-level gather: one fixed-size record per pair, what Frame::createStereo leaves
 * for the tracker -- int32 n_keypoints | int32 n_matches | 8 bytes pad | LEFT keypoints [n_features x 28 B] | LEFT descriptors
 * [n_features x 32 B] | right_u [n_features] f64 | depth [n_features] f64, entries past the count zeroed -- written to caller-provided
 * DEVICE memory (n_pairs * orbfe_record_bytes(ctx) bytes; e.g. a torch tensor handed to an RCCL gather).  Returns when the records
 * are complete.  The ticket must be live (submitted, and ticket + 3 not yet submitted).                                            */
size_t orbfe_record_bytes(const orbfe_ctx* ctx);
orbfe_status orbfe_stream_pack_records(orbfe_ctx* ctx, int64_t ticket, int32_t n_pairs, void* d_records);
/* Device pointers of the packed per-slot results, for gathers that never touch the host
 * (keypoints [max_images][n_features] orbfe_keypoint, descriptors [max_images][n_features][32],
 * counts [max_images] int32, right_u/depth [max_images/2][n_features] double).                       */
orbfe_status orbfe_device_results(orbfe_ctx* ctx, const void** d_kps, const void** d_desc, const void** d_counts,
                                  const void** d_right_u, const void** d_depth, const void** d_nmatch);

/* ---- brute-force Hamming-256 best / second-best (host buffers) --------------------------------
 * q: nq*32 bytes, t: nt*32 bytes.  Candidate lists in CSR form: query i scans train indices
 * cand_idx[cand_offsets[i] .. cand_offsets[i+1]) IN THAT ORDER (order matters, quirk Q6);
 * cand_offsets == NULL means "all nt, ascending".  Outputs per query: best_idx (-1 if no candidate),
 * best_dist, second_dist (INT32_MAX if none) exactly as ORBMatcher::getBestMatch computes them.       */
orbfe_status orbfe_match_bruteforce(orbfe_ctx* ctx, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt,
                                    const uint32_t* cand_offsets, const uint32_t* cand_idx, int32_t* best_idx,
                                    int32_t* best_dist, int32_t* second_dist);

/* ---- local-BA edge evaluation (fp64) ----------------------------------------------------------- */
typedef struct orbfe_ba_problem {
  int32_t n_poses, n_points, n_edges;
  const double* poses;       /* [n_poses][7]  qx,qy,qz,qw, tx,ty,tz  (g2o::SE3Quat of VertexSE3Expmap)   */
  const double* points;      /* [n_points][3] (VertexPointXYZ)                                          */
  const int32_t* edge_pose;  /* [n_edges]                                                               */
  const int32_t* edge_point; /* [n_edges]                                                               */
  const double* meas;        /* [n_edges][3]  u, v, u_right (u_right ignored for mono edges)            */
  const uint8_t* is_stereo;  /* [n_edges]     1: EdgeStereoSE3ProjectXYZ, 0: EdgeSE3ProjectXYZ           */
  const double* info;        /* [n_edges]     information = info * I                                    */
  const double* huber_delta; /* [n_edges]     <= 0: no robust kernel                                    */
  double fx, fy, cx, cy, bf;
} orbfe_ba_problem;

typedef struct orbfe_ba_edge_out {
  double* error;    /* [n_edges][3]  (3rd component 0 for mono)                                         */
  double* chi2;     /* [n_edges]                                                                        */
  double* rho;      /* [n_edges][2]  robustified chi2 rho(chi2), weight rho'(chi2)                      */
  double* j_point;  /* [n_edges][3][3] d e / d point (rows 0-1 used for mono), nullable                 */
  double* j_pose;   /* [n_edges][3][6] d e / d (omega, upsilon), nullable                               */
  uint8_t* depth_positive; /* [n_edges] isDepthPositive(), nullable                                     */
} orbfe_ba_edge_out;

/* DIAGNOSTIC entry points (this one and orbfe_ba_build_system): they return the per-edge Jacobians / the normal-equation blocks of ONE
 * linearisation to the host -- 4.8 MB over PCIe for the 15 597 edges of BASELINE config 5, which is why the host-array call takes about
 * as long as one CPU core needs for the arithmetic (0.42 vs 0.46 ms; the kernel itself 10 us).  They exist so that the edge formulas and
 * the quadratic form can be checked against the oracle; the optimisers (orbfe_ba_local_optimize, orbfe_pose_only_optimize,
 * orbfe_track_local_map) evaluate, build and solve on the device and move only poses, points and flags.                              */
orbfe_status orbfe_ba_eval_edges(orbfe_ctx* ctx, const orbfe_ba_problem* prob, const orbfe_ba_edge_out* out);

/* Normal-equation blocks of one linearisation, laid out like g2o's BlockSolver_6_3 (src/Optimizer.h:49-54): replaces
 * BaseBinaryEdge::constructQuadraticForm over all edges (what optimizer.optimize() runs per LM iteration under
 * Optimizer::OptimizeLocalMap, src/Optimizer.cc:336,361).  W = rho'(chi2) * info * I per edge (Huber as configured);
 * pose_fixed[k] != 0 => vertex k gets no blocks (setFixed, Optimizer.cc:248,288); points are the marginalised set.  */
typedef struct orbfe_ba_system_out {
  double* Hpp;  /* [n_poses][6][6]  sum B^T W B            (zero for fixed poses)                         */
  double* bp;   /* [n_poses][6]     - sum B^T W e                                                          */
  double* Hll;  /* [n_points][3][3] sum A^T W A                                                            */
  double* bl;   /* [n_points][3]    - sum A^T W e                                                          */
  double* Hpl;  /* [n_edges][6][3]  B^T W A of the edge (zero if its pose is fixed), nullable              */
} orbfe_ba_system_out;
orbfe_status orbfe_ba_build_system(orbfe_ctx* ctx, const orbfe_ba_problem* prob, const uint8_t* pose_fixed /*[n_poses], nullable*/,
                                   const orbfe_ba_system_out* out);

/* The g2o part of Optimizer::OptimizeLocalMap (include/ORB_SLAM2/Optimizer.h:69, src/Optimizer.cc:336-391) on the graph the
 * caller built (:232-330): optimize(iters_first = 5) with the problem's Huber deltas, then every edge with chi2 > 5.991 (mono) /
 * 7.815 (stereo) or non-positive depth goes to level 1 and ALL robust kernels are dropped (:338-359), optimize(iters_second = 10)
 * on level 0, final computeError() + the same test on every edge (:364-391).  BlockSolver_6_3 + OptimizationAlgorithmLevenberg
 * semantics (lambda0 = 1e-5 max diag, gain ratio, <= 10 trials per iteration, points marginalised by Schur complement); the
 * reduced system is factorised densely (6x6-blocked Cholesky: by one workgroup out of LDS up to 100 non-fixed keyframes, by a multi-workgroup
 * path beyond -- no bound on their number, as Optimizer.cc:232 has none).  stop_flag (nullable) is polled like g2o's
 * forceStopFlag (Optimizer.cc:230): it points at ONE BYTE, the reference's `bool mbAbortBA` (include/ORB_SLAM2/LocalMapping.h:185) passed as
 * `bool& isStop` (Optimizer.h:69) -- only that byte is read, non-zero = stop.  The map bookkeeping of :393-441 stays with the caller.                               */
typedef struct orbfe_ba_optimize_out {
  double* poses;         /* [n_poses][7]  optimised estimates (fixed poses unchanged)                          */
  double* points;        /* [n_points][3]                                                                     */
  uint8_t* level;        /* [n_edges] 1: excluded from the second round (setLevel(1)), nullable                 */
  double* chi2;          /* [n_edges] chi2 at the final estimates, nullable                                    */
  uint8_t* bad;          /* [n_edges] final chi2 / depth test failed (the vToProcess candidates), nullable       */
  int32_t* iterations;   /* [2] Levenberg iterations run by the two optimize() calls, nullable                  */
} orbfe_ba_optimize_out;
orbfe_status orbfe_ba_local_optimize(orbfe_ctx* ctx, const orbfe_ba_problem* prob, const uint8_t* pose_fixed /*[n_poses], nullable*/,
                                     int32_t iters_first, int32_t iters_second, const volatile uint8_t* stop_flag,
                                     const orbfe_ba_optimize_out* out);

/* ---- grid-guided matching against the features of one slot ---------------------------------------------------------
 * Replaces VirtualFrame::initGrid + findFeaturesInArea (src/Frame.cc:53-69, 286-311) + ORBMatcher::getBestMatch
 * (src/ORBMatcher.cc:967-990), the core of the guided searches (ORBMatcher::searchByProjection, src/ORBMatcher.cc:265-347 and
 * :561-612): query i searches the 64x48-px grid cells overlapping [x-r, x+r] x [y-r, y+r] around qxy[i] with r = radius[i] (the
 * caller multiplies by getScaledFactor2(octave), Frame.cc:289), cells rows-outer / columns-inner, a cell's features in index
 * order, keeps octaves in [min_level[i], max_level[i]] and features with exclude[idx] == 0 (nullable: e.g. "already has a
 * map point", ORBMatcher.cc:322-332), and applies the reference's order-dependent best / second-best scan.  Outputs as
 * orbfe_match_bruteforce plus the candidate count; best_idx = -1 when the list is empty.  Cell indices are clamped to the
 * grid (the reference indexes one column past the grid when the box touches x == width and width is a multiple of 64).      */
orbfe_status orbfe_search_in_area(orbfe_ctx* ctx, int32_t slot, int32_t nq, const float* qxy /*[nq][2]*/, const float* radius,
                                  const int8_t* min_level, const int8_t* max_level, const uint8_t* q_desc /*[nq][32]*/,
                                  const uint8_t* exclude /*[n_features], nullable*/, int32_t* best_idx, int32_t* best_dist,
                                  int32_t* second_dist, int32_t* n_cand);

/* The same search against a feature set the CALLER supplies -- a KeyFrame's mvFeatsLeft / descriptors: keyframes are not resident in a
 * slot.  This is the core of the remaining guided searches, ORBMatcher::searchBySim3 x2 (src/ORBMatcher.cc:370-559, via
 * KeyFrame::findFeaturesInArea), and of searchByProjection / fuse when the target is a KeyFrame (:561-734).  t_kps: x, y, octave are
 * read; exclude: [nt] flags, nullable; the grid is the context's width x height.                                                  */
orbfe_status orbfe_search_in_area_features(orbfe_ctx* ctx, int32_t nt, const orbfe_keypoint* t_kps, const uint8_t* t_desc /*[nt][32]*/,
                                           int32_t nq, const float* qxy /*[nq][2]*/, const float* radius, const int8_t* min_level,
                                           const int8_t* max_level, const uint8_t* q_desc /*[nq][32]*/, const uint8_t* exclude,
                                           int32_t* best_idx, int32_t* best_dist, int32_t* second_dist, int32_t* n_cand);
/* ... with the two things the frame-level searchByProjection (src/ORBMatcher.cc:265-347) needs besides: `bounds` = {min_u, max_u, min_v,
 * max_v}, the target frame's undistorted image bounds (VirtualFrame::mfMinU..mfMaxV, include/ORB_SLAM2/Frame.h:33-43): initGrid sizes the
 * grid from them and findFeaturesInArea clips the search box at (int)max_u / (int)max_v (src/Frame.cc:55-56, :289-293); NULL = the image,
 * i.e. a camera without distortion.  `excluded_hits` [nt], nullable: for every EXCLUDED feature how many queries had it inside their
 * window and octave range -- the reference calls MapPoint::addMatchInTrack once per such occurrence while it drops the feature from
 * the candidate list (ORBMatcher.cc:321-331); entries of features that are not excluded are 0.                                        */
orbfe_status orbfe_search_in_area_features_ex(orbfe_ctx* ctx, int32_t nt, const orbfe_keypoint* t_kps, const uint8_t* t_desc /*[nt][32]*/,
                                              const float* bounds /*[4], nullable*/, int32_t nq, const float* qxy /*[nq][2]*/,
                                              const float* radius, const int8_t* min_level, const int8_t* max_level,
                                              const uint8_t* q_desc /*[nq][32]*/, const uint8_t* exclude, int32_t* best_idx,
                                              int32_t* best_dist, int32_t* second_dist, int32_t* n_cand, int32_t* excluded_hits);

/* ---- pose-only optimisation of one frame (fp64), entirely on the device -------------------------------------------
 * Replaces the g2o part of Optimizer::OptimizePoseOnly (include/ORB_SLAM2/Optimizer.h:72, src/Optimizer.cc:33-178): one SE3 pose
 * vertex, one unary edge per observed map point -- EdgeSE3ProjectXYZOnlyPose when u_right < 0 (Optimizer.cc:77), else
 * EdgeStereoSE3ProjectXYZOnlyPose -- information info[i]*I (the caller passes getScaledFactorInv2(octave), :85,:106), Huber
 * sqrt(5.991)/sqrt(7.815); 4 rounds x optimize(10) from the initial pose with the re-classification chi2 > 5.991*sigma2[i] /
 * 7.815*sigma2[i] (sigma2 = getScaledFactor2(octave), :136,:157) and the kernels dropped in the third round (:149,:170).
 * Outputs the optimised pose (qx qy qz qw tx ty tz), the inlier flag of every edge and `edges - nBad` of the last round.
 * The projection post-check of Optimizer.cc:180-190 (Frame::project2UV with the frame's float pose) stays with the caller.  */
orbfe_status orbfe_pose_only_optimize(orbfe_ctx* ctx, int32_t n, const double* xw /*[n][3]*/, const double* meas /*[n][3] u v uR*/,
                                      const double* info /*[n]*/, const float* sigma2 /*[n]*/, const double* pose_in /*[7]*/,
                                      double fx, double fy, double cx, double cy, double bf, double* pose_out /*[7]*/,
                                      uint8_t* inlier_out /*[n], nullable*/, int32_t* n_good);

/* ---- frame-level glue either side of the ORB path -----------------------------------------------------
 * orbfe_extract_color: Tracking::grabFrame's cv::cvtColor(COLOR_RGB2GRAY / COLOR_BGR2GRAY) (src/Tracking.cc:55-68) fused in front of
 * orbfe_extract (slot 0): img is 8-bit 3-channel interleaved, color_order 1 = RGB, 2 = BGR (the reference's Camera.Color values);
 * the grey image becomes level 0 of the pyramid (orbfe_get_pyramid(ctx, 0, 0, 0, ..) returns it).
 * orbfe_frame_rgbd: for the keypoints of `slot`, Camera::undistortPoints (src/Camera.cc:29-39; no-op when k1 == 0) IN PLACE on the
 * device-resident keypoints -- so the feature grid of orbfe_search_in_area sees undistorted positions as VirtualFrame::initGrid does
 * (Frame.cc:108,148) -- and, when depth != NULL, the RGB-D tail of Frame::Frame (src/Frame.cc:136-158): depth read at the
 * DISTORTED keypoint with truncated indices, divided by depth_scale in float, depth_out = d and right_u_out = x_undist - bf / d
 * where d > 0, else -1.  depth_type 0: uint16 image, 1: float image; rows depth_stride_bytes apart (host memory).
 * Outputs [n_features] (entries past the keypoint count are -1), any may be NULL.  Stereo rigs are taken as rectified (k1 == 0,
 * the reference's only stereo configuration): orbfe_stereo_match does not undistort.                                          */
typedef struct orbfe_camera {
  float fx, fy, cx, cy;      /* Camera::mK as the reference stores it (float)                    */
  float k1, k2, p1, p2, k3;  /* Camera::mDistCoeff                                                */
  float bf;                  /* Camera::mfBf                                                      */
} orbfe_camera;
orbfe_status orbfe_extract_color(orbfe_ctx* ctx, const uint8_t* img, size_t stride_bytes, int32_t color_order, orbfe_keypoint* kps,
                                 uint8_t* desc, int32_t* n_out);
/* The device work of Frame::createRGBD (include/ORB_SLAM2/Frame.h:326-331) as ONE call: cv::cvtColor of Tracking::grabFrame when
 * color_order is 1 (RGB) or 2 (BGR) -- 0: the image is gray already -- (src/Tracking.cc:55-68), the extraction into `slot` (the RGB-D
 * Frame constructor, src/ORB_SLAM2/src/Frame.cc:125-135), then the constructor's tail: Camera::undistortPoints and the depth / rightU
 * lookup (:136-158).  One launch sequence on the slot's lane, one synchronisation; the depth image is not uploaded (the last kernel reads
 * one value per keypoint from the page-locked staging copy).  Results as orbfe_extract_color / orbfe_extract_slot followed by
 * orbfe_frame_rgbd (tests/test_frame_glue.py compares them): kps_undistorted / desc [n_features], depth_out / right_u_out [n_features]
 * (-1 where there is no depth), nullable.                                                                                          */
orbfe_status orbfe_frame_rgbd_image(orbfe_ctx* ctx, int32_t slot, const uint8_t* img, size_t stride_bytes, int32_t color_order,
                                    const orbfe_camera* cam, const void* depth, int32_t depth_type, size_t depth_stride_bytes,
                                    float depth_scale, orbfe_keypoint* kps_undistorted, uint8_t* desc, int32_t* n_out, double* depth_out,
                                    double* right_u_out);
orbfe_status orbfe_frame_rgbd(orbfe_ctx* ctx, int32_t slot, const orbfe_camera* cam, const void* depth, int32_t depth_type,
                              size_t depth_stride_bytes, float depth_scale, orbfe_keypoint* kps_undistorted, double* depth_out,
                              double* right_u_out);

/* MapPoint::isInVision + MapPoint::predictLevel (src/MapPoint.cc:141-201) for n map points against one frame -- the per-point
 * preamble of ORBMatcher::searchByProjection(frame, mapPoints, ...) (src/ORBMatcher.cc:575-580), whose outputs feed
 * orbfe_search_in_area.  pose: Rcw row-major + tcw as the reference's float cv::Mat; bounds: VirtualFrame::mfMinU, mfMaxU, mfMinV,
 * mfMaxV (Frame.h:259-264).  Outputs per point: projection uv, camera-centre distance, cosTheta, predicted level (clamped to
 * [0, 7] like the reference) and visible = isInVision's return value; a point that fails a test keeps visible = 0 and the
 * values computed up to that test.  Float arithmetic in the reference's order (documented in csrc/k_guided.hip).              */
typedef struct orbfe_frame_pose {
  float Rcw[9], tcw[3];
  float min_u, max_u, min_v, max_v;
} orbfe_frame_pose;
orbfe_status orbfe_project_map_points(orbfe_ctx* ctx, int32_t n, const float* pos /*[n][3]*/, const float* view_dir /*[n][3]*/,
                                      const float* max_dist /*[n]*/, const float* min_dist /*[n]*/, const orbfe_frame_pose* pose,
                                      const orbfe_camera* cam, float* uv /*[n][2]*/, float* distance /*[n]*/, float* cos_theta /*[n]*/,
                                      int8_t* level /*[n]*/, uint8_t* visible /*[n]*/);

/* ---- the tracking chain of one frame as ONE call -------------------------------------------------------------------------
 * Tracking::trackLocalMap (src/Tracking.cc:641-675) calls ORBMatcher::searchByProjection(frame, local map points, th)
 * (src/ORBMatcher.cc:561-612: MapPoint::isInVision + predictLevel per point, findFeaturesInArea + getBestMatch, then the assignment
 * in map-point order -- a feature that holds a good map point keeps it, a free one goes to the first point whose best match it is)
 * and then Optimizer::OptimizePoseOnly(frame) (src/Optimizer.cc:33-178) on what the frame holds.  Called one by one
 * (orbfe_project_map_points, orbfe_search_in_area, orbfe_pose_only_optimize) that is three host -> device -> host round trips with the
 * map points and the frame's features marshalled each time; here the frame's features are the device-resident results of `slot`, the
 * map points go up once, and the projections, windows, candidate lists, assignment and edge list stay on the device.
 * flags[i]: bit 0 = !isBad && isInMap, bit 1 = !isBad, bit 2 = member of the list searchByProjection walks (a point the frame holds
 * from an earlier stage but that is not in the local map carries bits 0-1 only: not searched, still an edge).
 * Outputs: assigned[f] = index of the map point feature f holds after the search (-1: none); n_matches = searchByProjection's return
 * value; when n_matches >= min_matches (else n_edges = -1, pose_out = pose_se3, no inliers -- trackLocalMap returns false there):
 * n_edges, n_good = edges - nBad after the four rounds, pose_out (qx qy qz qw tx ty tz), inlier[f] = 1 where feature f's edge ended
 * as an inlier.  The projection post-check (Optimizer.cc:180-190) and the map points' counters stay with the caller, as in
 * orbfe_pose_only_optimize.  At most 2048 features per frame.                                                                    */
typedef struct orbfe_track_input {
  int32_t n_mp;
  const float* pos;              /* [n_mp][3] MapPoint::getPos()                                                   */
  const float* view_dir;         /* [n_mp][3]                                                                      */
  const float* max_dist;         /* [n_mp]                                                                         */
  const float* min_dist;         /* [n_mp]                                                                         */
  const uint8_t* desc;           /* [n_mp][32] MapPoint::getDesc()                                                 */
  const uint8_t* flags;          /* [n_mp], see above                                                              */
  const int32_t* held;           /* [n_features], nullable: map point (index) a feature holds on entry, -1 none    */
  const double* right_u;         /* [n_features], nullable (every edge mono): Frame::getRightU                     */
  const float* level_sigma2;     /* [n_levels] VirtualFrame::getScaledFactor2(level)                               */
  const float* level_inv_sigma2; /* [n_levels] getScaledFactorInv2(level)                                          */
  const double* pose_se3;        /* [7] Converter::ConvertTcw2SE3(mRcw, mtcw): the optimisation's initial estimate */
  float th;                      /* searchByProjection's th (Tracking.cc:645-649: 3, or 5 after a relocalisation)  */
  float ratio;                   /* ORBMatcher::mfRatio (0.8 in trackLocalMap)                                     */
  int32_t min_threshold;         /* ORBMatcher::mnMinThreshold (50)                                                */
  int32_t min_matches;           /* Tracking.cc:656: below this many matches no optimisation (30)                  */
} orbfe_track_input;
typedef struct orbfe_track_output {
  int32_t* assigned;  /* [n_features]                                  */
  int32_t* edge_of;   /* [n_features], nullable: edge index of a feature in the optimisation (-1 none) */
  uint8_t* inlier;    /* [n_features]                                  */
  int32_t *n_matches, *n_edges, *n_good;
  double* pose_out;   /* [7]                                           */
} orbfe_track_output;
orbfe_status orbfe_track_local_map(orbfe_ctx* ctx, int32_t slot, const orbfe_frame_pose* pose, const orbfe_camera* cam,
                                   const orbfe_track_input* in, const orbfe_track_output* out);

/* The middle of Tracking::trackMotionModel (src/Tracking.cc:382-396) as ONE call: ORBMatcher::searchByProjection(frame, lastFrame, matches, th)
 * -- in this reference a search around the LAST frame's feature positions within th * getScaledFactor2(octave) pixels (VirtualFrame::
 * findFeaturesInArea, src/Frame.cc:286-311: by grid cell), octave window by the motion direction, among the
 * frame's features that hold no map point yet, accepted when ratio < mfRatio and distance < mnMinThreshold; every accepted query is a match
 * and setMapPoints assigns them in query order, so the last query that picked a feature keeps it (src/ORBMatcher.cc:265-347, :815-830) --
 * then, with fewer than min_matches matches, the same search with th_second among the features still free (Tracking.cc:388-391; the
 * matches of the first stay), then Optimizer::OptimizePoseOnly(frame).  The frame's features are the device-resident results of `slot`.
 * Queries = the last frame's features that hold a good map point, in feature order (the caller's filter, :286-289).
 * Outputs as orbfe_track_local_map (assigned[f] = query index); n_matches = the accepted queries of all passes; excluded_hits[f]
 * (nullable) = how many queries met feature f among their candidates while it held a map point (the addMatchInTrack calls of :322-331);
 * query_matches[i] (nullable, [n]) = in how many of the searches query i was accepted as a match (setMapPoints calls addMatchInTrack for
 * every match, whether a later query takes the feature over or not); passes (nullable) = 1 or 2.  The second search is decided on the
 * host.  At most 2048 features per frame.                                                                                            */
typedef struct orbfe_motion_input {
  int32_t n;
  const float* qxy;              /* [n][2] the last frame's (undistorted) feature positions: the search centres            */
  const int8_t* q_octave;        /* [n] their octaves: findFeaturesInArea searches within th * getScaledFactor2(octave)     */
  const int8_t* q_min_level;     /* [n] octave window (:300-315: [octave, 7] forward, [0, octave] backward, else +-1)       */
  const int8_t* q_max_level;     /* [n]                                                                                    */
  const uint8_t* desc;           /* [n][32] the last frame's descriptors                                                   */
  const float* pos;              /* [n][3] the map points' positions (the pose-only edges)                                 */
  const int32_t* held;           /* [n_features], nullable: query index a feature of the frame holds on entry, -1 none     */
  const double* right_u;         /* [n_features], nullable                                                                 */
  const float* level_sigma2;     /* [n_levels]                                                                             */
  const float* level_inv_sigma2; /* [n_levels]                                                                             */
  const double* pose_se3;        /* [7] the predicted pose: the optimisation's initial estimate                            */
  float th, th_second;           /* 15, 30 (Tracking.cc:387, 390); th_second <= 0: no second search                        */
  float ratio;                   /* ORBMatcher::mfRatio (0.9 in trackMotionModel)                                          */
  int32_t min_threshold;         /* ORBMatcher::mnMinThreshold                                                             */
  int32_t min_matches;           /* 20 (Tracking.cc:388, 392)                                                              */
} orbfe_motion_input;
orbfe_status orbfe_track_motion_model(orbfe_ctx* ctx, int32_t slot, const float* bounds4 /* mfMinU mfMaxU mfMinV mfMaxV */, const orbfe_camera* cam,
                                      const orbfe_motion_input* in, const orbfe_track_output* out, int32_t* excluded_hits, int32_t* query_matches,
                                      int32_t* passes);

/* ---- map.pb: the reference's on-disk map, and a local bundle adjustment on it -------------------------------
 * `orbslam2.MapData` as Map::saveToProtobuf writes it (src/Map.cc:200-250; proto/Map.proto, Keyframe.proto,
 * MapPoint.proto), read and written without libprotobuf (host/map_pb.hpp).  The first three calls are host-only
 * (no context, no device).  Output buffers follow the "size query" convention: *out_len receives the size needed;
 * with out == NULL or cap too small nothing is written and ORBFE_ECAPACITY is returned (ORBFE_OK for out == NULL). */
typedef struct orbfe_map_summary {
  uint64_t next_id;        /* KeyFrameList.next_id (KeyFrame::mnNextId)                                   */
  int32_t n_scale_factors; /* KeyFrameList.scale_factors (KeyFrame::mvfScaledFactors)                     */
  int32_t n_keyframes, n_mappoints;
  int64_t n_keypoints;     /* over all keyframes                                                          */
  int64_t n_observations;  /* keypoints that carry a map point id (map_points[i] != -1)                   */
} orbfe_map_summary;
/* Map::loadFromProtobuf's parse (src/Map.cc:263-267); ORBFE_EBADARG for a malformed file                  */
orbfe_status orbfe_map_pb_summary(const uint8_t* pb, size_t len, orbfe_map_summary* out);
/* parse + serialise: the canonical encoding libprotobuf's C++ serialiser produces for the same content    */
orbfe_status orbfe_map_pb_reencode(const uint8_t* pb, size_t len, uint8_t* out, size_t cap, size_t* out_len);

/* The reference's other on-disk format, the TEXT map of Map::saveToTxtFile / loadFromTxtFile (src/Map.cc:82-165): `KeyFrames.txt`
 * (one header line + 10 lines per keyframe, KeyFrame.cc:400-530) and `MapPoints.txt` (3 lines per map point, MapPoint.cc:538-596), as
 * conversions to and from map.pb (host/map_txt.hpp; numbers formatted by iostreams exactly like the reference: 6 significant digits,
 * so the text form is lossy).  Size-query convention as above, for both outputs.                                                  */
orbfe_status orbfe_map_pb_to_txt(const uint8_t* pb, size_t len, char* kf_out, size_t kf_cap, size_t* kf_len, char* mp_out, size_t mp_cap,
                                 size_t* mp_len);
orbfe_status orbfe_map_txt_to_pb(const char* kf_txt, size_t kf_len, const char* mp_txt, size_t mp_len, uint8_t* out, size_t cap,
                                 size_t* out_len);

/* The graph Optimizer::OptimizeLocalMap builds around keyframe kf_id (src/Optimizer.cc:232-330): vertices =
 * [covisible keyframes (weight > 15, descending) + kf_id | fixed observers], map points of the free group in
 * ascending id, one edge per observation (stereo iff right_u > 0; information getScaledFactorInv2 / Inv, quirk Q9).
 * sizes[4] = {n_poses, n_group, n_points, n_edges}; every array pointer may be NULL (size query), else it must
 * hold the counts a previous call returned.                                                                  */
typedef struct orbfe_map_graph {
  uint64_t* pose_kf_id;  /* [n_poses]                                                                       */
  uint8_t* pose_fixed;   /* [n_poses]  setFixed                                                             */
  double* poses;         /* [n_poses][7] Converter::ConvertTcw2SE3                                          */
  uint64_t* point_id;    /* [n_points]                                                                      */
  double* points;        /* [n_points][3]                                                                   */
  int32_t* edge_pose;    /* [n_edges] vertex index                                                          */
  int32_t* edge_point;   /* [n_edges] point index                                                           */
  int32_t* edge_feat;    /* [n_edges] keypoint index inside the observing keyframe                          */
  double* meas;          /* [n_edges][3]                                                                    */
  uint8_t* is_stereo;    /* [n_edges]                                                                       */
  double* info;          /* [n_edges]                                                                       */
  double* huber_delta;   /* [n_edges]                                                                       */
} orbfe_map_graph;
orbfe_status orbfe_map_local_graph(const uint8_t* pb, size_t len, uint64_t kf_id, int32_t sizes[4], const orbfe_map_graph* out);

/* Optimizer::OptimizeLocalMap(kf, isStop) (src/Optimizer.cc:225-441) on a map file: graph as above, the two
 * optimize() rounds on the device (orbfe_ba_local_optimize), then the reference's write-back policy (:363-441):
 * outlier observations erased and poses / points stored as float unless more than 20 % of the affected keyframes
 * would lose more than 30 % of their map points.  Camera::mfFx.. come from `cam` (fx fy cx cy bf).  Output: the
 * updated map.pb.  MapPoint::updateDescriptor / updateNormalAndDepth and KeyFrame::updateConnections (:436-440)
 * are map bookkeeping and are NOT applied.                                                                   */
typedef struct orbfe_map_ba_report {
  int32_t n_poses, n_group, n_points, n_edges;
  int32_t n_outlier_edges, n_keyframes_hit, n_bad_keyframes, written;
  int32_t iterations[2];
  double chi2_before, chi2_after; /* sum of robustified-free chi2 over all edges at the initial / final estimates */
} orbfe_map_ba_report;
orbfe_status orbfe_map_local_ba(orbfe_ctx* ctx, const uint8_t* pb, size_t len, uint64_t kf_id, const orbfe_camera* cam,
                                const volatile uint8_t* stop_flag, uint8_t* out, size_t cap, size_t* out_len,
                                orbfe_map_ba_report* report);

/* ---- instrumentation ---------------------------------------------------------------------------
 * Stage timing with HIP events on the context stream.  Enable, run, then read the accumulated
 * per-stage milliseconds and launch counts.  Stage ids: see orbfe_stage.                             */
typedef enum orbfe_stage {
  ORBFE_STAGE_RESIZE = 0,
  ORBFE_STAGE_BLUR = 1,
  ORBFE_STAGE_FAST = 2,
  ORBFE_STAGE_QUADTREE = 3,
  ORBFE_STAGE_BRIEF = 4,
  ORBFE_STAGE_STEREO = 5,
  ORBFE_STAGE_MATCH = 6,
  ORBFE_STAGE_BA = 7,
  ORBFE_STAGE_COUNT = 8
} orbfe_stage;
/* on = 0: off.  on = 1: every stage timed ALONE (the overlaps between streams and the graph replay are switched off, kernels run in
 * line).  on = 2 + stage: only that stage is timed, inside the production schedule (what rocprofv3 sees for its kernels).            */
orbfe_status orbfe_profile_enable(orbfe_ctx* ctx, int32_t on);
orbfe_status orbfe_profile_read(orbfe_ctx* ctx, double* ms /*[ORBFE_STAGE_COUNT]*/, int64_t* launches /*[..]*/, int32_t reset);
const char* orbfe_stage_name(int32_t stage);

/* ---- debugging / parity aids (used by tests; stable but not part of the drop-in surface) -------
 * FAST candidates of (slot, level) in the reference's order (cell-row-major, in-cell raster), region
 * coordinates: xyr[3*i] = x, y, response.  Returns the count through n_out even if cap is too small.  */
orbfe_status orbfe_debug_candidates(orbfe_ctx* ctx, int32_t slot, int32_t level, float* xyr, int32_t cap, int32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_H_ */
