// orbfe_shim.hpp -- C++17 host-side mirror of the reference's operator interface for the hot path, written on
// the C-ABI of include/orbfe.h.  Header only; link with -lorbfe_hip.
//
// Two layers:
//   namespace orbfe              OpenCV-free classes (ImageView / std::vector) with the reference's names, argument
//                                order and error behaviour.  Always available.
//   namespace ORB_SLAM2_ROS2     (orbfe_dropin.hpp) the drop-in classes with the reference's exact signatures (cv::Mat, cv::KeyPoint,
//                                Frame::SharedPtr, KeyFrame::SharedPtr), for the reference's own build.
//
// Error mapping (include/ORB_SLAM2/Error.h): ORBFE_EBADSIZE -> ImageSizeError, a missing template file ->
// FileNotOpenError, everything else -> std::runtime_error with orbfe_last_error().
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include <exception>
#include <thread>

#include "../../include/orbfe.h"
#include "map_pb.hpp"

namespace orbfe {

struct ImageSizeError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct FileNotOpenError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct ImageView {  // 8-bit single channel, like a CV_8UC1 cv::Mat header
  const uint8_t* data = nullptr;
  int cols = 0, rows = 0;
  size_t step = 0;
};

using Descriptor = std::array<uint8_t, ORBFE_DESC_BYTES>;

inline void check(orbfe_ctx* ctx, orbfe_status st) {
  if (st == ORBFE_OK) return;
  const std::string msg = orbfe_last_error(ctx);
  if (st == ORBFE_EBADSIZE) throw ImageSizeError(msg);
  throw std::runtime_error(msg);
}

// Parses config/brief_template.txt exactly like ORBExtractor::initBriefTemplate (src/ORBExtractor.cc:242-267): the header line is
// skipped, EVERY further line is a pair read with operator>> (a value that does not parse, and all after it, stay 0: a blank line is the
// pair (0,0)-(0,0)), no count is checked.  Only the first 256 pairs reach the 32-byte descriptor (computeBRIEF, :405-406, :426-456), so a
// longer file behaves like its first 256 lines; a shorter one makes the reference index past its template (undefined behaviour) and is
// refused here.
inline std::vector<int8_t> loadBriefTemplate(const std::string& path) {
  std::ifstream ifs(path);
  if (!ifs.is_open()) throw FileNotOpenError("BRIEF template file cannot be opened: " + path);
  std::vector<int8_t> out;
  std::string line;
  bool header = true;
  while (std::getline(ifs, line)) {
    if (header) {
      header = false;
      continue;
    }
    std::istringstream iss(line);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    iss >> v[0] >> v[1] >> v[2] >> v[3];
    for (float f : v) {
      if (!(f > -129.f && f < 128.f)) throw std::runtime_error("BRIEF template value outside [-128, 127] (line " + std::to_string(out.size() / 4 + 2) + "): " + path);
      out.push_back((int8_t)f);
    }
  }
  if (out.size() < 1024) throw std::runtime_error("BRIEF template holds fewer than the 256 pairs a descriptor needs: " + path);
  out.resize(1024);
  return out;
}

// One device context per (geometry, parameters, device); the reference constructs an extractor per image (src/Frame.cc:91-92,
// :132) -- the device buffers behind them are shared and re-used.  Every extractor object takes one image SLOT of that context when
// it extracts: slots are handed out round-robin, so the left and the right extractor of a frame (two objects, two threads,
// Frame.cc:100-105) always work in different slots, through orbfe_extract_slot, which is safe for exactly that.  A slot keeps its
// results on the device until it is handed out again (kSlots extractions later); a generation counter tells an object whether the
// slot still holds ITS results.
class ContextPool {
 public:
  static constexpr int kSlots = 4;  // two stereo frames' worth: the frame being built and the one before it
  using Key = std::tuple<int, int, int, int, float, int, int, std::string, int, int, int>;
  struct Lease {
    int slot = -1;
    uint64_t generation = 0;
  };
  static orbfe_ctx* get(int w, int h, int nFeatures, int nLevels, float scale, int maxTh, int minTh, const std::string& tplPath,
                        int device = 0, int maxImages = kSlots, int tag = 0) {
    // tag: contexts that differ in nothing else -- one per calling thread role (dropin::matcherContext / solverContext): a context
    // serves one call at a time, and the reference's matchers and optimisers run on three threads at once (System.cc:119-129)
    std::lock_guard<std::mutex> lk(mu());
    // (GPU_MAX_HW_QUEUES is left alone here: a process that builds one frame at a time -- this mirror's use -- is ~50 us per frame
    //  FASTER with the runtime's default of 4 hardware queues; the batch / sequence streaming API wants 16: INTEGRATION.md 5a)
    Key key{w, h, nFeatures, nLevels, scale, maxTh, minTh, tplPath, device, maxImages, tag};
    auto it = pool().find(key);
    if (it != pool().end()) return it->second.get();
    std::vector<int8_t> tpl;
    if (!tplPath.empty()) tpl = loadBriefTemplate(tplPath);
    orbfe_config cfg{};
    cfg.width = w;
    cfg.height = h;
    cfg.n_features = nFeatures;
    cfg.n_levels = nLevels;
    cfg.scale_factor = scale;
    cfg.fast_hi = maxTh;
    cfg.fast_lo = minTh;
    cfg.brief_pairs = tpl.empty() ? nullptr : tpl.data();
    cfg.device_id = device;
    cfg.max_images = maxImages;
    orbfe_ctx* ctx = nullptr;
    check(nullptr, orbfe_create(&cfg, &ctx));
    pool()[key] = std::shared_ptr<orbfe_ctx>(ctx, orbfe_destroy);
    slots()[ctx] = SlotTable{0, std::vector<uint64_t>((size_t)maxImages, 0)};
    return ctx;
  }
  // the next slot of the context, and the generation its new content carries.  Slots on which an eagerStart constructor has begun an
  // extraction that nobody has collected yet (setBusy) are passed over: the library refuses every other call on such a slot.  So at most
  // `maxImages` (kSlots) extractor objects of one context can be constructed-but-not-extracted at a time; the next one throws.
  static Lease acquire(orbfe_ctx* ctx) {
    std::lock_guard<std::mutex> lk(mu());
    SlotTable& t = table(ctx);
    const int n = (int)t.generation.size();
    for (int k = 0; k < n; ++k) {
      const int s = (t.next + k) % n;
      if (t.busy[(size_t)s]) continue;
      Lease l;
      l.slot = s;
      t.next = (s + 1) % n;
      l.generation = ++t.generation[(size_t)s];
      return l;
    }
    throw std::logic_error("ContextPool::acquire: every slot of the context holds a started extraction nobody has collected "
                           "(ORBExtractor::eagerStart: at most ContextPool::kSlots constructed-but-not-extracted objects per context)");
  }
  static void setBusy(orbfe_ctx* ctx, int slot, bool on) {
    std::lock_guard<std::mutex> lk(mu());
    SlotTable& t = table(ctx);
    if (slot >= 0 && (size_t)slot < t.busy.size()) t.busy[(size_t)slot] = on ? 1 : 0;
  }
  // the next slot PAIR (even, odd) of the context for orbfe_frame_stereo_slots: both slots of the pair get a new generation
  static std::pair<Lease, Lease> acquirePair(orbfe_ctx* ctx) {
    std::lock_guard<std::mutex> lk(mu());
    SlotTable& t = table(ctx);
    const int n = (int)t.generation.size();
    if (n < 2) throw std::logic_error("ContextPool::acquirePair: the context has fewer than two slots");
    int s = (t.next + 1) & ~1;
    if (s + 1 >= n) s = 0;
    for (int k = 0; t.busy[(size_t)s] || t.busy[(size_t)s + 1]; ++k) {  // (a pair with a started, uncollected extraction is passed over)
      if (2 * k >= n) throw std::logic_error("ContextPool::acquirePair: every slot pair of the context holds a started extraction nobody has collected");
      s += 2;
      if (s + 1 >= n) s = 0;
    }
    t.next = (s + 2) % n;
    Lease a, b;
    a.slot = s, b.slot = s + 1;
    a.generation = ++t.generation[(size_t)s];
    b.generation = ++t.generation[(size_t)s + 1];
    return {a, b};
  }
  static bool current(orbfe_ctx* ctx, const Lease& l) {
    std::lock_guard<std::mutex> lk(mu());
    auto it = slots().find(ctx);
    return l.slot >= 0 && it != slots().end() && (size_t)l.slot < it->second.generation.size() &&
           it->second.generation[(size_t)l.slot] == l.generation;
  }

 private:
  struct SlotTable {
    int next = 0;
    std::vector<uint64_t> generation;
    std::vector<uint8_t> busy;  // an orbfe_extract_slot_begin is outstanding on the slot (eagerStart)
  };
  static SlotTable& table(orbfe_ctx* ctx) {  // (under mu())
    SlotTable& t = slots()[ctx];
    if (t.generation.empty()) t.generation.assign(kSlots, 0);
    if (t.busy.size() != t.generation.size()) t.busy.assign(t.generation.size(), 0);
    return t;
  }
  static std::mutex& mu() {
    static std::mutex m;
    return m;
  }
  static std::map<Key, std::shared_ptr<orbfe_ctx>>& pool() {
    static std::map<Key, std::shared_ptr<orbfe_ctx>> p;
    return p;
  }
  static std::map<orbfe_ctx*, SlotTable>& slots() {
    static std::map<orbfe_ctx*, SlotTable> s;
    return s;
  }
};

class ORBExtractor {
 public:
  typedef std::shared_ptr<ORBExtractor> SharedPtr;
  static constexpr int mnBorderSize = 19;  // src/ORBExtractor.cc:523

  ORBExtractor(const ImageView& image, int nFeatures, int pyramidLevels, float scaleFactor, const std::string& bfTemFp,
               int maxThreshold, int minThreshold)
      : mImage(image), mnFeats(nFeatures), mnLevels(pyramidLevels) {
    if (!image.data || image.cols <= 0 || image.rows <= 0) throw std::invalid_argument("ORBExtractor: empty image");
    mCtx = ContextPool::get(image.cols, image.rows, nFeatures, pyramidLevels, scaleFactor, maxThreshold, minThreshold, bfTemFp);
    mnFeats = orbfe_get_capacity(mCtx);  // array stride: nFeatures, or more where the reference's rounded quotas exceed it
    mScales.resize(pyramidLevels);
    check(mCtx, orbfe_get_scale_factors(mCtx, mScales.data(), pyramidLevels));
    if (eagerStart()) {  // the device starts on the image NOW, as the reference's constructor builds the pyramid now (src/ORBExtractor.cc:205-214)
      mLease = ContextPool::acquire(mCtx);
      check(mCtx, orbfe_extract_slot_begin(mCtx, mLease.slot, mImage.data, mImage.step));
      ContextPool::setBusy(mCtx, mLease.slot, true);
      mStarted = true;
    }
  }
  ~ORBExtractor() {
    if (!mStarted) return;  // an extraction nobody collected: drain the slot (a destructor cannot throw: a device error is reported on stderr)
    const orbfe_status st = orbfe_extract_slot_end(mCtx, mLease.slot, nullptr, nullptr, nullptr);
    ContextPool::setBusy(mCtx, mLease.slot, false);
    if (st != ORBFE_OK) std::fprintf(stderr, "[orbfe] ~ORBExtractor: draining slot %d failed: %s\n", mLease.slot, orbfe_last_error(mCtx));
  }
  ORBExtractor(const ORBExtractor&) = delete;
  ORBExtractor& operator=(const ORBExtractor&) = delete;
  // Process-wide switch (default off): the constructor enqueues the extraction (orbfe_extract_slot_begin) and extract() only collects it.
  // For callers that keep the reference's Frame::Frame as it is -- two extractor objects built on the constructing thread, then one
  // std::thread per extract() (src/Frame.cc:91-105): the device works while the threads are being created and scheduled (30 - 50 us each,
  // the tail of that call shape).  Leave it off where a frame is built by extractStereo / extractRGBD (one device call: INTEGRATION 2b).
  static bool& eagerStart() {
    static bool on = false;
    return on;
  }

  // ORBExtractor::extract (src/ORBExtractor.cc:499-508).  Thread-safe against extract() of OTHER objects (Frame.cc:100-105).
  //
  // Errors.  The reference runs this member on a bare std::thread (`std::thread leftThread(std::bind(&ORBExtractor::extract, ...))`,
  // src/Frame.cc:100-105): an exception that leaves it there is std::terminate.  The reference's extract() cannot fail; this one can (a
  // device error), so on a thread OTHER than the one that constructed the object the failure is CAPTURED -- the call returns with no
  // keypoints -- and rethrown by the next call anybody makes on this object from another thread: rethrowPending(), called by
  // ORBMatcher::searchByStereo (the very next statement of Frame::createStereo, Frame.h:319), getPyramidLevel, extractStereo,
  // extractRGBD and by extract() itself.  On the constructing thread the exception propagates at once, as before.
  void extract(std::vector<orbfe_keypoint>& keyPoints, std::vector<Descriptor>& descriptors) {
    const bool foreign = std::this_thread::get_id() != mOwner;
    if (!foreign) rethrowPending();
    try {
      keyPoints.resize(mnFeats);
      descriptors.resize(mnFeats);
      int32_t n = 0;
      if (mStarted) {  // the constructor enqueued it (eagerStart): collect
        mStarted = false;
        const orbfe_status st = orbfe_extract_slot_end(mCtx, mLease.slot, keyPoints.data(), descriptors.data()->data(), &n);
        ContextPool::setBusy(mCtx, mLease.slot, false);
        check(mCtx, st);
      } else if (mDrained && resident()) {  // getPyramidLevel / searchByStereo came first and drained the started extraction: its results are in the slot
        check(mCtx, orbfe_fetch_features(mCtx, mLease.slot, keyPoints.data(), descriptors.data()->data(), &n));
      } else {
        mDrained = false;
        mLease = ContextPool::acquire(mCtx);
        check(mCtx, orbfe_extract_slot(mCtx, mLease.slot, mImage.data, mImage.step, keyPoints.data(), descriptors.data()->data(), &n));
      }
      keyPoints.resize(n);
      descriptors.resize(n);
      mnKeyPoints = n;
    } catch (...) {
      keyPoints.clear();
      descriptors.clear();
      mnKeyPoints = 0;
      if (!foreign) throw;
      std::lock_guard<std::mutex> lk(mPendingMutex);
      mPending = std::current_exception();
    }
  }
  // rethrows (once) what an extract() on a foreign thread captured; no-op otherwise
  void rethrowPending() const {
    std::exception_ptr e;
    {
      std::lock_guard<std::mutex> lk(mPendingMutex);
      std::swap(e, mPending);
    }
    if (e) std::rethrow_exception(e);
  }
  bool hasPendingError() const {
    std::lock_guard<std::mutex> lk(mPendingMutex);
    return (bool)mPending;
  }

  // Frame::createStereo's device work as ONE call (include/ORB_SLAM2/Frame.h:313-323: the two extractions of src/Frame.cc:100-105 and
  // ORBMatcher::searchByStereo): `this` is the left extractor, `right` the right one; both take a slot of one pair of the context, the
  // results are those of the two extract() calls followed by ORBMatcher::searchByStereo.  Returns the match count (Frame::mnN).
  int extractStereo(ORBExtractor& right, float fx, float bf, std::vector<orbfe_keypoint>& kpsLeft, std::vector<Descriptor>& descLeft,
                    std::vector<orbfe_keypoint>& kpsRight, std::vector<Descriptor>& descRight, std::vector<double>& rightU,
                    std::vector<double>& depths) {
    rethrowPending(), right.rethrowPending();
    drainStarted(), right.drainStarted();
    if (mCtx != right.mCtx) throw std::logic_error("extractStereo: the two extractors differ in geometry / parameters");
    if (mImage.step != right.mImage.step) throw std::logic_error("extractStereo: the two images differ in row stride");
    const size_t cap = (size_t)mnFeats;
    std::vector<orbfe_keypoint> k(2 * cap);
    std::vector<Descriptor> d(2 * cap);
    rightU.resize(cap), depths.resize(cap);
    int32_t n[2] = {0, 0}, nm = 0;
    auto pair = ContextPool::acquirePair(mCtx);
    mLease = pair.first, right.mLease = pair.second;
    mDrained = right.mDrained = false;  // (the slots hold a new extraction: a drained eager start is history)
    check(mCtx, orbfe_frame_stereo_slots(mCtx, mLease.slot, mImage.data, right.mImage.data, mImage.step, fx, bf, k.data(), d.data()->data(), n,
                                         rightU.data(), depths.data(), &nm));
    kpsLeft.assign(k.begin(), k.begin() + n[0]), descLeft.assign(d.begin(), d.begin() + n[0]);
    kpsRight.assign(k.begin() + cap, k.begin() + cap + n[1]), descRight.assign(d.begin() + cap, d.begin() + cap + n[1]);
    rightU.resize((size_t)n[0]), depths.resize((size_t)n[0]);
    mnKeyPoints = n[0], right.mnKeyPoints = n[1];
    return nm;
  }

  // Frame::createRGBD's device work as ONE call (include/ORB_SLAM2/Frame.h:326-331; the RGB-D Frame constructor, src/Frame.cc:125-158):
  // the extraction of this object's (gray) image, Camera::undistortPoints and the depth / rightU lookup -- the results of extract()
  // followed by orbfe_frame_rgbd.  depth: the image as read from the file (type 0: 16-bit, 1: float), rows depthStep bytes apart.
  void extractRGBD(const orbfe_camera& cam, const void* depth, int depthType, size_t depthStep, float depthScale,
                   std::vector<orbfe_keypoint>& undistorted, std::vector<Descriptor>& descriptors, std::vector<double>& depths,
                   std::vector<double>& rightU) {
    rethrowPending();
    drainStarted();
    undistorted.resize(mnFeats), descriptors.resize(mnFeats), depths.resize(mnFeats), rightU.resize(mnFeats);
    int32_t n = 0;
    mDrained = false;  // (the slot is about to hold UNDISTORTED keypoints: extract() must not hand those out as the raw ones)
    mLease = ContextPool::acquire(mCtx);
    check(mCtx, orbfe_frame_rgbd_image(mCtx, mLease.slot, mImage.data, mImage.step, 0, &cam, depth, depthType, depthStep, depthScale,
                                       undistorted.data(), descriptors.data()->data(), &n, depths.data(), rightU.data()));
    undistorted.resize(n), descriptors.resize(n), depths.resize(n), rightU.resize(n);
    mnKeyPoints = n;
  }

  // level `l` of the un-blurred pyramid, tight rows (what getPyramid()[l] holds in the reference).  Read from the device on demand:
  // if the slot has been handed to another extractor since, the pyramid is rebuilt from the image this object still refers to
  // (like the reference's pyramid, it lives as long as the image does).
  std::vector<uint8_t> getPyramidLevel(int l, int* w = nullptr, int* h = nullptr) {
    rethrowPending();
    drainStarted();  // (legal in the reference before extract(): its constructor has built the pyramid; a begun slot refuses orbfe_get_pyramid)
    orbfe_level_info li{};
    check(mCtx, orbfe_get_level_info(mCtx, l, &li));
    if (!resident()) {
      std::vector<orbfe_keypoint> k;
      std::vector<Descriptor> d;
      extract(k, d);
    }
    std::vector<uint8_t> out((size_t)li.width * li.height);
    check(mCtx, orbfe_get_pyramid(mCtx, mLease.slot, l, 0, out.data()));
    if (w) *w = li.width;
    if (h) *h = li.height;
    return out;
  }
  const std::vector<float>& getScaledFactors() const { return mScales; }
  orbfe_ctx* context() const { return mCtx; }
  int slot() const { return mLease.slot; }                                 // -1 before the first extract()
  bool resident() const { return ContextPool::current(mCtx, mLease); }    // the slot still holds this object's results
  int keyPointCount() const { return mnKeyPoints; }
  int levels() const { return mnLevels; }

 private:
  ImageView mImage;
  int mnFeats, mnLevels, mnKeyPoints = 0;
  orbfe_ctx* mCtx = nullptr;
  ContextPool::Lease mLease;
  std::vector<float> mScales;
  mutable bool mStarted = false;  // eagerStart: the constructor's orbfe_extract_slot_begin is outstanding
  mutable bool mDrained = false;  // ... and was collected without its features (drainStarted): extract() fetches them from the slot
  friend class ORBMatcher;
  void drainStarted() const {
    if (!mStarted) return;
    mStarted = false;
    int32_t n = 0;
    const orbfe_status st = orbfe_extract_slot_end(mCtx, mLease.slot, nullptr, nullptr, &n);
    ContextPool::setBusy(mCtx, mLease.slot, false);
    check(mCtx, st);
    mDrained = true;
    const_cast<ORBExtractor*>(this)->mnKeyPoints = n;
  }
  std::thread::id mOwner = std::this_thread::get_id();  // the constructing thread (Frame::Frame's)
  mutable std::mutex mPendingMutex;
  mutable std::exception_ptr mPending;                  // what extract() captured on a foreign thread
};

class ORBMatcher {
 public:
  static constexpr int mnMaxThreshold = 100, mnMinThreshold = 50, mnMeanThreshold = 75, mnW = 5, mnL = 5;  // ORBMatcher.cc:1086-1090
  explicit ORBMatcher(float ratio = 0.6f, bool checkOri = true) : mfRatio(ratio), mbCheckOri(checkOri) {}

  // ORBMatcher::descDistance (src/ORBMatcher.cc:941-956)
  static int descDistance(const Descriptor& a, const Descriptor& b) {
    int d = 0;
    for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
  }
  // ORBMatcher::searchByStereo (src/ORBMatcher.cc:18-81) over the device-resident results of the frame's two extractors (what
  // Frame::createStereo does right after Frame::Frame, include/ORB_SLAM2/Frame.h:313-322); fills mvFeatsRightU / mvDepths
  // (-1 where unmatched) and returns the match count (Frame::mnN).
  int searchByStereo(const ORBExtractor& left, const ORBExtractor& right, float fx, float bf, std::vector<double>& rightU,
                     std::vector<double>& depths) const {
    left.rethrowPending(), right.rethrowPending();  // a failure of one of Frame::Frame's extract() threads surfaces here (see extract())
    left.drainStarted(), right.drainStarted();      // eagerStart objects nobody has called extract() on: the match needs idle slots
    if (left.context() != right.context()) throw std::logic_error("searchByStereo: the two extractors differ in geometry / parameters");
    if (!left.resident() || !right.resident())
      throw std::logic_error("searchByStereo: the extractors' device results have been overwritten (more than "
                             "ContextPool::kSlots extractions since); match right after Frame construction as the reference does");
    orbfe_ctx* ctx = left.context();
    const int cap = std::max((int)orbfe_get_capacity(ctx), 1);
    std::vector<double> ru((size_t)cap), dp((size_t)cap);
    int32_t n = 0;
    check(ctx, orbfe_stereo_match(ctx, left.slot(), right.slot(), fx, bf, ru.data(), dp.data(), &n, nullptr, nullptr));
    rightU.assign(ru.begin(), ru.begin() + left.keyPointCount());
    depths.assign(dp.begin(), dp.begin() + left.keyPointCount());
    return n;
  }

  // The matching core of ORBMatcher::searchByProjection (src/ORBMatcher.cc:265-347, 561-612): for every projected point
  // Frame::findFeaturesInArea (src/Frame.cc:286-311) + getBestMatch (:967-990) against the features of `slot`.  The caller
  // projects, picks radius / octave window and applies mnMinThreshold, the ratio test and verifyAngle as the reference does.
  struct AreaMatch {
    std::vector<int32_t> bestIdx, bestDist, secondDist, nCand;
  };
  AreaMatch searchInArea(orbfe_ctx* ctx, int slot, const std::vector<float>& uv /*[n][2]*/, const std::vector<float>& radius,
                         const std::vector<int8_t>& minLevel, const std::vector<int8_t>& maxLevel,
                         const std::vector<Descriptor>& desc, const std::vector<uint8_t>* exclude = nullptr) const {
    const int32_t n = (int32_t)radius.size();
    AreaMatch m;
    m.bestIdx.resize(n), m.bestDist.resize(n), m.secondDist.resize(n), m.nCand.resize(n);
    std::vector<uint8_t> ex;  // the ABI reads one flag per possible feature of the slot
    if (exclude) {
      ex.assign((size_t)std::max<int>(orbfe_get_capacity(ctx), (int)exclude->size()), 0);
      std::copy(exclude->begin(), exclude->end(), ex.begin());
    }
    check(ctx, orbfe_search_in_area(ctx, slot, n, uv.data(), radius.data(), minLevel.data(), maxLevel.data(),
                                    n ? desc[0].data() : nullptr, exclude ? ex.data() : nullptr, m.bestIdx.data(),
                                    m.bestDist.data(), m.secondDist.data(), m.nCand.data()));
    return m;
  }

  struct DMatch {  // cv::DMatch fields the reference uses
    int queryIdx, trainIdx;
    int distance;
    bool operator==(const DMatch& o) const { return queryIdx == o.queryIdx && trainIdx == o.trainIdx && distance == o.distance; }
  };
  static constexpr int mnBinNum = 30, mnBinChoose = 3;  // ORBMatcher.cc:1091-1092

  // ORBMatcher::searchByProjection(pFrame1, pFrame2, matches, th, bFuse) (src/ORBMatcher.cc:265-347).  Frame 1 = the features of
  // `slot1` on the device; frame 2 enters as arrays.  valid2[idx]: map point of feature idx non-null and not bad (and, for
  // bFuse, in vision of frame 1); hasMp1[i]: feature i of frame 1 already carries a good map point (skipped unless bFuse; the
  // caller bumps addMatchInTrack and applies setMapPoints).  z = tlc.z (:274-276), bl = Camera::mfBl.
  std::vector<DMatch> searchByProjection(orbfe_ctx* ctx, int slot1, const std::vector<float>& scaleFactors2,
                                         const std::vector<orbfe_keypoint>& kps2, const std::vector<Descriptor>& desc2,
                                         const std::vector<uint8_t>& valid2, const std::vector<uint8_t>& hasMp1, float th, float z, float bl,
                                         bool bFuse) const {
    const bool up = std::abs(z) > bl && z > 0, down = std::abs(z) > bl && !(z > 0);
    std::vector<int> idx;
    std::vector<float> uv, radius;
    std::vector<int8_t> lo, hi;
    std::vector<Descriptor> qd;
    for (size_t i = 0; i < kps2.size(); ++i) {
      if (!valid2[i]) continue;
      const int o = kps2[i].octave;
      idx.push_back((int)i);
      uv.push_back(kps2[i].x), uv.push_back(kps2[i].y);
      radius.push_back(th * (scaleFactors2[o] * scaleFactors2[o]));  // findFeaturesInArea: radius * getScaledFactor2(octave)
      lo.push_back((int8_t)(up ? o : down ? 0 : std::max(0, o - 1)));
      hi.push_back((int8_t)(up ? 7 : down ? o : std::min(o + 1, 7)));
      qd.push_back(desc2[i]);
    }
    std::vector<DMatch> out;
    if (idx.empty()) return out;
    const AreaMatch m = searchInArea(ctx, slot1, uv, radius, lo, hi, qd, bFuse ? nullptr : &hasMp1);
    for (size_t k = 0; k < idx.size(); ++k) {
      if (m.nCand[k] <= 0) continue;
      const float ratio = (float)m.bestDist[k] / (float)m.secondDist[k];
      if (ratio < mfRatio && m.bestDist[k] < mnMinThreshold) out.push_back({m.bestIdx[k], idx[k], m.bestDist[k]});
    }
    return out;
  }

  // MapPoint::isInVision + predictLevel (src/MapPoint.cc:141-201) for all candidate map points of a frame in one call: the per-point
  // preamble of searchByProjection(pframe, mapPoints, ...) (src/ORBMatcher.cc:575-580).  pos / viewDir: [n][3] floats.
  struct Vision {
    std::vector<float> uv, distance, cosTheta;
    std::vector<int8_t> level;
    std::vector<uint8_t> visible;
  };
  static Vision projectMapPoints(orbfe_ctx* ctx, const std::vector<float>& pos, const std::vector<float>& viewDir,
                                 const std::vector<float>& maxDist, const std::vector<float>& minDist, const orbfe_frame_pose& pose,
                                 const orbfe_camera& cam) {
    const int32_t n = (int32_t)maxDist.size();
    Vision v;
    const size_t m = (size_t)std::max(n, 1);
    v.uv.resize(2 * m), v.distance.resize(m), v.cosTheta.resize(m), v.level.resize(m), v.visible.resize(m);
    check(ctx, orbfe_project_map_points(ctx, n, pos.data(), viewDir.data(), maxDist.data(), minDist.data(), &pose, &cam, v.uv.data(),
                                        v.distance.data(), v.cosTheta.data(), v.level.data(), v.visible.data()));
    v.uv.resize(2 * (size_t)n), v.distance.resize(n), v.cosTheta.resize(n), v.level.resize(n), v.visible.resize(n);
    return v;
  }
  // ORBMatcher::searchByProjection(pframe, mapPoints, th, matches, bFuse) (src/ORBMatcher.cc:561-612).  Per map point the caller
  // supplies the outputs of MapPoint::isInVision / predictLevel (uv, level, cosTheta) and usable = in map, not bad, in vision.
  // bFuse: matches (featIdx, mapPointIdx, distance).  Otherwise `matches` holds the assignments the caller applies with
  // setMapPoint / addMatchInTrack (first map point wins a feature, features with a good map point are left alone); returns nMatches.
  int searchByProjection(orbfe_ctx* ctx, int slot, const std::vector<float>& scaleFactors, int nLevels, const std::vector<float>& uv,
                         const std::vector<int>& level, const std::vector<float>& cosTheta, const std::vector<Descriptor>& mpDesc,
                         const std::vector<uint8_t>& usable, float th, std::vector<uint8_t> frameHasGoodMp, std::vector<DMatch>& matches,
                         bool bFuse) const {
    matches.clear();
    int nMatches = 0;
    if (!bFuse)
      for (uint8_t h : frameHasGoodMp) nMatches += h ? 1 : 0;
    std::vector<int> idx;
    std::vector<float> q, radius;
    std::vector<int8_t> lo, hi;
    std::vector<Descriptor> qd;
    for (size_t i = 0; i < usable.size(); ++i) {
      if (!usable[i]) continue;
      const int l = level[i];
      idx.push_back((int)i);
      q.push_back(uv[2 * i]), q.push_back(uv[2 * i + 1]);
      radius.push_back(((cosTheta[i] > 0.998f ? 2.5f : 4.0f) * th) * (scaleFactors[l] * scaleFactors[l]));
      lo.push_back((int8_t)std::max(0, l - 1));
      hi.push_back((int8_t)std::min(nLevels - 1, l + 1));
      qd.push_back(mpDesc[i]);
    }
    if (idx.empty()) return nMatches;
    const AreaMatch m = searchInArea(ctx, slot, q, radius, lo, hi, qd, nullptr);
    for (size_t k = 0; k < idx.size(); ++k) {
      if (m.nCand[k] <= 0) continue;
      const float ratio = (float)m.bestDist[k] / (float)m.secondDist[k];
      if (!(m.bestDist[k] < mnMinThreshold && ratio < mfRatio)) continue;
      const int f = m.bestIdx[k];
      if (bFuse) {
        matches.push_back({f, idx[k], m.bestDist[k]});
        ++nMatches;
      } else if (!frameHasGoodMp[f]) {
        frameHasGoodMp[f] = 1;  // pframe->setMapPoint(bestMatch.first, pMp)
        matches.push_back({f, idx[k], m.bestDist[k]});
        ++nMatches;
      }
    }
    return nMatches;
  }

  // ORBMatcher::searchByBow (src/ORBMatcher.cc:170-253) given the two DBoW feature vectors (node id -> feature ids, ordered maps
  // like DBoW3::FeatureVector); the BoW transform stays with DBoW3.  good* = map point non-null and not bad, inMap* = isInMap().
  std::vector<DMatch> searchByBow(orbfe_ctx* ctx, const std::vector<Descriptor>& descF, const std::vector<Descriptor>& descKF,
                                  const std::map<unsigned, std::vector<unsigned>>& featVecF,
                                  const std::map<unsigned, std::vector<unsigned>>& featVecKF, const std::vector<uint8_t>& goodF,
                                  const std::vector<uint8_t>& inMapF, const std::vector<uint8_t>& goodKF, const std::vector<uint8_t>& inMapKF,
                                  const std::vector<float>& anglesF, const std::vector<float>& anglesKF, bool bAddMPs, bool bLoop) const {
    std::vector<int> qIds;
    std::vector<uint32_t> off{0}, cand;
    std::vector<Descriptor> qd;
    auto f = featVecF.begin();
    auto k = featVecKF.begin();
    while (f != featVecF.end() && k != featVecKF.end()) {
      if (f->first > k->first) {
        ++k;
      } else if (f->first < k->first) {
        ++f;
      } else {
        std::vector<uint32_t> cf;
        for (unsigned p : f->second) {
          const bool g = goodF[p];
          if (bAddMPs ? !(g && inMapF[p]) : (bLoop || !g)) cf.push_back(p);
        }
        for (unsigned pk : k->second) {
          const bool g = goodKF[pk];
          if (bAddMPs) {
            if (g && inMapKF[pk]) continue;
          } else if (!bLoop && !g) {
            continue;
          }
          if (cf.empty()) continue;
          qIds.push_back((int)pk);
          qd.push_back(descKF[pk]);
          cand.insert(cand.end(), cf.begin(), cf.end());
          off.push_back((uint32_t)cand.size());
        }
        ++f;
        ++k;
      }
    }
    std::vector<DMatch> matches;
    if (qIds.empty()) return matches;
    const int32_t nq = (int32_t)qIds.size();
    std::vector<int32_t> bi(nq), bd(nq), sd(nq);
    check(ctx, orbfe_match_bruteforce(ctx, qd[0].data(), nq, descF.empty() ? nullptr : descF[0].data(), (int32_t)descF.size(), off.data(),
                                      cand.data(), bi.data(), bd.data(), sd.data()));
    for (int32_t i = 0; i < nq; ++i) {
      const float ratio = (float)bd[i] / (float)sd[i];
      if (bd[i] > mnMinThreshold || ratio > mfRatio) continue;
      matches.push_back({bi[i], qIds[i], bd[i]});
    }
    if (mbCheckOri) verifyAngle(matches, anglesF, anglesKF);
    return matches;
  }

  // ---- the remaining callers of getBestMatch (include/ORB_SLAM2/ORBMatcher.h:41-75) --------------------------------------------------
  // Their targets are KeyFrames, whose features are not resident in a slot: the feature set is uploaded with the call
  // (orbfe_search_in_area_features).  Map state enters as arrays, map side effects come back as data.  Float arithmetic follows the
  // reference's expressions; where it multiplies cv::Mat objects (cv::gemm, un-vendored) the convention of csrc/k_guided.hip applies: a
  // product element is the float sum a0 b0 + a1 b1 + a2 b2 taken left to right, `alpha A x + t` is (float)((double)alpha * sum + t).
  struct KeyFrameView {                       // what the searches read of a KeyFrame
    std::vector<orbfe_keypoint> kps;          // mvFeatsLeft
    std::vector<Descriptor> desc;             // mvLeftDescriptor
    std::vector<float> pos;                   // [n][3] world position of the feature's map point (ignored where !good)
    std::vector<uint8_t> good, inMap;         // map point non-null and not bad / isInMap()
    std::vector<float> maxDist, minDist;      // MapPoint::getDistance
  };
  struct Sim3 {                               // Sim3Ret (include/ORB_SLAM2/Sim3Solver.h:14-48): p_q = s R p_p + t
    float s = 1.f;
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
    Sim3 inv() const {                        // Sim3Ret::inv (:36-43)
      Sim3 o;
      o.s = 1.0f / s;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o.R[3 * r + c] = R[3 * c + r];
      float rt[3];
      matVec(o.R, t, rt);
      for (int r = 0; r < 3; ++r) o.t[r] = -o.s * rt[r];
      return o;
    }
  };
  static void matVec(const float* R, const float* x, float* out) {
    for (int r = 0; r < 3; ++r) out[r] = R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2];
  }
  static void affine(float alpha, const float* R, const float* x, const float* t, float* out) {
    float sm[3];
    matVec(R, x, sm);
    for (int r = 0; r < 3; ++r) out[r] = (float)((double)alpha * (double)sm[r] + (double)t[r]);
  }
  // MapPoint::predictLevel (src/MapPoint.cc:188-199)
  static int predictLevel(float maxDist, float d, float logScale) {
    const int level = (int)std::lrintf(std::log(maxDist / d) / logScale);
    return level < 0 ? 0 : (level > 7 ? 7 : level);
  }
  AreaMatch searchInAreaFeatures(orbfe_ctx* ctx, const std::vector<orbfe_keypoint>& tKps, const std::vector<Descriptor>& tDesc,
                                 const std::vector<float>& uv, const std::vector<float>& radius, const std::vector<int8_t>& minLevel,
                                 const std::vector<int8_t>& maxLevel, const std::vector<Descriptor>& desc,
                                 const std::vector<uint8_t>* exclude = nullptr) const {
    const int32_t n = (int32_t)radius.size();
    AreaMatch m;
    m.bestIdx.resize(n), m.bestDist.resize(n), m.secondDist.resize(n), m.nCand.resize(n);
    check(ctx, orbfe_search_in_area_features(ctx, (int32_t)tKps.size(), tKps.data(), tDesc.empty() ? nullptr : tDesc[0].data(), n, uv.data(),
                                             radius.data(), minLevel.data(), maxLevel.data(), n ? desc[0].data() : nullptr,
                                             exclude ? exclude->data() : nullptr, m.bestIdx.data(), m.bestDist.data(), m.secondDist.data(),
                                             m.nCand.data()));
    return m;
  }
  struct Intrinsics {
    float fx, fy, cx, cy, minU, maxU, minV, maxV;  // Camera::mfFx.. and the frame's undistorted bounds (VirtualFrame::isInImage)
  };
  // ORBMatcher::SIM3Project (src/ORBMatcher.cc:370-414) for the points `ids` of `src` into `dst`: best feature of dst per point, -1 = no match
  std::vector<int> sim3Project(orbfe_ctx* ctx, const KeyFrameView& src, const std::vector<int>& ids, const float* Rcw, const float* tcw,
                               const Sim3& S, const KeyFrameView& dst, float th, const Intrinsics& K, const std::vector<float>& scaleFactors) const {
    std::vector<int> out(ids.size(), -1), who;
    std::vector<float> uv, radius;
    std::vector<int8_t> lo, hi;
    std::vector<Descriptor> qd;
    const float logScale = std::log(scaleFactors.size() > 1 ? scaleFactors[1] : 1.2f);
    for (size_t k = 0; k < ids.size(); ++k) {
      const int i = ids[k];
      float pc[3], pm[3];
      if (Rcw)
        affine(1.0f, Rcw, &src.pos[3 * (size_t)i], tcw, pc);
      else
        for (int a = 0; a < 3; ++a) pc[a] = src.pos[3 * (size_t)i + a];
      affine(S.s, S.R, pc, S.t, pm);
      if (pm[2] <= 0) continue;
      const float u = K.fx * (pm[0] / pm[2]) + K.cx, v = K.fy * (pm[1] / pm[2]) + K.cy;  // Camera::project (src/Camera.cc:14-22)
      if (!(u < K.maxU && v < K.maxV && u > K.minU && v > K.minV)) continue;
      const float d = std::sqrt(pm[0] * pm[0] + pm[1] * pm[1] + pm[2] * pm[2]) / S.s;
      if (!(d < src.maxDist[i] && d > src.minDist[i])) continue;
      const int o = predictLevel(src.maxDist[i], d, logScale);
      who.push_back((int)k);
      uv.push_back(u), uv.push_back(v);
      radius.push_back(th * (scaleFactors[o] * scaleFactors[o]));
      lo.push_back((int8_t)(o - 1)), hi.push_back((int8_t)(o + 1));
      qd.push_back(src.desc[i]);
    }
    if (who.empty()) return out;
    std::vector<uint8_t> ex(dst.kps.size());
    for (size_t j = 0; j < ex.size(); ++j) ex[j] = !(dst.good[j] && dst.inMap[j]);  // vGoodIndices (:396-405)
    const AreaMatch m = searchInAreaFeatures(ctx, dst.kps, dst.desc, uv, radius, lo, hi, qd, &ex);
    for (size_t k = 0; k < who.size(); ++k) {
      if (m.nCand[k] <= 0) continue;
      const float ratio = (float)m.bestDist[k] / (float)m.secondDist[k];
      if (m.bestDist[k] <= mnMinThreshold && ratio <= mfRatio) out[who[k]] = m.bestIdx[k];
    }
    return out;
  }
  // ORBMatcher::searchBySim3(mpCurr, mpMatch, matches, g2oScm, th) (src/ORBMatcher.cc:424-484); matches: (queryIdx in C, trainIdx in M)
  int searchBySim3(orbfe_ctx* ctx, const KeyFrameView& C, const KeyFrameView& M, std::vector<std::pair<int, int>>& matches, const Sim3& Scm,
                   const float* RcwC, const float* tcwC, const float* RcwM, const float* tcwM, float th, const Intrinsics& K,
                   const std::vector<float>& scaleFactors) const {
    std::vector<uint8_t> flagC(C.kps.size(), 1), flagM(M.kps.size(), 1);
    for (const auto& m : matches) flagC[m.first] = 0, flagM[m.second] = 0;
    std::vector<int> idsC, idsM;
    for (size_t i = 0; i < C.kps.size(); ++i)
      if (flagC[i] && C.good[i] && C.inMap[i]) idsC.push_back((int)i);
    for (size_t i = 0; i < M.kps.size(); ++i)
      if (flagM[i] && M.good[i]) idsM.push_back((int)i);
    std::map<int, int> fresh;
    const std::vector<int> a = sim3Project(ctx, C, idsC, RcwC, tcwC, Scm.inv(), M, th, K, scaleFactors);
    for (size_t k = 0; k < idsC.size(); ++k)
      if (a[k] >= 0) fresh.insert({idsC[k], a[k]});
    const std::vector<int> b = sim3Project(ctx, M, idsM, RcwM, tcwM, Scm, C, th, K, scaleFactors);
    for (size_t k = 0; k < idsM.size(); ++k)
      if (b[k] >= 0) fresh.insert({b[k], idsM[k]});
    for (const auto& m : fresh) matches.push_back(m);
    return (int)matches.size();
  }
  // ORBMatcher::processFuseMps (src/ORBMatcher.cc:623-661): the decision per match; the caller applies it to its map
  struct FuseAction {
    enum Kind { Add, Replace } kind;
    long a, b;  // Add: feature index, map point index; Replace: identity kept, identity dropped
  };
  static int processFuseMps(const std::vector<DMatch>& matches, const std::vector<uint8_t>& fGood, const std::vector<long>& fId,
                            const std::vector<int>& fObs, const std::vector<uint8_t>& vGood, const std::vector<long>& vId,
                            const std::vector<int>& vObs, bool bLoop, std::vector<FuseAction>& actions) {
    int nFuse = 0;
    for (const DMatch& m : matches) {
      const int q = m.queryIdx, t = m.trainIdx;
      if (!vGood[t]) continue;
      if (!fGood[q]) {
        actions.push_back({FuseAction::Add, q, t});
        ++nFuse;
      } else if (fId[q] != vId[t]) {
        const bool keepV = bLoop || fObs[q] < vObs[t];
        actions.push_back({FuseAction::Replace, keepV ? vId[t] : fId[q], keepV ? fId[q] : vId[t]});
        ++nFuse;
      }
    }
    return nFuse;
  }

  // ORBMatcher::verifyAngle (src/ORBMatcher.cc:1013-1051)
  static void verifyAngle(std::vector<DMatch>& matches, const std::vector<float>& angles1, const std::vector<float>& angles2) {
    std::vector<std::vector<DMatch>> hist(mnBinNum);
    for (const DMatch& m : matches) {
      float diff = angles1[m.queryIdx] - angles2[m.trainIdx];
      diff = diff >= 0 ? diff : 360 + diff;
      int bin = diff / (360 / mnBinNum);
      if (bin == 30) bin = 0;
      hist[bin].push_back(m);
    }
    std::vector<bool> good(mnBinNum, false);
    for (int c = 0; c < mnBinChoose; ++c) {
      int maxSize = 0, maxId = -1;
      for (int id = 0; id < mnBinNum; ++id)
        if (!good[id] && (int)hist[id].size() > maxSize) maxId = id, maxSize = (int)hist[id].size();
      if (maxId >= 0) good[maxId] = true;
    }
    std::vector<DMatch> ret;
    for (int id = 0; id < mnBinNum; ++id)
      if (good[id]) ret.insert(ret.end(), hist[id].begin(), hist[id].end());
    matches.swap(ret);
  }

 private:
  float mfRatio;
  bool mbCheckOri;
};

// Mirror of the reference's Optimizer (include/ORB_SLAM2/Optimizer.h:69-72): the g2o parts run on the device, graph
// construction and map bookkeeping stay in the caller.
class Optimizer {
 public:
  static inline const float deltaMono = std::sqrt(5.991f), deltaStereo = std::sqrt(7.815f);  // src/Optimizer.cc:1084-1085

  struct LocalMapResult {
    std::vector<double> poses, points, chi2;  // [nPoses][7] qx qy qz qw tx ty tz, [nPoints][3], [nEdges]
    std::vector<uint8_t> level, bad;           // setLevel(1) after the first round; final chi2 / depth test
    int32_t iterations[2] = {0, 0};
  };
  // Optimizer::OptimizeLocalMap (src/Optimizer.cc:336-391) on the graph `prob` describes
  static LocalMapResult OptimizeLocalMap(orbfe_ctx* ctx, const orbfe_ba_problem& prob, const std::vector<uint8_t>& poseFixed,
                                         const volatile bool* isStop = nullptr) {
    LocalMapResult r;
    r.poses.resize((size_t)prob.n_poses * 7), r.points.resize((size_t)prob.n_points * 3);
    r.chi2.resize((size_t)std::max(prob.n_edges, 1)), r.level.resize(r.chi2.size()), r.bad.resize(r.chi2.size());
    orbfe_ba_optimize_out o = {r.poses.data(), r.points.data(), r.level.data(), r.chi2.data(), r.bad.data(), r.iterations};
    check(ctx, orbfe_ba_local_optimize(ctx, &prob, poseFixed.empty() ? nullptr : poseFixed.data(), 5, 10, (const volatile uint8_t*)isStop, &o));
    r.chi2.resize(prob.n_edges), r.level.resize(prob.n_edges), r.bad.resize(prob.n_edges);
    return r;
  }
  // Optimizer::OptimizeLocalMap(pkframe, isStop) (src/Optimizer.cc:225-441) on a map loaded from map.pb: graph construction,
  // the two optimize() rounds on the device, the write-back policy; `map` is updated in place
  static mappb::LocalBaReport OptimizeLocalMap(orbfe_ctx* ctx, mappb::MapRec& map, uint64_t kfId, const orbfe_camera& cam,
                                               const volatile bool* isStop = nullptr) {
    mappb::LocalGraph g;
    if (!mappb::build_local_graph(map, kfId, g)) throw std::runtime_error("OptimizeLocalMap: keyframe id is not in the map");
    orbfe_ba_problem p{};
    p.n_poses = (int32_t)g.pose_kf_id.size(), p.n_points = (int32_t)g.point_id.size(), p.n_edges = (int32_t)g.edge_pose.size();
    if (p.n_edges == 0) return mappb::LocalBaReport{p.n_poses, g.n_group, p.n_points, 0, 0, 0, 0, 0};
    p.poses = g.poses.data(), p.points = g.points.data(), p.edge_pose = g.edge_pose.data(), p.edge_point = g.edge_point.data();
    p.meas = g.meas.data(), p.is_stereo = g.is_stereo.data(), p.info = g.info.data(), p.huber_delta = g.huber.data();
    p.fx = cam.fx, p.fy = cam.fy, p.cx = cam.cx, p.cy = cam.cy, p.bf = cam.bf;
    const LocalMapResult r = OptimizeLocalMap(ctx, p, g.pose_fixed, isStop);
    return mappb::apply_local_ba(map, g, r.poses.data(), r.points.data(), r.bad.data());
  }
  // the g2o part of Optimizer::OptimizePoseOnly (src/Optimizer.cc:33-178); returns edges - nBad, pose and inlier flags in place
  static int OptimizePoseOnly(orbfe_ctx* ctx, const std::vector<double>& pointsWorld, const std::vector<double>& meas /*u v uR*/,
                              const std::vector<double>& invSigma2, const std::vector<float>& sigma2, double fx, double fy, double cx,
                              double cy, double bf, double pose[7], std::vector<uint8_t>& inlier) {
    const int32_t n = (int32_t)invSigma2.size();
    inlier.assign((size_t)std::max(n, 1), 0);
    int32_t good = 0;
    double out[7];
    check(ctx, orbfe_pose_only_optimize(ctx, n, pointsWorld.data(), meas.data(), invSigma2.data(), sigma2.data(), pose, fx, fy, cx, cy, bf,
                                        out, inlier.data(), &good));
    for (int i = 0; i < 7; ++i) pose[i] = out[i];
    inlier.resize(n);
    return good;
  }
};

}  // namespace orbfe

// The drop-in classes with the reference's exact signatures (cv::Mat, cv::KeyPoint, Frame::SharedPtr, KeyFrame::SharedPtr) are in
// orbfe_dropin.hpp (needs <opencv2/core.hpp>, i.e. the reference's build).
