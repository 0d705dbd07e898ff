// orbfe_shim.hpp -- C++17 host-side mirror of the reference's operator interface for the hot path, written on
// the C-ABI of include/orbfe.h.  Header only; link with -lorbfe_hip.
//
// Two layers:
//   namespace orbfe              OpenCV-free classes (ImageView / std::vector) with the reference's names, argument
//                                order and error behaviour.  Always available.
//   namespace ORB_SLAM2_ROS2     the drop-in classes with the reference's exact signatures (cv::Mat, cv::KeyPoint),
//                                compiled only when ORBFE_WITH_OPENCV is defined, i.e. inside the reference's build:
//                                  ORBExtractor(const cv::Mat&, int, int, float, const std::string&, int, int)
//                                  void extract(std::vector<cv::KeyPoint>&, std::vector<cv::Mat>&)
//                                  const std::vector<cv::Mat>& getPyramid() const
//                                  static const std::vector<float>& getScaledFactors()
//                                (include/ORB_SLAM2/ORBExtractor.h:107-116) and the stereo-match entry used by
//                                Frame::createStereo (include/ORB_SLAM2/Frame.h:316-319).
//
// Error mapping (include/ORB_SLAM2/Error.h): ORBFE_EBADSIZE -> ImageSizeError, a missing template file ->
// FileNotOpenError, everything else -> std::runtime_error with orbfe_last_error().
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
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

#include "../../include/orbfe.h"

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

// Parses config/brief_template.txt exactly like ORBExtractor::initBriefTemplate (src/ORBExtractor.cc:242-267).
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
    float v[4];
    if (!(iss >> v[0] >> v[1] >> v[2] >> v[3])) continue;
    for (float f : v) out.push_back((int8_t)f);
  }
  if (out.size() != 1024) throw std::runtime_error("BRIEF template must hold 256 pairs: " + path);
  return out;
}

// One device context per (geometry, parameters, device); the reference constructs an extractor per image
// (src/Frame.cc:91-92) -- the device buffers behind it are shared and re-used.
class ContextPool {
 public:
  using Key = std::tuple<int, int, int, int, float, int, int, std::string, int, int>;
  static orbfe_ctx* get(int w, int h, int nFeatures, int nLevels, float scale, int maxTh, int minTh, const std::string& tplPath,
                        int device = 0, int maxImages = 2) {
    static std::mutex mu;
    static std::map<Key, std::shared_ptr<orbfe_ctx>> pool;
    std::lock_guard<std::mutex> lk(mu);
    Key key{w, h, nFeatures, nLevels, scale, maxTh, minTh, tplPath, device, maxImages};
    auto it = pool.find(key);
    if (it != pool.end()) return it->second.get();
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
    pool[key] = std::shared_ptr<orbfe_ctx>(ctx, orbfe_destroy);
    return ctx;
  }
};

class ORBExtractor {
 public:
  typedef std::shared_ptr<ORBExtractor> SharedPtr;
  static constexpr int mnBorderSize = 19;  // src/ORBExtractor.cc:523

  ORBExtractor(const ImageView& image, int nFeatures, int pyramidLevels, float scaleFactor, const std::string& bfTemFp,
               int maxThreshold, int minThreshold, int slot = 0)
      : mImage(image), mnFeats(nFeatures), mnLevels(pyramidLevels), mSlot(slot) {
    if (!image.data || image.cols <= 0 || image.rows <= 0) throw std::invalid_argument("ORBExtractor: empty image");
    mCtx = ContextPool::get(image.cols, image.rows, nFeatures, pyramidLevels, scaleFactor, maxThreshold, minThreshold, bfTemFp);
    mScales.resize(pyramidLevels);
    check(mCtx, orbfe_get_scale_factors(mCtx, mScales.data(), pyramidLevels));
  }

  void extract(std::vector<orbfe_keypoint>& keyPoints, std::vector<Descriptor>& descriptors) {
    keyPoints.resize(mnFeats);
    descriptors.resize(mnFeats);
    int32_t n = 0;
    if (mSlot == 0) {
      check(mCtx, orbfe_extract(mCtx, mImage.data, mImage.step, keyPoints.data(), descriptors.data()->data(), &n));
    } else {  // left/right of a stereo frame live in slots 0/1 of one context
      std::vector<const uint8_t*> imgs(mSlot + 1, mImage.data);
      std::vector<orbfe_keypoint> k((size_t)(mSlot + 1) * mnFeats);
      std::vector<uint8_t> d((size_t)(mSlot + 1) * mnFeats * 32);
      std::vector<int32_t> cnt(mSlot + 1);
      check(mCtx, orbfe_extract_batch(mCtx, mSlot + 1, imgs.data(), mImage.step, k.data(), d.data(), cnt.data()));
      n = cnt[mSlot];
      std::memcpy(keyPoints.data(), k.data() + (size_t)mSlot * mnFeats, sizeof(orbfe_keypoint) * n);
      std::memcpy(descriptors.data(), d.data() + (size_t)mSlot * mnFeats * 32, (size_t)32 * n);
    }
    keyPoints.resize(n);
    descriptors.resize(n);
  }

  // level `l` of the un-blurred pyramid, tight rows (what getPyramid()[l] holds in the reference)
  std::vector<uint8_t> getPyramidLevel(int l, int* w = nullptr, int* h = nullptr) const {
    orbfe_level_info li{};
    check(mCtx, orbfe_get_level_info(mCtx, l, &li));
    std::vector<uint8_t> out((size_t)li.width * li.height);
    check(mCtx, orbfe_get_pyramid(mCtx, mSlot, l, 0, out.data()));
    if (w) *w = li.width;
    if (h) *h = li.height;
    return out;
  }
  const std::vector<float>& getScaledFactors() const { return mScales; }
  orbfe_ctx* context() const { return mCtx; }

 private:
  ImageView mImage;
  int mnFeats, mnLevels, mSlot;
  orbfe_ctx* mCtx = nullptr;
  std::vector<float> mScales;
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
  // ORBMatcher::searchByStereo (src/ORBMatcher.cc:18-81) over the features extracted into slots 0 (left) / 1 (right);
  // fills mvFeatsRightU / mvDepths (-1 where unmatched) and returns the match count (Frame::mnN).
  int searchByStereo(orbfe_ctx* ctx, int nFeatures, int nLeft, float fx, float bf, std::vector<double>& rightU,
                     std::vector<double>& depths) const {
    std::vector<double> ru((size_t)std::max(nFeatures, 1)), dp((size_t)std::max(nFeatures, 1));
    int32_t n = 0;
    check(ctx, orbfe_stereo_match(ctx, 0, 1, fx, bf, ru.data(), dp.data(), &n, nullptr, nullptr));
    rightU.assign(ru.begin(), ru.begin() + nLeft);
    depths.assign(dp.begin(), dp.begin() + nLeft);
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
    check(ctx, orbfe_search_in_area(ctx, slot, n, uv.data(), radius.data(), minLevel.data(), maxLevel.data(),
                                    n ? desc[0].data() : nullptr, exclude ? exclude->data() : nullptr, m.bestIdx.data(),
                                    m.bestDist.data(), m.secondDist.data(), m.nCand.data()));
    return m;
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
                                         const volatile int32_t* isStop = nullptr) {
    LocalMapResult r;
    r.poses.resize((size_t)prob.n_poses * 7), r.points.resize((size_t)prob.n_points * 3);
    r.chi2.resize((size_t)std::max(prob.n_edges, 1)), r.level.resize(r.chi2.size()), r.bad.resize(r.chi2.size());
    orbfe_ba_optimize_out o = {r.poses.data(), r.points.data(), r.level.data(), r.chi2.data(), r.bad.data(), r.iterations};
    check(ctx, orbfe_ba_local_optimize(ctx, &prob, poseFixed.empty() ? nullptr : poseFixed.data(), 5, 10, isStop, &o));
    r.chi2.resize(prob.n_edges), r.level.resize(prob.n_edges), r.bad.resize(prob.n_edges);
    return r;
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

#ifdef ORBFE_WITH_OPENCV
#include <opencv2/core.hpp>
namespace ORB_SLAM2_ROS2 {
// Drop-in for include/ORB_SLAM2/ORBExtractor.h:100-160 -- same constructor, extract(), getPyramid(), statics.
class ORBExtractor {
 public:
  typedef std::shared_ptr<ORBExtractor> SharedPtr;
  ORBExtractor(const cv::Mat& image, int nFeatures, int pyramidLevels, float scaleFactor, const std::string& bfTemFp, int maxThreshold,
               int minThreshold)
      : mImpl(orbfe::ImageView{image.data, image.cols, image.rows, image.step}, nFeatures, pyramidLevels, scaleFactor, bfTemFp,
              maxThreshold, minThreshold) {
    CV_Assert(image.type() == CV_8UC1);
    mnLevels = pyramidLevels;
    mfScaledFactor = scaleFactor;
    mvfScaledFactors = mImpl.getScaledFactors();
  }
  void extract(std::vector<cv::KeyPoint>& keyPoints, std::vector<cv::Mat>& descriptors) {
    std::vector<orbfe_keypoint> k;
    std::vector<orbfe::Descriptor> d;
    mImpl.extract(k, d);
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
    keyPoints.resize(k.size());
    std::memcpy((void*)keyPoints.data(), k.data(), sizeof(orbfe_keypoint) * k.size());
    descriptors.clear();
    for (auto& row : d) descriptors.push_back(cv::Mat(1, 32, CV_8U, row.data()).clone());  // one 1x32 Mat per keypoint (:402-412)
    mvPyramids.clear();
    for (int l = 0; l < mnLevels; ++l) {
      int w, h;
      auto buf = mImpl.getPyramidLevel(l, &w, &h);
      mvPyramids.push_back(cv::Mat(h, w, CV_8U, buf.data()).clone());
    }
  }
  const std::vector<cv::Mat>& getPyramid() const { return mvPyramids; }
  static const std::vector<float>& getScaledFactors() { return mvfScaledFactors; }
  static inline int mnLevels = 0, mnBorderSize = 19;
  static inline float mfScaledFactor = 0.f;

 private:
  orbfe::ORBExtractor mImpl;
  std::vector<cv::Mat> mvPyramids;
  static inline std::vector<float> mvfScaledFactors;
};
}  // namespace ORB_SLAM2_ROS2
#endif  // ORBFE_WITH_OPENCV
