// orbfe_dropin.hpp -- the reference's own signatures over the C-ABI (include/orbfe.h), for the reference's build (needs
// <opencv2/core.hpp>; nothing else of OpenCV, no g2o, no Eigen).
//
//   ORB_SLAM2_ROS2::ORBExtractor          include/ORB_SLAM2/ORBExtractor.h:100-160: ORBExtractor(const cv::Mat&, int, int, float, const
//                                         std::string&, int, int), extract(std::vector<cv::KeyPoint>&, std::vector<cv::Mat>&), getPyramid(),
//                                         getScaledFactors(), the public statics.  Two objects may extract on two threads (Frame.cc:100-105).
//   orbfe::dropin::searchByStereo         the body of `int ORBMatcher::searchByStereo(Frame::SharedPtr)` (ORBMatcher.h:38, src/ORBMatcher.cc:18-81)
//   orbfe::dropin::createStereo           the two extract() threads of `Frame::Frame` (src/Frame.cc:100-105) and the searchByStereo of
//                                         `Frame::createStereo` (Frame.h:319) as ONE device call (ORBExtractor::extractStereo)
//   orbfe::dropin::descDistance           `static int ORBMatcher::descDistance(const cv::Mat&, const cv::Mat&)` (ORBMatcher.h:77)
//   orbfe::dropin::OptimizePoseOnly       the body of `static int Optimizer::OptimizePoseOnly(Frame::SharedPtr)` (Optimizer.h:72, src/Optimizer.cc:33-203)
//   orbfe::dropin::OptimizeLocalMap       the body of `static void Optimizer::OptimizeLocalMap(KeyFrame::SharedPtr, bool&)` (Optimizer.h:69,
//                                         src/Optimizer.cc:225-442)
//   orbfe::dropin::searchByBow            the body of `int ORBMatcher::searchByBow(VirtualFrame::SharedPtr, VirtualFrame::SharedPtr,
//                                         std::vector<cv::DMatch>&, bool, bool)` (ORBMatcher.h:42, src/ORBMatcher.cc:170-253)
//   orbfe::dropin::searchByProjection x2  frame <- frame (ORBMatcher.h:49, src/ORBMatcher.cc:265-347) and frame <- map points (ORBMatcher.h:52,
//                                         :561-612)
//   orbfe::dropin::frameRGBD              the tail of `Frame::Frame` for RGB-D input (src/Frame.cc:130-131, :139-157)
//   orbfe::dropin::createRGBD             extract() AND that tail as ONE device call (ORBExtractor::extractRGBD -> orbfe_frame_rgbd_image)
//   orbfe::dropin::searchBySim3 x2, searchForTriangulation, fuse x2   the back-end matchers (ORBMatcher.h:55-67, src/ORBMatcher.cc:424-559, 691-787)
//   orbfe::dropin::trackMotionModel       the middle of `Tracking::trackMotionModel` (src/Tracking.cc:385-396): searchByProjection(frame, last frame,
//                                         15) [+ 30] + OptimizePoseOnly(frame) as ONE device call
//   orbfe::dropin::trackLocalMap          the middle of `Tracking::trackLocalMap` (src/Tracking.cc:650-658): searchByProjection(frame, local map
//                                         points, th) + OptimizePoseOnly(frame) as ONE device call
//
// The three bodies are templates over the reference's Frame / KeyFrame / MapPoint / Camera types (they only use the accessors
// the reference's own function bodies use), so this header does not have to see the reference's headers; INTEGRATION.md shows the
// one-line member functions a maintainer writes around them.  tests/cpp/test_dropin.cpp instantiates them with stand-in classes of
// the same accessors over tests/cpp/stubs/opencv2/core.hpp -- that checks the templates compile and that their logic agrees with the
// array-level path; it pins nothing about OpenCV.
#pragma once
#include <opencv2/core.hpp>

#include <atomic>
#include <set>

#include "orbfe_shim.hpp"

namespace ORB_SLAM2_ROS2 {
// Drop-in for include/ORB_SLAM2/ORBExtractor.h:100-160 -- same constructor, extract(), getPyramid(), statics.
class ORBExtractor {
 public:
  typedef std::shared_ptr<ORBExtractor> SharedPtr;
  ORBExtractor(const cv::Mat& image, int nFeatures, int pyramidLevels, float scaleFactor, const std::string& bfTemFp, int maxThreshold,
               int minThreshold)
      : mImage(image),  // a header on the caller's pixels, like the reference's mvPyramids[0] source (Frame keeps mLeftIm alive)
        mImpl(orbfe::ImageView{mImage.data, mImage.cols, mImage.rows, (size_t)mImage.step}, nFeatures, pyramidLevels, scaleFactor, bfTemFp,
              maxThreshold, minThreshold) {
    CV_Assert(image.type() == CV_8UC1);
    static std::once_flag once;  // the reference's unsynchronised static-init flags (ORBExtractor.cc:219,244,283)
    std::call_once(once, [&] {
      mnLevels = pyramidLevels;
      mfScaledFactor = scaleFactor;
      mvfScaledFactors = mImpl.getScaledFactors();
    });
  }
  void extract(std::vector<cv::KeyPoint>& keyPoints, std::vector<cv::Mat>& descriptors) {
    std::vector<orbfe_keypoint> k;
    std::vector<orbfe::Descriptor> d;
    mImpl.extract(k, d);
    deliver(k, d, keyPoints, descriptors);
  }
  // Both extractions of a stereo frame and its stereo match as ONE device call (orbfe_frame_stereo_slots): what Frame::createStereo does
  // with two extract() threads (src/Frame.cc:100-105) and ORBMatcher::searchByStereo (include/ORB_SLAM2/Frame.h:319), for a caller that
  // changes those lines (orbfe::dropin::createStereo below is the body).  `this` is the left extractor.
  int extractStereo(ORBExtractor& right, float fx, float bf, std::vector<cv::KeyPoint>& kpsLeft, std::vector<cv::Mat>& descLeft,
                    std::vector<cv::KeyPoint>& kpsRight, std::vector<cv::Mat>& descRight, std::vector<double>& rightU,
                    std::vector<double>& depths) {
    std::vector<orbfe_keypoint> kl, kr;
    std::vector<orbfe::Descriptor> dl, dr;
    const int n = mImpl.extractStereo(right.mImpl, fx, bf, kl, dl, kr, dr, rightU, depths);
    deliver(kl, dl, kpsLeft, descLeft);
    right.deliver(kr, dr, kpsRight, descRight);
    return n;
  }

  // The extraction and the RGB-D tail of the Frame constructor (src/Frame.cc:125-158) as ONE device call (orbfe_frame_rgbd_image): what the
  // constructor does with extract(), Camera::undistortPoints and the depth loop (orbfe::dropin::createRGBD below is the body).
  void extractRGBD(const orbfe_camera& cam, const cv::Mat& depthImg, float dScale, std::vector<cv::KeyPoint>& undistorted,
                   std::vector<cv::Mat>& descriptors, std::vector<double>& depths, std::vector<double>& rightU) {
    if (depthImg.type() != CV_32F && depthImg.type() != CV_16U) throw std::invalid_argument("extractRGBD: depth image must be CV_16U or CV_32F");
    std::vector<orbfe_keypoint> k;
    std::vector<orbfe::Descriptor> d;
    mImpl.extractRGBD(cam, depthImg.data, depthImg.type() == CV_32F ? 1 : 0, (size_t)depthImg.step, dScale, k, d, depths, rightU);
    deliver(k, d, undistorted, descriptors);
  }

 private:
  void deliver(const std::vector<orbfe_keypoint>& k, std::vector<orbfe::Descriptor>& d, std::vector<cv::KeyPoint>& keyPoints,
               std::vector<cv::Mat>& descriptors) {
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
    keyPoints.resize(k.size());
    std::memcpy((void*)keyPoints.data(), k.data(), sizeof(orbfe_keypoint) * k.size());
    descriptors.clear();
    descriptors.reserve(d.size());
#ifdef ORBFE_DROPIN_CLONE_DESCRIPTORS
    for (auto& row : d) descriptors.push_back(cv::Mat(1, 32, CV_8U, row.data()).clone());  // one allocation per keypoint, as the reference makes them (:402-412)
#else
    // one 1x32 Mat per keypoint as in the reference (:402-412), but as ROW HEADERS of one n x 32 block: one allocation per image instead of
    // 2000 (~0.1 ms of a 0.3 ms call).  A descriptor that outlives its frame (a MapPoint's copy made with clone() / copyTo() does not)
    // keeps the frame's whole block alive; define ORBFE_DROPIN_CLONE_DESCRIPTORS for independent allocations.
    if (!d.empty()) {
      cv::Mat all((int)d.size(), 32, CV_8U);
      std::memcpy(all.data, d[0].data(), d.size() * 32);
      for (int i = 0; i < (int)d.size(); ++i) descriptors.push_back(all.row(i));
    }
#endif
    std::lock_guard<std::mutex> lk(mPyrMutex);
    mvPyramids.clear();  // a new extraction: the planes are fetched again when somebody asks
  }

 public:
  // The reference fills mvPyramids in the constructor; its only reader is ORBMatcher::searchByStereo (src/ORBMatcher.cc:27-28), which
  // runs on the device here.  The 8 planes (1.4 MB) therefore cross PCIe only if somebody calls this.
  const std::vector<cv::Mat>& getPyramid() const {
    std::lock_guard<std::mutex> lk(mPyrMutex);
    if (mvPyramids.empty())
      for (int l = 0; l < mImpl.levels(); ++l) {
        int w = 0, h = 0;
        auto buf = const_cast<orbfe::ORBExtractor&>(mImpl).getPyramidLevel(l, &w, &h);
        mvPyramids.push_back(cv::Mat(h, w, CV_8U, buf.data()).clone());
      }
    return mvPyramids;
  }
  static const std::vector<float>& getScaledFactors() { return mvfScaledFactors; }
  static inline int mnLevels = 0, mnBorderSize = 19;
  static inline float mfScaledFactor = 0.f;

  const orbfe::ORBExtractor& device() const { return mImpl; }  // the slot-holding object (for the stereo match)

 private:
  cv::Mat mImage;
  orbfe::ORBExtractor mImpl;
  mutable std::mutex mPyrMutex;
  mutable std::vector<cv::Mat> mvPyramids;
  static inline std::vector<float> mvfScaledFactors;
};
}  // namespace ORB_SLAM2_ROS2

namespace orbfe {
namespace dropin {

// Contexts for the solver entry points: one per calling thread role, since OptimizePoseOnly runs on the Tracking thread while
// OptimizeLocalMap runs on the LocalMapping thread (System.cc:128) and one context serves one thread at a time.
inline orbfe_ctx* solverContext(int role /*0: tracking, 1: local mapping*/) {
  return ContextPool::get(160, 120, 16, 1, 1.2f, 20, 7, "", 0, 1 + role);  // a token geometry; the BA calls only use its stream and scratch
}

// ORBMatcher::descDistance (src/ORBMatcher.cc:941-956) on two 1x32 CV_8U rows
inline int descDistance(const cv::Mat& a, const cv::Mat& b) {
  int d = 0;
  for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a.data[i] ^ b.data[i]));
  return d;
}

// A context of its own for the guided searches of every calling thread (Tracking, LocalMapping and LoopClosing all match: System.cc:119-129);
// the searches below upload the target's feature set and pass the target frame's own bounds, so the context's geometry is a token.
inline orbfe_ctx* matcherContext() {
  static std::atomic<int> nextRole{0};
  thread_local const int role = nextRole.fetch_add(1) % 16;
  return ContextPool::get(160, 120, 16, 1, 1.2f, 20, 7, "", 0, 1, 1000 + role);
}

inline void matToPose(const cv::Mat& Rcw, const cv::Mat& tcw, double out[7]) {  // Converter::ConvertTcw2SE3 (src/Optimizer.cc:628-641)
  float R[9], t[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) R[3 * r + c] = Rcw.template at<float>(r, c);
    t[r] = tcw.template at<float>(r, 0);
  }
  mappb::tcw_to_se3(R, t, out);
}
inline cv::Mat poseToMat(const double p[7]) {  // Converter::ConvertSE32Tcw (:649-672)
  float R[9], t[3];
  mappb::se3_to_tcw(p, R, t);
  cv::Mat T(4, 4, CV_32F);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T.template at<float>(r, c) = R[3 * r + c];
    T.template at<float>(r, 3) = t[r];
    T.template at<float>(3, r) = 0.0f;
  }
  T.template at<float>(3, 3) = 1.0f;
  return T;
}

// The bodies.  They read and write what the reference's own member functions read and write, protected members included
// (mvFeatsLeft, mvLeftDescriptor, mvpMapPoints, mFeatVec, mvDepths, mvFeatsRightU, mpExtractorLeft ...): ORBMatcher / Optimizer are
// friends of the frame classes (Frame.h:23, :302-303), but friendship does not reach a function they call -- so the bodies are static
// members of ONE struct, and the frame classes name it next to their existing friend lines:
//     friend struct orbfe::dropin::Bodies;       // VirtualFrame, Frame, KeyFrame (beside `friend class ORBMatcher;`)
// tests/cpp/test_dropin.cpp keeps the stand-in classes' members protected with exactly that line.  The free functions of the same names
// after the struct forward to it.
struct Bodies {
// int ORBMatcher::searchByStereo(Frame::SharedPtr pFrame)  (src/ORBMatcher.cc:18-81).  Uses pFrame->mvFeatsLeft, mvDepths, mvFeatsRightU,
// mpExtractorLeft / mpExtractorRight (ORBMatcher is a friend of Frame, Frame.h:302-303) and Camera::mfFx / mfBf.
template <class CameraT, class FramePtr>
static int searchByStereo(FramePtr pFrame) {
  const size_t nLeft = pFrame->mvFeatsLeft.size();
  std::vector<double> ru, dp;
  const int n = orbfe::ORBMatcher().searchByStereo(pFrame->mpExtractorLeft->device(), pFrame->mpExtractorRight->device(), CameraT::mfFx,
                                                   CameraT::mfBf, ru, dp);
  ru.resize(nLeft, -1.0);
  dp.resize(nLeft, -1.0);
  pFrame->mvFeatsRightU.assign(ru.begin(), ru.end());
  pFrame->mvDepths.assign(dp.begin(), dp.end());
  return n;
}

// The device work of Frame::createStereo (include/ORB_SLAM2/Frame.h:313-323) as one call: stands for the two extract() threads of the
// Frame constructor (src/Frame.cc:100-105) AND the searchByStereo of Frame.h:319 -- fills mvFeatsLeft / mvLeftDescriptor / mvFeatsRight /
// mRightDescriptor / mvFeatsRightU / mvDepths and returns the match count (Frame::mnN).  INTEGRATION.md 2b shows the two edits.
template <class CameraT, class FrameT>
static int createStereo(FrameT* self) {
  return self->mpExtractorLeft->extractStereo(*self->mpExtractorRight, CameraT::mfFx, CameraT::mfBf, self->mvFeatsLeft, self->mvLeftDescriptor,
                                              self->mvFeatsRight, self->mRightDescriptor, self->mvFeatsRightU, self->mvDepths);
}

// static int Optimizer::OptimizePoseOnly(Frame::SharedPtr pFrame)  (src/Optimizer.cc:33-203)
template <class CameraT, class FramePtr>
static int OptimizePoseOnly(FramePtr pFrame) {
  auto mapPoints = pFrame->getMapPoints();
  const auto& kps = pFrame->getLeftKeyPoints();
  const size_t N = pFrame->mvFeatsLeft.size();
  std::vector<uint8_t> inLier(N, 1);
  std::vector<int> edgeOf(N, -1);
  std::vector<cv::Mat> mapPointPoses;
  std::vector<double> Xw, meas, info;
  std::vector<float> sigma2;
  int edges = 0;
  for (size_t idx = 0; idx < mapPoints.size(); ++idx) {
    auto& pMp = mapPoints[idx];
    cv::Mat pos;
    if (pMp && !pMp->isBad()) {
      pos = pMp->getPos();
      const double rightU = pFrame->getRightU(idx);
      const auto& kp = kps[idx];
      for (int a = 0; a < 3; ++a) Xw.push_back((double)pos.template at<float>(a));
      meas.push_back((double)kp.pt.x), meas.push_back((double)kp.pt.y), meas.push_back(rightU < 0 ? -1.0 : rightU);  // < 0: mono edge (:77)
      info.push_back((double)pFrame->getScaledFactorInv2(kp.octave));                                                // :85, :106
      sigma2.push_back(pFrame->getScaledFactor2(kp.octave));                                                         // :136, :157
      edgeOf[idx] = edges++;
    } else {
      inLier[idx] = 0;
    }
    mapPointPoses.push_back(pos);
  }
  double pose[7], out[7];
  matToPose(pFrame->mRcw, pFrame->mtcw, pose);
  std::vector<uint8_t> edgeInlier((size_t)std::max(edges, 1), 0);
  int32_t good = 0;
  orbfe_ctx* ctx = solverContext(0);
  check(ctx, orbfe_pose_only_optimize(ctx, edges, Xw.data(), meas.data(), info.data(), sigma2.data(), pose, CameraT::mfFx, CameraT::mfFy,
                                      CameraT::mfCx, CameraT::mfCy, CameraT::mfBf, out, edgeInlier.data(), &good));
  int nBad = edges - good;
  for (size_t idx = 0; idx < N; ++idx)
    if (edgeOf[idx] >= 0) inLier[idx] = edgeInlier[(size_t)edgeOf[idx]];
  for (size_t idx = 0; idx < inLier.size(); ++idx) {  // :180-190 -- with the frame's pose as it still is (setPose comes last)
    if (!inLier[idx]) continue;
    bool isPositive = false;
    auto uv = pFrame->project2UV(mapPointPoses[idx], isPositive);
    if (!isPositive || uv.x > pFrame->mfMaxU || uv.x < 0 || uv.y > pFrame->mfMaxV || uv.y < 0) {
      inLier[idx] = 0;
      ++nBad;
    }
  }
  for (size_t idx = 0; idx < inLier.size(); ++idx) {
    if (!inLier[idx])
      pFrame->mvpMapPoints[idx] = nullptr;
    else
      pFrame->mvpMapPoints[idx]->addInlierInTrack();
  }
  pFrame->setPose(poseToMat(out));
  return edges - nBad;
}

// static void Optimizer::OptimizeLocalMap(KeyFrame::SharedPtr pkframe, bool& isStop)  (src/Optimizer.cc:225-442).
// Vertex / edge order: the reference walks a std::set of map-point pointers (address order); here the map points go in ascending
// getID() so that a run is reproducible -- the order only permutes the sums of the normal equations.
template <class CameraT, class KeyFramePtr>
static void OptimizeLocalMap(KeyFramePtr pkframe, bool& isStop) {
  using KeyFrameT = typename KeyFramePtr::element_type;
  auto group = pkframe->getConnectedKfs(0);
  group.push_back(pkframe);
  using MapPointPtr = typename std::decay<decltype(pkframe->getMapPoints()[0])>::type;

  std::vector<KeyFramePtr> frames;  // vertex order: the free group, then the fixed observers as they are met
  std::map<KeyFrameT*, int> vertexOf;
  std::vector<uint8_t> fixed;
  std::vector<double> poses;
  auto addFrame = [&](const KeyFramePtr& kf, bool fix) {
    cv::Mat Rcw, tcw;
    kf->getPose(Rcw, tcw);
    double p[7];
    matToPose(Rcw, tcw, p);
    poses.insert(poses.end(), p, p + 7);
    fixed.push_back(fix ? 1 : 0);
    frames.push_back(kf);
    vertexOf[kf.get()] = (int)frames.size() - 1;
    return (int)frames.size() - 1;
  };
  std::map<size_t, MapPointPtr> groupMps;  // by id
  for (auto& kf : group) {
    if (vertexOf.count(kf.get())) continue;
    addFrame(kf, kf->getID() == 0);                    // :248
    for (auto& pMp : kf->getMapPoints())
      if (pMp && !pMp->isBad()) groupMps.insert({(size_t)pMp->getID(), pMp});
  }
  const int nGroup = (int)frames.size();

  std::vector<MapPointPtr> landmarks;
  std::vector<double> points, meas, info, huber;
  std::vector<int32_t> edgePose, edgePoint;
  std::vector<uint8_t> isStereo;
  struct EdgeRec {
    MapPointPtr mp;
    KeyFramePtr kf;
    size_t idx;
  };
  std::vector<EdgeRec> edgeDB;
  for (auto& item : groupMps) {
    const MapPointPtr& pMp = item.second;
    const int pv = (int)landmarks.size();
    landmarks.push_back(pMp);
    const cv::Mat pos = pMp->getPos();
    for (int a = 0; a < 3; ++a) points.push_back((double)pos.template at<float>(a));  // ConvertPw2Vector3 (:683-689)
    auto obs = pMp->getObservation();
    for (auto& o : obs) {
      KeyFramePtr pkf = o.first.lock();
      if (!pkf || pkf->isBad()) continue;
      auto it = vertexOf.find(pkf.get());
      const int v = it != vertexOf.end() ? it->second : addFrame(pkf, true);  // :281-290
      const double rightU = pkf->getRightU(o.second);
      const auto& kp = pkf->getLeftKeyPoint(o.second);
      edgePose.push_back(v), edgePoint.push_back(pv);
      meas.push_back((double)kp.pt.x), meas.push_back((double)kp.pt.y);
      if (rightU > 0) {  // :296-312
        meas.push_back(rightU), isStereo.push_back(1);
        info.push_back((double)pkf->getScaledFactorInv2(kp.octave)), huber.push_back((double)Optimizer::deltaStereo);
      } else {           // :314-329 (getScaledFactorInv, not squared: quirk Q9)
        meas.push_back(0.0), isStereo.push_back(0);
        info.push_back((double)pkf->getScaledFactorInv(kp.octave)), huber.push_back((double)Optimizer::deltaMono);
      }
      edgeDB.push_back({pMp, pkf, o.second});
    }
  }
  if (isStop) return;  // :331-332

  orbfe_ba_problem prob{};
  prob.n_poses = (int32_t)frames.size(), prob.n_points = (int32_t)landmarks.size(), prob.n_edges = (int32_t)edgeDB.size();
  prob.poses = poses.data(), prob.points = points.data(), prob.edge_pose = edgePose.data(), prob.edge_point = edgePoint.data();
  prob.meas = meas.data(), prob.is_stereo = isStereo.data(), prob.info = info.data(), prob.huber_delta = huber.data();
  prob.fx = CameraT::mfFx, prob.fy = CameraT::mfFy, prob.cx = CameraT::mfCx, prob.cy = CameraT::mfCy, prob.bf = CameraT::mfBf;
  const Optimizer::LocalMapResult r = Optimizer::OptimizeLocalMap(solverContext(1), prob, fixed, (const volatile bool*)&isStop);

  std::map<KeyFramePtr, std::vector<std::pair<MapPointPtr, size_t>>> vToProcess;  // :363-388
  for (size_t e = 0; e < edgeDB.size(); ++e)
    if (r.bad[e]) vToProcess[edgeDB[e].kf].push_back({edgeDB[e].mp, edgeDB[e].idx});
  int nBad = 0;
  for (auto& item : vToProcess) {
    int nGoodMp = 0;
    for (auto& pMp : item.first->getMapPoints())
      if (pMp && !pMp->isBad()) ++nGoodMp;
    if (item.second.size() / (float)nGoodMp > 0.3) ++nBad;
  }
  if (nBad / (vToProcess.size() + 1e-5) > 0.2) return;  // bSetAndErase = false

  for (auto& item : vToProcess)
    for (auto& era : item.second) {
      item.first->setMapPoint(era.second, nullptr);
      era.first->eraseObservetion(item.first);
    }
  for (int v = 0; v < nGroup; ++v)
    if (frames[v] && !frames[v]->isBad()) frames[v]->setPose(poseToMat(r.poses.data() + (size_t)v * 7));
  for (size_t p = 0; p < landmarks.size(); ++p) {
    auto& pMp = landmarks[p];
    if (pMp && !pMp->isBad() && pMp->isInMap()) {
      cv::Mat pos(3, 1, CV_32F);  // ConvertVector32Pw (:697-703)
      for (int a = 0; a < 3; ++a) pos.template at<float>(a) = (float)r.points[p * 3 + a];
      pMp->setPos(pos);
      pMp->updateDescriptor();
      pMp->updateNormalAndDepth();
    }
  }
  KeyFrameT::updateConnections(pkframe);
}

  // ---- the per-frame guided matchers (include/ORB_SLAM2/ORBMatcher.h:42,49,52) ---------------------------------------------------------
  static void descRows(const std::vector<cv::Mat>& rows, std::vector<Descriptor>& out) {
    out.resize(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) std::memcpy(out[i].data(), rows[i].data, 32);
  }
  static void toKeypoints(const std::vector<cv::KeyPoint>& kps, std::vector<orbfe_keypoint>& out) {
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
    out.resize(kps.size());
    if (!kps.empty()) std::memcpy((void*)out.data(), kps.data(), sizeof(orbfe_keypoint) * kps.size());
  }
  // ORBMatcher::verifyAngle (src/ORBMatcher.cc:1013-1051)
  static void verifyAngle(std::vector<cv::DMatch>& matches, const std::vector<cv::KeyPoint>& keyPoints1,
                          const std::vector<cv::KeyPoint>& keyPoints2) {
    const int nBins = orbfe::ORBMatcher::mnBinNum;
    std::vector<std::vector<cv::DMatch>> hist((size_t)nBins);
    for (const auto& dmatch : matches) {
      float diff = keyPoints1[dmatch.queryIdx].angle - keyPoints2[dmatch.trainIdx].angle;
      diff = diff >= 0 ? diff : 360 + diff;
      int bin = diff / (360 / nBins);
      if (bin == 30) bin = 0;
      hist[(size_t)bin].push_back(dmatch);
    }
    std::set<std::size_t> goodBinIds;
    for (int c = 0; c < orbfe::ORBMatcher::mnBinChoose; ++c) {
      int maxSize = 0;
      std::size_t maxId = 0;
      bool bInit = false;
      for (std::size_t id = 0; id < (std::size_t)nBins; ++id) {
        if (goodBinIds.count(id)) continue;
        if ((int)hist[id].size() > maxSize) maxId = id, maxSize = (int)hist[id].size(), bInit = true;
      }
      if (bInit) goodBinIds.insert(maxId);
    }
    std::vector<cv::DMatch> ret;
    for (auto id : goodBinIds) ret.insert(ret.end(), hist[id].begin(), hist[id].end());
    matches.swap(ret);
  }
  // ORBMatcher::setMapPoints (src/ORBMatcher.cc:815-830)
  template <class MapPointsA, class MapPointsB>
  static void setMapPoints(MapPointsA& toMatchMps, MapPointsB& matchMps, const std::vector<cv::DMatch>& matches) {
    for (const auto& dmatch : matches) {
      auto& matchPMp = matchMps[dmatch.trainIdx];
      if (matchPMp && !matchPMp->isBad()) {
        matchPMp->addMatchInTrack();
        toMatchMps[dmatch.queryIdx] = matchPMp;
      } else {
        matchPMp = nullptr;
      }
    }
  }

  // int ORBMatcher::searchByBow(VirtualFrame::SharedPtr pFrame, VirtualFrame::SharedPtr pKframe, std::vector<cv::DMatch>& matches,
  //                             bool bAddMPs, bool bLoop)   (ORBMatcher.h:42, src/ORBMatcher.cc:170-253); mfRatio / mbCheckOri: the matcher's
  // members.  All getBestMatch scans of the call (one per keyframe feature that shares a vocabulary node with the frame) run as ONE
  // orbfe_match_bruteforce launch over CSR candidate lists in the reference's order.
  template <class FramePtrF, class FramePtrK>
  static int searchByBow(FramePtrF pFrame, FramePtrK pKframe, std::vector<cv::DMatch>& matches, bool bAddMPs, bool bLoop, float mfRatio,
                         bool mbCheckOri) {
    pFrame->computeBow();
    pKframe->computeBow();
    auto mapPointsF = pFrame->getMapPoints();
    auto mapPointsKF = pKframe->getMapPoints();
    std::map<unsigned, std::vector<unsigned>> fvF, fvK;  // DBoW3::FeatureVector: node id -> feature ids, ordered by node id
    for (const auto& node : pFrame->mFeatVec) fvF[(unsigned)node.first].assign(node.second.begin(), node.second.end());
    for (const auto& node : pKframe->mFeatVec) fvK[(unsigned)node.first].assign(node.second.begin(), node.second.end());
    auto flags = [](decltype(mapPointsF)& mps, std::vector<uint8_t>& good, std::vector<uint8_t>& inMap) {
      good.assign(mps.size(), 0), inMap.assign(mps.size(), 0);
      for (size_t i = 0; i < mps.size(); ++i) {
        good[i] = mps[i] && !mps[i]->isBad();
        inMap[i] = good[i] && mps[i]->isInMap();
      }
    };
    std::vector<uint8_t> goodF, inMapF, goodK, inMapK;
    flags(mapPointsF, goodF, inMapF);
    flags(mapPointsKF, goodK, inMapK);
    std::vector<Descriptor> descF, descK;
    descRows(pFrame->mvLeftDescriptor, descF);
    descRows(pKframe->mvLeftDescriptor, descK);
    const std::vector<float> noAngles;  // the orientation check runs below, on the caller's whole `matches` as the reference does
    if (!fvF.empty() && !fvK.empty()) {
      const auto found = orbfe::ORBMatcher(mfRatio, false).searchByBow(matcherContext(), descF, descK, fvF, fvK, goodF, inMapF, goodK, inMapK,
                                                                       noAngles, noAngles, bAddMPs, bLoop);
      for (const auto& m : found) matches.emplace_back(m.queryIdx, m.trainIdx, (float)m.distance);
    }
    if (mbCheckOri) verifyAngle(matches, pFrame->getLeftKeyPoints(), pKframe->getLeftKeyPoints());
    if (!bAddMPs && !bLoop) setMapPoints(pFrame->mvpMapPoints, mapPointsKF, matches);
    return (int)matches.size();
  }

  // what findFeaturesInArea needs of the TARGET frame, uploaded with the call (orbfe_search_in_area_features_ex): its features, its
  // descriptors and its undistorted bounds (VirtualFrame::mfMinU .. mfMaxV)
  struct Target {
    std::vector<orbfe_keypoint> kps;
    std::vector<Descriptor> desc;
    float bounds[4];
  };
  template <class FramePtr>
  static void target(FramePtr f, Target& t) {
    toKeypoints(f->mvFeatsLeft, t.kps);
    descRows(f->mvLeftDescriptor, t.desc);
    t.bounds[0] = f->mfMinU, t.bounds[1] = f->mfMaxU, t.bounds[2] = f->mfMinV, t.bounds[3] = f->mfMaxV;
  }
  struct Queries {
    std::vector<int> who;
    std::vector<float> uv, radius;
    std::vector<int8_t> lo, hi;
    std::vector<Descriptor> desc;
    void add(int id, float u, float v, float r, int minLevel, int maxLevel, const cv::Mat& d) {
      who.push_back(id), uv.push_back(u), uv.push_back(v), radius.push_back(r), lo.push_back((int8_t)minLevel), hi.push_back((int8_t)maxLevel);
      desc.emplace_back();
      std::memcpy(desc.back().data(), d.data, 32);
    }
  };
  static orbfe::ORBMatcher::AreaMatch searchTarget(const Target& t, const Queries& q, const std::vector<uint8_t>* exclude,
                                                   std::vector<int32_t>* excludedHits) {
    const int32_t n = (int32_t)q.who.size();
    orbfe::ORBMatcher::AreaMatch m;
    m.bestIdx.resize(n), m.bestDist.resize(n), m.secondDist.resize(n), m.nCand.resize(n);
    if (excludedHits) excludedHits->assign(std::max<size_t>(t.kps.size(), 1), 0);
    orbfe_ctx* ctx = matcherContext();
    check(ctx, orbfe_search_in_area_features_ex(ctx, (int32_t)t.kps.size(), t.kps.data(), t.desc.empty() ? nullptr : t.desc[0].data(), t.bounds, n,
                                                q.uv.data(), q.radius.data(), q.lo.data(), q.hi.data(), n ? q.desc[0].data() : nullptr,
                                                exclude ? exclude->data() : nullptr, m.bestIdx.data(), m.bestDist.data(), m.secondDist.data(),
                                                m.nCand.data(), excludedHits ? excludedHits->data() : nullptr));
    return m;
  }

  // forward / backward motion between two frames (src/ORBMatcher.cc:270-283)
  template <class CameraT, class FramePtr1, class FramePtr2>
  static void motionDirection(FramePtr1 pFrame1, FramePtr2 pFrame2, bool& up, bool& down) {
    cv::Mat Rcw1, tcw1, Rcw2, tcw2;
    pFrame1->getPose(Rcw1, tcw1);
    pFrame2->getPose(Rcw2, tcw2);
    // twc1 = -Rcw1.t() * tcw1;  tlc = Rcw2 * twc1 + tcw2   (cv::gemm on 3x3 / 3x1 floats: products summed in float, alpha / beta in double)
    float twc1[3], z = 0.f;
    for (int r = 0; r < 3; ++r) {
      const float sm = Rcw1.template at<float>(0, r) * tcw1.template at<float>(0, 0) + Rcw1.template at<float>(1, r) * tcw1.template at<float>(1, 0) +
                       Rcw1.template at<float>(2, r) * tcw1.template at<float>(2, 0);
      twc1[r] = (float)(-1.0 * (double)sm);
    }
    {
      const float sm = Rcw2.template at<float>(2, 0) * twc1[0] + Rcw2.template at<float>(2, 1) * twc1[1] + Rcw2.template at<float>(2, 2) * twc1[2];
      z = (float)((double)sm + (double)tcw2.template at<float>(2, 0));
    }
    const float zabs = std::abs(z);
    up = down = false;
    if (zabs > CameraT::mfBl) z > 0 ? up = true : down = true;
  }

  // int ORBMatcher::searchByProjection(VirtualFrame::SharedPtr pFrame1, VirtualFrame::SharedPtr pFrame2, std::vector<cv::DMatch>& matches,
  //                                    float th, bool bFuse)   (ORBMatcher.h:49, src/ORBMatcher.cc:265-347)
  template <class CameraT, class FramePtr1, class FramePtr2>
  static int searchByProjection(FramePtr1 pFrame1, FramePtr2 pFrame2, std::vector<cv::DMatch>& matches, float th, bool bFuse, float mfRatio) {
    matches.clear();
    bool up = false, down = false;
    motionDirection<CameraT>(pFrame1, pFrame2, up, down);
    auto mps1 = pFrame1->getMapPoints();
    auto mps2 = pFrame2->getMapPoints();
    Queries q;
    for (std::size_t idx = 0; idx < mps2.size(); ++idx) {
      auto pMp2 = mps2[idx];
      if (!pMp2 || pMp2->isBad()) continue;
      if (bFuse) {
        float vecDistance, cosTheta;
        cv::Point2f uv;
        if (!pMp2->isInVision(pFrame1, vecDistance, uv, cosTheta)) continue;
      }
      const auto& feature = pFrame2->mvFeatsLeft[idx];
      int minOctave, maxOctave;
      if (up)
        minOctave = feature.octave, maxOctave = 7;
      else if (down)
        minOctave = 0, maxOctave = feature.octave;
      else
        minOctave = std::max(0, feature.octave - 1), maxOctave = std::min(feature.octave + 1, 7);
      // findFeaturesInArea (src/Frame.cc:286-311): radius * getScaledFactor2(kp.octave)
      q.add((int)idx, feature.pt.x, feature.pt.y, th * pFrame1->getScaledFactor2(feature.octave), minOctave, maxOctave, pFrame2->mvLeftDescriptor[idx]);
    }
    if (!q.who.empty()) {
      Target t;
      target(pFrame1, t);
      std::vector<uint8_t> hasGood;  // !bFuse: features of frame 1 that keep their map point are no candidates (:321-331) ...
      std::vector<int32_t> hits;
      if (!bFuse) {
        hasGood.assign(std::max<size_t>(t.kps.size(), 1), 0);
        for (size_t c = 0; c < mps1.size() && c < hasGood.size(); ++c) hasGood[c] = mps1[c] && !mps1[c]->isBad();
      }
      const auto m = searchTarget(t, q, bFuse ? nullptr : &hasGood, bFuse ? nullptr : &hits);
      if (!bFuse)  // ... and addMatchInTrack is called once for every query that had such a feature in its window
        for (size_t c = 0; c < mps1.size() && c < hits.size(); ++c)
          for (int32_t k = 0; k < hits[c]; ++k) mps1[c]->addMatchInTrack();
      for (size_t k = 0; k < q.who.size(); ++k) {
        if (m.nCand[k] <= 0) continue;
        const float ratio = (float)m.bestDist[k] / (float)m.secondDist[k];
        if (ratio < mfRatio && m.bestDist[k] < orbfe::ORBMatcher::mnMinThreshold) matches.emplace_back(m.bestIdx[k], q.who[k], (float)m.bestDist[k]);
      }
    }
    if (!bFuse) setMapPoints(pFrame1->mvpMapPoints, pFrame2->mvpMapPoints, matches);
    return (int)matches.size();
  }

  // int ORBMatcher::searchByProjection(VirtualFrame::SharedPtr pframe, const std::vector<MapPoint::SharedPtr>& mapPoints, float th,
  //                                    std::vector<cv::DMatch>& matches, bool bFuse)   (ORBMatcher.h:52, src/ORBMatcher.cc:561-612)
  // nLevels = ORBExtractor::mnLevels.  isInVision / predictLevel are the map point's own methods, per point as there; the
  // findFeaturesInArea + getBestMatch of ALL visible points are one launch.
  template <class FramePtr, class MapPointPtr>
  static int searchByProjection(FramePtr pframe, const std::vector<MapPointPtr>& mapPoints, float th, std::vector<cv::DMatch>& matches, bool bFuse,
                                float mfRatio, int nLevels) {
    int nMatches = 0;
    auto pFrameMapPoints = pframe->getMapPoints();
    if (!bFuse)
      for (auto& pMp : pFrameMapPoints)
        if (pMp && !pMp->isBad()) ++nMatches;
    Queries q;
    for (std::size_t idx = 0; idx < mapPoints.size(); ++idx) {
      auto pMp = mapPoints[idx];
      if (!pMp || pMp->isBad() || !pMp->isInMap()) continue;
      float distance, cosTheta;
      cv::KeyPoint kp;
      if (!pMp->isInVision(pframe, distance, kp.pt, cosTheta)) continue;
      kp.octave = pMp->predictLevel(distance);
      const float radius = cosTheta > 0.998f ? 2.5f : 4.0f;
      const int minLevel = std::max(0, kp.octave - 1), maxLevel = std::min(nLevels - 1, kp.octave + 1);
      q.add((int)idx, kp.pt.x, kp.pt.y, (radius * th) * pframe->getScaledFactor2(kp.octave), minLevel, maxLevel, pMp->getDesc());
    }
    if (q.who.empty()) return nMatches;
    Target t;
    target(pframe, t);
    const auto m = searchTarget(t, q, nullptr, nullptr);
    for (size_t k = 0; k < q.who.size(); ++k) {
      if (m.nCand[k] <= 0) continue;
      const float fRatio = (float)m.bestDist[k] / (float)m.secondDist[k];
      if (!(m.bestDist[k] < orbfe::ORBMatcher::mnMinThreshold && fRatio < mfRatio)) continue;
      auto pMp = mapPoints[(size_t)q.who[k]];
      if (!bFuse) {
        auto pMpInF = pframe->getMapPoint((std::size_t)m.bestIdx[k]);  // sees the assignments made earlier in this loop, as the reference's does
        if (!pMpInF || pMpInF->isBad() || !pMpInF->isInMap()) {
          pframe->setMapPoint(m.bestIdx[k], pMp);
          pMp->addMatchInTrack();
          ++nMatches;
        }
      } else {
        matches.push_back(cv::DMatch(m.bestIdx[k], q.who[k], (float)m.bestDist[k]));
        ++nMatches;
      }
    }
    return nMatches;
  }

  // ---- the back-end matchers (LocalMapping / LoopClosing callers), include/ORB_SLAM2/ORBMatcher.h:55-67 ----------------------------------
  // A KeyFrame as the array-level mirror reads it (orbfe::ORBMatcher::KeyFrameView)
  template <class KeyFramePtr>
  static void keyFrameView(KeyFramePtr kf, orbfe::ORBMatcher::KeyFrameView& v) {
    toKeypoints(kf->mvFeatsLeft, v.kps);
    descRows(kf->getLeftDescriptor(), v.desc);
    auto mps = kf->getMapPoints();
    const size_t n = v.kps.size();
    v.pos.assign(3 * n, 0.f), v.good.assign(n, 0), v.inMap.assign(n, 0), v.maxDist.assign(n, 0.f), v.minDist.assign(n, 0.f);
    for (size_t i = 0; i < n && i < mps.size(); ++i) {
      const auto& p = mps[i];
      if (!p || p->isBad()) continue;
      v.good[i] = 1, v.inMap[i] = p->isInMap() ? 1 : 0;
      const cv::Mat X = p->getPos();
      for (int a = 0; a < 3; ++a) v.pos[3 * i + a] = X.template at<float>(a);
      p->getDistance(v.maxDist[i], v.minDist[i]);
    }
  }
  template <class Sim3T>
  static orbfe::ORBMatcher::Sim3 toSim3(const Sim3T& g) {  // Sim3Ret: mRqp, mtqp, mfS (include/ORB_SLAM2/Sim3Solver.h:14-48)
    orbfe::ORBMatcher::Sim3 o;
    o.s = g.mfS;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) o.R[3 * r + c] = g.mRqp.template at<float>(r, c);
      o.t[r] = g.mtqp.template at<float>(r, 0);
    }
    return o;
  }
  template <class CameraT, class FramePtr>
  static orbfe::ORBMatcher::Intrinsics intrinsics(FramePtr f) {
    return {CameraT::mfFx, CameraT::mfFy, CameraT::mfCx, CameraT::mfCy, f->mfMinU, f->mfMaxU, f->mfMinV, f->mfMaxV};
  }
  static void poseFloats(const cv::Mat& Rcw, const cv::Mat& tcw, float* R, float* t) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) R[3 * r + c] = Rcw.template at<float>(r, c);
      t[r] = tcw.template at<float>(r, 0);
    }
  }

  // int ORBMatcher::searchBySim3(KeyFramePtr mpCurr, KeyFramePtr mpMatch, std::vector<cv::DMatch>& matches, Sim3Ret& g2oScm, float th)
  // (ORBMatcher.h:55, src/ORBMatcher.cc:424-484): both SIM3Project sweeps as two device searches
  template <class CameraT, class KeyFramePtr, class Sim3T>
  static int searchBySim3(KeyFramePtr mpCurr, KeyFramePtr mpMatch, std::vector<cv::DMatch>& matches, Sim3T& g2oScm, float th, float mfRatio) {
    orbfe::ORBMatcher::KeyFrameView C, M;
    keyFrameView(mpCurr, C), keyFrameView(mpMatch, M);
    cv::Mat Rcw, tcw, Rmw, tmw;
    mpCurr->getPose(Rcw, tcw), mpMatch->getPose(Rmw, tmw);
    float Rc[9], tc[3], Rm[9], tm[3];
    poseFloats(Rcw, tcw, Rc, tc), poseFloats(Rmw, tmw, Rm, tm);
    std::vector<std::pair<int, int>> pairs;
    for (const auto& m : matches) pairs.emplace_back(m.queryIdx, m.trainIdx);
    const size_t had = pairs.size();
    orbfe::ORBMatcher(mfRatio).searchBySim3(matcherContext(), C, M, pairs, toSim3(g2oScm), Rc, tc, Rm, tm, th, intrinsics<CameraT>(mpCurr),
                                            mpCurr->mvfScaledFactors);
    for (size_t k = had; k < pairs.size(); ++k) {
      cv::DMatch m;
      m.queryIdx = pairs[k].first, m.trainIdx = pairs[k].second;
      matches.push_back(m);
    }
    return (int)matches.size();
  }

  // int ORBMatcher::searchBySim3(KeyFramePtr pCurr, const std::vector<MapPointPtr>& vLoopGroupMps, std::vector<MapPointPtr>& vMatchedMps,
  //                              Sim3Ret& g2oScw, float th)   (ORBMatcher.h:58, src/ORBMatcher.cc:501-559)
  // The projection tests run on the host in the reference's float / double mix (Sim3Ret * p = (float)(s * (R p) + t) with cv::gemm's alpha /
  // beta in double; cv::norm and Mat::dot accumulate in double), every findFeaturesInArea + getBestMatch as ONE device search.
  template <class CameraT, class KeyFramePtr, class MapPointPtr, class Sim3T>
  static int searchBySim3(KeyFramePtr pCurr, const std::vector<MapPointPtr>& vLoopGroupMps, std::vector<MapPointPtr>& vMatchedMps, Sim3T& g2oScw,
                          float th, float mfRatio) {
    int nMatches = 0;
    std::set<MapPointPtr> sAlreadyMatched;
    for (const auto& p : vMatchedMps)
      if (p && !p->isBad() && p->isInMap()) sAlreadyMatched.insert(p), ++nMatches;
    const orbfe::ORBMatcher::Sim3 S = toSim3(g2oScw);
    const auto& sf = pCurr->mvfScaledFactors;
    const float logScale = std::log(sf.size() > 1 ? sf[1] : 1.2f);
    Queries q;
    for (size_t i = 0; i < vLoopGroupMps.size(); ++i) {
      const auto& pMp = vLoopGroupMps[i];
      if (!pMp || pMp->isBad() || !pMp->isInMap()) continue;
      if (sAlreadyMatched.count(pMp)) continue;
      const cv::Mat Xw = pMp->getPos();
      const float pw[3] = {Xw.template at<float>(0), Xw.template at<float>(1), Xw.template at<float>(2)};
      float pc[3];
      orbfe::ORBMatcher::affine(S.s, S.R, pw, S.t, pc);
      if (pc[2] <= 0) continue;
      const float u = CameraT::mfFx * (pc[0] / pc[2]) + CameraT::mfCx, v = CameraT::mfFy * (pc[1] / pc[2]) + CameraT::mfCy;  // Camera::project
      if (!(u < pCurr->mfMaxU && v < pCurr->mfMaxV && u > pCurr->mfMinU && v > pCurr->mfMinV)) continue;                 // isInImage
      const float dWithS = (float)std::sqrt((double)pc[0] * pc[0] + (double)pc[1] * pc[1] + (double)pc[2] * pc[2]);         // cv::norm: double sum
      const float d = dWithS / S.s;
      float mx = 0.f, mn = 0.f;
      pMp->getDistance(mx, mn);
      if (!(d < mx && d > mn)) continue;                                                                                     // isGoodDistance
      const cv::Mat vd = pMp->getViewDirection();
      const float vw[3] = {vd.template at<float>(0), vd.template at<float>(1), vd.template at<float>(2)};
      float rv[3];
      orbfe::ORBMatcher::matVec(S.R, vw, rv);
      const double dot = (double)rv[0] * pc[0] + (double)rv[1] * pc[1] + (double)rv[2] * pc[2];                              // Mat::dot
      if (dot < 0.5 * dWithS) continue;
      const int o = orbfe::ORBMatcher::predictLevel(mx, d, logScale);
      q.add((int)i, u, v, th * pCurr->getScaledFactor2(o), o - 1, o + 1, pMp->getDesc());
    }
    if (q.who.empty()) return nMatches;
    Target t;
    target(pCurr, t);
    const auto m = searchTarget(t, q, nullptr, nullptr);
    for (size_t k = 0; k < q.who.size(); ++k) {
      if (m.nCand[k] <= 0) continue;
      const float ratio = (float)m.bestDist[k] / (float)m.secondDist[k];
      if (m.bestDist[k] <= orbfe::ORBMatcher::mnMinThreshold && ratio <= mfRatio) {
        vMatchedMps[(size_t)m.bestIdx[k]] = vLoopGroupMps[(size_t)q.who[k]];
        ++nMatches;
      }
    }
    return nMatches;
  }

  // int ORBMatcher::processFuseMps(matches, fMapPoints, vMapPoints, pkf1, map, bLoop)   (src/ORBMatcher.cc:623-661): the reference's policy on the
  // reference's objects (setMapPoint / addObservation / MapPoint::replace / getObsNum)
  template <class KeyFramePtr, class FMapPoints, class VMapPoints, class MapPtr>
  static int processFuseMps(const std::vector<cv::DMatch>& matches, FMapPoints& fMapPoints, VMapPoints& vMapPoints, KeyFramePtr& pkf1, MapPtr& map,
                            bool bLoop) {
    using MapPointT = typename std::decay<decltype(*fMapPoints[0])>::type;
    int nFuse = 0;
    for (const auto& match : matches) {
      auto& pMp1 = fMapPoints[(size_t)match.queryIdx];
      auto pMp2 = vMapPoints[(size_t)match.trainIdx];
      if (!pMp2 || pMp2->isBad()) continue;
      if (!pMp1 || pMp1->isBad()) {
        pkf1->setMapPoint(match.queryIdx, pMp2);
        pMp2->addObservation(pkf1, match.queryIdx);
        ++nFuse;
      } else {
        if (pMp1 == pMp2) continue;
        if (bLoop) {
          MapPointT::replace(pMp2, pMp1, map);
          ++nFuse;
        } else {
          const int obs1 = pMp1->getObsNum(), obs2 = pMp2->getObsNum();
          if (obs1 >= obs2)
            MapPointT::replace(pMp1, pMp2, map);
          else
            MapPointT::replace(pMp2, pMp1, map);
          ++nFuse;
        }
      }
    }
    return nFuse;
  }

  // int ORBMatcher::fuse(KeyFramePtr pkf1, const std::vector<MapPointPtr>& mapPoints, MapPtr map, bool bLoop, float th)
  // (ORBMatcher.h:64, src/ORBMatcher.cc:691-714)
  template <class KeyFramePtr, class MapPointPtr, class MapPtr>
  static int fuse(KeyFramePtr pkf1, const std::vector<MapPointPtr>& mapPoints, MapPtr map, bool bLoop, float th, float mfRatio, int nLevels) {
    std::vector<cv::DMatch> matches;
    std::set<MapPointPtr> sMapPoints;
    std::vector<MapPointPtr> vMapPoints;
    auto fMapPoints = pkf1->getMapPoints();
    for (auto& p : fMapPoints)
      if (p && !p->isBad()) sMapPoints.insert(p);
    for (auto& p : mapPoints) {
      if (sMapPoints.find(p) != sMapPoints.end()) continue;
      vMapPoints.push_back(p);
    }
    searchByProjection(pkf1, vMapPoints, th, matches, true, mfRatio, nLevels);
    fMapPoints = pkf1->getMapPoints();
    return processFuseMps(matches, fMapPoints, vMapPoints, pkf1, map, bLoop);
  }

  // int ORBMatcher::fuse(KeyFramePtr pkf1, KeyFramePtr pkf2, MapPtr map)   (ORBMatcher.h:67, src/ORBMatcher.cc:724-732)
  template <class CameraT, class KeyFramePtr, class MapPtr>
  static int fuse(KeyFramePtr pkf1, KeyFramePtr pkf2, MapPtr map, float mfRatio) {
    std::vector<cv::DMatch> matches;
    searchByProjection<CameraT>(pkf1, pkf2, matches, 3.0f, true, mfRatio);
    auto fMapPoints = pkf1->getMapPoints();
    auto vMapPoints = pkf2->getMapPoints();
    return processFuseMps(matches, fMapPoints, vMapPoints, pkf1, map, false);
  }

  // int ORBMatcher::searchForTriangulation(KeyFramePtr pkf1, KeyFramePtr pkf2, std::vector<cv::DMatch>& matches)
  // (ORBMatcher.h:61, src/ORBMatcher.cc:736-787): searchByBow(pkf1, pkf2, matches, true) on the device, then the mutual epipolar test with the
  // reference's float cv::Mat products (elements summed in float left to right; Mat::dot in double): F21 = KInv^T [t21]x R21 KInv
  template <class CameraT, class KeyFramePtr>
  static int searchForTriangulation(KeyFramePtr pkf1, KeyFramePtr pkf2, std::vector<cv::DMatch>& matches, float mfRatio, bool mbCheckOri) {
    const int nAddMatches = searchByBow(pkf1, pkf2, matches, true, false, mfRatio, mbCheckOri);
    if (!nAddMatches) return 0;
    auto mul = [](const float* A, int n, int k, const float* B, int m2, float* out) {  // out[n][m2] = A[n][k] B[k][m2]
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < m2; ++j) {
          float acc = A[i * k] * B[j];
          for (int q = 1; q < k; ++q) acc = acc + A[i * k + q] * B[q * m2 + j];
          out[i * m2 + j] = acc;
        }
    };
    auto mat16 = [](const cv::Mat& T, float* o) {
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) o[4 * r + c] = T.template at<float>(r, c);
    };
    float P1[16], P1i[16], P2[16], P2i[16], T21[16], T12[16], Ki[9], KiT[9];
    mat16(pkf1->getPose(), P1), mat16(pkf1->getPoseInv(), P1i), mat16(pkf2->getPose(), P2), mat16(pkf2->getPoseInv(), P2i);
    mul(P2, 4, 4, P1i, 4, T21), mul(P1, 4, 4, P2i, 4, T12);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Ki[3 * r + c] = CameraT::mKInv.template at<float>(r, c), KiT[3 * c + r] = Ki[3 * r + c];
    auto fundamental = [&](const float* T, float* F) {
      const float x = T[3], y = T[7], z = T[11];
      const float ssm[9] = {0, -z, y, z, 0, -x, -y, x, 0};
      float R[9], a[9], b[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[3 * r + c] = T[4 * r + c];
      mul(KiT, 3, 3, ssm, 3, a), mul(a, 3, 3, R, 3, b), mul(b, 3, 3, Ki, 3, F);
    };
    float F21[9], F12[9];
    fundamental(T21, F21), fundamental(T12, F12);
    auto dist = [&](const float* pl, const float* F, const float* pt) {  // point2LineDistance(pl^T F, pt) (:789-795)
      float prm[3];
      mul(pl, 1, 3, F, 3, prm);
      const double dot = (double)prm[0] * pt[0] + (double)prm[1] * pt[1] + (double)prm[2] * pt[2];
      return (float)std::abs(dot) / std::sqrt(prm[0] * prm[0] + prm[1] * prm[1]);
    };
    std::vector<cv::DMatch> goodMatches;
    for (int idx = 0; idx < nAddMatches; ++idx) {
      const auto& match = matches[(size_t)idx];
      const auto kpt1 = pkf1->mvFeatsLeft[(size_t)match.queryIdx];
      const auto kpt2 = pkf2->mvFeatsLeft[(size_t)match.trainIdx];
      const float pt1[3] = {kpt1.pt.x, kpt1.pt.y, 1.f}, pt2[3] = {kpt2.pt.x, kpt2.pt.y, 1.f};
      const float th1 = 5.991 * pkf1->getScaledFactor2(kpt1.octave);
      if (dist(pt2, F21, pt1) > th1) continue;
      const float th2 = 5.991 * pkf2->getScaledFactor2(kpt2.octave);
      if (dist(pt1, F12, pt2) > th2) continue;
      goodMatches.push_back(match);
    }
    std::swap(goodMatches, matches);
    return (int)matches.size();
  }

  // The middle of Tracking::trackLocalMap (src/Tracking.cc:650-658) as ONE device call (orbfe_track_local_map):
  //     nMatches = matcher.searchByProjection(mpCurrFrame, mvpLocalMps, th, matches);     (src/ORBMatcher.cc:561-612, bFuse = false)
  //     if (nMatches < 30) return false;
  //     Optimizer::OptimizePoseOnly(mpCurrFrame);                                          (src/Optimizer.cc:33-203)
  // Returns nMatches; nGood = OptimizePoseOnly's return value, or -1 when nMatches < minMatches and nothing was optimised.  The frame's
  // features are the device-resident results of its left extractor's slot (extract() of THIS frame must be among the last four of the
  // extractor pool: ContextPool::kSlots); when the slot has been re-used the two bodies above run one after the other instead.  The
  // map points' own methods are NOT called for the visibility test here: pos / view direction / distance range go to the device, which
  // evaluates MapPoint::isInVision and predictLevel (MapPoint.cc:141-201) in the reference's float / double mix.
  template <class CameraT, class FramePtr, class MapPointPtr>
  static int trackLocalMap(FramePtr pframe, const std::vector<MapPointPtr>& mapPoints, float th, float mfRatio, int nLevels, int& nGood,
                           int minMatches = 30) {
    const auto& ext = pframe->mpExtractorLeft->device();
    const size_t N = pframe->mvFeatsLeft.size();
    if (!ext.resident() || (size_t)orbfe_get_capacity(ext.context()) > 2048) {
      std::vector<cv::DMatch> matches;
      const int nMatches = searchByProjection(pframe, mapPoints, th, matches, false, mfRatio, nLevels);
      nGood = nMatches < minMatches ? -1 : OptimizePoseOnly<CameraT>(pframe);
      return nMatches;
    }
    orbfe_ctx* ctx = ext.context();
    const size_t NF = (size_t)orbfe_get_capacity(ctx);
    using BasePtr = typename std::decay<decltype(pframe->getMapPoint(0))>::type;
    // the arrays: the list as given (null entries dropped), then the points the frame holds that are not in it
    std::vector<float> pos, vdir, maxD, minD;
    std::vector<uint8_t> desc, flags;
    std::vector<int> whoList;           // array index -> index in mapPoints (-1: an extra)
    std::vector<BasePtr> ptrOf;         // array index -> the map point
    std::map<const void*, int> indexOf;
    auto push = [&](const BasePtr& base, const cv::Mat& p, const cv::Mat& v, float mx, float mn, const cv::Mat& d, uint8_t fl, int who) {
      for (int a = 0; a < 3; ++a) pos.push_back(p.template at<float>(a)), vdir.push_back(v.empty() ? 0.f : v.template at<float>(a));
      maxD.push_back(mx), minD.push_back(mn);
      const size_t o = desc.size();
      desc.resize(o + 32, 0);
      if (!d.empty()) std::memcpy(desc.data() + o, d.data, 32);
      flags.push_back(fl);
      whoList.push_back(who);
      indexOf[(const void*)base.get()] = (int)ptrOf.size();
      ptrOf.push_back(base);
    };
    for (size_t i = 0; i < mapPoints.size(); ++i) {
      const auto& pMp = mapPoints[i];
      if (!pMp || indexOf.count((const void*)pMp.get())) continue;  // (a point listed twice is searched once: the second visit finds its feature taken)
      const bool bad = pMp->isBad(), inMap = pMp->isInMap();
      float mx = 0.f, mn = 0.f;
      pMp->getDistance(mx, mn);
      push(BasePtr(pMp), pMp->getPos(), pMp->getViewDirection(), mx, mn, pMp->getDesc(), (uint8_t)(4 | (bad ? 0 : 2) | ((!bad && inMap) ? 1 : 0)), (int)i);
    }
    auto frameMps = pframe->getMapPoints();
    std::vector<int32_t> held(NF, -1);
    for (size_t f = 0; f < N && f < NF; ++f) {
      const auto& h = frameMps[f];
      if (!h) continue;
      auto it = indexOf.find((const void*)h.get());
      if (it == indexOf.end()) {
        const bool bad = h->isBad(), inMap = h->isInMap();
        push(h, h->getPos(), cv::Mat(), 0.f, 0.f, cv::Mat(), (uint8_t)((bad ? 0 : 2) | ((!bad && inMap) ? 1 : 0)), -1);
        it = indexOf.find((const void*)h.get());
      }
      held[f] = it->second;
    }
    std::vector<double> rightU(NF, -1.0);
    for (size_t f = 0; f < N && f < NF; ++f) rightU[f] = pframe->getRightU(f);
    std::vector<float> sig2((size_t)nLevels), invSig2((size_t)nLevels);
    for (int l = 0; l < nLevels; ++l) sig2[(size_t)l] = pframe->getScaledFactor2(l), invSig2[(size_t)l] = pframe->getScaledFactorInv2(l);
    orbfe_frame_pose fp{};
    for (int r = 0; r < 3; ++r) {
      for (int c2 = 0; c2 < 3; ++c2) fp.Rcw[3 * r + c2] = pframe->mRcw.template at<float>(r, c2);
      fp.tcw[r] = pframe->mtcw.template at<float>(r, 0);
    }
    fp.min_u = pframe->mfMinU, fp.max_u = pframe->mfMaxU, fp.min_v = pframe->mfMinV, fp.max_v = pframe->mfMaxV;
    orbfe_camera cam{};
    cam.fx = CameraT::mfFx, cam.fy = CameraT::mfFy, cam.cx = CameraT::mfCx, cam.cy = CameraT::mfCy, cam.bf = CameraT::mfBf;
    double pose0[7], poseOut[7];
    matToPose(pframe->mRcw, pframe->mtcw, pose0);
    orbfe_track_input in{};
    in.n_mp = (int32_t)ptrOf.size();
    in.pos = pos.data(), in.view_dir = vdir.data(), in.max_dist = maxD.data(), in.min_dist = minD.data(), in.desc = desc.data(), in.flags = flags.data();
    in.held = held.data(), in.right_u = rightU.data(), in.level_sigma2 = sig2.data(), in.level_inv_sigma2 = invSig2.data(), in.pose_se3 = pose0;
    in.th = th, in.ratio = mfRatio, in.min_threshold = orbfe::ORBMatcher::mnMinThreshold, in.min_matches = minMatches;
    std::vector<int32_t> assigned(NF, -1);
    std::vector<uint8_t> inl(NF, 0);
    int32_t nMatches = 0, nEdges = 0, good = 0;
    orbfe_track_output out{};
    out.assigned = assigned.data(), out.inlier = inl.data(), out.n_matches = &nMatches, out.n_edges = &nEdges, out.n_good = &good, out.pose_out = poseOut;
    check(ctx, orbfe_track_local_map(ctx, ext.slot(), &fp, &cam, &in, &out));
    // searchByProjection's side effects (:595-599): the new assignments, in map-point order for the counters' sake
    for (size_t f = 0; f < N && f < NF; ++f)
      if (assigned[f] != held[f] && assigned[f] >= 0) {
        pframe->setMapPoint((int)f, ptrOf[(size_t)assigned[f]]);
        ptrOf[(size_t)assigned[f]]->addMatchInTrack();
      }
    if (nEdges < 0) {  // nMatches < minMatches: trackLocalMap returns false before the optimisation
      nGood = -1;
      return nMatches;
    }
    // OptimizePoseOnly's tail (:180-203): the projection post-check with the frame's pose as it still is, the counters, the pose
    int nBad = nEdges - good;
    for (size_t f = 0; f < N && f < NF; ++f) {
      auto cur = pframe->getMapPoint(f);
      bool keep = inl[f] != 0;
      if (keep) {
        bool isPositive = false;
        auto uv = pframe->project2UV(cur->getPos(), isPositive);
        if (!isPositive || uv.x > pframe->mfMaxU || uv.x < 0 || uv.y > pframe->mfMaxV || uv.y < 0) keep = false, ++nBad;
      }
      if (!keep)
        pframe->mvpMapPoints[f] = nullptr;
      else
        pframe->mvpMapPoints[f]->addInlierInTrack();
    }
    pframe->setPose(poseToMat(poseOut));
    nGood = nEdges - nBad;
    return nMatches;
  }

  // The middle of Tracking::trackMotionModel (src/Tracking.cc:385-396) as ONE device call (orbfe_track_motion_model):
  //     nMatches = matcher.searchByProjection(mpCurrFrame, mpLastFrame, matches, 15);
  //     if (nMatches < 20) nMatches += matcher.searchByProjection(mpCurrFrame, mpLastFrame, matches, 30);
  //     if (nMatches < 20) return false;   Optimizer::OptimizePoseOnly(mpCurrFrame);
  // with the side effects of the bodies above: the matches' map points set in the frame (setMapPoints, in query order: the last one
  // stays) with addMatchInTrack for every match and for every query that met a feature which already held a good point; then
  // OptimizePoseOnly's tail (the projection post-check, addInlierInTrack, setPose).  Returns nMatches; nGood = -1 when they were too few.
  // The current frame's features are the device-resident results of its extractor's slot; falls back to the two bodies when they are not.
  template <class CameraT, class FramePtr1, class FramePtr2>
  static int trackMotionModel(FramePtr1 pCurr, FramePtr2 pLast, float mfRatio, int& nGood, float th = 15.f, float thSecond = 30.f, int minMatches = 20) {
    const auto& ext = pCurr->mpExtractorLeft->device();
    const size_t N = pCurr->mvFeatsLeft.size();
    if (!ext.resident() || (size_t)orbfe_get_capacity(ext.context()) > 2048) {
      std::vector<cv::DMatch> matches;
      int nMatches = searchByProjection<CameraT>(pCurr, pLast, matches, th, false, mfRatio);
      if (nMatches < minMatches) nMatches += searchByProjection<CameraT>(pCurr, pLast, matches, thSecond, false, mfRatio);
      nGood = nMatches < minMatches ? -1 : OptimizePoseOnly<CameraT>(pCurr);
      return nMatches;
    }
    orbfe_ctx* ctx = ext.context();
    const size_t NF = (size_t)orbfe_get_capacity(ctx);
    bool up = false, down = false;
    motionDirection<CameraT>(pCurr, pLast, up, down);
    auto mps1 = pCurr->getMapPoints();
    auto mps2 = pLast->getMapPoints();
    std::vector<float> qxy, pos;
    std::vector<int8_t> qOct, qLo, qHi;
    std::vector<uint8_t> desc;
    std::vector<int> who;  // query -> index in the last frame
    for (std::size_t idx = 0; idx < mps2.size(); ++idx) {
      auto pMp2 = mps2[idx];
      if (!pMp2 || pMp2->isBad()) continue;
      const auto& feature = pLast->mvFeatsLeft[idx];
      int lo, hi;
      if (up)
        lo = feature.octave, hi = 7;
      else if (down)
        lo = 0, hi = feature.octave;
      else
        lo = std::max(0, feature.octave - 1), hi = std::min(feature.octave + 1, 7);
      qxy.push_back(feature.pt.x), qxy.push_back(feature.pt.y);
      qOct.push_back((int8_t)feature.octave), qLo.push_back((int8_t)lo), qHi.push_back((int8_t)hi);
      const size_t o = desc.size();
      desc.resize(o + 32);
      std::memcpy(desc.data() + o, pLast->mvLeftDescriptor[idx].data, 32);
      const cv::Mat p = pMp2->getPos();
      for (int a = 0; a < 3; ++a) pos.push_back(p.template at<float>(a));
      who.push_back((int)idx);
    }
    // what the current frame's features hold on entry: a good point makes the feature no candidate; it is an edge of the optimisation either
    // way, so such points go in as extra "queries" that search nothing (an empty octave window)
    std::vector<int32_t> held(NF, -1);
    using BasePtr = typename std::decay<decltype(pCurr->getMapPoint(0))>::type;
    std::vector<BasePtr> extra;
    for (size_t f = 0; f < N && f < NF && f < mps1.size(); ++f) {
      if (!mps1[f] || mps1[f]->isBad()) continue;
      held[f] = (int32_t)who.size();
      qxy.push_back(0.f), qxy.push_back(0.f), qOct.push_back(0), qLo.push_back(1), qHi.push_back(0);
      desc.resize(desc.size() + 32, 0);
      const cv::Mat p = mps1[f]->getPos();
      for (int a = 0; a < 3; ++a) pos.push_back(p.template at<float>(a));
      who.push_back(-1);
      extra.push_back(mps1[f]);
    }
    const int nLevels = 8;
    std::vector<float> sig2((size_t)nLevels), invSig2((size_t)nLevels);
    for (int l = 0; l < nLevels; ++l) sig2[(size_t)l] = pCurr->getScaledFactor2(l), invSig2[(size_t)l] = pCurr->getScaledFactorInv2(l);
    std::vector<double> rightU(NF, -1.0);
    for (size_t f = 0; f < N && f < NF; ++f) rightU[f] = pCurr->getRightU(f);
    const float bounds[4] = {pCurr->mfMinU, pCurr->mfMaxU, pCurr->mfMinV, pCurr->mfMaxV};
    orbfe_camera cam{};
    cam.fx = CameraT::mfFx, cam.fy = CameraT::mfFy, cam.cx = CameraT::mfCx, cam.cy = CameraT::mfCy, cam.bf = CameraT::mfBf;
    double pose0[7], poseOut[7];
    matToPose(pCurr->mRcw, pCurr->mtcw, pose0);
    orbfe_motion_input in{};
    in.n = (int32_t)who.size();
    in.qxy = qxy.data(), in.q_octave = qOct.data(), in.q_min_level = qLo.data(), in.q_max_level = qHi.data(), in.desc = desc.data(), in.pos = pos.data();
    in.held = held.data(), in.right_u = rightU.data(), in.level_sigma2 = sig2.data(), in.level_inv_sigma2 = invSig2.data(), in.pose_se3 = pose0;
    in.th = th, in.th_second = thSecond, in.ratio = mfRatio, in.min_threshold = orbfe::ORBMatcher::mnMinThreshold, in.min_matches = minMatches;
    std::vector<int32_t> assigned(NF, -1), hits(NF, 0), qMatches(std::max<size_t>(who.size(), 1), 0);
    std::vector<uint8_t> inl(NF, 0);
    int32_t nMatches = 0, nEdges = 0, good = 0, passes = 0;
    orbfe_track_output out{};
    out.assigned = assigned.data(), out.inlier = inl.data(), out.n_matches = &nMatches, out.n_edges = &nEdges, out.n_good = &good, out.pose_out = poseOut;
    check(ctx, orbfe_track_motion_model(ctx, ext.slot(), bounds, &cam, &in, &out, hits.data(), qMatches.data(), &passes));
    // the side effects of the searches (:322-331, :815-830)
    // (a feature that held a good point on entry is met in both searches; one that the FIRST search assigned is met in the second -- it
    //  keeps that point, no later query can take it -- so its visits belong to the point it ends up with)
    for (size_t f = 0; f < N && f < NF && f < mps1.size(); ++f)
      for (int32_t k = 0; k < hits[f]; ++k) {
        if (held[f] >= 0)
          mps1[f]->addMatchInTrack();
        else if (assigned[f] >= 0)
          mps2[(size_t)who[(size_t)assigned[f]]]->addMatchInTrack();
      }
    for (size_t k = 0; k < who.size(); ++k)
      for (int32_t r = 0; who[k] >= 0 && r < qMatches[k]; ++r) mps2[(size_t)who[k]]->addMatchInTrack();
    for (size_t f = 0; f < N && f < NF; ++f)
      if (assigned[f] >= 0 && assigned[f] != held[f]) pCurr->mvpMapPoints[f] = mps2[(size_t)who[(size_t)assigned[f]]];
    if (nEdges < 0) {
      nGood = -1;
      return nMatches;
    }
    // OptimizePoseOnly's tail (src/Optimizer.cc:180-203), as in trackLocalMap above
    int nBad = nEdges - good;
    // (every entry that is not an inlier is cleared -- also a bad point that never became an edge: `if (!inLier[idx])
    //  mvpMapPoints[idx] = nullptr` runs over the whole vector, :190-199 -- exactly as in trackLocalMap above)
    for (size_t f = 0; f < N && f < NF; ++f) {
      auto cur = pCurr->getMapPoint(f);
      bool keep = inl[f] != 0 && cur && !cur->isBad();
      if (keep) {
        bool isPositive = false;
        auto uv = pCurr->project2UV(cur->getPos(), isPositive);
        if (!isPositive || uv.x > pCurr->mfMaxU || uv.x < 0 || uv.y > pCurr->mfMaxV || uv.y < 0) keep = false, ++nBad;
      }
      if (!keep)
        pCurr->mvpMapPoints[f] = nullptr;
      else
        pCurr->mvpMapPoints[f]->addInlierInTrack();
    }
    pCurr->setPose(poseToMat(poseOut));
    nGood = nEdges - nBad;
    return nMatches;
  }

  // The tail of Frame::Frame (RGB-D) after extract() (src/Frame.cc:130-131, :139-157): depthImg.convertTo(CV_32F) / dScale, the copy of the
  // distorted keypoints, Camera::undistortPoints(mvFeatsLeft), the depth / rightU lookup -- as one call on the extractor's slot.  depthImg
  // as read from the file: CV_16U (TUM) or CV_32F.  initGrid() stays with the caller (the grid is rebuilt on the device per search).
  template <class CameraT>
  static orbfe_camera cameraOf() {
    orbfe_camera cam{};
    cam.fx = CameraT::mfFx, cam.fy = CameraT::mfFy, cam.cx = CameraT::mfCx, cam.cy = CameraT::mfCy, cam.bf = CameraT::mfBf;
    if (!CameraT::mDistCoeff.empty()) {
      const int nd = CameraT::mDistCoeff.rows * CameraT::mDistCoeff.cols;
      float d[5] = {0, 0, 0, 0, 0};
      for (int i = 0; i < nd && i < 5; ++i) d[i] = CameraT::mDistCoeff.template at<float>(i);
      cam.k1 = d[0], cam.k2 = d[1], cam.p1 = d[2], cam.p2 = d[3], cam.k3 = d[4];
    }
    return cam;
  }
  // The whole device work of the RGB-D Frame constructor (src/Frame.cc:125-158, what Frame::createRGBD runs, include/ORB_SLAM2/Frame.h:
  // 326-331) as ONE call: stands for extract() at :138 AND the tail frameRGBD covers -- fills mvFeatsLeft (undistorted), mvLeftDescriptor,
  // mvDepths, mvFeatsRightU; mvpMapPoints is resized.  INTEGRATION.md 4b shows the edit.
  template <class CameraT, class FrameT>
  static void createRGBD(FrameT* self, const cv::Mat& depthImg, float dScale) {
    self->mpExtractorLeft->extractRGBD(cameraOf<CameraT>(), depthImg, dScale, self->mvFeatsLeft, self->mvLeftDescriptor, self->mvDepths,
                                       self->mvFeatsRightU);
    self->mvpMapPoints.resize(self->mvFeatsLeft.size(), nullptr);
  }
  template <class CameraT, class FrameT>
  static void frameRGBD(FrameT* self, const cv::Mat& depthImg, float dScale) {
    const auto& ext = self->mpExtractorLeft->device();
    if (!ext.resident()) throw std::logic_error("frameRGBD: the extractor's slot has been re-used since extract()");
    const orbfe_camera cam = cameraOf<CameraT>();
    const size_t n = self->mvFeatsLeft.size();
    const size_t cap = (size_t)std::max<int>(orbfe_get_capacity(ext.context()), 1);
    std::vector<orbfe_keypoint> und(cap);
    std::vector<double> depth(cap), rightU(cap);
    const int type = depthImg.type() == CV_32F ? 1 : 0;
    if (depthImg.type() != CV_32F && depthImg.type() != CV_16U) throw std::invalid_argument("frameRGBD: depth image must be CV_16U or CV_32F");
    check(ext.context(), orbfe_frame_rgbd(ext.context(), ext.slot(), &cam, depthImg.data, type, depthImg.step, dScale, und.data(), depth.data(),
                                          rightU.data()));
    std::memcpy((void*)self->mvFeatsLeft.data(), und.data(), sizeof(orbfe_keypoint) * n);
    self->mvpMapPoints.resize(n, nullptr);
    self->mvDepths.assign(depth.begin(), depth.begin() + n);
    self->mvFeatsRightU.assign(rightU.begin(), rightU.begin() + n);
  }
};  // struct Bodies

// the same bodies as free functions (what INTEGRATION.md's one-line members call)
template <class CameraT, class FramePtr>
int searchByStereo(FramePtr pFrame) {
  return Bodies::template searchByStereo<CameraT>(pFrame);
}
template <class CameraT, class FrameT>
int createStereo(FrameT* self) {
  return Bodies::template createStereo<CameraT>(self);
}
template <class CameraT, class FramePtr1, class FramePtr2>
int trackMotionModel(FramePtr1 pCurr, FramePtr2 pLast, float mfRatio, int& nGood) {
  return Bodies::template trackMotionModel<CameraT>(pCurr, pLast, mfRatio, nGood);
}
template <class CameraT, class FrameT>
void createRGBD(FrameT* self, const cv::Mat& depthImg, float dScale) {
  Bodies::template createRGBD<CameraT>(self, depthImg, dScale);
}
template <class CameraT, class FramePtr>
int OptimizePoseOnly(FramePtr pFrame) {
  return Bodies::template OptimizePoseOnly<CameraT>(pFrame);
}
template <class CameraT, class KeyFramePtr>
void OptimizeLocalMap(KeyFramePtr pkframe, bool& isStop) {
  Bodies::template OptimizeLocalMap<CameraT>(pkframe, isStop);
}
template <class CameraT, class KeyFramePtr, class Sim3T>
int searchBySim3(KeyFramePtr mpCurr, KeyFramePtr mpMatch, std::vector<cv::DMatch>& matches, Sim3T& g2oScm, float th, float mfRatio = 0.6f) {
  return Bodies::template searchBySim3<CameraT>(mpCurr, mpMatch, matches, g2oScm, th, mfRatio);
}
template <class CameraT, class KeyFramePtr, class MapPointPtr, class Sim3T>
int searchBySim3(KeyFramePtr pCurr, const std::vector<MapPointPtr>& vLoopGroupMps, std::vector<MapPointPtr>& vMatchedMps, Sim3T& g2oScw, float th,
                 float mfRatio = 0.6f) {
  return Bodies::template searchBySim3<CameraT>(pCurr, vLoopGroupMps, vMatchedMps, g2oScw, th, mfRatio);
}
template <class CameraT, class KeyFramePtr>
int searchForTriangulation(KeyFramePtr pkf1, KeyFramePtr pkf2, std::vector<cv::DMatch>& matches, float mfRatio = 0.6f, bool mbCheckOri = true) {
  return Bodies::template searchForTriangulation<CameraT>(pkf1, pkf2, matches, mfRatio, mbCheckOri);
}
template <class KeyFramePtr, class MapPointPtr, class MapPtr>
int fuse(KeyFramePtr pkf1, const std::vector<MapPointPtr>& mapPoints, MapPtr map, bool bLoop = false, float th = 3.0f, float mfRatio = 0.6f,
         int nLevels = 8) {
  return Bodies::fuse(pkf1, mapPoints, map, bLoop, th, mfRatio, nLevels);
}
template <class CameraT, class KeyFramePtr, class MapPtr>
int fuse(KeyFramePtr pkf1, KeyFramePtr pkf2, MapPtr map, float mfRatio = 0.6f) {
  return Bodies::template fuse<CameraT>(pkf1, pkf2, map, mfRatio);
}
template <class CameraT, class FramePtr, class MapPointPtr>
int trackLocalMap(FramePtr pframe, const std::vector<MapPointPtr>& mapPoints, float th, int& nGood, float mfRatio = 0.8f, int nLevels = 8,
                  int minMatches = 30) {
  return Bodies::template trackLocalMap<CameraT>(pframe, mapPoints, th, mfRatio, nLevels, nGood, minMatches);
}
template <class FramePtrF, class FramePtrK>
int searchByBow(FramePtrF pFrame, FramePtrK pKframe, std::vector<cv::DMatch>& matches, bool bAddMPs, bool bLoop, float mfRatio = 0.6f,
                bool mbCheckOri = true) {
  return Bodies::searchByBow(pFrame, pKframe, matches, bAddMPs, bLoop, mfRatio, mbCheckOri);
}
template <class CameraT, class FramePtr1, class FramePtr2>
int searchByProjection(FramePtr1 pFrame1, FramePtr2 pFrame2, std::vector<cv::DMatch>& matches, float th, bool bFuse, float mfRatio = 0.6f) {
  return Bodies::template searchByProjection<CameraT>(pFrame1, pFrame2, matches, th, bFuse, mfRatio);
}
template <class FramePtr, class MapPointPtr>
int searchByProjection(FramePtr pframe, const std::vector<MapPointPtr>& mapPoints, float th, std::vector<cv::DMatch>& matches, bool bFuse,
                       float mfRatio, int nLevels) {
  return Bodies::searchByProjection(pframe, mapPoints, th, matches, bFuse, mfRatio, nLevels);
}
template <class CameraT, class FrameT>
void frameRGBD(FrameT* self, const cv::Mat& depthImg, float dScale) {
  Bodies::template frameRGBD<CameraT>(self, depthImg, dScale);
}

}  // namespace dropin
}  // namespace orbfe
