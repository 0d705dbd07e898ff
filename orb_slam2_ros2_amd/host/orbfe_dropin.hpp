// orbfe_dropin.hpp -- the reference's own signatures over the C-ABI (include/orbfe.h), for the reference's build (needs
// <opencv2/core.hpp>; nothing else of OpenCV, no g2o, no Eigen).
//
//   ORB_SLAM2_ROS2::ORBExtractor          include/ORB_SLAM2/ORBExtractor.h:100-160: ORBExtractor(const cv::Mat&, int, int, float, const
//                                         std::string&, int, int), extract(std::vector<cv::KeyPoint>&, std::vector<cv::Mat>&), getPyramid(),
//                                         getScaledFactors(), the public statics.  Two objects may extract on two threads (Frame.cc:100-105).
//   orbfe::dropin::searchByStereo         the body of `int ORBMatcher::searchByStereo(Frame::SharedPtr)` (ORBMatcher.h:38, src/ORBMatcher.cc:18-81)
//   orbfe::dropin::descDistance           `static int ORBMatcher::descDistance(const cv::Mat&, const cv::Mat&)` (ORBMatcher.h:77)
//   orbfe::dropin::OptimizePoseOnly       the body of `static int Optimizer::OptimizePoseOnly(Frame::SharedPtr)` (Optimizer.h:72, src/Optimizer.cc:33-203)
//   orbfe::dropin::OptimizeLocalMap       the body of `static void Optimizer::OptimizeLocalMap(KeyFrame::SharedPtr, bool&)` (Optimizer.h:69,
//                                         src/Optimizer.cc:225-442)
//
// The three bodies are templates over the reference's Frame / KeyFrame / MapPoint / Camera types (they only use the accessors
// the reference's own function bodies use), so this header does not have to see the reference's headers; INTEGRATION.md shows the
// one-line member functions a maintainer writes around them.  tests/cpp/test_dropin.cpp instantiates them with stand-in classes of
// the same accessors over tests/cpp/stubs/opencv2/core.hpp -- that checks the templates compile and that their logic agrees with the
// array-level path; it pins nothing about OpenCV.
#pragma once
#include <opencv2/core.hpp>

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
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
    keyPoints.resize(k.size());
    std::memcpy((void*)keyPoints.data(), k.data(), sizeof(orbfe_keypoint) * k.size());
    descriptors.clear();
    descriptors.reserve(d.size());
    for (auto& row : d) descriptors.push_back(cv::Mat(1, 32, CV_8U, row.data()).clone());  // one 1x32 Mat per keypoint (:402-412)
    std::lock_guard<std::mutex> lk(mPyrMutex);
    mvPyramids.clear();  // a new extraction: the planes are fetched again when somebody asks
  }
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

// int ORBMatcher::searchByStereo(Frame::SharedPtr pFrame)  (src/ORBMatcher.cc:18-81).  Uses pFrame->mvFeatsLeft, mvDepths, mvFeatsRightU,
// mpExtractorLeft / mpExtractorRight (ORBMatcher is a friend of Frame, Frame.h:302-303) and Camera::mfFx / mfBf.
template <class CameraT, class FramePtr>
int searchByStereo(FramePtr pFrame) {
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

// static int Optimizer::OptimizePoseOnly(Frame::SharedPtr pFrame)  (src/Optimizer.cc:33-203)
template <class CameraT, class FramePtr>
int OptimizePoseOnly(FramePtr pFrame) {
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
void OptimizeLocalMap(KeyFramePtr pkframe, bool& isStop) {
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

}  // namespace dropin
}  // namespace orbfe
