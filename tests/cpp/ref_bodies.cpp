// ref_bodies.cpp -- the member-function bodies INTEGRATION.md (sections 2b, 3, 3b, 4) tells a maintainer to write, compiled against the
// REFERENCE'S OWN class declarations (include/ORB_SLAM2/{Frame,KeyFrame,MapPoint,Map,ORBMatcher,Optimizer,Camera,Sim3Solver}.h of
// /root/reference, with ORBExtractor.h replaced by the one-line include of orbfe_dropin.hpp and the friend line of INTEGRATION section 3
// added to VirtualFrame / Frame / KeyFrame): every accessor, member and type the drop-in's templates touch is type-checked by the
// compiler against what the reference really declares.  `g++ -fsyntax-only` only (tests/test_reference_compile.py): third-party headers are
// stand-ins (tests/cpp/stubs), nothing is linked or run.  Not part of the product.
#include "ORB_SLAM2/Camera.h"
#include "ORB_SLAM2/Frame.h"
#include "ORB_SLAM2/KeyFrame.h"
#include "ORB_SLAM2/Map.h"
#include "ORB_SLAM2/MapPoint.h"
#include "ORB_SLAM2/ORBMatcher.h"
#include "ORB_SLAM2/Optimizer.h"
#include "ORB_SLAM2/Sim3Solver.h"

namespace ORB_SLAM2_ROS2 {

// ---- ORBMatcher (include/ORB_SLAM2/ORBMatcher.h:38-67): INTEGRATION section 3 ----
int ORBMatcher::searchByStereo(FramePtr pFrame) { return orbfe::dropin::searchByStereo<Camera>(pFrame); }
int ORBMatcher::searchByBow(VirtualFramePtr pFrame, VirtualFramePtr pKframe, std::vector<cv::DMatch> &matches, bool bAddMPs, bool bLoop) {
  return orbfe::dropin::searchByBow(pFrame, pKframe, matches, bAddMPs, bLoop, mfRatio, mbCheckOri);
}
int ORBMatcher::searchByProjection(VirtualFramePtr pFrame1, VirtualFramePtr pFrame2, std::vector<cv::DMatch> &matches, float th, bool bFuse) {
  return orbfe::dropin::searchByProjection<Camera>(pFrame1, pFrame2, matches, th, bFuse, mfRatio);
}
int ORBMatcher::searchByProjection(VirtualFramePtr pframe, const std::vector<MapPointPtr> &mapPoints, float th, std::vector<cv::DMatch> &matches, bool bFuse) {
  return orbfe::dropin::searchByProjection(pframe, mapPoints, th, matches, bFuse, mfRatio, ORBExtractor::mnLevels);
}
int ORBMatcher::searchBySim3(KeyFramePtr mpCurr, KeyFramePtr mpMatch, std::vector<cv::DMatch> &matches, Sim3Ret &g2oScm, float th) {
  return orbfe::dropin::searchBySim3<Camera>(mpCurr, mpMatch, matches, g2oScm, th, mfRatio);
}
int ORBMatcher::searchBySim3(KeyFramePtr pCurr, const std::vector<MapPointPtr> &vLoopGroupMps, std::vector<MapPointPtr> &vMatchedMps, Sim3Ret &g2oScw, float th) {
  return orbfe::dropin::searchBySim3<Camera>(pCurr, vLoopGroupMps, vMatchedMps, g2oScw, th, mfRatio);
}
int ORBMatcher::searchForTriangulation(KeyFramePtr pkf1, KeyFramePtr pkf2, std::vector<cv::DMatch> &matches) {
  return orbfe::dropin::searchForTriangulation<Camera>(pkf1, pkf2, matches, mfRatio, mbCheckOri);
}
int ORBMatcher::fuse(KeyFramePtr pkf1, const std::vector<MapPointPtr> &mapPoints, MapPtr map, bool bLoop, float th) {
  return orbfe::dropin::fuse(pkf1, mapPoints, map, bLoop, th, mfRatio, ORBExtractor::mnLevels);
}
int ORBMatcher::fuse(KeyFramePtr pkf1, KeyFramePtr pkf2, MapPtr map) { return orbfe::dropin::fuse<Camera>(pkf1, pkf2, map, mfRatio); }
int ORBMatcher::descDistance(const cv::Mat &a, const cv::Mat &b) { return orbfe::dropin::descDistance(a, b); }

// ---- Optimizer (include/ORB_SLAM2/Optimizer.h:69-72): INTEGRATION section 4 ----
int Optimizer::OptimizePoseOnly(FramePtr pFrame) { return orbfe::dropin::OptimizePoseOnly<Camera>(pFrame); }
void Optimizer::OptimizeLocalMap(KeyFramePtr pkframe, bool &isStop) { orbfe::dropin::OptimizeLocalMap<Camera>(pkframe, isStop); }

// ---- the call sites, as the reference writes them (src/Tracking.cc:361-372, :385-396, :650-658; src/LocalMapping.cc:95-97) ----
bool trackReferenceAsInTheReference(Frame::SharedPtr mpCurrFrame, KeyFrame::SharedPtr mpRefKf) {
  ORBMatcher matcher(0.7, true);
  std::vector<cv::DMatch> matches;
  int nMatches = matcher.searchByBow(mpCurrFrame, mpRefKf, matches);
  if (nMatches < 10) return false;
  int nInliers = Optimizer::OptimizePoseOnly(mpCurrFrame);
  return nInliers >= 10;
}
int trackMotionModelAsInTheReference(Frame::SharedPtr mpCurrFrame, Frame::SharedPtr mpLastFrame) {
  std::vector<cv::DMatch> matches;
  ORBMatcher matcher(0.9, true);
  int nMatches = matcher.searchByProjection(mpCurrFrame, mpLastFrame, matches, 15);
  if (nMatches < 20) nMatches += matcher.searchByProjection(mpCurrFrame, mpLastFrame, matches, 30);
  if (nMatches < 20) return -1;
  return Optimizer::OptimizePoseOnly(mpCurrFrame);
}
int trackLocalMapAsInTheReference(Frame::SharedPtr mpCurrFrame, std::vector<MapPoint::SharedPtr> &mvpLocalMps, float th) {
  ORBMatcher matcher(0.8, true);
  std::vector<cv::DMatch> matches;
  int nMatches = matcher.searchByProjection(mpCurrFrame, mvpLocalMps, th, matches);
  if (nMatches < 30) return -1;
  return Optimizer::OptimizePoseOnly(mpCurrFrame);
}
void localMappingAsInTheReference(KeyFrame::SharedPtr mpCurrKeyFrame, bool &mbAbortBA) { Optimizer::OptimizeLocalMap(mpCurrKeyFrame, mbAbortBA); }

// ---- the fused call shapes (INTEGRATION sections 2b, 3b): what a maintainer who touches Tracking.cc / Frame.cc writes instead ----
int trackMotionModelFused(Frame::SharedPtr mpCurrFrame, Frame::SharedPtr mpLastFrame) {
  int nGood = 0;
  const int nMatches = orbfe::dropin::trackMotionModel<Camera>(mpCurrFrame, mpLastFrame, /*mfRatio=*/0.9f, nGood);
  return nMatches < 20 ? -1 : nGood;
}
int trackLocalMapFused(Frame::SharedPtr mpCurrFrame, std::vector<MapPoint::SharedPtr> &mvpLocalMps, float th) {
  int nGood = 0;
  const int nMatches = orbfe::dropin::trackLocalMap<Camera>(mpCurrFrame, mvpLocalMps, th, nGood);
  return nMatches < 30 ? -1 : nGood;
}

// the whole stereo / RGB-D frame as one device call from inside the Frame constructors (src/Frame.cc:100-105, :130-157)
int createStereoInTheConstructor(Frame *self) { return orbfe::dropin::createStereo<Camera>(self); }
void createRGBDInTheConstructor(Frame *self, const cv::Mat &depthImg, float dScale) { orbfe::dropin::createRGBD<Camera>(self, depthImg, dScale); }
void rgbdTailInTheConstructor(Frame *self, const cv::Mat &depthImg, float dScale) { orbfe::dropin::frameRGBD<Camera>(self, depthImg, dScale); }

}  // namespace ORB_SLAM2_ROS2
