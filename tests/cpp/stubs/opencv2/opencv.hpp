// A STAND-IN for <opencv2/opencv.hpp>, only for tests/test_reference_compile.py: DECLARATIONS (no bodies) of the OpenCV names the
// reference's Frame / KeyFrame / MapPoint / Map / Camera / ORBMatcher headers and src/Frame.cc mention, on top of the stand-in core
// (opencv2/core.hpp).  It lets `g++ -fsyntax-only` type-check the reference's own translation units against
// orb_slam2_ros2_amd/host/orbfe_dropin.hpp in an image without OpenCV; it is NOT OpenCV and nothing is ever linked against it.
#pragma once
#include <iostream>  // (the real header pulls it in: src/Frame.cc uses std::cout without including it)
#include <string>

#include "core.hpp"

namespace cv {
enum { COLOR_GRAY2BGR = 8, COLOR_BGR2GRAY = 6, COLOR_RGB2GRAY = 7, NORM_L1 = 2, NORM_L2 = 4 };
struct DrawMatchesFlags {
  enum { DEFAULT = 0 };
};
void hconcat(const std::vector<Mat>& src, Mat& dst);
void cvtColor(const Mat& src, Mat& dst, int code);
void line(Mat& img, Point2f a, Point2f b, const Scalar& color, int thickness = 1);
void drawKeypoints(const Mat& img, const std::vector<KeyPoint>& kps, Mat& out, const Scalar& color = Scalar(), int flags = 0);
void imshow(const std::string& name, const Mat& img);
int waitKey(int delay = 0);
void destroyAllWindows();
void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& dist, const Mat& R = Mat(), const Mat& P = Mat());
double norm(const Mat& a, int type = NORM_L2);
double norm(const Mat& a, const Mat& b, int type = NORM_L2);
}  // namespace cv
