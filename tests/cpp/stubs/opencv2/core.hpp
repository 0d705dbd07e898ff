// A STAND-IN for <opencv2/core.hpp>, only for tests/cpp/test_dropin.cpp: the few members of cv::Mat / cv::KeyPoint / cv::Point2f that
// orb_slam2_ros2_amd/host/orbfe_dropin.hpp touches, with OpenCV's names, layouts and meanings.  It lets the drop-in header go through a
// compiler and its logic run in this image (which has no OpenCV); it is NOT OpenCV and pins nothing about OpenCV's behaviour.
#pragma once
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_16U 2
#define CV_32F 5
#define CV_Assert(expr) assert(expr)

namespace cv {

struct Point2f {
  float x = 0.f, y = 0.f;
  Point2f() = default;
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct KeyPoint {  // same members, order and size (28 bytes) as cv::KeyPoint
  Point2f pt;
  float size = 0.f, angle = -1.f, response = 0.f;
  int octave = 0, class_id = -1;
};

struct DMatch {  // same members as cv::DMatch
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 3.402823466e+38f;
  DMatch() = default;
  DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
};

class Mat {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;
  uint8_t* data = nullptr;

  Mat() = default;
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type) {
    step = (size_t)c * elem();
    own_ = std::make_shared<std::vector<uint8_t>>(step * (size_t)r, 0);
    data = own_->data();
  }
  Mat(int r, int c, int type, void* d, size_t st = 0) : rows(r), cols(c), data((uint8_t*)d), type_(type) {  // a header on foreign memory
    step = st ? st : (size_t)c * elem();
  }
  int type() const { return type_; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int r = 0; r < rows; ++r) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elem());
    return m;
  }
  void copyTo(Mat& o) const { o = clone(); }
  Mat row(int r) const {  // a header on row r that shares the block (and keeps it alive), as cv::Mat::row does
    Mat m;
    m.rows = 1, m.cols = cols, m.step = step, m.data = data + (size_t)r * step, m.type_ = type_, m.own_ = own_;
    return m;
  }
  template <class T>
  T& at(int r, int c) {
    return *(T*)(data + (size_t)r * step + (size_t)c * sizeof(T));
  }
  template <class T>
  const T& at(int r, int c) const {
    return *(const T*)(data + (size_t)r * step + (size_t)c * sizeof(T));
  }
  template <class T>
  T& at(int i) {  // single index: row i of a column vector / element i of a row vector, as OpenCV does
    return cols == 1 ? at<T>(i, 0) : at<T>(0, i);
  }
  template <class T>
  const T& at(int i) const {
    return cols == 1 ? at<T>(i, 0) : at<T>(0, i);
  }

 private:
  size_t elem() const { return type_ == CV_32F ? 4 : (type_ == CV_16U ? 2 : 1); }
  int type_ = CV_8U;
  std::shared_ptr<std::vector<uint8_t>> own_;
};

}  // namespace cv
