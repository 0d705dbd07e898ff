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
#define CV_32FC1 5
#define CV_64F 6
#define CV_Assert(expr) assert(expr)

// OpenCV's rounding helpers (round half to even / floor / ceil to int)
#include <cmath>
static inline int cvRound(double v) { return (int)std::lrint(v); }
static inline int cvFloor(double v) { return (int)std::floor(v); }
static inline int cvCeil(double v) { return (int)std::ceil(v); }

namespace cv {

struct Point2f {
  float x = 0.f, y = 0.f;
  Point2f() = default;
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct Point3f {
  float x = 0.f, y = 0.f, z = 0.f;
  Point3f() = default;
  Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

struct Size {
  int width = 0, height = 0;
  Size() = default;
  Size(int w, int h) : width(w), height(h) {}
};

struct Range {
  int start = 0, end = 0;
  Range() = default;
  Range(int s, int e) : start(s), end(e) {}
};

struct Scalar {
  double val[4] = {0, 0, 0, 0};
  Scalar() = default;
  Scalar(double a, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {}
};

struct KeyPoint {  // same members, order and size (28 bytes) as cv::KeyPoint
  Point2f pt;
  float size = 0.f, angle = -1.f, response = 0.f;
  int octave = 0, class_id = -1;
  KeyPoint() = default;
  KeyPoint(Point2f pt_, float size_, float angle_ = -1.f, float response_ = 0.f, int octave_ = 0, int class_id_ = -1)
      : pt(pt_), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
  KeyPoint(float x, float y, float size_, float angle_ = -1.f, float response_ = 0.f, int octave_ = 0, int class_id_ = -1)
      : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
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
  // ---- DECLARED ONLY (what the reference's headers and src/Frame.cc use of cv::Mat beyond the above): enough for `g++ -fsyntax-only`
  //      of the reference's translation units in tests/test_reference_compile.py; tests/cpp/test_dropin.cpp never calls them ----
  Mat(int r, int c, int type, const Scalar& fill);
  static Mat eye(int r, int c, int type);
  static Mat zeros(int r, int c, int type);
  static Mat ones(int r, int c, int type);
  void copyTo(Mat&& view) const;  // (copying into a temporary header: `R.copyTo(T.rowRange(0, 3).colRange(0, 3))`)
  Mat rowRange(int a, int b) const;
  Mat colRange(int a, int b) const;
  Mat col(int c) const;
  Mat operator()(const Range& rows_, const Range& cols_) const;
  Mat& operator/=(double s);
  Mat& operator*=(double s);
  Mat& operator+=(const Mat& o);
  Mat& operator-=(const Mat& o);
  Mat t() const;
  Mat inv(int method = 0) const;
  Mat mul(const Mat& o, double scale = 1) const;
  Mat cross(const Mat& o) const;
  double dot(const Mat& o) const;
  Mat reshape(int cn, int rows_ = 0) const;
  void convertTo(Mat& o, int type, double alpha = 1, double beta = 0) const;
  void release();
  size_t total() const;
  Size size() const;
  int channels() const;
  bool isContinuous() const;
  template <class T>
  T* ptr(int r = 0);
  template <class T>
  const T* ptr(int r = 0) const;
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

// (declared only, see above)
Mat operator*(const Mat& a, const Mat& b);
Mat operator*(const Mat& a, double s);
Mat operator*(double s, const Mat& a);
Mat operator/(const Mat& a, double s);
Mat operator+(const Mat& a, const Mat& b);
Mat operator-(const Mat& a, const Mat& b);
Mat operator-(const Mat& a);

}  // namespace cv
