// minicv.h — the small part of the OpenCV 3.4 C++ surface that appears in the reference's five
// public headers (include/{System,Tracking,Frame,ORBextractor,Optimizer,Map}.h), so that the
// host classes of this directory keep the reference's signatures without OpenCV (which is not
// available in the build image).  When real OpenCV is present a maintainer uses the
// stubs of INTEGRATION.md instead; nothing here is on the GPU hot path.
#pragma once
#include <cassert>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace cv {

enum { CV_8U = 0, CV_32S = 4, CV_32F = 5 };
#define VDO_CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
enum {
  CV_8UC1 = VDO_CV_MAKETYPE(CV_8U, 1), CV_8UC3 = VDO_CV_MAKETYPE(CV_8U, 3), CV_8UC4 = VDO_CV_MAKETYPE(CV_8U, 4),
  CV_32SC1 = VDO_CV_MAKETYPE(CV_32S, 1), CV_32FC1 = VDO_CV_MAKETYPE(CV_32F, 1), CV_32FC2 = VDO_CV_MAKETYPE(CV_32F, 2)
};

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point3f { float x = 0, y = 0, z = 0; Point3f() {} Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {} };
struct Vec2f { float v[2]; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };

struct KeyPoint {
  Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
      : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};

// Dense 2-D matrix with shared storage (shallow copies like cv::Mat); only what the host classes use.
class Mat {
 public:
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  size_t step = 0;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  void create(int r, int c, int type) {
    rows = r; cols = c; type_ = type;
    step = (size_t)c * elemSize();
    buf_ = std::shared_ptr<std::vector<uint8_t>>(new std::vector<uint8_t>(step * (size_t)r));
    data = buf_->data();
  }
  // wraps external memory (no ownership), like cv::Mat(rows, cols, type, void*)
  Mat(int r, int c, int type, void* ext) : rows(r), cols(c), data((uint8_t*)ext), type_(type) { step = (size_t)c * elemSize(); }
  static Mat eye(int r, int c, int type) { Mat m = zeros(r, c, type); for (int i = 0; i < r && i < c; ++i) m.at<float>(i, i) = 1.f; return m; }
  static Mat zeros(int r, int c, int type) { Mat m(r, c, type); std::memset(m.data, 0, m.step * (size_t)r); return m; }
  int type() const { return type_; }
  int depth() const { return type_ & 7; }
  int channels() const { return (type_ >> 3) + 1; }
  size_t elemSize() const { return (size_t)channels() * (depth() == CV_8U ? 1 : 4); }
  size_t total() const { return (size_t)rows * cols; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  Mat clone() const { Mat m(rows, cols, type_); for (int r = 0; r < rows; ++r) std::memcpy(m.data + r * m.step, data + r * step, m.step); return m; }
  template <class T> T& at(int r, int c) { return *(T*)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
  template <class T> const T& at(int r, int c) const { return *(const T*)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
  template <class T> T& at(int i) { return ((T*)data)[i]; }
  template <class T> const T& at(int i) const { return ((const T*)data)[i]; }

 private:
  int type_ = 0;
  std::shared_ptr<std::vector<uint8_t>> buf_;
};

// float matrix product (3x3, 3x1, 4x4 ... as used by the reference on small cv::Mat): UNTRANSPOSED products 2..4 wide take cv::gemm's small-matrix
// fast path (OpenCV 3.4 modules/core/src/matmul.cpp), which accumulates in float, k ascending - this loop.  (Products with a transposed operand
// take the double-accumulating generic path: Converter::toInvMatrix spells that one out.)
inline Mat operator*(const Mat& a, const Mat& b) {
  assert(a.cols == b.rows);
  Mat o = Mat::zeros(a.rows, b.cols, CV_32F);
  for (int i = 0; i < a.rows; ++i)
    for (int j = 0; j < b.cols; ++j) {
      float s = 0;
      for (int k = 0; k < a.cols; ++k) s += a.at<float>(i, k) * b.at<float>(k, j);
      o.at<float>(i, j) = s;
    }
  return o;
}

typedef const Mat& InputArray;
typedef Mat& OutputArray;

}  // namespace cv
