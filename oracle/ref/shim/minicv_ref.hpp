// TEST INFRASTRUCTURE (checker side only) - a minimal `cv` namespace so that first-party sources of the reference
// (/root/reference/src/ORBextractor.cc) compile VERBATIM, from where they lie, into oracle/_ref/ (recipe: oracle/ref/Makefile).
//
// What is first-party (and therefore pinned by that build): everything ORBextractor.cc does itself - the feature budget per
// level, the 30-px cell grid with its threshold fallback, the quadtree with its std::list order and (size, pointer) sort,
// IC_Angle, computeOrbDescriptor, the key-point rescale, the pyramid's level sizes and ROI arithmetic.
// What is NOT pinned by it: the five OpenCV 3.4.0 primitives it calls - cv::FAST, cv::resize, cv::copyMakeBorder,
// cv::GaussianBlur, cv::fastAtan2 (+ cvRound) - OpenCV is absent, so they are forwarded to the oracle's restatements
// (libvdo_oracle.so, frontend_oracle.cpp); "parity unpinned" stays true for those.
//
// The container types below carry only the members the reference's sources touch, with OpenCV's semantics (ref-counted
// shallow copies, ROI views sharing the parent's buffer, step in bytes).
#ifndef VDO_REF_MINICV_HPP_
#define VDO_REF_MINICV_HPP_
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

typedef unsigned char uchar;

// the oracle's restatements of the OpenCV primitives (oracle/frontend_oracle.cpp)
extern "C" {
int vdo_oracle_fast_image(const uint8_t* img, int w, int h, int thr, float* x, float* y, float* resp, int cap);
float vdo_oracle_fast_atan2(float y, float x);
void vdo_oracle_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst);
void vdo_oracle_resize_linear_8u(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
int vdo_oracle_border_reflect101(int p, int len);
}

#define CV_PI 3.1415926535897932384626433832795
#define CV_8U 0
#define CV_8UC1 0
#define CV_Assert(expr) assert(expr)

inline int cvRound(double v) { return (int)lrint(v); }     // OpenCV on x86-64: cvtsd2si = round half to even
inline int cvRound(float v) { return (int)lrintf(v); }
inline int cvRound(int v) { return v; }
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvFloor(float v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }
inline int cvCeil(float v) { int i = (int)v; return i + (i < v); }

namespace cv {

template <typename T> struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T _x, T _y) : x(_x), y(_y) {}
  Point_& operator*=(float b) { x = (T)(x * b); y = (T)(y * b); return *this; }   // saturate_cast<float> of a float product
};
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;

struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x, y, width, height; Rect(int _x, int _y, int w, int h) : x(_x), y(_y), width(w), height(h) {} };

struct KeyPoint {
  Point2f pt; float size, angle, response; int octave, class_id;
  KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
  KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1)
      : pt(x, y), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {}
};

enum { BORDER_REFLECT_101 = 4, BORDER_ISOLATED = 16, INTER_LINEAR = 1 };

// single-channel 8-bit matrix header over a shared buffer (the only type ORBextractor.cc handles)
class Mat {
 public:
  struct Step { size_t v; operator size_t() const { return v; } };
  int rows = 0, cols = 0;
  uchar* data = nullptr;
  Step step{0};
  Mat() {}
  Mat(Size sz, int type) { create(sz.height, sz.width, type); }
  Mat(int r, int c, int type) { create(r, c, type); }
  void create(int r, int c, int type) {
    assert(type == CV_8UC1);
    if (r == rows && c == cols && data) return;   // cv::Mat::create keeps a header of the right size and type - also a ROI (ComputePyramid relies on it)
    buf_ = std::shared_ptr<std::vector<uchar>>(new std::vector<uchar>((size_t)r * c));
    base_ = buf_->data(); base_rows_ = r; base_cols_ = c;
    rows = r; cols = c; step.v = (size_t)c; data = base_;
  }
  static Mat zeros(int r, int c, int type) { Mat m(r, c, type); if (m.data) std::memset(m.data, 0, (size_t)r * c); return m; }
  int type() const { return CV_8UC1; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  bool isContinuous() const { return step.v == (size_t)cols || rows == 1; }
  size_t step1() const { return step.v; }
  template <typename T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step.v + x * sizeof(T)); }
  template <typename T> const T& at(int y, int x) const { return *(const T*)(data + (size_t)y * step.v + x * sizeof(T)); }
  uchar* ptr(int y = 0) { return data + (size_t)y * step.v; }
  const uchar* ptr(int y = 0) const { return data + (size_t)y * step.v; }
  Mat operator()(const Rect& r) const {
    assert(r.x >= 0 && r.y >= 0 && r.x + r.width <= cols && r.y + r.height <= rows);
    Mat m(*this); m.rows = r.height; m.cols = r.width; m.data = data + (size_t)r.y * step.v + r.x; return m;
  }
  Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
  Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
  Mat clone() const {
    Mat m(rows, cols, CV_8UC1);
    for (int y = 0; y < rows; ++y) std::memcpy(m.ptr(y), ptr(y), (size_t)cols);
    return m;
  }
  void release() { *this = Mat(); }
  // where this header sits inside the allocation it views (cv::Mat::locateROI); used by copyMakeBorder
  void locateROI(Size& whole, Point& ofs) const {
    const size_t d = (size_t)(data - base_);
    whole = Size(base_cols_, base_rows_); ofs = Point((int)(d % base_cols_), (int)(d / base_cols_));
  }
  bool sameBuffer(const Mat& o) const { return buf_ && buf_ == o.buf_; }

 private:
  std::shared_ptr<std::vector<uchar>> buf_;
  uchar* base_ = nullptr; int base_rows_ = 0, base_cols_ = 0;
};

class _InputArray {
 public:
  _InputArray() : m_(nullptr) {}
  _InputArray(const Mat& m) : m_(&m) {}
  bool empty() const { return !m_ || m_->empty(); }
  Mat getMat() const { return m_ ? *m_ : Mat(); }
 protected:
  const Mat* m_;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat& m) : _InputArray(m), w_(&m) {}
  void create(int r, int c, int type) const { w_->create(r, c, type); }
  void release() const { w_->release(); }
 private:
  Mat* w_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
inline _InputArray noArray() { return _InputArray(); }

inline float fastAtan2(float y, float x) { return vdo_oracle_fast_atan2(y, x); }

// cv::FAST(image, keypoints, threshold, nonmaxSuppression = true): TYPE_9_16, key points (x, y, 7.f, -1, score) in raster order
inline void FAST(InputArray _img, std::vector<KeyPoint>& kps, int threshold, bool nonmax = true) {
  assert(nonmax);
  const Mat img = _img.getMat();
  std::vector<uchar> c((size_t)img.rows * img.cols);
  for (int y = 0; y < img.rows; ++y) std::memcpy(&c[(size_t)y * img.cols], img.ptr(y), (size_t)img.cols);
  const int cap = img.rows * img.cols;
  std::vector<float> x(cap), yv(cap), s(cap);
  const int n = vdo_oracle_fast_image(c.data(), img.cols, img.rows, threshold, x.data(), yv.data(), s.data(), cap);
  kps.clear();
  for (int i = 0; i < n; ++i) kps.push_back(KeyPoint(x[i], yv[i], 7.f, -1, s[i]));
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR), CV_8UC1: dst keeps its buffer when it already has dsize (here: the ROI of `temp`)
inline void resize(InputArray _src, OutputArray _dst, Size dsize, double fx = 0, double fy = 0, int interp = INTER_LINEAR) {
  assert(fx == 0 && fy == 0 && interp == INTER_LINEAR);
  const Mat src = _src.getMat();
  _dst.create(dsize.height, dsize.width, CV_8UC1);
  Mat dst = _dst.getMat();
  std::vector<uchar> s((size_t)src.rows * src.cols), d((size_t)dsize.width * dsize.height);
  for (int y = 0; y < src.rows; ++y) std::memcpy(&s[(size_t)y * src.cols], src.ptr(y), (size_t)src.cols);
  vdo_oracle_resize_linear_8u(s.data(), src.cols, src.rows, d.data(), dsize.width, dsize.height);
  for (int y = 0; y < dst.rows; ++y) std::memcpy(dst.ptr(y), &d[(size_t)y * dst.cols], (size_t)dst.cols);
}

// cv::copyMakeBorder(src, dst, t, b, l, r, BORDER_REFLECT_101 [+ BORDER_ISOLATED]).  Without BORDER_ISOLATED OpenCV reads the pixels
// a ROI has around it in its parent before it mirrors; with it, only the ROI counts.
inline void copyMakeBorder(InputArray _src, OutputArray _dst, int top, int bottom, int left, int right, int borderType) {
  Mat src = _src.getMat();
  const bool isolated = (borderType & BORDER_ISOLATED) != 0;
  assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
  // snapshot of the source (dst may be the parent of src, as in ComputePyramid)
  int ox = 0, oy = 0; Size whole(src.cols, src.rows);
  std::vector<uchar> s((size_t)src.rows * src.cols);
  for (int y = 0; y < src.rows; ++y) std::memcpy(&s[(size_t)y * src.cols], src.ptr(y), (size_t)src.cols);
  std::vector<uchar> par; int pw = 0, ph = 0;
  if (!isolated) {
    Point ofs; src.locateROI(whole, ofs); ox = ofs.x; oy = ofs.y;
    if (whole.width != src.cols || whole.height != src.rows) {           // pixels outside the ROI exist: take what is there
      pw = whole.width; ph = whole.height; par.resize((size_t)pw * ph);
      const uchar* base = src.data - (size_t)oy * src.step.v - ox;
      for (int y = 0; y < ph; ++y) std::memcpy(&par[(size_t)y * pw], base + (size_t)y * src.step.v, (size_t)pw);
    }
  }
  _dst.create(src.rows + top + bottom, src.cols + left + right, CV_8UC1);
  Mat dst = _dst.getMat();
  for (int y = 0; y < dst.rows; ++y)
    for (int x = 0; x < dst.cols; ++x) {
      const int sy = y - top, sx = x - left;
      uchar v;
      if (!par.empty() && oy + sy >= 0 && oy + sy < ph && ox + sx >= 0 && ox + sx < pw && (sy < 0 || sy >= src.rows || sx < 0 || sx >= src.cols)) {
        v = par[(size_t)(oy + sy) * pw + ox + sx];
      } else {
        v = s[(size_t)vdo_oracle_border_reflect101(sy, src.rows) * src.cols + vdo_oracle_border_reflect101(sx, src.cols)];
      }
      dst.at<uchar>(y, x) = v;
    }
}

// cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101), CV_8UC1 (in place in the reference)
inline void GaussianBlur(InputArray _src, OutputArray _dst, Size k, double sx, double sy, int borderType) {
  assert(k.width == 7 && k.height == 7 && sx == 2 && sy == 2 && borderType == BORDER_REFLECT_101);
  const Mat src = _src.getMat();
  std::vector<uchar> s((size_t)src.rows * src.cols), d(s.size());
  for (int y = 0; y < src.rows; ++y) std::memcpy(&s[(size_t)y * src.cols], src.ptr(y), (size_t)src.cols);
  vdo_oracle_gaussian_blur7(s.data(), src.cols, src.rows, d.data());
  _dst.create(src.rows, src.cols, CV_8UC1);
  Mat dst = _dst.getMat();
  for (int y = 0; y < dst.rows; ++y) std::memcpy(dst.ptr(y), &d[(size_t)y * dst.cols], (size_t)dst.cols);
}

// only reached from ComputeKeyPointsOld, which nothing calls (it has to link)
struct KeyPointsFilter {
  static void retainBest(std::vector<KeyPoint>& k, int n) {
    if (n >= 0 && (int)k.size() > n) {
      std::stable_sort(k.begin(), k.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
      k.resize(n);
    }
  }
};

}  // namespace cv
#endif
