// TEST INFRASTRUCTURE (checker side only) - a minimal `cv` namespace so that first-party sources of the reference compile VERBATIM, from
// where they lie under /root/reference, into oracle/_ref/ (recipe: oracle/ref/Makefile):
//   src/ORBextractor.cc                                   -> libref_orb.so   
//   src/Tracking.cc, Frame.cc, System.cc, Map.cc (+ ORB)  -> libref_track.so (rows a7-a14, the control flow of Track(), GetInitModelCam/Obj)
//
// What is first-party (and therefore pinned by those builds): everything those files do themselves.  What is NOT: the OpenCV 3.4.0 calls
// underneath - OpenCV is absent from this image - which this header supplies:
//   * containers (Mat with ref-counted ROI views, KeyPoint, Point, Vec, Scalar ...): OpenCV's semantics for the members the sources touch;
//   * cv::FAST, resize, copyMakeBorder, GaussianBlur, fastAtan2, cvtColor, solvePnPRansac: forwarded to the oracle's restatements
//     (libvdo_oracle.so) - "parity unpinned" stays true for those;
//   * cv::gemm behind the Mat expressions (A*B, A*B + C, -A.t()*B): restated here by path as OpenCV 3.4 modules/core/src/matmul.cpp runs them -
//     untransposed products 2..4 wide through the small-matrix fast path (float, left to right), everything else through the generic
//     GEMMSingleMul<float, double> (double accumulation, one rounding); cv::Rodrigues both ways in double;
//   * drawing / window / plotting calls: no-ops.
#ifndef VDO_REF_MINICV_HPP_
#define VDO_REF_MINICV_HPP_
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

typedef unsigned char uchar;
typedef unsigned short ushort;

// the oracle's restatements of the OpenCV primitives (oracle/frontend_oracle.cpp, p3p_oracle.cpp)
extern "C" {
int vdo_oracle_fast_image(const uint8_t* img, int w, int h, int thr, float* x, float* y, float* resp, int cap);
float vdo_oracle_fast_atan2(float y, float x);
void vdo_oracle_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst);
void vdo_oracle_resize_linear_8u(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
int vdo_oracle_border_reflect101(int p, int len);
void vdo_oracle_rgb2gray(const uint8_t* rgb, int64_t n_pixels, int channels, int rgb_order, uint8_t* gray);
int vdo_oracle_pnp_ransac_refit(int n, const double* X, const double* uv, const double* K4, int max_iters, double thr, double confidence, int refit,
                                double* T_out, uint8_t* inlier_out, int32_t* iters_run, int32_t* best_iter);
}

#define CV_PI 3.1415926535897932384626433832795
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_MAT_DEPTH(t) ((t)&7)
#define CV_MAT_CN(t) ((((t) >> CV_CN_SHIFT) & 63) + 1)
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)
#define CV_Assert(expr) assert(expr)
#define CV_RGB2GRAY 7
#define CV_BGR2GRAY 6
#define CV_RGBA2GRAY 11
#define CV_BGRA2GRAY 10
#define CV_AA 16
#define CV_FILLED -1
#define CV_GRAY2RGB 8
#define CV_GRAY2BGR 8
#define CV_RGB(r, g, b) cv::Scalar((b), (g), (r), 0)

inline int cvRound(double v) { return (int)lrint(v); }     // OpenCV on x86-64: cvtsd2si = round half to even
inline int cvRound(float v) { return (int)lrintf(v); }
inline int cvRound(int v) { return v; }
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvFloor(float v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }
inline int cvCeil(float v) { int i = (int)v; return i + (i < v); }

namespace cv {

typedef std::string String;

template <typename T> struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T _x, T _y) : x(_x), y(_y) {}
  template <typename U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {}
  Point_& operator*=(float b) { x = (T)(x * b); y = (T)(y * b); return *this; }   // saturate_cast<float> of a float product
  Point_& operator+=(const Point_& o) { x += o.x; y += o.y; return *this; }
};
template <typename T> inline Point_<T> operator+(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x + b.x, a.y + b.y); }
template <typename T> inline Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x - b.x, a.y - b.y); }
template <typename T> inline Point_<T> operator*(const Point_<T>& a, double s) { return Point_<T>((T)(a.x * s), (T)(a.y * s)); }
template <typename T> inline bool operator==(const Point_<T>& a, const Point_<T>& b) { return a.x == b.x && a.y == b.y; }
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;

template <typename T> struct Point3_ {
  T x, y, z;
  Point3_() : x(0), y(0), z(0) {}
  Point3_(T _x, T _y, T _z) : x(_x), y(_y), z(_z) {}
};
template <typename T> inline Point3_<T> operator-(const Point3_<T>& a, const Point3_<T>& b) { return Point3_<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> inline Point3_<T> operator+(const Point3_<T>& a, const Point3_<T>& b) { return Point3_<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;
inline double norm(const Point3f& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y + (double)p.z * p.z); }
inline double norm(const Point2f& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }

struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x, y, width, height; Rect() : x(0), y(0), width(0), height(0) {} Rect(int _x, int _y, int w, int h) : x(_x), y(_y), width(w), height(h) {} };

template <typename T, int N> struct Vec {
  T val[N];
  Vec() { for (int i = 0; i < N; ++i) val[i] = T(0); }
  Vec(T a, T b) { static_assert(N >= 2, ""); for (int i = 0; i < N; ++i) val[i] = T(0); val[0] = a; val[1] = b; }
  Vec(T a, T b, T c) { static_assert(N >= 3, ""); for (int i = 0; i < N; ++i) val[i] = T(0); val[0] = a; val[1] = b; val[2] = c; }
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
template <typename T, int N> inline Vec<T, N> operator*(double s, const Vec<T, N>& v) { Vec<T, N> o; for (int i = 0; i < N; ++i) o.val[i] = (T)(s * v.val[i]); return o; }
template <typename T, int N> inline Vec<T, N> operator*(const Vec<T, N>& v, double s) { return s * v; }
template <typename T, int N> inline Vec<T, N> operator+(const Vec<T, N>& a, const Vec<T, N>& b) { Vec<T, N> o; for (int i = 0; i < N; ++i) o.val[i] = (T)(a.val[i] + b.val[i]); return o; }
typedef Vec<float, 2> Vec2f;
typedef Vec<float, 3> Vec3f;
typedef Vec<uchar, 3> Vec3b;
typedef Vec<double, 3> Vec3d;

struct Scalar {
  double val[4];
  Scalar() { val[0] = val[1] = val[2] = val[3] = 0; }
  Scalar(double a, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  static Scalar all(double v) { return Scalar(v, v, v, v); }
  double& operator[](int i) { return val[i]; }
  const double& operator[](int i) const { return val[i]; }
};

struct KeyPoint {
  Point2f pt; float size, angle, response; int octave, class_id;
  KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
  KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1)
      : pt(x, y), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {}
};
struct DMatch { int queryIdx, trainIdx, imgIdx; float distance; DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(0) {} };

enum { BORDER_REFLECT_101 = 4, BORDER_ISOLATED = 16, INTER_LINEAR = 1, INTER_NEAREST = 0 };
enum { GEMM_1_T = 1, GEMM_2_T = 2, GEMM_3_T = 4 };
enum { SOLVEPNP_ITERATIVE = 0, SOLVEPNP_EPNP = 1, SOLVEPNP_P3P = 2, SOLVEPNP_DLS = 3, SOLVEPNP_UPNP = 4, SOLVEPNP_AP3P = 5 };
enum { FONT_HERSHEY_SIMPLEX = 0, FONT_HERSHEY_PLAIN = 1, FONT_HERSHEY_DUPLEX = 2, FONT_HERSHEY_COMPLEX = 3, FONT_HERSHEY_COMPLEX_SMALL = 5 };
enum { WINDOW_NORMAL = 0, WINDOW_AUTOSIZE = 1 };
enum { COLOR_BGR2GRAY = 6, COLOR_RGB2GRAY = 7, COLOR_BGRA2GRAY = 10, COLOR_RGBA2GRAY = 11 };
struct DrawMatchesFlags { enum { DEFAULT = 0, DRAW_OVER_OUTIMG = 1, NOT_DRAW_SINGLE_POINTS = 2, DRAW_RICH_KEYPOINTS = 4 }; };

class MatExpr;

// dense matrix header over a shared buffer: depth CV_8U .. CV_64F, 1..4 channels, ROI views share the parent's buffer (step in bytes)
class Mat {
 public:
  struct Step { size_t v; operator size_t() const { return v; } };
  int rows = 0, cols = 0;
  uchar* data = nullptr;
  Step step{0};
  Mat() {}
  Mat(Size sz, int type) { create(sz.height, sz.width, type); }
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, const Scalar& s) { create(r, c, type); setTo(s); }
  Mat(Size sz, int type, const Scalar& s) { create(sz.height, sz.width, type); setTo(s); }
  Mat(int r, int c, int type, void* ext, size_t st = 0) : rows(r), cols(c), data((uchar*)ext), type_(type) {      // a header over the caller's memory
    step.v = st ? st : (size_t)c * elemSize(); base_ = data; base_rows_ = r; base_cols_ = c;
  }
  Mat(const MatExpr& e);
  Mat& operator=(const MatExpr& e);
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && data) return;   // cv::Mat::create keeps a header of the right size and type - also a ROI (ComputePyramid relies on it)
    type_ = type;
    buf_ = std::shared_ptr<std::vector<uchar>>(new std::vector<uchar>((size_t)r * c * elemSize() + 16));
    base_ = buf_->data(); base_rows_ = r; base_cols_ = c;
    rows = r; cols = c; step.v = (size_t)c * elemSize(); data = base_;
  }
  void create(Size sz, int type) { create(sz.height, sz.width, type); }
  static Mat zeros(int r, int c, int type) { Mat m(r, c, type); return m; }      // (fresh buffers are value-initialised)
  static Mat zeros(Size s, int type) { return zeros(s.height, s.width, type); }
  static Mat ones(int r, int c, int type) { Mat m(r, c, type); m.setTo(Scalar::all(1)); return m; }
  static Mat eye(int r, int c, int type) { Mat m(r, c, type); for (int i = 0; i < std::min(r, c); ++i) m.set_elem(i, i, 0, 1.0); return m; }
  int type() const { return type_; }
  int depth() const { return CV_MAT_DEPTH(type_); }
  int channels() const { return CV_MAT_CN(type_); }
  size_t elemSize1() const { static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 0}; return (size_t)sz[depth()]; }
  size_t elemSize() const { return elemSize1() * channels(); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  bool isContinuous() const { return step.v == (size_t)cols * elemSize() || rows == 1; }
  size_t step1() const { return step.v / elemSize1(); }
  size_t total() const { return (size_t)rows * cols; }
  Size size() const { return Size(cols, rows); }
  template <typename T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step.v + (size_t)x * sizeof(T)); }
  template <typename T> const T& at(int y, int x) const { return *(const T*)(data + (size_t)y * step.v + (size_t)x * sizeof(T)); }
  template <typename T> T& at(int i) { return rows == 1 ? at<T>(0, i) : (cols == 1 ? at<T>(i, 0) : at<T>(i / cols, i % cols)); }
  template <typename T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : (cols == 1 ? at<T>(i, 0) : at<T>(i / cols, i % cols)); }
  template <typename T> T& at(Point p) { return at<T>(p.y, p.x); }
  template <typename T> const T& at(Point p) const { return at<T>(p.y, p.x); }
  uchar* ptr(int y = 0) { return data + (size_t)y * step.v; }
  const uchar* ptr(int y = 0) const { return data + (size_t)y * step.v; }
  template <typename T> T* ptr(int y = 0) { return (T*)(data + (size_t)y * step.v); }
  template <typename T> const T* ptr(int y = 0) const { return (const T*)(data + (size_t)y * step.v); }
  Mat operator()(const Rect& r) const {
    assert(r.x >= 0 && r.y >= 0 && r.x + r.width <= cols && r.y + r.height <= rows);
    Mat m(*this); m.rows = r.height; m.cols = r.width; m.data = data + (size_t)r.y * step.v + (size_t)r.x * elemSize(); return m;
  }
  Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
  Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
  Mat row(int i) const { return rowRange(i, i + 1); }
  Mat col(int j) const { return colRange(j, j + 1); }
  Mat clone() const { Mat m; copyTo(m); return m; }
  void copyTo(Mat& dst) const {
    if (empty()) { dst.release(); return; }
    dst.create(rows, cols, type_);
    for (int y = 0; y < rows; ++y) std::memmove(dst.ptr(y), ptr(y), (size_t)cols * elemSize());
  }
  void copyTo(const Mat& dst_view) const { Mat d(dst_view); copyTo(d); }      // (a temporary ROI header, e.g. M.copyTo(T.rowRange(0,3).colRange(0,3)))
  void release() { *this = Mat(); }
  Mat reshape(int cn, int r = 0) const {      // (only reached for distorted cameras, which the reference's settings never are)
    assert(isContinuous());
    Mat m(*this); const size_t tot = total() * channels();
    m.type_ = CV_MAKETYPE(depth(), cn); m.rows = r ? r : rows; m.cols = (int)(tot / cn / m.rows); m.step.v = (size_t)m.cols * m.elemSize(); return m;
  }
  void resize(size_t nrows) {                 // DistCoef.resize(5): rows appended, contents kept
    Mat m(nrows ? (int)nrows : 0, cols, type_);
    for (int y = 0; y < std::min(rows, (int)nrows); ++y) std::memcpy(m.ptr(y), ptr(y), (size_t)cols * elemSize());
    *this = m;
  }
  double get_elem(int y, int x, int c = 0) const {
    const uchar* p = data + (size_t)y * step.v + (size_t)x * elemSize() + c * elemSize1();
    switch (depth()) { case CV_8U: return *p; case CV_8S: return *(const signed char*)p; case CV_16U: return *(const ushort*)p; case CV_16S: return *(const short*)p;
                       case CV_32S: return *(const int*)p; case CV_32F: return *(const float*)p; default: return *(const double*)p; }
  }
  void set_elem(int y, int x, int c, double v) {
    uchar* p = data + (size_t)y * step.v + (size_t)x * elemSize() + c * elemSize1();
    switch (depth()) { case CV_8U: *p = (uchar)std::min(255.0, std::max(0.0, std::nearbyint(v))); break; case CV_8S: *(signed char*)p = (signed char)std::nearbyint(v); break;
                       case CV_16U: *(ushort*)p = (ushort)std::min(65535.0, std::max(0.0, std::nearbyint(v))); break; case CV_16S: *(short*)p = (short)std::nearbyint(v); break;
                       case CV_32S: *(int*)p = (int)std::nearbyint(v); break; case CV_32F: *(float*)p = (float)v; break; default: *(double*)p = v; }
  }
  Mat& setTo(const Scalar& s) { for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) for (int c = 0; c < channels(); ++c) set_elem(y, x, c, s[c]); return *this; }
  Mat& operator=(const Scalar& s) { return setTo(s); }
  void convertTo(Mat& dst, int rtype, double alpha = 1, double beta = 0) const {      // saturate_cast<dst>(src * alpha + beta) per element
    Mat out(rows, cols, CV_MAKETYPE(CV_MAT_DEPTH(rtype), channels()));
    const bool f2f = depth() == CV_32F && out.depth() == CV_32F, u2f = depth() != CV_64F && out.depth() == CV_32F;
    for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) for (int c = 0; c < channels(); ++c) {
      const double v = get_elem(y, x, c);
      if ((f2f || u2f) && alpha == 1 && beta == 0) out.set_elem(y, x, c, v);
      else if (u2f) out.set_elem(y, x, c, (double)((float)v * (float)alpha + (float)beta));      // cvt32f: float arithmetic
      else out.set_elem(y, x, c, v * alpha + beta);
    }
    dst = out;
  }
  MatExpr t() const;
  MatExpr inv() const;
  MatExpr mul(const Mat& o) const;
  // where this header sits inside the allocation it views (cv::Mat::locateROI); used by copyMakeBorder
  void locateROI(Size& whole, Point& ofs) const {
    const size_t d = (size_t)(data - base_), row_bytes = (size_t)base_cols_ * elemSize();
    whole = Size(base_cols_, base_rows_); ofs = Point((int)((d % row_bytes) / elemSize()), (int)(d / row_bytes));
  }

 protected:
  int type_ = 0;
  std::shared_ptr<std::vector<uchar>> buf_;
  uchar* base_ = nullptr; int base_rows_ = 0, base_cols_ = 0;
};

// cv::gemm(A, B, alpha, C, beta, D, flags) for CV_32F / CV_64F, by path as OpenCV 3.4 modules/core/src/matmul.cpp runs it
inline Mat gemm_eval(const Mat& A, const Mat& B, double alpha, const Mat* C, double beta, int flags) {
  assert(A.type() == B.type() && (A.type() == CV_32F || A.type() == CV_64F));
  const bool tA = flags & GEMM_1_T, tB = flags & GEMM_2_T;
  const int M = tA ? A.cols : A.rows, Kd = tA ? A.rows : A.cols, N = tB ? B.rows : B.cols;
  assert((tB ? B.cols : B.rows) == Kd);
  Mat D(M, N, A.type());
  auto a = [&](int i, int k) { return tA ? A.get_elem(k, i) : A.get_elem(i, k); };
  auto b = [&](int k, int j) { return tB ? B.get_elem(j, k) : B.get_elem(k, j); };
  auto c = [&](int i, int j) { return C ? C->get_elem(i, j) : 0.0; };
  if (C) assert(C->rows == M && C->cols == N);
  const bool fast = flags == 0 && 2 <= Kd && Kd <= 4 && (Kd == N || Kd == M) && (Kd == N || N <= 16);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      if (A.type() == CV_32F) {
        if (fast) {                                   // float t = a0*b0 + a1*b1 + ...; d = (float)(t*alpha + c*beta)
          float t = (float)a(i, 0) * (float)b(0, j);
          for (int k = 1; k < Kd; ++k) t = t + (float)a(i, k) * (float)b(k, j);
          D.at<float>(i, j) = (float)((double)t * alpha + c(i, j) * beta);
        } else {                                      // GEMMSingleMul<float, double>: double accumulation, k ascending
          double s = 0;
          for (int k = 0; k < Kd; ++k) s += a(i, k) * b(k, j);
          D.at<float>(i, j) = (float)(C ? s * alpha + c(i, j) * beta : s * alpha);
        }
      } else {
        double s = 0;
        if (fast) { s = a(i, 0) * b(0, j); for (int k = 1; k < Kd; ++k) s = s + a(i, k) * b(k, j); }
        else for (int k = 0; k < Kd; ++k) s += a(i, k) * b(k, j);
        D.at<double>(i, j) = C ? s * alpha + c(i, j) * beta : s * alpha;
      }
    }
  return D;
}

// lazily evaluated expressions, as far as the reference's sources use them: alpha*A, alpha*A^T, alpha*op(A)*op(B) [+ beta*C], A +- B
class MatExpr {
 public:
  enum Kind { IDENT, SCALE, TRANSP, GEMM, ADDSUB, INV, MULELEM };
  Kind kind = IDENT;
  Mat a, b, c;
  double alpha = 1, beta = 0;
  int flags = 0;
  bool has_c = false;
  MatExpr() {}
  MatExpr(const Mat& m) : a(m) {}
  Mat eval() const {
    switch (kind) {
      case IDENT: return a;
      case SCALE: { Mat o; a.convertTo(o, a.type(), alpha, 0); return o; }
      case TRANSP: { Mat o(a.cols, a.rows, a.type()); for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) for (int ch = 0; ch < a.channels(); ++ch) o.set_elem(j, i, ch, a.get_elem(i, j, ch) * alpha); return o; }
      case GEMM: return gemm_eval(a, b, alpha, has_c ? &c : nullptr, beta, flags);
      case ADDSUB: {                                  // cv::add / cv::subtract of equal types: the operation in the element type
        assert(a.rows == b.rows && a.cols == b.cols && a.type() == b.type());
        Mat o(a.rows, a.cols, a.type());
        for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) for (int ch = 0; ch < a.channels(); ++ch) {
          const double x = a.get_elem(i, j, ch), y = b.get_elem(i, j, ch);
          if (a.depth() == CV_32F) o.set_elem(i, j, ch, beta > 0 ? (double)((float)x + (float)y) : (double)((float)x - (float)y));
          else o.set_elem(i, j, ch, beta > 0 ? x + y : x - y);
        }
        return o;
      }
      case MULELEM: { Mat o(a.rows, a.cols, a.type()); for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) o.set_elem(i, j, 0, a.depth() == CV_32F ? (double)((float)a.get_elem(i, j) * (float)b.get_elem(i, j)) : a.get_elem(i, j) * b.get_elem(i, j)); return o; }
      case INV: {                                     // DECOMP_LU on a small square matrix, in double
        const int n = a.rows; assert(n == a.cols);
        std::vector<double> m((size_t)n * 2 * n, 0.0);
        for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) m[(size_t)i * 2 * n + j] = a.get_elem(i, j); m[(size_t)i * 2 * n + n + i] = 1; }
        for (int k = 0; k < n; ++k) {
          int p = k; for (int i = k + 1; i < n; ++i) if (std::fabs(m[(size_t)i * 2 * n + k]) > std::fabs(m[(size_t)p * 2 * n + k])) p = i;
          if (p != k) for (int j = 0; j < 2 * n; ++j) std::swap(m[(size_t)p * 2 * n + j], m[(size_t)k * 2 * n + j]);
          const double d = m[(size_t)k * 2 * n + k];
          for (int j = 0; j < 2 * n; ++j) m[(size_t)k * 2 * n + j] /= d;
          for (int i = 0; i < n; ++i) if (i != k) { const double f = m[(size_t)i * 2 * n + k]; for (int j = 0; j < 2 * n; ++j) m[(size_t)i * 2 * n + j] -= f * m[(size_t)k * 2 * n + j]; }
        }
        Mat o(n, n, a.type());
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) o.set_elem(i, j, 0, m[(size_t)i * 2 * n + n + j]);
        return o;
      }
    }
    return Mat();
  }
  // (conversion to Mat: the converting constructor Mat(const MatExpr&))
  // members the sources call on expressions directly
  Mat clone() const { return eval().clone(); }
  Mat rowRange(int s, int e) const { return eval().rowRange(s, e); }
  Mat colRange(int s, int e) const { return eval().colRange(s, e); }
  Mat row(int i) const { return eval().row(i); }
  Mat col(int j) const { return eval().col(j); }
  MatExpr t() const { return eval().t(); }
  void copyTo(Mat& m) const { eval().copyTo(m); }
  void copyTo(const Mat& m) const { eval().copyTo(m); }
  template <typename T> T at(int i, int j) const { return eval().at<T>(i, j); }
  template <typename T> T at(int i) const { return eval().at<T>(i); }
};
inline Mat::Mat(const MatExpr& e) { *this = e.eval(); }
inline Mat& Mat::operator=(const MatExpr& e) { Mat m = e.eval(); *this = m; return *this; }
inline MatExpr Mat::t() const { MatExpr e(*this); e.kind = MatExpr::TRANSP; return e; }
inline MatExpr Mat::inv() const { MatExpr e(*this); e.kind = MatExpr::INV; return e; }
inline MatExpr Mat::mul(const Mat& o) const { MatExpr e(*this); e.kind = MatExpr::MULELEM; e.b = o; return e; }

inline MatExpr make_gemm(const MatExpr& x, const MatExpr& y) {
  MatExpr e; e.kind = MatExpr::GEMM; e.alpha = 1; e.flags = 0;
  auto operand = [&](const MatExpr& q, int tflag, Mat& dst) {
    if (q.kind == MatExpr::TRANSP) { dst = q.a; e.flags |= tflag; e.alpha *= q.alpha; }
    else if (q.kind == MatExpr::SCALE) { dst = q.a; e.alpha *= q.alpha; }
    else dst = q.eval();
  };
  operand(x, GEMM_1_T, e.a); operand(y, GEMM_2_T, e.b);
  return e;
}
inline MatExpr operator*(const Mat& a, const Mat& b) { return make_gemm(MatExpr(a), MatExpr(b)); }
inline MatExpr operator*(const MatExpr& a, const Mat& b) { return make_gemm(a, MatExpr(b)); }
inline MatExpr operator*(const Mat& a, const MatExpr& b) { return make_gemm(MatExpr(a), b); }
inline MatExpr operator*(const MatExpr& a, const MatExpr& b) { return make_gemm(a, b); }
inline MatExpr scale_expr(const MatExpr& x, double s) {
  MatExpr e = x;
  if (x.kind == MatExpr::IDENT) { e.kind = MatExpr::SCALE; e.alpha = s; }
  else if (x.kind == MatExpr::SCALE || x.kind == MatExpr::TRANSP || x.kind == MatExpr::GEMM) { e.alpha *= s; if (x.kind == MatExpr::GEMM && x.has_c) e.beta *= s; }
  else { e = MatExpr(x.eval()); e.kind = MatExpr::SCALE; e.alpha = s; }
  return e;
}
inline MatExpr operator*(const Mat& a, double s) { return scale_expr(MatExpr(a), s); }
inline MatExpr operator*(double s, const Mat& a) { return scale_expr(MatExpr(a), s); }
inline MatExpr operator*(const MatExpr& a, double s) { return scale_expr(a, s); }
inline MatExpr operator*(double s, const MatExpr& a) { return scale_expr(a, s); }
inline MatExpr operator/(const Mat& a, double s) { return scale_expr(MatExpr(a), 1.0 / s); }
inline MatExpr operator-(const Mat& a) { return scale_expr(MatExpr(a), -1.0); }
inline MatExpr operator-(const MatExpr& a) { return scale_expr(a, -1.0); }
inline MatExpr addsub(const MatExpr& x, const MatExpr& y, double sign) {
  if (x.kind == MatExpr::GEMM && !x.has_c) { MatExpr e = x; e.c = y.eval(); e.has_c = true; e.beta = sign; return e; }      // A*B +- C: one gemm (MatOp_GEMM::add)
  if (y.kind == MatExpr::GEMM && !y.has_c && sign > 0) { MatExpr e = y; e.c = x.eval(); e.has_c = true; e.beta = 1; return e; }
  MatExpr e; e.kind = MatExpr::ADDSUB; e.a = x.eval(); e.b = y.eval(); e.beta = sign; return e;
}
inline MatExpr operator+(const Mat& a, const Mat& b) { return addsub(MatExpr(a), MatExpr(b), 1); }
inline MatExpr operator+(const MatExpr& a, const Mat& b) { return addsub(a, MatExpr(b), 1); }
inline MatExpr operator+(const Mat& a, const MatExpr& b) { return addsub(MatExpr(a), b, 1); }
inline MatExpr operator+(const MatExpr& a, const MatExpr& b) { return addsub(a, b, 1); }
inline MatExpr operator-(const Mat& a, const Mat& b) { return addsub(MatExpr(a), MatExpr(b), -1); }
inline MatExpr operator-(const MatExpr& a, const Mat& b) { return addsub(a, MatExpr(b), -1); }
inline MatExpr operator-(const Mat& a, const MatExpr& b) { return addsub(MatExpr(a), b, -1); }
inline MatExpr operator-(const MatExpr& a, const MatExpr& b) { return addsub(a, b, -1); }

template <typename T> struct DepthOf;
template <> struct DepthOf<uchar> { enum { v = CV_8U }; };
template <> struct DepthOf<int> { enum { v = CV_32S }; };
template <> struct DepthOf<float> { enum { v = CV_32F }; };
template <> struct DepthOf<double> { enum { v = CV_64F }; };
template <typename T> class Mat_;
template <typename T> struct MatCommaInit {
  Mat_<T>* m; int idx;
  MatCommaInit(Mat_<T>* _m) : m(_m), idx(0) {}
  template <typename U> MatCommaInit& operator,(U v);
  operator Mat_<T>() const;
  operator Mat() const;
};
template <typename T> class Mat_ : public Mat {
 public:
  Mat_() {}
  Mat_(int r, int c) : Mat(r, c, DepthOf<T>::v) {}
  Mat_(const Mat& m) : Mat(m) {}
  T& operator()(int i, int j) { return this->template at<T>(i, j); }
  const T& operator()(int i, int j) const { return this->template at<T>(i, j); }
  T& operator()(int i) { return this->template at<T>(i); }
};
template <typename T, typename U> inline MatCommaInit<T> operator<<(const Mat_<T>& m, U v) {
  MatCommaInit<T> ci(const_cast<Mat_<T>*>(&m));      // (the temporary lives to the end of the full expression, like cv::MatCommaInitializer_'s)
  return ci, v;
}
template <typename T> template <typename U> inline MatCommaInit<T>& MatCommaInit<T>::operator,(U v) { m->template at<T>(idx / m->cols, idx % m->cols) = (T)v; ++idx; return *this; }
template <typename T> inline MatCommaInit<T>::operator Mat_<T>() const { return *m; }
template <typename T> inline MatCommaInit<T>::operator Mat() const { return *m; }

inline double norm(const Mat& m) { double s = 0; for (int i = 0; i < m.rows; ++i) for (int j = 0; j < m.cols; ++j) { const double v = m.get_elem(i, j); s += v * v; } return std::sqrt(s); }
inline double norm(const MatExpr& e) { return norm(e.eval()); }
inline double norm(const Mat& a, const Mat& b) { return norm(Mat(a - b)); }
inline std::ostream& operator<<(std::ostream& o, const Mat& m) {
  o << "[";
  for (int i = 0; i < m.rows; ++i) { for (int j = 0; j < m.cols; ++j) for (int c = 0; c < m.channels(); ++c) o << (j || c ? ", " : "") << m.get_elem(i, j, c); o << (i + 1 < m.rows ? ";\n " : ""); }
  return o << "]";
}
inline std::ostream& operator<<(std::ostream& o, const MatExpr& e) { return o << e.eval(); }
template <typename T> inline std::ostream& operator<<(std::ostream& o, const Point_<T>& p) { return o << "[" << p.x << ", " << p.y << "]"; }
template <typename T> inline std::ostream& operator<<(std::ostream& o, const Point3_<T>& p) { return o << "[" << p.x << ", " << p.y << ", " << p.z << "]"; }

class _InputArray {
 public:
  _InputArray() : m_(nullptr) {}
  _InputArray(const Mat& m) : m_(&m) {}
  _InputArray(const MatExpr& e) : own_(new Mat(e.eval())), m_(own_.get()) {}
  bool empty() const { return !m_ || m_->empty(); }
  Mat getMat() const { return m_ ? *m_ : Mat(); }
 protected:
  std::shared_ptr<Mat> own_;
  const Mat* m_;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat& m) : _InputArray(m), w_(&m) {}
  void create(int r, int c, int type) const { w_->create(r, c, type); }
  void create(Size s, int type) const { w_->create(s.height, s.width, type); }
  void release() const { w_->release(); }
  Mat& getMatRef() const { return *w_; }
 private:
  Mat* w_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _OutputArray& InputOutputArray;
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
  const Mat src = _src.getMat();
  assert(fx == 0 && fy == 0 && interp == INTER_LINEAR && src.type() == CV_8UC1);
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
  assert(src.type() == CV_8UC1);
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
  assert(src.type() == CV_8UC1);
  std::vector<uchar> s((size_t)src.rows * src.cols), d(s.size());
  for (int y = 0; y < src.rows; ++y) std::memcpy(&s[(size_t)y * src.cols], src.ptr(y), (size_t)src.cols);
  vdo_oracle_gaussian_blur7(s.data(), src.cols, src.rows, d.data());
  _dst.create(src.rows, src.cols, CV_8UC1);
  Mat dst = _dst.getMat();
  for (int y = 0; y < dst.rows; ++y) std::memcpy(dst.ptr(y), &d[(size_t)y * dst.cols], (size_t)dst.cols);
}

// cv::cvtColor(src, dst, CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY), 8-bit (dst may be src: Tracking.cc:209-222)
inline void cvtColor(InputArray _src, OutputArray _dst, int code) {
  const Mat src = _src.getMat();
  const int cn = src.channels();
  assert(src.depth() == CV_8U && (cn == 3 || cn == 4));
  std::vector<uchar> s((size_t)src.rows * src.cols * cn), g((size_t)src.rows * src.cols);
  for (int y = 0; y < src.rows; ++y) std::memcpy(&s[(size_t)y * src.cols * cn], src.ptr(y), (size_t)src.cols * cn);
  vdo_oracle_rgb2gray(s.data(), (int64_t)src.rows * src.cols, cn, (code == CV_RGB2GRAY || code == CV_RGBA2GRAY) ? 1 : 0, g.data());
  Mat out(src.rows, src.cols, CV_8UC1);
  std::memcpy(out.data, g.data(), g.size());
  _dst.getMatRef() = out;
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

// cv::RNG (modules/core/include/opencv2/core/operations.hpp): multiply-with-carry, uniform(a, b) = next() % (b - a) + a
// (tests silence the reference's TEST noise - UnprojectStereoStat(i, addnoise = 1) inside the non-joint optimisers, src/Frame.cc:484-519 - to compare the
//  rest of those functions: vdo_ref_set_gaussian_scale, oracle/ref/ref_g2o_entry.cc)
inline double& rng_gaussian_scale() { static double s = 1.0; return s; }
class RNG {
 public:
  uint64_t state;
  RNG() : state(0xffffffff) {}
  RNG(uint64_t s) : state(s ? s : 0xffffffff) {}
  unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
  operator unsigned() { return next(); }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
  float uniform(float a, float b) { return ((float)*this) * (b - a) + a; }
  double uniform(double a, double b) { return ((double)*this) * (b - a) + a; }
  operator float() { return next() * 2.3283064365386962890625e-10f; }
  operator double() { unsigned t = next(); return (((uint64_t)t << 32) | next()) * 5.4210108624275221700372640043497e-20; }
  double gaussian(double sigma) {               // (Track() never gets here: its callers pass addnoise = false; a Box-Muller draw, not OpenCV's ziggurat)
    const double u1 = std::max(1e-12, (double)uniform(0.0, 1.0)), u2 = uniform(0.0, 1.0);
    return rng_gaussian_scale() * sigma * std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * CV_PI * u2);
  }
};

// cv::Rodrigues, both directions, CV_64F (calib3d/src/calibration.cpp cvRodrigues2 without the Jacobian)
inline void Rodrigues(InputArray _src, OutputArray _dst) {
  const Mat src = _src.getMat();
  if (src.rows * src.cols == 3) {                // vector -> matrix
    const double rx = src.get_elem(src.rows == 1 ? 0 : 0, 0), ry = src.rows == 1 ? src.get_elem(0, 1) : src.get_elem(1, 0), rz = src.rows == 1 ? src.get_elem(0, 2) : src.get_elem(2, 0);
    const double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
    Mat R = Mat::eye(3, 3, CV_64F);
    if (theta >= 2.2204460492503131e-16) {
      const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
      const double x = rx * itheta, y = ry * itheta, z = rz * itheta;
      const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
      const double rx_[9] = {0, -z, y, z, 0, -x, -y, x, 0};
      const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      for (int k = 0; k < 9; ++k) R.at<double>(k / 3, k % 3) = c * I[k] + c1 * rrt[k] + s * rx_[k];
    }
    Mat out; R.convertTo(out, src.depth() == CV_32F ? CV_32F : CV_64F);
    _dst.getMatRef() = out;
    return;
  }
  assert(src.rows == 3 && src.cols == 3);         // matrix -> vector (no SVD re-orthogonalisation: the inputs here are rotations to 1e-16)
  double R[9];
  for (int k = 0; k < 9; ++k) R[k] = src.get_elem(k / 3, k % 3);
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  const double theta = std::acos(c);
  if (s < 1e-5) {
    if (c > 0) rx = ry = rz = 0;
    else {
      double t;
      t = (R[0] + 1) * 0.5; rx = std::sqrt(std::max(t, 0.));
      t = (R[4] + 1) * 0.5; ry = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
      t = (R[8] + 1) * 0.5; rz = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
      if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
      const double nrm = theta / std::sqrt(rx * rx + ry * ry + rz * rz);
      rx *= nrm; ry *= nrm; rz *= nrm;
    }
  } else {
    const double vth = 1 / (2 * s) * theta;
    rx *= vth; ry *= vth; rz *= vth;
  }
  Mat out(3, 1, CV_64F);
  out.at<double>(0) = rx; out.at<double>(1) = ry; out.at<double>(2) = rz;
  _dst.getMatRef() = out;
}

// cv::solvePnPRansac(objectPoints, imagePoints, K, dist, rvec, tvec, false, iterations, reprojectionError, confidence, inliers, SOLVEPNP_AP3P) as
// Tracking::GetInitModelCam / GetInitModelObj call it: the oracle's restatement (cv::RNG subsets, P3P on 3 + 1 points, vote, adaptive budget, EPnP
// re-estimation of the winner on its inliers - p3p_oracle.cpp), handed back the way OpenCV does: rvec by Rodrigues, inliers as a column of int.
inline bool solvePnPRansac(const std::vector<Point3f>& obj, const std::vector<Point2f>& img, const Mat& K, const Mat& dist, Mat& rvec, Mat& tvec, bool useExtrinsicGuess,
                           int iterationsCount, float reprojectionError, double confidence, Mat& inliers, int flags) {
  assert(!useExtrinsicGuess && flags == SOLVEPNP_AP3P && obj.size() == img.size());
  (void)dist;
  const int n = (int)obj.size();
  std::vector<double> X(3 * (size_t)std::max(n, 1)), uv(2 * (size_t)std::max(n, 1));
  for (int i = 0; i < n; ++i) { X[3 * i] = obj[i].x; X[3 * i + 1] = obj[i].y; X[3 * i + 2] = obj[i].z; uv[2 * i] = img[i].x; uv[2 * i + 1] = img[i].y; }
  const double K4[4] = {K.get_elem(0, 0), K.get_elem(1, 1), K.get_elem(0, 2), K.get_elem(1, 2)};
  double T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::vector<uint8_t> inl((size_t)std::max(n, 1), 0);
  int good = 0;
  if (n >= 4) good = vdo_oracle_pnp_ransac_refit(n, X.data(), uv.data(), K4, iterationsCount, (double)reprojectionError, confidence, 1, T, inl.data(), nullptr, nullptr);
  Mat R(3, 3, CV_64F);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R.at<double>(i, j) = T[4 * i + j];
  Rodrigues(R, rvec);
  tvec = Mat(3, 1, CV_64F);
  for (int i = 0; i < 3; ++i) tvec.at<double>(i) = T[4 * i + 3];
  Mat il(good, 1, CV_32SC1);
  for (int i = 0, k = 0; i < n && k < good; ++i) if (inl[i]) il.at<int>(k++) = i;
  inliers = il;
  return good > 0;
}
inline void undistortPoints(InputArray, OutputArray, InputArray, InputArray, InputArray = noArray(), InputArray = noArray()) {
  std::fprintf(stderr, "minicv_ref: cv::undistortPoints is not restated (the reference's settings are distortion-free)\n"); std::abort();
}

// ---- settings: cv::FileStorage over the flat "key: value" YAML 1.0 files of example/ -------------------------------------------------
class FileNode {
 public:
  FileNode() : has_(false), v_(0) {}
  FileNode(double v, const std::string& s) : has_(true), v_(v), s_(s) {}
  operator int() const { return (int)v_; }                  // (a real node converts through cvRound for ints written as reals; the files write ints as ints)
  operator float() const { return (float)v_; }
  operator double() const { return v_; }
  operator std::string() const { return s_; }
  bool empty() const { return !has_; }
 private:
  bool has_; double v_; std::string s_;
};
class FileStorage {
 public:
  enum { READ = 0 };
  FileStorage() {}
  FileStorage(const std::string& path, int) { open(path); }
  bool open(const std::string& path) {
    std::ifstream f(path.c_str());
    ok_ = f.is_open();
    std::string line;
    while (ok_ && std::getline(f, line)) {
      const size_t h = line.find('#');
      if (h != std::string::npos) line = line.substr(0, h);
      if (line.empty() || line[0] == '%') continue;
      const size_t c = line.find(':');
      if (c == std::string::npos) continue;
      std::string key = line.substr(0, c), val = line.substr(c + 1);
      while (!key.empty() && (key.back() == ' ' || key.back() == '\t')) key.pop_back();
      while (!val.empty() && (val.front() == ' ' || val.front() == '\t')) val.erase(val.begin());
      while (!val.empty() && (val.back() == ' ' || val.back() == '\t' || val.back() == '\r')) val.pop_back();
      char* end = nullptr;
      const double v = std::strtod(val.c_str(), &end);
      kv_[key] = FileNode(end != val.c_str() ? v : 0.0, val);
    }
    return ok_;
  }
  bool isOpened() const { return ok_; }
  void release() {}
  FileNode operator[](const std::string& k) const { auto it = kv_.find(k); return it == kv_.end() ? FileNode() : it->second; }
  FileNode operator[](const char* k) const { return (*this)[std::string(k)]; }
 private:
  bool ok_ = false;
  std::map<std::string, FileNode> kv_;
};

// ---- drawing / windows: the reference's visualisation is out of scope - every call is a no-op ------------------------------------------------
inline void imshow(const std::string&, InputArray) {}
inline int waitKey(int = 0) { return -1; }
inline void namedWindow(const std::string&, int = 0) {}
inline void destroyAllWindows() {}
inline void moveWindow(const std::string&, int, int) {}
inline void resizeWindow(const std::string&, int, int) {}
inline bool imwrite(const std::string&, InputArray) { return true; }
inline void drawKeypoints(InputArray, const std::vector<KeyPoint>&, const Mat&, const Scalar& = Scalar::all(-1), int = 0) {}
inline void circle(const Mat&, Point, int, const Scalar&, int = 1, int = 8, int = 0) {}
inline void line(const Mat&, Point, Point, const Scalar&, int = 1, int = 8, int = 0) {}
inline void arrowedLine(const Mat&, Point, Point, const Scalar&, int = 1, int = 8, int = 0, double = 0.1) {}
inline void rectangle(const Mat&, Point, Point, const Scalar&, int = 1, int = 8, int = 0) {}
inline void rectangle(const Mat&, Rect, const Scalar&, int = 1, int = 8, int = 0) {}
inline void putText(const Mat&, const std::string&, Point, int, double, Scalar, int = 1, int = 8, bool = false) {}
inline bool clipLine(Size, Point&, Point&) { return true; }
inline bool clipLine(Rect, Point&, Point&) { return true; }
inline void flip(InputArray src, OutputArray dst, int) { dst.getMatRef() = src.getMat().clone(); }

template <typename T> struct Ptr : public std::shared_ptr<T> { Ptr() {} Ptr(T* p) : std::shared_ptr<T>(p) {} };
class Feature2D { public: virtual ~Feature2D() {} };

}  // namespace cv
#endif
