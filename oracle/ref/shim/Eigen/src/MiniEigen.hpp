// TEST INFRASTRUCTURE - a small dense-algebra library with Eigen3's interface, written for this repository (Eigen is not installed in this
// image and is not vendored by the reference: SURVEY.md §8c).  It exists so that the reference's OWN optimiser sources - src/Optimizer.cc,
// src/Converter.cc and the vendored g2o under dependencies/g2o/g2o/{core,types,solvers,stuff} - compile VERBATIM from /root/reference
// (oracle/ref/Makefile) and can be compared with the oracle's restatements and with the HIP product.  It is NOT Eigen and shares no code with
// it: everything is evaluated eagerly (an arithmetic expression returns a plain Matrix), storage is always column-major (what g2o's raw-pointer
// Map aliasing - SURVEY.md F3 - depends on), inner products are accumulated left to right in the scalar type without fused multiply-adds
// (-ffp-contract=off).  Real Eigen vectorises and, under the reference's -march=native, contracts to FMA: the last bits of a result are a property
// of the build there too, so what this pins is g2o's control flow, formulas, memory layout and the decompositions' PUBLISHED algorithms
// (pivoted LDLT with the sign rules of Eigen 3.2.9x / 3.3 as LinearSolverDense consults them, Cholesky, Shepperd's matrix->quaternion branch order,
// Eigen's quaternion->matrix and quaternion*vector formulas), not Eigen's instruction order.  Only what those sources use is here.
#ifndef VDO_REF_MINI_EIGEN_HPP_
#define VDO_REF_MINI_EIGEN_HPP_
#include <algorithm>
#include <cassert>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW_IF_VECTORIZABLE_FIXED_SIZE(a, b)
#define EIGEN_DEFINE_STL_VECTOR_SPECIALIZATION(...)
#define EIGEN_STRONG_INLINE inline
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 2
#define EIGEN_MINOR_VERSION 92
#define EIGEN_VERSION_AT_LEAST(x, y, z) (EIGEN_WORLD_VERSION > x || (EIGEN_WORLD_VERSION >= x && (EIGEN_MAJOR_VERSION > y || (EIGEN_MAJOR_VERSION >= y && EIGEN_MINOR_VERSION >= z))))
#define EIGEN_PI 3.141592653589793238462643383279502884197169399375105820974944592307816406L
// bounds / size checks of this library: only with -DMINI_EIGEN_CHECKS.  The reference is built Release (-DNDEBUG, dependencies/g2o/CMakeLists.txt:9-11,
// CMakeLists.txt:4-7) and RELIES on Eigen's checks being off: BlockSolver_6_3 allocates `new Matrix<double,3,3>(2, 2)` for the 2-DoF flow vertices
// (core/sparse_block_matrix.hpp:101) - a fixed-size matrix keeps its size there.
#ifdef MINI_EIGEN_CHECKS
#define mini_eigen_assert(x) assert(x)
#else
#define mini_eigen_assert(x) ((void)0)
#endif
#define eigen_assert(x) mini_eigen_assert(x)

namespace Eigen {
typedef std::ptrdiff_t DenseIndex;
typedef DenseIndex Index;
const int Dynamic = -1;
enum { ColMajor = 0, RowMajor = 0x1, AutoAlign = 0, DontAlign = 0x2 };
enum { Unaligned = 0, Aligned = 1 };
const unsigned int AlignedBit = 0x80;
enum { Lower = 0x1, Upper = 0x2, UnitDiag = 0x4, ZeroDiag = 0x8, UnitLower = UnitDiag | Lower, UnitUpper = UnitDiag | Upper, StrictlyLower = ZeroDiag | Lower,
       StrictlyUpper = ZeroDiag | Upper, SelfAdjoint = 0x10, Symmetric = 0x20 };
enum { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20, EigenvaluesOnly = 0x40, ComputeEigenvectors = 0x80 };
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };
enum TransformTraits { Isometry = 0x1, Affine = 0x2, AffineCompact = 0x10 | Affine, Projective = 0x20 };
inline void initParallel() {}

template <typename T> struct NumTraits {
  typedef T Real;
  static T epsilon() { return std::numeric_limits<T>::epsilon(); }
  static T dummy_precision() { return T(1e-12); }
  static T highest() { return (std::numeric_limits<T>::max)(); }
  static T lowest() { return std::numeric_limits<T>::lowest(); }
};
template <> inline float NumTraits<float>::dummy_precision() { return 1e-5f; }

template <typename T> class aligned_allocator : public std::allocator<T> {
 public:
  template <class U> struct rebind { typedef aligned_allocator<U> other; };
  aligned_allocator() {}
  aligned_allocator(const aligned_allocator& o) : std::allocator<T>(o) {}
  template <class U> aligned_allocator(const aligned_allocator<U>&) {}
};

template <typename S, int R, int C, int Opt = 0, int MaxR = R, int MaxC = C> class Matrix;
template <typename Plain, int MapOpt = Unaligned, typename Stride = void> class Map;
template <typename Xpr, int R, int C> class Block;
template <typename Xpr> class Transpose;
template <typename Xpr> class DiagonalView;
template <typename Xpr> class ArrayWrapper;
template <typename Derived> class MatrixBase;
template <typename S> class Quaternion;
template <typename S> class AngleAxis;
template <typename S, int Dim, int Mode, int Opt = 0> class Transform;
template <typename MatT, int UpLo = Lower> class LLT;
template <typename MatT, int UpLo = Lower> class LDLT;

namespace internal {
template <typename T> struct traits;
template <typename T> struct traits<const T> : traits<T> {};
template <typename S, int R, int C, int O, int MR, int MC> struct traits<Matrix<S, R, C, O, MR, MC> > {
  typedef S Scalar;
  enum { Rows = R, Cols = C, IsConst = 0 };
};
template <typename Plain, int MO, typename St> struct traits<Map<Plain, MO, St> > {
  typedef typename traits<typename std::remove_const<Plain>::type>::Scalar Scalar;
  enum { Rows = traits<typename std::remove_const<Plain>::type>::Rows, Cols = traits<typename std::remove_const<Plain>::type>::Cols, IsConst = std::is_const<Plain>::value ? 1 : 0 };
};
template <typename Xpr, int R, int C> struct traits<Block<Xpr, R, C> > {
  typedef typename traits<typename std::remove_const<Xpr>::type>::Scalar Scalar;
  enum { Rows = R, Cols = C, IsConst = (std::is_const<Xpr>::value || traits<typename std::remove_const<Xpr>::type>::IsConst) ? 1 : 0 };
};
template <typename Xpr> struct traits<Transpose<Xpr> > {
  typedef typename traits<typename std::remove_const<Xpr>::type>::Scalar Scalar;
  enum { Rows = traits<typename std::remove_const<Xpr>::type>::Cols, Cols = traits<typename std::remove_const<Xpr>::type>::Rows,
         IsConst = (std::is_const<Xpr>::value || traits<typename std::remove_const<Xpr>::type>::IsConst) ? 1 : 0 };
};
template <typename Xpr> struct traits<DiagonalView<Xpr> > {
  typedef typename traits<typename std::remove_const<Xpr>::type>::Scalar Scalar;
  enum { Rows = (traits<typename std::remove_const<Xpr>::type>::Rows == Dynamic || traits<typename std::remove_const<Xpr>::type>::Cols == Dynamic) ? Dynamic
                    : (traits<typename std::remove_const<Xpr>::type>::Rows < traits<typename std::remove_const<Xpr>::type>::Cols ? traits<typename std::remove_const<Xpr>::type>::Rows
                                                                                                                                 : traits<typename std::remove_const<Xpr>::type>::Cols),
         Cols = 1, IsConst = (std::is_const<Xpr>::value || traits<typename std::remove_const<Xpr>::type>::IsConst) ? 1 : 0 };
};
template <typename Xpr> struct traits<ArrayWrapper<Xpr> > : traits<Xpr> {};
template <int A, int B> struct same_dim { enum { value = (A == Dynamic) ? B : A }; };   // the compile-time size of two sizes that must agree
template <typename T> struct is_arith : std::is_arithmetic<T> {};
// writable access through a view: a view of a const expression hands out const references
template <typename X, bool C = std::is_const<X>::value> struct cref { template <typename R> static R get(X& x, std::ptrdiff_t i, std::ptrdiff_t j) { return x.coeffRef(i, j); } };
template <typename X> struct cref<X, true> { template <typename R> static R get(X& x, std::ptrdiff_t i, std::ptrdiff_t j) { return x.coeff(i, j); } };
}  // namespace internal

// ---------------------------------------------------------------------------------------------------------------------------------------
template <typename Derived> class MatrixBase {
 public:
  typedef typename internal::traits<Derived>::Scalar Scalar;
  typedef Scalar RealScalar;
  typedef Eigen::Index Index;
  enum { RowsAtCompileTime = internal::traits<Derived>::Rows, ColsAtCompileTime = internal::traits<Derived>::Cols,
         SizeAtCompileTime = (RowsAtCompileTime == Dynamic || ColsAtCompileTime == Dynamic) ? Dynamic : RowsAtCompileTime * ColsAtCompileTime,
         IsVectorAtCompileTime = (RowsAtCompileTime == 1 || ColsAtCompileTime == 1) ? 1 : 0, IsConstXpr = internal::traits<Derived>::IsConst, Flags = 0,
         MaxRowsAtCompileTime = RowsAtCompileTime, MaxColsAtCompileTime = ColsAtCompileTime };
  typedef Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime> PlainObject;
  typedef PlainObject PlainMatrix;
  typedef typename std::conditional<IsConstXpr != 0, const Scalar&, Scalar&>::type CoeffRef;

  Derived& derived() { return *static_cast<Derived*>(this); }
  const Derived& derived() const { return *static_cast<const Derived*>(this); }
  Derived& const_cast_derived() const { return *const_cast<Derived*>(static_cast<const Derived*>(this)); }
  Index rows() const { return derived().rows_(); }
  Index cols() const { return derived().cols_(); }
  Index size() const { return rows() * cols(); }
  Index innerSize() const { return rows(); }
  Index outerSize() const { return cols(); }

  // coefficient access (column-major linear index on matrices; along the vector on vectors)
  const Scalar& coeff(Index i, Index j) const { return derived().at_(i, j); }
  CoeffRef coeffRef(Index i, Index j) { return derived().at_(i, j); }
  const Scalar& operator()(Index i, Index j) const { return derived().at_(i, j); }
  CoeffRef operator()(Index i, Index j) { return derived().at_(i, j); }
  const Scalar& coeff(Index i) const { return cols() == 1 ? derived().at_(i, 0) : (rows() == 1 ? derived().at_(0, i) : derived().at_(i % rows(), i / rows())); }
  CoeffRef coeffRef(Index i) { return cols() == 1 ? derived().at_(i, 0) : (rows() == 1 ? derived().at_(0, i) : derived().at_(i % rows(), i / rows())); }
  const Scalar& operator()(Index i) const { return coeff(i); }
  CoeffRef operator()(Index i) { return coeffRef(i); }
  const Scalar& operator[](Index i) const { return coeff(i); }
  CoeffRef operator[](Index i) { return coeffRef(i); }
  const Scalar& x() const { return coeff(0); }
  const Scalar& y() const { return coeff(1); }
  const Scalar& z() const { return coeff(2); }
  const Scalar& w() const { return coeff(3); }
  CoeffRef x() { return coeffRef(0); }
  CoeffRef y() { return coeffRef(1); }
  CoeffRef z() { return coeffRef(2); }
  CoeffRef w() { return coeffRef(3); }
  Scalar value() const { return coeff(0, 0); }

  PlainObject eval() const { return PlainObject(derived()); }

  // assignment from any expression (the right-hand sides of this library are plain values or views; a view of the destination itself is copied first)
  template <typename Other> Derived& operator=(const MatrixBase<Other>& o) { return assign_(o); }
  Derived& operator=(const MatrixBase& o) { return assign_(o); }      // (not the implicit member-wise one: a base-class reference still copies coefficients)
  MatrixBase() {}
  MatrixBase(const MatrixBase&) {}
  template <typename Other> Derived& assign_(const MatrixBase<Other>& o) {
    derived().resize_like_(o.rows(), o.cols());
    const Index r = rows(), c = cols();
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) coeffRef(i, j) = o.coeff(i, j);
    return derived();
  }
  template <typename Other> Derived& operator+=(const MatrixBase<Other>& o) {
    mini_eigen_assert(rows() == o.rows() && cols() == o.cols());
    const Index r = rows(), c = cols();
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) coeffRef(i, j) += o.coeff(i, j);
    return derived();
  }
  template <typename Other> Derived& operator-=(const MatrixBase<Other>& o) {
    mini_eigen_assert(rows() == o.rows() && cols() == o.cols());
    const Index r = rows(), c = cols();
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) coeffRef(i, j) -= o.coeff(i, j);
    return derived();
  }
  Derived& operator*=(const Scalar& s) { const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) coeffRef(i, j) *= s; return derived(); }
  Derived& operator/=(const Scalar& s) { const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) coeffRef(i, j) /= s; return derived(); }
  template <typename Other> Derived& operator*=(const MatrixBase<Other>& o) { PlainObject t = (*this) * o; return assign_(t); }
  Derived& noalias() { return derived(); }
  void resize(Index r, Index c) { derived().resize_like_(r, c); }       // (views and fixed sizes: must already have that size)
  void resize(Index n) { mini_eigen_assert(n == size()); (void)n; }
  template <typename Other> void swap(const MatrixBase<Other>& o_) {
    Other& o = o_.const_cast_derived();
    const Index r = rows(), c = cols();
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) { Scalar t = coeff(i, j); coeffRef(i, j) = o.coeff(i, j); o.coeffRef(i, j) = t; }
  }

  // setters
  Derived& setConstant(const Scalar& s) { const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) coeffRef(i, j) = s; return derived(); }
  Derived& fill(const Scalar& s) { return setConstant(s); }
  Derived& setZero() { return setConstant(Scalar(0)); }
  Derived& setOnes() { return setConstant(Scalar(1)); }
  Derived& setIdentity() { const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) coeffRef(i, j) = (i == j) ? Scalar(1) : Scalar(0); return derived(); }
  Derived& setIdentity(Index r, Index c) { derived().resize_like_(r, c); return setIdentity(); }
  Derived& setZero(Index r, Index c) { derived().resize_like_(r, c); return setZero(); }
  Derived& setZero(Index n) { derived().resize(n); return setZero(); }
  static PlainObject Zero() { PlainObject m; m.setZero(); return m; }
  static PlainObject Zero(Index n) { PlainObject m(n); m.setZero(); return m; }
  static PlainObject Zero(Index r, Index c) { PlainObject m(r, c); m.setZero(); return m; }
  static PlainObject Ones() { PlainObject m; m.setOnes(); return m; }
  static PlainObject Ones(Index n) { PlainObject m(n); m.setOnes(); return m; }
  static PlainObject Ones(Index r, Index c) { PlainObject m(r, c); m.setOnes(); return m; }
  static PlainObject Constant(const Scalar& s) { PlainObject m; m.setConstant(s); return m; }
  static PlainObject Constant(Index n, const Scalar& s) { PlainObject m(n); m.setConstant(s); return m; }
  static PlainObject Constant(Index r, Index c, const Scalar& s) { PlainObject m(r, c); m.setConstant(s); return m; }
  static PlainObject Identity() { PlainObject m; m.setIdentity(); return m; }
  static PlainObject Identity(Index r, Index c) { PlainObject m(r, c); m.setIdentity(); return m; }
  static PlainObject Unit(Index i) { PlainObject m; m.setZero(); m.coeffRef(i) = Scalar(1); return m; }
  static PlainObject UnitX() { return Unit(0); }
  static PlainObject UnitY() { return Unit(1); }
  static PlainObject UnitZ() { return Unit(2); }
  static PlainObject UnitW() { return Unit(3); }

  // views
  Transpose<Derived> transpose() { return Transpose<Derived>(derived()); }
  Transpose<const Derived> transpose() const { return Transpose<const Derived>(derived()); }
  Transpose<const Derived> adjoint() const { return Transpose<const Derived>(derived()); }
  void transposeInPlace() { PlainObject t(transpose()); assign_(t); }
  Block<Derived, Dynamic, Dynamic> block(Index i, Index j, Index r, Index c) { return Block<Derived, Dynamic, Dynamic>(derived(), i, j, r, c); }
  Block<const Derived, Dynamic, Dynamic> block(Index i, Index j, Index r, Index c) const { return Block<const Derived, Dynamic, Dynamic>(derived(), i, j, r, c); }
  template <int R, int C> Block<Derived, R, C> block(Index i, Index j) { return Block<Derived, R, C>(derived(), i, j, R, C); }
  template <int R, int C> Block<const Derived, R, C> block(Index i, Index j) const { return Block<const Derived, R, C>(derived(), i, j, R, C); }
  template <int R, int C> Block<Derived, R, C> block(Index i, Index j, Index r, Index c) { return Block<Derived, R, C>(derived(), i, j, r, c); }
  template <int R, int C> Block<const Derived, R, C> block(Index i, Index j, Index r, Index c) const { return Block<const Derived, R, C>(derived(), i, j, r, c); }
  template <int R, int C> Block<Derived, R, C> topLeftCorner() { return Block<Derived, R, C>(derived(), 0, 0, R, C); }
  template <int R, int C> Block<const Derived, R, C> topLeftCorner() const { return Block<const Derived, R, C>(derived(), 0, 0, R, C); }
  Block<Derived, Dynamic, Dynamic> topLeftCorner(Index r, Index c) { return block(0, 0, r, c); }
  Block<const Derived, Dynamic, Dynamic> topLeftCorner(Index r, Index c) const { return block(0, 0, r, c); }
  template <int R, int C> Block<Derived, R, C> topRightCorner() { return Block<Derived, R, C>(derived(), 0, cols() - C, R, C); }
  template <int R, int C> Block<const Derived, R, C> topRightCorner() const { return Block<const Derived, R, C>(derived(), 0, cols() - C, R, C); }
  template <int R, int C> Block<Derived, R, C> bottomLeftCorner() { return Block<Derived, R, C>(derived(), rows() - R, 0, R, C); }
  template <int R, int C> Block<const Derived, R, C> bottomLeftCorner() const { return Block<const Derived, R, C>(derived(), rows() - R, 0, R, C); }
  template <int R, int C> Block<Derived, R, C> bottomRightCorner() { return Block<Derived, R, C>(derived(), rows() - R, cols() - C, R, C); }
  template <int R, int C> Block<const Derived, R, C> bottomRightCorner() const { return Block<const Derived, R, C>(derived(), rows() - R, cols() - C, R, C); }
  Block<Derived, RowsAtCompileTime, 1> col(Index j) { return Block<Derived, RowsAtCompileTime, 1>(derived(), 0, j, rows(), 1); }
  Block<const Derived, RowsAtCompileTime, 1> col(Index j) const { return Block<const Derived, RowsAtCompileTime, 1>(derived(), 0, j, rows(), 1); }
  Block<Derived, 1, ColsAtCompileTime> row(Index i) { return Block<Derived, 1, ColsAtCompileTime>(derived(), i, 0, 1, cols()); }
  Block<const Derived, 1, ColsAtCompileTime> row(Index i) const { return Block<const Derived, 1, ColsAtCompileTime>(derived(), i, 0, 1, cols()); }
  Block<Derived, Dynamic, ColsAtCompileTime> topRows(Index n) { return Block<Derived, Dynamic, ColsAtCompileTime>(derived(), 0, 0, n, cols()); }
  Block<const Derived, Dynamic, ColsAtCompileTime> topRows(Index n) const { return Block<const Derived, Dynamic, ColsAtCompileTime>(derived(), 0, 0, n, cols()); }
  Block<Derived, RowsAtCompileTime, Dynamic> leftCols(Index n) { return Block<Derived, RowsAtCompileTime, Dynamic>(derived(), 0, 0, rows(), n); }
  Block<const Derived, RowsAtCompileTime, Dynamic> leftCols(Index n) const { return Block<const Derived, RowsAtCompileTime, Dynamic>(derived(), 0, 0, rows(), n); }
  // vector segments (of a column vector, or of a row vector)
  enum { SegR_ = ColsAtCompileTime == 1 ? 1 : 0 };
  template <int N> struct SegT { typedef Block<Derived, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> type; typedef Block<const Derived, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> ctype; };
  template <int N> typename SegT<N>::type segment(Index s) { return cols() == 1 ? typename SegT<N>::type(derived(), s, 0, N, 1) : typename SegT<N>::type(derived(), 0, s, 1, N); }
  template <int N> typename SegT<N>::ctype segment(Index s) const { return cols() == 1 ? typename SegT<N>::ctype(derived(), s, 0, N, 1) : typename SegT<N>::ctype(derived(), 0, s, 1, N); }
  template <int N> typename SegT<N>::type segment(Index s, Index n) { return cols() == 1 ? typename SegT<N>::type(derived(), s, 0, n, 1) : typename SegT<N>::type(derived(), 0, s, 1, n); }
  template <int N> typename SegT<N>::ctype segment(Index s, Index n) const { return cols() == 1 ? typename SegT<N>::ctype(derived(), s, 0, n, 1) : typename SegT<N>::ctype(derived(), 0, s, 1, n); }
  typename SegT<Dynamic>::type segment(Index s, Index n) { return cols() == 1 ? typename SegT<Dynamic>::type(derived(), s, 0, n, 1) : typename SegT<Dynamic>::type(derived(), 0, s, 1, n); }
  typename SegT<Dynamic>::ctype segment(Index s, Index n) const { return cols() == 1 ? typename SegT<Dynamic>::ctype(derived(), s, 0, n, 1) : typename SegT<Dynamic>::ctype(derived(), 0, s, 1, n); }
  template <int N> typename SegT<N>::type head() { return segment<N>(0); }
  template <int N> typename SegT<N>::ctype head() const { return segment<N>(0); }
  template <int N> typename SegT<N>::type tail() { return segment<N>(size() - N); }
  template <int N> typename SegT<N>::ctype tail() const { return segment<N>(size() - N); }
  typename SegT<Dynamic>::type head(Index n) { return segment(0, n); }
  typename SegT<Dynamic>::ctype head(Index n) const { return segment(0, n); }
  typename SegT<Dynamic>::type tail(Index n) { return segment(size() - n, n); }
  typename SegT<Dynamic>::ctype tail(Index n) const { return segment(size() - n, n); }
  DiagonalView<Derived> diagonal() { return DiagonalView<Derived>(derived()); }
  DiagonalView<const Derived> diagonal() const { return DiagonalView<const Derived>(derived()); }
  ArrayWrapper<Derived> array() { return ArrayWrapper<Derived>(derived()); }
  ArrayWrapper<const Derived> array() const { return ArrayWrapper<const Derived>(derived()); }
  Derived& matrix() { return derived(); }
  const Derived& matrix() const { return derived(); }

  // reductions
  Scalar squaredNorm() const { Scalar s(0); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) s += coeff(i, j) * coeff(i, j); return s; }
  Scalar norm() const { using std::sqrt; return sqrt(squaredNorm()); }
  Scalar stableNorm() const { return norm(); }
  void normalize() { (*this) /= norm(); }
  PlainObject normalized() const { PlainObject t(derived()); t /= norm(); return t; }
  Scalar sum() const { Scalar s(0); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) s += coeff(i, j); return s; }
  Scalar prod() const { Scalar s(1); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) s *= coeff(i, j); return s; }
  Scalar mean() const { return sum() / Scalar(size()); }
  Scalar trace() const { Scalar s(0); const Index n = rows() < cols() ? rows() : cols(); for (Index i = 0; i < n; ++i) s += coeff(i, i); return s; }
  Scalar maxCoeff() const { Scalar s = coeff(0, 0); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) if (coeff(i, j) > s) s = coeff(i, j); return s; }
  Scalar minCoeff() const { Scalar s = coeff(0, 0); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) if (coeff(i, j) < s) s = coeff(i, j); return s; }
  template <typename I> Scalar maxCoeff(I* idx) const { Index b = 0; const Index n = size(); for (Index i = 1; i < n; ++i) if (coeff(i) > coeff(b)) b = i; *idx = (I)b; return coeff(b); }
  template <typename I> Scalar maxCoeff(I* ri, I* ci) const {
    Index br = 0, bc = 0; const Index r = rows(), c = cols();
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) if (coeff(i, j) > coeff(br, bc)) { br = i; bc = j; }
    *ri = (I)br; *ci = (I)bc; return coeff(br, bc);
  }
  bool hasNaN() const { const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) if (coeff(i, j) != coeff(i, j)) return true; return false; }
  bool allFinite() const { const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) if (!std::isfinite((double)coeff(i, j))) return false; return true; }
  template <typename Other> Scalar dot(const MatrixBase<Other>& o) const { Scalar s(0); const Index n = size(); mini_eigen_assert(n == o.size()); for (Index i = 0; i < n; ++i) s += coeff(i) * o.coeff(i); return s; }
  template <typename Other> Matrix<Scalar, 3, 1> cross(const MatrixBase<Other>& o) const {
    Matrix<Scalar, 3, 1> r;
    r.coeffRef(0) = coeff(1) * o.coeff(2) - coeff(2) * o.coeff(1);
    r.coeffRef(1) = coeff(2) * o.coeff(0) - coeff(0) * o.coeff(2);
    r.coeffRef(2) = coeff(0) * o.coeff(1) - coeff(1) * o.coeff(0);
    return r;
  }
  template <typename Other> bool isApprox(const MatrixBase<Other>& o, const Scalar& prec = NumTraits<Scalar>::dummy_precision()) const {
    PlainObject d(derived()); d -= o;
    const Scalar a = squaredNorm(), b = o.squaredNorm();
    return d.squaredNorm() <= prec * prec * (a < b ? a : b);
  }
  PlainObject cwiseAbs() const { PlainObject t(derived()); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) t.coeffRef(i, j) = std::abs(coeff(i, j)); return t; }
  template <typename Other> PlainObject cwiseProduct(const MatrixBase<Other>& o) const { PlainObject t(derived()); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) t.coeffRef(i, j) *= o.coeff(i, j); return t; }
  template <typename Other> PlainObject cwiseQuotient(const MatrixBase<Other>& o) const { PlainObject t(derived()); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) t.coeffRef(i, j) /= o.coeff(i, j); return t; }
  PlainObject cwiseSqrt() const { PlainObject t(derived()); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) t.coeffRef(i, j) = std::sqrt(coeff(i, j)); return t; }
  PlainObject cwiseInverse() const { PlainObject t(derived()); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) t.coeffRef(i, j) = Scalar(1) / coeff(i, j); return t; }
  template <typename U> Matrix<U, RowsAtCompileTime, ColsAtCompileTime> cast() const {
    Matrix<U, RowsAtCompileTime, ColsAtCompileTime> t; t.resize_like_(rows(), cols());
    const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) t.coeffRef(i, j) = static_cast<U>(coeff(i, j));
    return t;
  }

  // determinant / inverse: closed forms up to 3x3 as Eigen has them (cofactors), 4x4 by cofactors of 2x2 minors, larger by partial-pivot LU
  Scalar determinant() const;
  PlainObject inverse() const;
  LLT<PlainObject> llt() const;
  struct PartialPivLU_ {                                 // .lu(): LU with partial pivoting, kept as the factorised inverse
    PlainObject inv;
    template <typename D> typename MatrixBase<D>::PlainObject solve(const MatrixBase<D>& b) const { return typename MatrixBase<D>::PlainObject(inv * b); }
    PlainObject inverse() const { return inv; }
  };
  PartialPivLU_ lu() const { return PartialPivLU_{inverse_lu_()}; }
  PartialPivLU_ partialPivLu() const { return lu(); }
  PlainObject inverse_lu_() const;
  LDLT<PlainObject> ldlt() const;
  template <unsigned int UpLo> PlainObject selfadjointView() const {      // the full symmetric matrix the stored triangle stands for
    PlainObject t(derived()); const Index n = rows();
    for (Index j = 0; j < n; ++j) for (Index i = 0; i < n; ++i) {
      const bool stored = (UpLo & Upper) ? (i <= j) : (i >= j);
      t.coeffRef(i, j) = stored ? coeff(i, j) : coeff(j, i);
    }
    return t;
  }
  template <unsigned int Mode> PlainObject triangularView() const {       // (read-only use: the triangle, zeros elsewhere)
    PlainObject t(derived()); const Index r = rows(), c = cols();
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) {
      bool keep = (Mode & Upper) ? (i <= j) : (i >= j);
      if ((Mode & ZeroDiag) && i == j) keep = false;
      if (!keep) t.coeffRef(i, j) = Scalar(0);
      if ((Mode & UnitDiag) && i == j) t.coeffRef(i, j) = Scalar(1);
    }
    return t;
  }

  // comma initialiser (row by row, scalars only)
  struct CommaInit {
    Derived* m; Index k;
    CommaInit& operator,(const Scalar& s) { const Index c = m->cols(); m->coeffRef(k / c, k % c) = s; ++k; return *this; }
    template <typename Other> CommaInit& operator,(const MatrixBase<Other>& o) {        // a block: only whole columns vectors stacked / row pieces of a single-row layout are used
      const Index c = m->cols();
      if (c == 1) { for (Index i = 0; i < o.size(); ++i) { m->coeffRef(k, 0) = o.coeff(i); ++k; } }
      else { mini_eigen_assert(o.rows() == m->rows()); const Index j0 = k % c; for (Index j = 0; j < o.cols(); ++j) for (Index i = 0; i < o.rows(); ++i) m->coeffRef(i, j0 + j) = o.coeff(i, j); k += o.cols(); }
      return *this;
    }
    Derived& finished() { return *m; }
  };
  CommaInit operator<<(const Scalar& s) { coeffRef(0, 0) = s; return CommaInit{&derived(), 1}; }
  template <typename Other> CommaInit operator<<(const MatrixBase<Other>& o) { CommaInit ci{&derived(), 0}; ci, o; return ci; }

  // scalar products (the scalar is not deduced, so that `2 * v` works on a double vector)
  friend PlainObject operator*(const MatrixBase& a, const Scalar& s) { PlainObject t(a.derived()); t *= s; return t; }
  friend PlainObject operator*(const Scalar& s, const MatrixBase& a) { PlainObject t(a.derived()); const Index r = t.rows(), c = t.cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) t.coeffRef(i, j) = s * a.coeff(i, j); return t; }
  friend PlainObject operator/(const MatrixBase& a, const Scalar& s) { PlainObject t(a.derived()); t /= s; return t; }
  PlainObject operator-() const { PlainObject t(derived()); const Index r = rows(), c = cols(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) t.coeffRef(i, j) = -coeff(i, j); return t; }
};

// ---------------------------------------------------------------------------------------------------------------------------------------
// storage
namespace internal {
template <typename S, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)> struct Storage;
template <typename S, int R, int C> struct Storage<S, R, C, false> {
  S d[R * C > 0 ? R * C : 1];
  Storage() {}
  Index rows() const { return R; }
  Index cols() const { return C; }
  void resize(Index r, Index c) { mini_eigen_assert(r == R && c == C); (void)r; (void)c; }
  S* data() { return d; }
  const S* data() const { return d; }
};
template <typename S, int R, int C> struct Storage<S, R, C, true> {
  std::vector<S> d; Index r_, c_;
  Storage() : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C) {}
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  void resize(Index r, Index c) { if (r * c != (Index)d.size()) d.assign((size_t)(r * c), S()); r_ = r; c_ = c; }     // (Eigen leaves resized storage uninitialised; zero here)
  S* data() { return d.data(); }
  const S* data() const { return d.data(); }
};
}  // namespace internal

template <typename S, int R, int C, int Opt, int MaxR, int MaxC> class Matrix : public MatrixBase<Matrix<S, R, C, Opt, MaxR, MaxC> > {
 public:
  typedef MatrixBase<Matrix> Base;
  typedef S Scalar;
  typedef Eigen::Index Index;
  using Base::operator=;
  enum { Options = Opt };
  typedef Map<Matrix, Unaligned> MapType;
  typedef const Map<const Matrix, Unaligned> ConstMapType;
  typedef Map<Matrix, Aligned> AlignedMapType;
  typedef const Map<const Matrix, Aligned> ConstAlignedMapType;
  Matrix() {}
  Matrix(const Matrix& o) : st_(o.st_) {}
  Matrix& operator=(const Matrix& o) { st_ = o.st_; return *this; }
  template <typename Other> Matrix(const MatrixBase<Other>& o) { this->assign_(o); }
  // (Eigen: a single integer sizes a dynamic vector; on a fixed 1x1 it would be the value - not used by the sources compiled here)
  template <typename T1, typename = typename std::enable_if<std::is_arithmetic<T1>::value>::type> explicit Matrix(const T1& n) {
    if (R == Dynamic && C == Dynamic) st_.resize((Index)n, (Index)n > 0 ? 1 : 0); else if (R == Dynamic) st_.resize((Index)n, C); else if (C == Dynamic) st_.resize(R, (Index)n);
    else if (R * C == 1) st_.data()[0] = S(n);          // a fixed 1x1: the value
  }
  explicit Matrix(const S* p) { std::memcpy(st_.data(), p, sizeof(S) * R * C); }
  // two arguments: sizes of a dynamic matrix, or the two coefficients of a fixed 2-vector
  template <typename A, typename B, typename = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>::type> Matrix(const A& a, const B& b) {
    if (R == Dynamic || C == Dynamic) st_.resize((Index)a, (Index)b);
    else if (R * C == 2 && !(R == 2 && C == 1 && false)) { st_.data()[0] = S(a); st_.data()[1 % (R * C > 0 ? R * C : 1)] = S(b); }
    // else: a fixed-size matrix given sizes keeps its own (Eigen checks them in debug builds only; g2o passes (2, 2) to a 3x3: F3)
  }
  Matrix(const S& a, const S& b, const S& c) { static_assert(R * C == 3, "3 coefficients"); st_.d[0] = a; st_.d[1] = b; st_.d[2] = c; }
  Matrix(const S& a, const S& b, const S& c, const S& d) { static_assert(R * C == 4, "4 coefficients"); st_.d[0] = a; st_.d[1] = b; st_.d[2] = c; st_.d[3] = d; }
  Index rows_() const { return st_.rows(); }
  Index cols_() const { return st_.cols(); }
  S& at_(Index i, Index j) { mini_eigen_assert(i >= 0 && i < rows_() && j >= 0 && j < cols_()); return st_.data()[i + j * st_.rows()]; }
  const S& at_(Index i, Index j) const { mini_eigen_assert(i >= 0 && i < rows_() && j >= 0 && j < cols_()); return st_.data()[i + j * st_.rows()]; }
  void resize_like_(Index r, Index c) { st_.resize(r, c); }
  void resize(Index r, Index c) { st_.resize(r, c); }
  void resize(Index n) { if (C == 1) st_.resize(n, 1); else if (R == 1) st_.resize(1, n); else st_.resize(n, n > 0 ? 1 : 0); }
  void conservativeResize(Index r, Index c) {
    Matrix t(*this); const Index orr = t.rows(), oc = t.cols(); st_.resize(r, c);
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) at_(i, j) = (i < orr && j < oc) ? t.at_(i, j) : S();
  }
  void conservativeResize(Index n) { if (C == 1) conservativeResize(n, 1); else conservativeResize(1, n); }
  S* data() { return st_.data(); }
  const S* data() const { return st_.data(); }
 private:
  internal::Storage<S, R, C> st_;
};

template <typename Plain, int MapOpt, typename Stride> class Map : public MatrixBase<Map<Plain, MapOpt, Stride> > {
 public:
  typedef MatrixBase<Map> Base;
  typedef typename internal::traits<Map>::Scalar Scalar;
  typedef typename std::conditional<std::is_const<Plain>::value, const Scalar, Scalar>::type Elem;
  typedef Eigen::Index Index;
  enum { R = internal::traits<Map>::Rows, C = internal::traits<Map>::Cols };
  using Base::operator=;
  Map(Elem* p) : p_(p), r_(R), c_(C) { static_assert(R != Dynamic && C != Dynamic, "size needed"); }
  Map(Elem* p, Index n) : p_(p), r_(R == Dynamic ? n : R), c_(R == Dynamic ? (C == Dynamic ? 1 : C) : (C == Dynamic ? n : C)) {}
  Map(Elem* p, Index r, Index c) : p_(p), r_(r), c_(c) {}
  Map(const Map& o) : p_(o.p_), r_(o.r_), c_(o.c_) {}
  Map& operator=(const Map& o) { return this->assign_(o); }       // copies the coefficients (Eigen's semantics; g2o re-seats by placement new)
  Index rows_() const { return r_; }
  Index cols_() const { return c_; }
  Elem& at_(Index i, Index j) { return p_[i + j * r_]; }
  const Scalar& at_(Index i, Index j) const { return p_[i + j * r_]; }
  void resize_like_(Index r, Index c) { mini_eigen_assert(r == r_ && c == c_); (void)r; (void)c; }
  Elem* data() { return p_; }
  const Scalar* data() const { return p_; }
 private:
  Elem* p_; Index r_, c_;
};

template <typename Xpr, int BR, int BC> class Block : public MatrixBase<Block<Xpr, BR, BC> > {
 public:
  typedef MatrixBase<Block> Base;
  typedef typename internal::traits<Block>::Scalar Scalar;
  typedef Eigen::Index Index;
  using Base::operator=;
  Block(Xpr& x, Index i0, Index j0, Index r, Index c) : x_(x), i0_(i0), j0_(j0), r_(r), c_(c) { mini_eigen_assert(i0 >= 0 && j0 >= 0 && i0 + r <= x.rows() && j0 + c <= x.cols()); }
  Block(const Block& o) : x_(o.x_), i0_(o.i0_), j0_(o.j0_), r_(o.r_), c_(o.c_) {}
  Block& operator=(const Block& o) { return this->assign_(o); }
  Index rows_() const { return r_; }
  Index cols_() const { return c_; }
  typename Base::CoeffRef at_(Index i, Index j) { return internal::cref<Xpr>::template get<typename Base::CoeffRef>(x_, i0_ + i, j0_ + j); }
  const Scalar& at_(Index i, Index j) const { return const_cast<const Xpr&>(x_).coeff(i0_ + i, j0_ + j); }
  void resize_like_(Index r, Index c) { mini_eigen_assert(r == r_ && c == c_); (void)r; (void)c; }
 private:
  Xpr& x_; Index i0_, j0_, r_, c_;
};

template <typename Xpr> class Transpose : public MatrixBase<Transpose<Xpr> > {
 public:
  typedef MatrixBase<Transpose> Base;
  typedef typename internal::traits<Transpose>::Scalar Scalar;
  typedef Eigen::Index Index;
  using Base::operator=;
  explicit Transpose(Xpr& x) : x_(x) {}
  Transpose(const Transpose& o) : x_(o.x_) {}
  Transpose& operator=(const Transpose& o) { return this->assign_(o); }
  Index rows_() const { return x_.cols(); }
  Index cols_() const { return x_.rows(); }
  typename Base::CoeffRef at_(Index i, Index j) { return internal::cref<Xpr>::template get<typename Base::CoeffRef>(x_, j, i); }
  const Scalar& at_(Index i, Index j) const { return const_cast<const Xpr&>(x_).coeff(j, i); }
  void resize_like_(Index r, Index c) { mini_eigen_assert(r == rows_() && c == cols_()); (void)r; (void)c; }
  const Xpr& nestedExpression() const { return x_; }
 private:
  Xpr& x_;
};

template <typename Xpr> class DiagonalView : public MatrixBase<DiagonalView<Xpr> > {
 public:
  typedef MatrixBase<DiagonalView> Base;
  typedef typename internal::traits<DiagonalView>::Scalar Scalar;
  typedef Eigen::Index Index;
  using Base::operator=;
  explicit DiagonalView(Xpr& x) : x_(x) {}
  DiagonalView(const DiagonalView& o) : x_(o.x_) {}
  DiagonalView& operator=(const DiagonalView& o) { return this->assign_(o); }
  Index rows_() const { return x_.rows() < x_.cols() ? x_.rows() : x_.cols(); }
  Index cols_() const { return 1; }
  typename Base::CoeffRef at_(Index i, Index) { return internal::cref<Xpr>::template get<typename Base::CoeffRef>(x_, i, i); }
  const Scalar& at_(Index i, Index) const { return const_cast<const Xpr&>(x_).coeff(i, i); }
  void resize_like_(Index r, Index c) { mini_eigen_assert(r == rows_() && c == 1); (void)r; (void)c; }
 private:
  Xpr& x_;
};

// .array(): coefficient-wise view with scalar += / -= and abs()
template <typename Xpr> class ArrayWrapper : public MatrixBase<ArrayWrapper<Xpr> > {
 public:
  typedef MatrixBase<ArrayWrapper> Base;
  typedef typename internal::traits<ArrayWrapper>::Scalar Scalar;
  typedef Eigen::Index Index;
  typedef typename Base::PlainObject PlainObject;
  using Base::operator=;
  using Base::operator+=;
  using Base::operator-=;
  explicit ArrayWrapper(Xpr& x) : x_(x) {}
  ArrayWrapper(const ArrayWrapper& o) : x_(o.x_) {}
  Index rows_() const { return x_.rows(); }
  Index cols_() const { return x_.cols(); }
  typename Base::CoeffRef at_(Index i, Index j) { return internal::cref<Xpr>::template get<typename Base::CoeffRef>(x_, i, j); }
  const Scalar& at_(Index i, Index j) const { return const_cast<const Xpr&>(x_).coeff(i, j); }
  void resize_like_(Index r, Index c) { mini_eigen_assert(r == rows_() && c == cols_()); (void)r; (void)c; }
  ArrayWrapper& operator+=(const Scalar& s) { const Index r = rows_(), c = cols_(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) at_(i, j) += s; return *this; }
  ArrayWrapper& operator-=(const Scalar& s) { const Index r = rows_(), c = cols_(); for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) at_(i, j) -= s; return *this; }
  PlainObject abs() const { return this->cwiseAbs(); }
  PlainObject sqrt() const { return this->cwiseSqrt(); }
  PlainObject square() const { return this->cwiseProduct(*this); }
  PlainObject inverse() const { return this->cwiseInverse(); }
 private:
  Xpr& x_;
};

// ---------------------------------------------------------------------------------------------------------------------------------------
// arithmetic: every operator returns a plain matrix
template <typename A, typename B>
inline Matrix<typename internal::traits<A>::Scalar, internal::same_dim<internal::traits<A>::Rows, internal::traits<B>::Rows>::value, internal::same_dim<internal::traits<A>::Cols, internal::traits<B>::Cols>::value>
operator+(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  Matrix<typename internal::traits<A>::Scalar, internal::same_dim<internal::traits<A>::Rows, internal::traits<B>::Rows>::value, internal::same_dim<internal::traits<A>::Cols, internal::traits<B>::Cols>::value> t(a);
  t += b; return t;
}
template <typename A, typename B>
inline Matrix<typename internal::traits<A>::Scalar, internal::same_dim<internal::traits<A>::Rows, internal::traits<B>::Rows>::value, internal::same_dim<internal::traits<A>::Cols, internal::traits<B>::Cols>::value>
operator-(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  Matrix<typename internal::traits<A>::Scalar, internal::same_dim<internal::traits<A>::Rows, internal::traits<B>::Rows>::value, internal::same_dim<internal::traits<A>::Cols, internal::traits<B>::Cols>::value> t(a);
  t -= b; return t;
}
template <typename A, typename B>
inline Matrix<typename internal::traits<A>::Scalar, internal::traits<A>::Rows, internal::traits<B>::Cols> operator*(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  typedef typename internal::traits<A>::Scalar S;
  mini_eigen_assert(a.cols() == b.rows());
  Matrix<S, internal::traits<A>::Rows, internal::traits<B>::Cols> t; t.resize_like_(a.rows(), b.cols());
  const Index r = a.rows(), c = b.cols(), n = a.cols();
  for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) {
    S s(0);
    if (n > 0) { s = a.coeff(i, 0) * b.coeff(0, j); for (Index k = 1; k < n; ++k) s += a.coeff(i, k) * b.coeff(k, j); }
    t.coeffRef(i, j) = s;
  }
  return t;
}
template <typename A, typename B> inline bool operator==(const MatrixBase<A>& a, const MatrixBase<B>& b) {
  if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) if (!(a.coeff(i, j) == b.coeff(i, j))) return false;
  return true;
}
template <typename A, typename B> inline bool operator!=(const MatrixBase<A>& a, const MatrixBase<B>& b) { return !(a == b); }
template <typename D> inline std::ostream& operator<<(std::ostream& os, const MatrixBase<D>& m) {
  for (Index i = 0; i < m.rows(); ++i) { for (Index j = 0; j < m.cols(); ++j) { if (j) os << " "; os << m.coeff(i, j); } if (i + 1 < m.rows()) os << "\n"; }
  return os;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// decompositions
namespace internal {
// partial-pivot LU of a square matrix in place; returns the permutation sign, 0 when singular
template <typename M> int lu_inplace(M& a, std::vector<Index>& piv) {
  typedef typename M::Scalar S;
  const Index n = a.rows(); piv.resize((size_t)n); int sign = 1;
  for (Index k = 0; k < n; ++k) {
    Index p = k; S best = std::abs(a.coeff(k, k));
    for (Index i = k + 1; i < n; ++i) if (std::abs(a.coeff(i, k)) > best) { best = std::abs(a.coeff(i, k)); p = i; }
    piv[(size_t)k] = p;
    if (best == S(0)) { sign = 0; continue; }
    if (p != k) { for (Index j = 0; j < n; ++j) std::swap(a.coeffRef(k, j), a.coeffRef(p, j)); sign = -sign; }
    for (Index i = k + 1; i < n; ++i) {
      a.coeffRef(i, k) /= a.coeff(k, k);
      const S f = a.coeff(i, k);
      for (Index j = k + 1; j < n; ++j) a.coeffRef(i, j) -= f * a.coeff(k, j);
    }
  }
  return sign;
}
}  // namespace internal

template <typename Derived> typename MatrixBase<Derived>::Scalar MatrixBase<Derived>::determinant() const {
  const Index n = rows(); mini_eigen_assert(n == cols());
  const MatrixBase& m = *this;
  if (n == 0) return Scalar(1);
  if (n == 1) return m(0, 0);
  if (n == 2) return m(0, 0) * m(1, 1) - m(1, 0) * m(0, 1);
  if (n == 3) {
    // Eigen: bruteforce_det3_helper(m,0,1,2) - (m,1,0,2) + (m,2,0,1), helper(a,b,c) = m(0,a) * (m(1,b) m(2,c) - m(1,c) m(2,b))
    return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) + m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
  }
  Matrix<Scalar, Dynamic, Dynamic> a(derived()); std::vector<Index> piv;
  const int sign = internal::lu_inplace(a, piv);
  if (sign == 0) return Scalar(0);
  Scalar d = Scalar(sign); for (Index i = 0; i < n; ++i) d *= a.coeff(i, i);
  return d;
}
template <typename Derived> typename MatrixBase<Derived>::PlainObject MatrixBase<Derived>::inverse() const {
  const Index n = rows(); mini_eigen_assert(n == cols());
  const MatrixBase& m = *this;
  PlainObject r; r.resize_like_(n, n);
  if (n == 1) { r(0, 0) = Scalar(1) / m(0, 0); return r; }
  if (n == 2) {
    const Scalar invdet = Scalar(1) / determinant();
    r(0, 0) = m(1, 1) * invdet; r(1, 0) = -m(1, 0) * invdet; r(0, 1) = -m(0, 1) * invdet; r(1, 1) = m(0, 0) * invdet;
    return r;
  }
  if (n == 3) {
    // Eigen's compute_inverse<3>: cofactors of the first column give the determinant, result = cofactor transpose * (1 / det)
    auto cof = [&](int i, int j) { const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return m(i1, j1) * m(i2, j2) - m(i1, j2) * m(i2, j1); };
    const Scalar c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const Scalar det = (c00 * m(0, 0) + c10 * m(1, 0)) + c20 * m(2, 0);
    const Scalar invdet = Scalar(1) / det;
    r(0, 0) = c00 * invdet; r(0, 1) = c10 * invdet; r(0, 2) = c20 * invdet;
    r(1, 0) = cof(0, 1) * invdet; r(1, 1) = cof(1, 1) * invdet; r(1, 2) = cof(2, 1) * invdet;
    r(2, 0) = cof(0, 2) * invdet; r(2, 1) = cof(1, 2) * invdet; r(2, 2) = cof(2, 2) * invdet;
    return r;
  }
  return inverse_lu_();
}
template <typename Derived> typename MatrixBase<Derived>::PlainObject MatrixBase<Derived>::inverse_lu_() const {
  const Index n = rows(); mini_eigen_assert(n == cols());
  PlainObject r; r.resize_like_(n, n);
  Matrix<Scalar, Dynamic, Dynamic> a(derived()); std::vector<Index> piv;
  internal::lu_inplace(a, piv);
  for (Index c = 0; c < n; ++c) {
    std::vector<Scalar> x((size_t)n, Scalar(0)); x[(size_t)c] = Scalar(1);
    for (Index k = 0; k < n; ++k) if (piv[(size_t)k] != k) std::swap(x[(size_t)k], x[(size_t)piv[(size_t)k]]);
    for (Index i = 0; i < n; ++i) for (Index k = 0; k < i; ++k) x[(size_t)i] -= a.coeff(i, k) * x[(size_t)k];
    for (Index i = n - 1; i >= 0; --i) { for (Index k = i + 1; k < n; ++k) x[(size_t)i] -= a.coeff(i, k) * x[(size_t)k]; x[(size_t)i] /= a.coeff(i, i); }
    for (Index i = 0; i < n; ++i) r.coeffRef(i, c) = x[(size_t)i];
  }
  return r;
}

// Cholesky A = L L^T (unblocked, column by column: the diagonal entry, then the column below it)
template <typename MatT, int UpLo> class LLT {
 public:
  typedef typename MatT::Scalar Scalar;
  LLT() : ok_(false) {}
  template <typename D> explicit LLT(const MatrixBase<D>& a) { compute(a); }
  template <typename D> LLT& compute(const MatrixBase<D>& a) {
    const Index n = a.rows(); l_ = a; ok_ = true;
    for (Index k = 0; k < n; ++k) {
      Scalar x = l_.coeff(k, k);
      for (Index j = 0; j < k; ++j) x -= l_.coeff(k, j) * l_.coeff(k, j);
      if (!(x > Scalar(0))) { ok_ = false; return *this; }
      x = std::sqrt(x); l_.coeffRef(k, k) = x;
      for (Index i = k + 1; i < n; ++i) {
        Scalar s = l_.coeff(i, k);
        for (Index j = 0; j < k; ++j) s -= l_.coeff(i, j) * l_.coeff(k, j);
        l_.coeffRef(i, k) = s / x;
      }
    }
    for (Index j = 0; j < n; ++j) for (Index i = 0; i < j; ++i) l_.coeffRef(i, j) = Scalar(0);
    return *this;
  }
  template <typename D> typename MatrixBase<D>::PlainObject solve(const MatrixBase<D>& b) const {
    typename MatrixBase<D>::PlainObject x(b); const Index n = l_.rows();
    for (Index c = 0; c < x.cols(); ++c) {
      for (Index i = 0; i < n; ++i) { Scalar s = x.coeff(i, c); for (Index k = 0; k < i; ++k) s -= l_.coeff(i, k) * x.coeff(k, c); x.coeffRef(i, c) = s / l_.coeff(i, i); }
      for (Index i = n - 1; i >= 0; --i) { Scalar s = x.coeff(i, c); for (Index k = i + 1; k < n; ++k) s -= l_.coeff(k, i) * x.coeff(k, c); x.coeffRef(i, c) = s / l_.coeff(i, i); }
    }
    return x;
  }
  MatT matrixL() const { return l_; }
  MatT matrixU() const { return MatT(l_.transpose()); }
  const MatT& matrixLLT() const { return l_; }
  ComputationInfo info() const { return ok_ ? Success : NumericalIssue; }
 private:
  MatT l_; bool ok_;
};

// Robust Cholesky with diagonal pivoting, P A P^T = L D L^T: the published algorithm of Eigen's LDLT (unblocked, lower) - at step k the largest
// remaining |diagonal| entry is brought to position k by a symmetric transposition; sign bookkeeping as Eigen 3.2.9x / 3.3 keep it (zero -> the
// sign of the first non-zero pivot -> indefinite on a pivot of the other sign); isPositive() is what g2o's LinearSolverDense consults
// (dependencies/g2o/g2o/solvers/linear_solver_dense.h:100-107).  solve(): P, L^-1, D^-1 with pivots not above 1 / highest() treated as zero, L^-T, P^T.
template <typename MatT, int UpLo> class LDLT {
 public:
  typedef typename MatT::Scalar Scalar;
  enum Sign { PositiveSemiDef, NegativeSemiDef, ZeroSign, Indefinite };
  LDLT() : sign_(ZeroSign), ok_(false) {}
  template <typename D> explicit LDLT(const MatrixBase<D>& a) { compute(a); }
  template <typename D> LDLT& compute(const MatrixBase<D>& a) {
    const Index n = a.rows(); m_ = a; tr_.assign((size_t)n, 0); sign_ = ZeroSign; ok_ = true;
    std::vector<Scalar> temp((size_t)n);
    if (n <= 1) { if (n == 1) { tr_[0] = 0; const Scalar d = m_.coeff(0, 0); sign_ = d > Scalar(0) ? PositiveSemiDef : (d < Scalar(0) ? NegativeSemiDef : ZeroSign); } return *this; }
    for (Index k = 0; k < n; ++k) {
      Index p = k; Scalar big = std::abs(m_.coeff(k, k));
      for (Index i = k + 1; i < n; ++i) if (std::abs(m_.coeff(i, i)) > big) { big = std::abs(m_.coeff(i, i)); p = i; }
      tr_[(size_t)k] = p;
      if (k != p) {
        const Index s = n - p - 1;
        for (Index j = 0; j < k; ++j) std::swap(m_.coeffRef(k, j), m_.coeffRef(p, j));
        for (Index i = 0; i < s; ++i) std::swap(m_.coeffRef(p + 1 + i, k), m_.coeffRef(p + 1 + i, p));
        std::swap(m_.coeffRef(k, k), m_.coeffRef(p, p));
        for (Index i = k + 1; i < p; ++i) { const Scalar t = m_.coeff(i, k); m_.coeffRef(i, k) = m_.coeff(p, i); m_.coeffRef(p, i) = t; }
      }
      const Index rs = n - k - 1;
      if (k > 0) {
        for (Index j = 0; j < k; ++j) temp[(size_t)j] = m_.coeff(j, j) * m_.coeff(k, j);
        Scalar acc(0); for (Index j = 0; j < k; ++j) acc += m_.coeff(k, j) * temp[(size_t)j];
        m_.coeffRef(k, k) -= acc;
        for (Index i = 0; i < rs; ++i) { Scalar a2(0); for (Index j = 0; j < k; ++j) a2 += m_.coeff(k + 1 + i, j) * temp[(size_t)j]; m_.coeffRef(k + 1 + i, k) -= a2; }
      }
      const Scalar akk = m_.coeff(k, k);
      const bool pivot_valid = std::abs(akk) > Scalar(0);
      if (k == 0 && !pivot_valid) {          // the whole matrix is zero (the first pivot is the largest diagonal entry)
        sign_ = ZeroSign;
        for (Index j = 0; j < n; ++j) tr_[(size_t)j] = j;
        return *this;
      }
      if (rs > 0 && pivot_valid) for (Index i = 0; i < rs; ++i) m_.coeffRef(k + 1 + i, k) /= akk;
      if (sign_ == PositiveSemiDef) { if (akk < Scalar(0)) sign_ = Indefinite; }
      else if (sign_ == NegativeSemiDef) { if (akk > Scalar(0)) sign_ = Indefinite; }
      else if (sign_ == ZeroSign) { if (akk > Scalar(0)) sign_ = PositiveSemiDef; else if (akk < Scalar(0)) sign_ = NegativeSemiDef; }
    }
    return *this;
  }
  bool isPositive() const { return sign_ == PositiveSemiDef || sign_ == ZeroSign; }
  bool isNegative() const { return sign_ == NegativeSemiDef || sign_ == ZeroSign; }
  template <typename D> typename MatrixBase<D>::PlainObject solve(const MatrixBase<D>& b) const {
    typename MatrixBase<D>::PlainObject x(b); const Index n = m_.rows();
    const Scalar tol = Scalar(1) / NumTraits<Scalar>::highest();
    for (Index c = 0; c < x.cols(); ++c) {
      for (Index k = 0; k < n; ++k) if (tr_[(size_t)k] != k) std::swap(x.coeffRef(k, c), x.coeffRef(tr_[(size_t)k], c));
      for (Index i = 0; i < n; ++i) { Scalar s = x.coeff(i, c); for (Index k = 0; k < i; ++k) s -= m_.coeff(i, k) * x.coeff(k, c); x.coeffRef(i, c) = s; }
      for (Index i = 0; i < n; ++i) { if (std::abs(m_.coeff(i, i)) > tol) x.coeffRef(i, c) /= m_.coeff(i, i); else x.coeffRef(i, c) = Scalar(0); }
      for (Index i = n - 1; i >= 0; --i) { Scalar s = x.coeff(i, c); for (Index k = i + 1; k < n; ++k) s -= m_.coeff(k, i) * x.coeff(k, c); x.coeffRef(i, c) = s; }
      for (Index k = n - 1; k >= 0; --k) if (tr_[(size_t)k] != k) std::swap(x.coeffRef(k, c), x.coeffRef(tr_[(size_t)k], c));
    }
    return x;
  }
  Matrix<Scalar, MatT::RowsAtCompileTime, 1> vectorD() const { Matrix<Scalar, MatT::RowsAtCompileTime, 1> d; d.resize_like_(m_.rows(), 1); for (Index i = 0; i < m_.rows(); ++i) d.coeffRef(i) = m_.coeff(i, i); return d; }
  MatT matrixL() const { MatT l(m_); for (Index j = 0; j < l.cols(); ++j) for (Index i = 0; i <= j && i < l.rows(); ++i) l.coeffRef(i, j) = (i == j) ? Scalar(1) : Scalar(0); return l; }
  const MatT& matrixLDLT() const { return m_; }
  ComputationInfo info() const { return ok_ ? Success : NumericalIssue; }
 private:
  MatT m_; std::vector<Index> tr_; Sign sign_; bool ok_;
};
template <typename Derived> LLT<typename MatrixBase<Derived>::PlainObject> MatrixBase<Derived>::llt() const { return LLT<PlainObject>(derived()); }
template <typename Derived> LDLT<typename MatrixBase<Derived>::PlainObject> MatrixBase<Derived>::ldlt() const { return LDLT<PlainObject>(derived()); }

// symmetric eigenvalues by cyclic Jacobi rotations, ascending (only optimizable_graph.cpp's verifyInformationMatrices asks: "is it SPD")
template <typename MatT> class SelfAdjointEigenSolver {
 public:
  typedef typename MatT::Scalar Scalar;
  typedef Matrix<Scalar, MatT::RowsAtCompileTime, 1> RealVectorType;
  SelfAdjointEigenSolver() {}
  template <typename D> explicit SelfAdjointEigenSolver(const MatrixBase<D>& a, int = ComputeEigenvectors) { compute(a); }
  template <typename D> SelfAdjointEigenSolver& compute(const MatrixBase<D>& a_, int = ComputeEigenvectors) {
    MatT a(a_); const Index n = a.rows(); v_ = a; v_.setIdentity();
    for (int sweep = 0; sweep < 60; ++sweep) {
      Scalar off(0); for (Index j = 0; j < n; ++j) for (Index i = 0; i < j; ++i) off += a(i, j) * a(i, j);
      if (off < Scalar(1e-300)) break;
      for (Index p = 0; p < n; ++p) for (Index q = p + 1; q < n; ++q) {
        if (a(p, q) == Scalar(0)) continue;
        const Scalar th = (a(q, q) - a(p, p)) / (Scalar(2) * a(p, q));
        const Scalar t = (th >= 0 ? Scalar(1) : Scalar(-1)) / (std::abs(th) + std::sqrt(th * th + Scalar(1)));
        const Scalar c = Scalar(1) / std::sqrt(t * t + Scalar(1)), s = t * c;
        for (Index k = 0; k < n; ++k) { const Scalar akp = a(k, p), akq = a(k, q); a(k, p) = c * akp - s * akq; a(k, q) = s * akp + c * akq; }
        for (Index k = 0; k < n; ++k) { const Scalar apk = a(p, k), aqk = a(q, k); a(p, k) = c * apk - s * aqk; a(q, k) = s * apk + c * aqk; }
        for (Index k = 0; k < n; ++k) { const Scalar vkp = v_(k, p), vkq = v_(k, q); v_(k, p) = c * vkp - s * vkq; v_(k, q) = s * vkp + c * vkq; }
      }
    }
    w_.resize_like_(n, 1); for (Index i = 0; i < n; ++i) w_(i) = a(i, i);
    for (Index i = 0; i < n; ++i) { Index b = i; for (Index j = i + 1; j < n; ++j) if (w_(j) < w_(b)) b = j; if (b != i) { std::swap(w_(i), w_(b)); for (Index k = 0; k < n; ++k) std::swap(v_(k, i), v_(k, b)); } }
    return *this;
  }
  const RealVectorType& eigenvalues() const { return w_; }
  const MatT& eigenvectors() const { return v_; }
  ComputationInfo info() const { return Success; }
 private:
  RealVectorType w_; MatT v_;
};

// SVD by one-sided Jacobi rotations (only isometry3d_mappings.h's nearestOrthogonalMatrix asks, for a 3x3)
template <typename MatT, int QRPre = 0> class JacobiSVD {
 public:
  typedef typename MatT::Scalar Scalar;
  typedef Matrix<Scalar, MatT::RowsAtCompileTime, 1> SingularValuesType;
  JacobiSVD() {}
  template <typename D> explicit JacobiSVD(const MatrixBase<D>& a, unsigned int = 0) { compute(a); }
  template <typename D> JacobiSVD& compute(const MatrixBase<D>& a_, unsigned int = 0) {
    MatT a(a_); const Index n = a.cols(), m = a.rows(); v_.resize_like_(n, n); v_.setIdentity();
    for (int sweep = 0; sweep < 60; ++sweep) {
      bool rotated = false;
      for (Index p = 0; p < n; ++p) for (Index q = p + 1; q < n; ++q) {
        Scalar al(0), be(0), ga(0);
        for (Index k = 0; k < m; ++k) { al += a(k, p) * a(k, p); be += a(k, q) * a(k, q); ga += a(k, p) * a(k, q); }
        if (std::abs(ga) <= Scalar(1e-300) || std::abs(ga) <= NumTraits<Scalar>::epsilon() * std::sqrt(al * be)) continue;
        rotated = true;
        const Scalar ze = (be - al) / (Scalar(2) * ga);
        const Scalar t = (ze >= 0 ? Scalar(1) : Scalar(-1)) / (std::abs(ze) + std::sqrt(Scalar(1) + ze * ze));
        const Scalar c = Scalar(1) / std::sqrt(Scalar(1) + t * t), s = c * t;
        for (Index k = 0; k < m; ++k) { const Scalar x = a(k, p), y = a(k, q); a(k, p) = c * x - s * y; a(k, q) = s * x + c * y; }
        for (Index k = 0; k < n; ++k) { const Scalar x = v_(k, p), y = v_(k, q); v_(k, p) = c * x - s * y; v_(k, q) = s * x + c * y; }
      }
      if (!rotated) break;
    }
    w_.resize_like_(n, 1); u_ = a;
    for (Index j = 0; j < n; ++j) { Scalar s(0); for (Index k = 0; k < m; ++k) s += a(k, j) * a(k, j); s = std::sqrt(s); w_(j) = s; if (s > Scalar(0)) for (Index k = 0; k < m; ++k) u_(k, j) = a(k, j) / s; }
    for (Index i = 0; i < n; ++i) { Index b = i; for (Index j = i + 1; j < n; ++j) if (w_(j) > w_(b)) b = j; if (b != i) { std::swap(w_(i), w_(b)); for (Index k = 0; k < n; ++k) std::swap(v_(k, i), v_(k, b)); for (Index k = 0; k < m; ++k) std::swap(u_(k, i), u_(k, b)); } }
    return *this;
  }
  const MatT& matrixU() const { return u_; }
  const MatT& matrixV() const { return v_; }
  const SingularValuesType& singularValues() const { return w_; }
 private:
  MatT u_, v_; SingularValuesType w_;
};

// ---------------------------------------------------------------------------------------------------------------------------------------
// geometry
template <typename S> class AngleAxis {
 public:
  typedef Matrix<S, 3, 1> Vector3;
  typedef Matrix<S, 3, 3> Matrix3;
  AngleAxis() : angle_(0) { axis_.setZero(); }
  template <typename D> AngleAxis(const S& a, const MatrixBase<D>& ax) : angle_(a), axis_(ax) {}
  explicit AngleAxis(const Quaternion<S>& q) { *this = q; }
  template <typename D> explicit AngleAxis(const MatrixBase<D>& m) { *this = Quaternion<S>(m); }
  AngleAxis& operator=(const Quaternion<S>& q);
  S angle() const { return angle_; }
  S& angle() { return angle_; }
  const Vector3& axis() const { return axis_; }
  Vector3& axis() { return axis_; }
  Matrix3 toRotationMatrix() const {
    // Eigen: c*I + s*[axis]x + (1-c) axis axis^T, with cos_axis = (1-c) axis
    Matrix3 res; const S s = std::sin(angle_), c = std::cos(angle_);
    Vector3 cos1_axis = (S(1) - c) * axis_;
    S tmp;
    tmp = cos1_axis.x() * axis_.y(); res(0, 1) = tmp - s * axis_.z(); res(1, 0) = tmp + s * axis_.z();
    tmp = cos1_axis.x() * axis_.z(); res(0, 2) = tmp + s * axis_.y(); res(2, 0) = tmp - s * axis_.y();
    tmp = cos1_axis.y() * axis_.z(); res(1, 2) = tmp - s * axis_.x(); res(2, 1) = tmp + s * axis_.x();
    res(0, 0) = cos1_axis.x() * axis_.x() + c; res(1, 1) = cos1_axis.y() * axis_.y() + c; res(2, 2) = cos1_axis.z() * axis_.z() + c;
    return res;
  }
  Matrix3 matrix() const { return toRotationMatrix(); }
  AngleAxis inverse() const { return AngleAxis(-angle_, axis_); }
 private:
  S angle_; Vector3 axis_;
};

template <typename S> class Quaternion {
 public:
  typedef S Scalar;
  typedef Matrix<S, 4, 1> Coefficients;
  typedef Matrix<S, 3, 1> Vector3;
  typedef Matrix<S, 3, 3> Matrix3;
  typedef AngleAxis<S> AngleAxisType;
  Quaternion() {}
  Quaternion(const S& w, const S& x, const S& y, const S& z) { c_(0) = x; c_(1) = y; c_(2) = z; c_(3) = w; }
  explicit Quaternion(const S* d) { c_(0) = d[0]; c_(1) = d[1]; c_(2) = d[2]; c_(3) = d[3]; }
  Quaternion(const Quaternion& o) : c_(o.c_) {}
  explicit Quaternion(const AngleAxis<S>& aa) { *this = aa; }
  // a 3x3 expression is a rotation matrix, a 4-vector the coefficients (x, y, z, w)
  template <typename D> explicit Quaternion(const MatrixBase<D>& m) { set_(m); }
  template <typename U> explicit Quaternion(const Quaternion<U>& o) { c_(0) = S(o.x()); c_(1) = S(o.y()); c_(2) = S(o.z()); c_(3) = S(o.w()); }
  Quaternion& operator=(const Quaternion& o) { c_ = o.c_; return *this; }
  Quaternion& operator=(const AngleAxis<S>& aa) {
    const S ha = S(0.5) * aa.angle();
    c_(3) = std::cos(ha);
    const S sh = std::sin(ha);
    c_(0) = sh * aa.axis()(0); c_(1) = sh * aa.axis()(1); c_(2) = sh * aa.axis()(2);
    return *this;
  }
  template <typename D> Quaternion& operator=(const MatrixBase<D>& m) { set_(m); return *this; }
  static Quaternion Identity() { return Quaternion(S(1), S(0), S(0), S(0)); }
  Quaternion& setIdentity() { c_(0) = S(0); c_(1) = S(0); c_(2) = S(0); c_(3) = S(1); return *this; }
  const S& x() const { return c_(0); }
  const S& y() const { return c_(1); }
  const S& z() const { return c_(2); }
  const S& w() const { return c_(3); }
  S& x() { return c_(0); }
  S& y() { return c_(1); }
  S& z() { return c_(2); }
  S& w() { return c_(3); }
  const Coefficients& coeffs() const { return c_; }
  Coefficients& coeffs() { return c_; }
  Block<Coefficients, 3, 1> vec() { return c_.template head<3>(); }
  Block<const Coefficients, 3, 1> vec() const { return c_.template head<3>(); }
  S squaredNorm() const { return c_.squaredNorm(); }
  S norm() const { return c_.norm(); }
  void normalize() { c_.normalize(); }
  Quaternion normalized() const { Quaternion q(*this); q.normalize(); return q; }
  S dot(const Quaternion& o) const { return c_.dot(o.c_); }
  Quaternion conjugate() const { return Quaternion(w(), -x(), -y(), -z()); }
  Quaternion inverse() const {
    const S n2 = squaredNorm();
    if (n2 > S(0)) { Quaternion q = conjugate(); q.c_ /= n2; return q; }
    Quaternion q; q.c_.setZero(); return q;
  }
  // Hamilton product, in the order of Eigen's scalar path
  Quaternion operator*(const Quaternion& b) const {
    const Quaternion& a = *this;
    return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                      a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                      a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                      a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
  }
  Quaternion& operator*=(const Quaternion& b) { *this = (*this) * b; return *this; }
  // rotation of a vector: v + w * (2 q_v x v) + q_v x (2 q_v x v)   (Eigen's _transformVector)
  template <typename D> Vector3 operator*(const MatrixBase<D>& v) const { return _transformVector(Vector3(v)); }
  Vector3 _transformVector(const Vector3& v) const {
    Vector3 qv(x(), y(), z());
    Vector3 uv = qv.cross(v);
    uv += uv;
    return v + w() * uv + qv.cross(uv);
  }
  Matrix3 toRotationMatrix() const {
    Matrix3 res;
    const S tx = S(2) * x(), ty = S(2) * y(), tz = S(2) * z();
    const S twx = tx * w(), twy = ty * w(), twz = tz * w();
    const S txx = tx * x(), txy = ty * x(), txz = tz * x();
    const S tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
    res(0, 0) = S(1) - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
    res(1, 0) = txy + twz; res(1, 1) = S(1) - (txx + tzz); res(1, 2) = tyz - twx;
    res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = S(1) - (txx + tyy);
    return res;
  }
  Matrix3 matrix() const { return toRotationMatrix(); }
  S angularDistance(const Quaternion& o) const {
    Quaternion d = (*this) * o.conjugate();
    return S(2) * std::atan2(d.vec().norm(), std::abs(d.w()));
  }
  bool isApprox(const Quaternion& o, const S& prec = NumTraits<S>::dummy_precision()) const { return c_.isApprox(o.c_, prec); }
  template <typename U> Quaternion<U> cast() const { return Quaternion<U>(U(w()), U(x()), U(y()), U(z())); }
 private:
  template <typename D> void set_(const MatrixBase<D>& m) {
    if (m.rows() == 3 && m.cols() == 3) {
      // rotation matrix -> quaternion as Eigen's quaternionbase_assign_impl<Other,3,3> (Shoemake): the trace branch first, otherwise the largest diagonal entry
      S t = m.coeff(0, 0) + m.coeff(1, 1) + m.coeff(2, 2);
      if (t > S(0)) {
        t = std::sqrt(t + S(1.0));
        w() = S(0.5) * t;
        t = S(0.5) / t;
        x() = (m.coeff(2, 1) - m.coeff(1, 2)) * t;
        y() = (m.coeff(0, 2) - m.coeff(2, 0)) * t;
        z() = (m.coeff(1, 0) - m.coeff(0, 1)) * t;
      } else {
        Index i = 0;
        if (m.coeff(1, 1) > m.coeff(0, 0)) i = 1;
        if (m.coeff(2, 2) > m.coeff(i, i)) i = 2;
        const Index j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m.coeff(i, i) - m.coeff(j, j) - m.coeff(k, k) + S(1.0));
        c_(i) = S(0.5) * t;
        t = S(0.5) / t;
        w() = (m.coeff(k, j) - m.coeff(j, k)) * t;
        c_(j) = (m.coeff(j, i) + m.coeff(i, j)) * t;
        c_(k) = (m.coeff(k, i) + m.coeff(i, k)) * t;
      }
    } else {
      mini_eigen_assert(m.size() == 4);
      for (Index i = 0; i < 4; ++i) c_(i) = m.coeff(i);
    }
  }
  Coefficients c_;
};
template <typename S> AngleAxis<S>& AngleAxis<S>::operator=(const Quaternion<S>& q) {
  S n = q.vec().norm();
  if (n < NumTraits<S>::epsilon()) n = q.vec().stableNorm();
  if (n != S(0)) { angle_ = S(2) * std::atan2(n, std::abs(q.w())); if (q.w() < S(0)) n = -n; axis_ = Vector3(q.vec()) / n; }
  else { angle_ = S(0); axis_ = Vector3(S(1), S(0), S(0)); }
  return *this;
}
template <typename S> inline std::ostream& operator<<(std::ostream& os, const Quaternion<S>& q) { return os << q.x() << "i + " << q.y() << "j + " << q.z() << "k + " << q.w(); }

template <typename S, int Dim> class Translation {
 public:
  typedef Matrix<S, Dim, 1> VectorType;
  Translation() {}
  explicit Translation(const VectorType& v) : v_(v) {}
  Translation(const S& x, const S& y, const S& z) : v_(x, y, z) {}
  const VectorType& vector() const { return v_; }
  const VectorType& translation() const { return v_; }
 private:
  VectorType v_;
};

template <typename S> class Rotation2D {
 public:
  Rotation2D() : a_(0) {}
  explicit Rotation2D(const S& a) : a_(a) {}
  S angle() const { return a_; }
  S& angle() { return a_; }
  Matrix<S, 2, 2> toRotationMatrix() const { Matrix<S, 2, 2> m; const S s = std::sin(a_), c = std::cos(a_); m(0, 0) = c; m(0, 1) = -s; m(1, 0) = s; m(1, 1) = c; return m; }
  Matrix<S, 2, 2> matrix() const { return toRotationMatrix(); }
  Rotation2D inverse() const { return Rotation2D(-a_); }
  Rotation2D operator*(const Rotation2D& o) const { return Rotation2D(a_ + o.a_); }
  template <typename D> Matrix<S, 2, 1> operator*(const MatrixBase<D>& v) const { return toRotationMatrix() * v; }
  template <typename D> Rotation2D& fromRotationMatrix(const MatrixBase<D>& m) { a_ = std::atan2(m.coeff(1, 0), m.coeff(0, 0)); return *this; }
 private:
  S a_;
};

// homogeneous transform stored as a (Dim+1)x(Dim+1) matrix
template <typename S, int Dim, int Mode, int Opt> class Transform {
 public:
  typedef S Scalar;
  enum { HDim = Dim + 1 };
  typedef Matrix<S, Dim + 1, Dim + 1> MatrixType;
  typedef Matrix<S, Dim, Dim> LinearMatrixType;
  typedef Matrix<S, Dim, 1> VectorType;
  typedef Block<MatrixType, Dim, Dim> LinearPart;
  typedef Block<const MatrixType, Dim, Dim> ConstLinearPart;
  typedef Block<MatrixType, Dim, 1> TranslationPart;
  typedef Block<const MatrixType, Dim, 1> ConstTranslationPart;
  typedef Block<MatrixType, Dim, Dim + 1> AffinePart;
  typedef Block<const MatrixType, Dim, Dim + 1> ConstAffinePart;
  Transform() { m_.setZero(); m_(Dim, Dim) = S(1); }      // (Eigen leaves the upper part uninitialised and sets the last row for non-projective modes)
  Transform(const Transform& o) : m_(o.m_) {}
  template <typename D> explicit Transform(const MatrixBase<D>& m) { *this = m; }
  explicit Transform(const Quaternion<S>& q) { *this = q; }
  explicit Transform(const AngleAxis<S>& a) { *this = a; }
  explicit Transform(const Translation<S, Dim>& t) { setIdentity(); translation() = t.vector(); }
  template <int OMode, int OOpt> Transform(const Transform<S, Dim, OMode, OOpt>& o) : m_(o.matrix()) {}
  Transform& operator=(const Transform& o) { m_ = o.m_; return *this; }
  // a (Dim+1)^2 matrix is the whole transform, a Dim^2 matrix the linear part (translation zeroed) - Eigen's transform_construct_from_matrix
  template <typename D> Transform& operator=(const MatrixBase<D>& m) {
    if (m.rows() == Dim + 1 && m.cols() == Dim + 1) m_ = m;
    else if (m.rows() == Dim && m.cols() == Dim + 1) { m_.setIdentity(); affine() = m; }
    else { mini_eigen_assert(m.rows() == Dim && m.cols() == Dim); m_.setIdentity(); linear() = m; }
    return *this;
  }
  Transform& operator=(const Quaternion<S>& q) { m_.setIdentity(); linear() = q.toRotationMatrix(); return *this; }
  Transform& operator=(const AngleAxis<S>& a) { m_.setIdentity(); linear() = a.toRotationMatrix(); return *this; }
  Transform& operator=(const Translation<S, Dim>& t) { m_.setIdentity(); translation() = t.vector(); return *this; }
  static Transform Identity() { Transform t; t.setIdentity(); return t; }
  void setIdentity() { m_.setIdentity(); }
  Index rows() const { return Dim + 1; }
  Index cols() const { return Dim + 1; }
  const MatrixType& matrix() const { return m_; }
  MatrixType& matrix() { return m_; }
  S operator()(Index i, Index j) const { return m_(i, j); }
  S& operator()(Index i, Index j) { return m_(i, j); }
  ConstLinearPart linear() const { return ConstLinearPart(m_, 0, 0, Dim, Dim); }
  LinearPart linear() { return LinearPart(m_, 0, 0, Dim, Dim); }
  ConstAffinePart affine() const { return ConstAffinePart(m_, 0, 0, Dim, Dim + 1); }
  AffinePart affine() { return AffinePart(m_, 0, 0, Dim, Dim + 1); }
  ConstTranslationPart translation() const { return ConstTranslationPart(m_, 0, Dim, Dim, 1); }
  TranslationPart translation() { return TranslationPart(m_, 0, Dim, Dim, 1); }
  // Isometry mode: the rotation IS the linear part (Eigen returns it without a decomposition); other modes are not used by the sources compiled here
  LinearMatrixType rotation() const { return LinearMatrixType(linear()); }
  void makeAffine() { for (int j = 0; j < Dim; ++j) m_(Dim, j) = S(0); m_(Dim, Dim) = S(1); }
  const S* data() const { return m_.data(); }
  S* data() { return m_.data(); }
  // composition: affine parts only (last row 0 .. 0 1), as Eigen does for non-projective modes
  Transform operator*(const Transform& o) const {
    Transform r;
    r.linear() = linear() * o.linear();
    r.translation() = linear() * o.translation() + translation();
    r.makeAffine();
    return r;
  }
  Transform& operator*=(const Transform& o) { *this = (*this) * o; return *this; }
  // applied to a point (Dim) / a homogeneous vector or matrix (Dim+1 rows)
  template <typename D> Matrix<S, internal::traits<D>::Rows, internal::traits<D>::Cols> operator*(const MatrixBase<D>& v) const {
    Matrix<S, internal::traits<D>::Rows, internal::traits<D>::Cols> r;
    if (v.rows() == Dim) {
      r.resize_like_(v.rows(), v.cols());
      for (Index c = 0; c < v.cols(); ++c) { VectorType t = linear() * v.col(c) + translation(); for (Index i = 0; i < Dim; ++i) r.coeffRef(i, c) = t(i); }
    } else { mini_eigen_assert(v.rows() == Dim + 1); r = m_ * v; }
    return r;
  }
  Transform operator*(const Quaternion<S>& q) const { Transform r(*this); r.linear() = linear() * q.toRotationMatrix(); return r; }
  Transform operator*(const Translation<S, Dim>& t) const { Transform r(*this); r.translate(t.vector()); return r; }
  template <typename D> Transform& translate(const MatrixBase<D>& v) { VectorType t = linear() * v; translation() += t; return *this; }
  template <typename D> Transform& pretranslate(const MatrixBase<D>& v) { translation() += v; return *this; }
  template <typename D> Transform& rotate(const MatrixBase<D>& r) { LinearMatrixType t = linear() * r; linear() = t; return *this; }
  Transform& rotate(const Quaternion<S>& q) { return rotate(q.toRotationMatrix()); }
  Transform& rotate(const AngleAxis<S>& a) { return rotate(a.toRotationMatrix()); }
  template <typename D> Transform& prerotate(const MatrixBase<D>& r) { Matrix<S, Dim, Dim + 1> t = r * affine(); affine() = t; return *this; }
  Transform& prerotate(const Quaternion<S>& q) { return prerotate(q.toRotationMatrix()); }
  Transform inverse(TransformTraits hint = (TransformTraits)Mode) const {
    Transform r;
    if (hint == Projective) { r.m_ = m_.inverse(); return r; }
    LinearMatrixType li;
    if (hint == Isometry) li = linear().transpose(); else li = linear().inverse();
    r.linear() = li;
    r.translation() = -(li * translation());
    r.makeAffine();
    return r;
  }
  bool isApprox(const Transform& o, const S& prec = NumTraits<S>::dummy_precision()) const { return m_.isApprox(o.m_, prec); }
  template <typename U> Transform<U, Dim, Mode, Opt> cast() const { Transform<U, Dim, Mode, Opt> r; r.matrix() = m_.template cast<U>(); return r; }
 private:
  MatrixType m_;
};
template <typename S, int Dim> inline Transform<S, Dim, Affine> operator*(const Translation<S, Dim>& t, const Quaternion<S>& q) { Transform<S, Dim, Affine> r; r.setIdentity(); r.linear() = q.toRotationMatrix(); r.translation() = t.vector(); return r; }

// ---------------------------------------------------------------------------------------------------------------------------------------
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<double, 1, 2> RowVector2d;
typedef Matrix<double, 1, 3> RowVector3d;
typedef Matrix<double, 1, Dynamic> RowVectorXd;
typedef Matrix<float, 2, 1> Vector2f;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<float, Dynamic, 1> VectorXf;
typedef Matrix<int, 2, 1> Vector2i;
typedef Matrix<int, 3, 1> Vector3i;
typedef Matrix<int, 4, 1> Vector4i;
typedef Matrix<int, Dynamic, 1> VectorXi;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<float, 2, 2> Matrix2f;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<float, Dynamic, Dynamic> MatrixXf;
typedef Matrix<int, Dynamic, Dynamic> MatrixXi;
typedef Quaternion<double> Quaterniond;
typedef Quaternion<float> Quaternionf;
typedef AngleAxis<double> AngleAxisd;
typedef AngleAxis<float> AngleAxisf;
typedef Rotation2D<double> Rotation2Dd;
typedef Transform<double, 2, Isometry> Isometry2d;
typedef Transform<double, 3, Isometry> Isometry3d;
typedef Transform<double, 2, Affine> Affine2d;
typedef Transform<double, 3, Affine> Affine3d;
typedef Translation<double, 3> Translation3d;
}  // namespace Eigen
#endif
