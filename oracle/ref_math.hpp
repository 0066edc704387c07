// TEST INFRASTRUCTURE — CPU oracle for the VDO-SLAM hot path.  Not shipped, not
// linked by the product library.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use anything under oracle/.
//
// ref_math.hpp: dependency-free restatement (fp64) of the small fixed-size
// algebra the reference gets from Eigen3 + g2o:
//   * Eigen::Quaterniond(Matrix3d) / toRotationMatrix / operator* / normalize
//     (Eigen 3.x Geometry/Quaternion.h — NOT vendored in /root/reference, restated
//     from the published algorithm; parity unpinned)
//   * g2o::SE3Quat            (dependencies/g2o/g2o/types/se3quat.h:41-301)
//   * g2o::internal mappings  (dependencies/g2o/g2o/types/isometry3d_mappings.cpp:33-160)
//   * se3_ops skew/deltaR     (dependencies/g2o/g2o/types/se3_ops.hpp)
#pragma once
#include <cmath>
#include <cstring>

namespace vdo_oracle {

struct V3 { double x, y, z; };
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Row-major 3x3.
struct M3 {
  double m[9];
  double& operator()(int r, int c) { return m[3 * r + c]; }
  double operator()(int r, int c) const { return m[3 * r + c]; }
};
inline M3 m3_identity() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
inline M3 m3_zero() { return M3{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
inline M3 mul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += a(i, k) * b(k, j);
      r(i, j) = s;
    }
  return r;
}
inline V3 mul(const M3& a, V3 v) {
  return {a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z,
          a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z,
          a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z};
}
inline M3 transpose(const M3& a) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r(i, j) = a(j, i);
  return r;
}
inline M3 add(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i];
  return r;
}
inline M3 scale(double s, const M3& a) {
  M3 r;
  for (int i = 0; i < 9; ++i) r.m[i] = s * a.m[i];
  return r;
}
// se3_ops.hpp skew(): [0 -z y; z 0 -x; -y x 0]
inline M3 skew(V3 v) { return M3{{0, -v.z, v.y, v.z, 0, -v.x, -v.y, v.x, 0}}; }

// ---- Eigen::Quaterniond restatement (x,y,z,w) -------------------------------
struct Quat { double x, y, z, w; };

// Eigen quaternion-from-rotation-matrix (trace branch first, then largest diagonal).
inline Quat quat_from_matrix(const M3& m) {
  Quat q;
  double t = m(0, 0) + m(1, 1) + m(2, 2);
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m(2, 1) - m(1, 2)) * t;
    q.y = (m(0, 2) - m(2, 0)) * t;
    q.z = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    double c[3];
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m(k, j) - m(j, k)) * t;
    c[j] = (m(j, i) + m(i, j)) * t;
    c[k] = (m(k, i) + m(i, k)) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}

inline M3 quat_to_matrix(const Quat& q) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3 r;
  r(0, 0) = 1 - (tyy + tzz); r(0, 1) = txy - twz;       r(0, 2) = txz + twy;
  r(1, 0) = txy + twz;       r(1, 1) = 1 - (txx + tzz); r(1, 2) = tyz - twx;
  r(2, 0) = txz - twy;       r(2, 1) = tyz + twx;       r(2, 2) = 1 - (txx + tyy);
  return r;
}

inline Quat quat_mul(const Quat& a, const Quat& b) {
  return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
              a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline double quat_norm(const Quat& q) {
  return std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
}
inline void quat_normalize(Quat& q) {
  double n = quat_norm(q);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
// Eigen _transformVector: v + w*uv + qv x uv,  uv = 2 * (qv x v)
inline V3 quat_rotate(const Quat& q, V3 v) {
  V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + cross(qv, uv);
}
inline Quat quat_conj(const Quat& q) { return Quat{-q.x, -q.y, -q.z, q.w}; }

// ---- g2o::SE3Quat (se3quat.h) ------------------------------------------------
struct SE3Quat {
  Quat r{0, 0, 0, 1};
  V3 t{0, 0, 0};
  // se3quat.h:289-294
  void normalizeRotation() {
    if (r.w < 0) { r.x = -r.x; r.y = -r.y; r.z = -r.z; r.w = -r.w; }
    quat_normalize(r);
  }
  static SE3Quat fromRt(const M3& R, V3 t) {  // se3quat.h:58-60
    SE3Quat s; s.r = quat_from_matrix(R); s.t = t; s.normalizeRotation(); return s;
  }
  V3 map(V3 p) const { return quat_rotate(r, p) + t; }  // :218-221
  SE3Quat compose(const SE3Quat& o) const {                // :106-112
    SE3Quat res = *this;
    res.t = res.t + quat_rotate(r, o.t);
    res.r = quat_mul(r, o.r);
    res.normalizeRotation();
    return res;
  }
  SE3Quat inverse() const {                              // :126-131
    SE3Quat ret; ret.r = quat_conj(r); ret.t = quat_rotate(ret.r, -1.0 * t); return ret;
  }
  // se3quat.h:229-262: update = (omega, upsilon)
  static SE3Quat exp(const double u[6]) {
    V3 omega{u[0], u[1], u[2]}, upsilon{u[3], u[4], u[5]};
    double theta = std::sqrt(dot(omega, omega));
    M3 Omega = skew(omega);
    M3 R, V;
    if (theta < 0.00001) {
      R = add(add(m3_identity(), Omega), mul(Omega, Omega));
      V = R;
    } else {
      M3 Omega2 = mul(Omega, Omega);
      R = add(add(m3_identity(), scale(std::sin(theta) / theta, Omega)),
              scale((1 - std::cos(theta)) / (theta * theta), Omega2));
      V = add(add(m3_identity(), scale((1 - std::cos(theta)) / (theta * theta), Omega)),
              scale((theta - std::sin(theta)) / (std::pow(theta, 3)), Omega2));
    }
    SE3Quat s; s.r = quat_from_matrix(R); s.t = mul(V, upsilon); s.normalizeRotation();
    return s;
  }
  void toMatrix4(double T[16]) const {                   // :276-284
    M3 R = quat_to_matrix(r);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[4 * i + j] = R(i, j); }
    T[3] = t.x; T[7] = t.y; T[11] = t.z; T[12] = T[13] = T[14] = 0; T[15] = 1;
  }
};

// ---- Isometry3 (Eigen::Transform<double,3,Isometry>) as R|t -------------------
struct Iso {
  M3 R = m3_identity();
  V3 t{0, 0, 0};
};
inline Iso iso_mul(const Iso& a, const Iso& b) { return Iso{mul(a.R, b.R), mul(a.R, b.t) + a.t}; }
inline Iso iso_inv(const Iso& a) {
  Iso r; r.R = transpose(a.R); r.t = -1.0 * mul(r.R, a.t); return r;
}
inline V3 iso_apply(const Iso& a, V3 p) { return mul(a.R, p) + a.t; }
// SE3Quat::operator Isometry3d (se3quat.h:299-304)
inline Iso iso_from_se3quat(const SE3Quat& s) { return Iso{quat_to_matrix(s.r), s.t}; }
inline void iso_to12(const Iso& a, double* o) { std::memcpy(o, a.R.m, 72); o[9] = a.t.x; o[10] = a.t.y; o[11] = a.t.z; }
inline Iso iso_from12(const double* o) { Iso a; std::memcpy(a.R.m, o, 72); a.t = {o[9], o[10], o[11]}; return a; }

// isometry3d_mappings.cpp:75-80  (Quaternion(R); normalize; w>=0; xyz)
inline V3 toCompactQuaternion(const M3& R) {
  Quat q = quat_from_matrix(R);
  quat_normalize(q);
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  return {q.x, q.y, q.z};
}
// isometry3d_mappings.cpp:82-89
inline M3 fromCompactQuaternion(V3 v) {
  double w = 1 - dot(v, v);
  if (w < 0) return m3_identity();
  w = std::sqrt(w);
  return quat_to_matrix(Quat{v.x, v.y, v.z, w});
}
// :92-97, e = (t, q_xyz)
inline void toVectorMQT(const Iso& T, double e[6]) {
  V3 q = toCompactQuaternion(T.R);
  e[0] = T.t.x; e[1] = T.t.y; e[2] = T.t.z; e[3] = q.x; e[4] = q.y; e[5] = q.z;
}
// :118-123
inline Iso fromVectorMQT(const double v[6]) {
  Iso T; T.R = fromCompactQuaternion({v[3], v[4], v[5]}); T.t = {v[0], v[1], v[2]}; return T;
}
// isometry3d_mappings.h approximateNearestOrthogonalMatrix: R -= 0.5 R (R^T R - I)
inline void approximateNearestOrthogonalMatrix(M3& R) {
  M3 E = mul(transpose(R), R);
  E(0, 0) -= 1; E(1, 1) -= 1; E(2, 2) -= 1;
  M3 RE = mul(R, E);
  for (int i = 0; i < 9; ++i) R.m[i] -= 0.5 * RE.m[i];
}

// Huber kernel, robust_kernel_impl.cpp:65-91.  NB `dsqr` is a *float* member
// (robust_kernel_impl.h:84) so delta^2 is rounded to fp32 — replicated.
struct Huber {
  double delta = 1.0;
  double dsqr = 1.0;
  void setDelta(double d) { delta = d; dsqr = (double)(float)(d * d); }
  // returns rho[0], rho[1]
  inline void robustify(double e, double& rho0, double& rho1) const {
    if (e <= dsqr) { rho0 = e; rho1 = 1.0; }
    else { double s = std::sqrt(e); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
  }
};

}  // namespace vdo_oracle
