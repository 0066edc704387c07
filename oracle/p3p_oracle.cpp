// TEST INFRASTRUCTURE — CPU oracle (see vdo_oracle.h).  RANSAC initialiser of the per-frame pose problems:
// Tracking::GetInitModelCam / GetInitModelObj call cv::solvePnPRansac(..., 500 its, 0.4 px, 0.98, SOLVEPNP_AP3P)
// (reference src/Tracking.cc:1614-1715, 1717-1849).  OpenCV 3.4 is not available here, so this restates its
// published scheme (parity unpinned):
//   * cv::RNG (multiply-with-carry, coefficient 4164903690, seed (uint64)-1 in RANSACPointSetRegistrator::run),
//     subsets of 4 distinct indices drawn by rejection (getSubset);
//   * minimal solver on the first 3 points, the 4th picks among the <= 4 solutions by reprojection error
//     (p3p::solve / ap3p::solve with 4 points).  Default since round 5: AP3P (ap3p_oracle.cpp, the libm-free form).
//     `refit & 2`: Grunert's quartic in v = d3/d1 (coefficients derived symbolically, see tests/test_oracle_p3p.py) +
//     absolute orientation of the 3 points - rounds 1-4's solver; AP3P returns the same geometric solutions (+ mirrored ones);
//   * inliers: squared reprojection error <= thr^2 (findInliers); the iteration budget shrinks with
//     RANSACUpdateNumIters(confidence, outlier ratio, 4, niters) whenever a better model is found.
//   * the final re-estimation of the winning model on its inliers by EPnP (OpenCV >= 3.3): epnp_oracle.hpp,
//     vdo_oracle_pnp_ransac_refit below.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "epnp_oracle.hpp"
#include "vdo_oracle.h"

namespace {

struct RNG {
  uint64_t state;
  explicit RNG(uint64_t s) : state(s ? s : 0xffffffffULL) {}
  unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

inline double from_bits(uint64_t b) { double d; std::memcpy(&d, &b, 8); return d; }
inline uint64_t to_bits(double d) { uint64_t b; std::memcpy(&b, &d, 8); return b; }

// Cube root and the largest root of a three-real-root depressed cubic written with +, -, *, / and sqrt only (IEEE, correctly
// rounded everywhere) and integer arithmetic on the exponent field - deliberately NOT std::cbrt / std::cos / std::acos: the
// GPU build (vdo_slam_amd/csrc/ransac.hip) runs the same sequence of operations, and libm and the device maths library do not
// round their transcendentals alike.  With this the RANSAC pose is the same bit pattern on both sides.
double cbrt_exact(double x) {
  if (x == 0 || x != x) return x;
  uint64_t bits = to_bits(x);
  const uint64_t sign = bits & 0x8000000000000000ULL;
  bits &= 0x7fffffffffffffffULL;
  int bexp = (int)(bits >> 52), adj = 0;
  if (bexp == 0x7ff) return x;
  if (bexp == 0) { bits = to_bits(from_bits(bits) * 18014398509481984.0 /* 2^54 */); bexp = (int)(bits >> 52); adj = -18; }
  const int e = bexp - 1023;
  const int k = (e >= 0 ? e : e - 2) / 3, r = e - 3 * k;
  const uint64_t frac = bits & 0x000fffffffffffffULL;
  const double f = from_bits(frac | (1023ULL << 52));
  const double m = from_bits(frac | ((uint64_t)(1023 + r) << 52));
  double y = (1.0 + (f - 1.0) * 0.26) * (r == 0 ? 1.0 : r == 1 ? 1.2599210498948732 : 1.5874010519681994);
  for (int it = 0; it < 6; ++it) y = y - (y * y * y - m) / (3.0 * (y * y));
  const double s = from_bits((uint64_t)(1023 + k + adj) << 52);
  return from_bits(to_bits(y * s) | sign);
}

double cubic3_largest_root(double P, double Q) {
  double t = 2 * std::sqrt(-P / 3);
  for (int it = 0; it < 64; ++it) {
    const double f = (t * t + P) * t + Q, fp = 3 * (t * t) + P;
    if (!(fp > 0)) break;
    const double tn = t - f / fp;
    if (!(tn < t)) break;
    t = tn;
  }
  return t;
}

// real roots of x^4 + b x^3 + c x^2 + d x + e (Ferrari via the resolvent cubic), polished by Newton
int solve_quartic_monic(double b, double c, double d, double e, double* roots) {
  const double p = c - 3 * b * b / 8, q = d - b * c / 2 + b * b * b / 8, r = e - b * d / 4 + b * b * c / 16 - 3 * b * b * b * b / 256;
  double y[4];
  int n = 0;
  const double scale = std::fabs(p) + std::fabs(r) + 1e-300;
  if (std::fabs(q) < 1e-14 * scale) {            // biquadratic
    const double disc = p * p - 4 * r;
    if (disc >= 0) {
      const double s = std::sqrt(disc);
      for (double w : {(-p + s) / 2, (-p - s) / 2})
        if (w >= 0) { y[n++] = std::sqrt(w); y[n++] = -std::sqrt(w); }
    }
  } else {
    // resolvent cubic z^3 + 2p z^2 + (p^2 - 4r) z - q^2 = 0 has a positive real root
    const double A = 2 * p, B = p * p - 4 * r, C = -q * q;
    const double P = B - A * A / 3, Q = 2 * A * A * A / 27 - A * B / 3 + C;
    const double disc = Q * Q / 4 + P * P * P / 27;
    double t;
    if (disc >= 0) {
      const double s = std::sqrt(disc);
      t = cbrt_exact(-Q / 2 + s) + cbrt_exact(-Q / 2 - s);
    } else {
      t = cubic3_largest_root(P, Q);                 // largest root
    }
    double z = t - A / 3;
    for (int it = 0; it < 3; ++it) {                  // Newton on the cubic
      const double f = ((z + A) * z + B) * z + C, fp = (3 * z + 2 * A) * z + B;
      if (fp != 0) z -= f / fp;
    }
    if (z > 0) {
      const double s = std::sqrt(z);
      const double h1 = (p + z - q / s) / 2, h2 = (p + z + q / s) / 2;
      const double d1 = s * s - 4 * h1, d2 = s * s - 4 * h2;
      if (d1 >= 0) { const double w = std::sqrt(d1); y[n++] = (-s + w) / 2; y[n++] = (-s - w) / 2; }
      if (d2 >= 0) { const double w = std::sqrt(d2); y[n++] = (s + w) / 2; y[n++] = (s - w) / 2; }
    }
  }
  for (int i = 0; i < n; ++i) {
    double x = y[i] - b / 4;
    for (int it = 0; it < 2; ++it) {
      const double f = (((x + b) * x + c) * x + d) * x + e, fp = ((4 * x + 3 * b) * x + 2 * c) * x + d;
      if (fp != 0) x -= f / fp;
    }
    roots[i] = x;
  }
  return n;
}

struct Pose { double R[9], t[3]; };

inline void cross3(const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline bool normalize3(double* a) { const double n = std::sqrt(dot3(a, a)); if (!(n > 1e-300)) return false; a[0] /= n; a[1] /= n; a[2] /= n; return true; }

// orthonormal frame of three points: e1 along P2-P1, e3 normal of the triangle
bool frame3(const double* P1, const double* P2, const double* P3, double* B /*columns e1 e2 e3, row-major 3x3*/) {
  double e1[3] = {P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2]}, w[3] = {P3[0] - P1[0], P3[1] - P1[1], P3[2] - P1[2]}, e3[3], e2[3];
  if (!normalize3(e1)) return false;
  cross3(e1, w, e3);
  if (!normalize3(e3)) return false;
  cross3(e3, e1, e2);
  for (int i = 0; i < 3; ++i) { B[3 * i] = e1[i]; B[3 * i + 1] = e2[i]; B[3 * i + 2] = e3[i]; }
  return true;
}

// up to 4 poses (camera from world) from 3 bearings f (unit) and 3 world points
int p3p(const double* f1, const double* f2, const double* f3, const double* P1, const double* P2, const double* P3, Pose* out) {
  double d23[3] = {P2[0] - P3[0], P2[1] - P3[1], P2[2] - P3[2]}, d13[3] = {P1[0] - P3[0], P1[1] - P3[1], P1[2] - P3[2]}, d12[3] = {P1[0] - P2[0], P1[1] - P2[1], P1[2] - P2[2]};
  const double a2 = dot3(d23, d23), b2 = dot3(d13, d13), c2 = dot3(d12, d12);
  const double ca = dot3(f2, f3), cb = dot3(f1, f3), cg = dot3(f1, f2);
  if (!(a2 > 0 && b2 > 0 && c2 > 0)) return 0;
  const double A4 = a2 * a2 - 2 * a2 * b2 - 2 * a2 * c2 + b2 * b2 - 4 * b2 * c2 * ca * ca + 2 * b2 * c2 + c2 * c2;
  const double A3 = -4 * (a2 * a2 * cb - a2 * b2 * ca * cg - a2 * b2 * cb - 2 * a2 * c2 * cb + b2 * b2 * ca * cg - 2 * b2 * c2 * ca * ca * cb - b2 * c2 * ca * cg + b2 * c2 * cb + c2 * c2 * cb);
  const double A2 = 2 * (2 * a2 * a2 * cb * cb + a2 * a2 - 4 * a2 * b2 * ca * cb * cg - 2 * a2 * b2 * cg * cg - 4 * a2 * c2 * cb * cb - 2 * a2 * c2 + 2 * b2 * b2 * ca * ca +
                         2 * b2 * b2 * cg * cg - b2 * b2 - 2 * b2 * c2 * ca * ca - 4 * b2 * c2 * ca * cb * cg + 2 * c2 * c2 * cb * cb + c2 * c2);
  const double A1 = -4 * (a2 * a2 * cb - a2 * b2 * ca * cg - 2 * a2 * b2 * cb * cg * cg + a2 * b2 * cb - 2 * a2 * c2 * cb + b2 * b2 * ca * cg - b2 * c2 * ca * cg - b2 * c2 * cb + c2 * c2 * cb);
  const double A0 = a2 * a2 - 4 * a2 * b2 * cg * cg + 2 * a2 * b2 - 2 * a2 * c2 + b2 * b2 - 2 * b2 * c2 + c2 * c2;
  if (!(std::fabs(A4) > 1e-300)) return 0;
  double vr[4];
  const int nr = solve_quartic_monic(A3 / A4, A2 / A4, A1 / A4, A0 / A4, vr);
  double Bw[9];
  if (!frame3(P1, P2, P3, Bw)) return 0;
  int n = 0;
  for (int k = 0; k < nr; ++k) {
    const double v = vr[k];
    if (!(v > 0)) continue;
    const double den = 2 * b2 * (ca * v - cg);
    if (!(std::fabs(den) > 1e-300)) continue;
    const double u = -(-2 * a2 * cb * v + a2 * v * v + a2 - b2 * v * v + b2 + 2 * c2 * cb * v - c2 * v * v - c2) / den;
    if (!(u > 0)) continue;
    const double w = 1 + v * v - 2 * v * cb;
    if (!(w > 0)) continue;
    const double d1 = std::sqrt(b2 / w), d2 = u * d1, d3 = v * d1;
    const double X1[3] = {d1 * f1[0], d1 * f1[1], d1 * f1[2]}, X2[3] = {d2 * f2[0], d2 * f2[1], d2 * f2[2]}, X3[3] = {d3 * f3[0], d3 * f3[1], d3 * f3[2]};
    double Bc[9];
    if (!frame3(X1, X2, X3, Bc)) continue;
    Pose& o = out[n];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) o.R[3 * i + j] = Bc[3 * i] * Bw[3 * j] + Bc[3 * i + 1] * Bw[3 * j + 1] + Bc[3 * i + 2] * Bw[3 * j + 2];
    for (int i = 0; i < 3; ++i) o.t[i] = X1[i] - (o.R[3 * i] * P1[0] + o.R[3 * i + 1] * P1[1] + o.R[3 * i + 2] * P1[2]);
    ++n;
  }
  return n;
}

inline double reproj2(const Pose& T, const double* K4, const double* X, const double* uv) {
  const double x = T.R[0] * X[0] + T.R[1] * X[1] + T.R[2] * X[2] + T.t[0], y = T.R[3] * X[0] + T.R[4] * X[1] + T.R[5] * X[2] + T.t[1],
               z = T.R[6] * X[0] + T.R[7] * X[1] + T.R[8] * X[2] + T.t[2];
  const double du = K4[0] * x / z + K4[2] - uv[0], dv = K4[1] * y / z + K4[3] - uv[1];
  return du * du + dv * dv;
}

// one hypothesis: pose from points idx[0..2], disambiguated by idx[3]; false when the minimal solver has no solution
bool hypothesis(const double* X, const double* uv, const double* K4, const int* idx, Pose* out) {
  double f[3][3];
  for (int k = 0; k < 3; ++k) {
    f[k][0] = (uv[2 * idx[k]] - K4[2]) / K4[0]; f[k][1] = (uv[2 * idx[k] + 1] - K4[3]) / K4[1]; f[k][2] = 1.0;
    normalize3(f[k]);
  }
  Pose sol[4];
  const int ns = p3p(f[0], f[1], f[2], X + 3 * idx[0], X + 3 * idx[1], X + 3 * idx[2], sol);
  if (ns == 0) return false;
  int best = 0;
  double be = DBL_MAX;
  for (int s = 0; s < ns; ++s) {
    const double e = reproj2(sol[s], K4, X + 3 * idx[3], uv + 2 * idx[3]);
    if (e < be) { be = e; best = s; }
  }
  *out = sol[best];
  return true;
}

int update_num_iters(double p, double ep, int model_points, int max_iters) {
  p = std::min(1.0, std::max(0.0, p)); ep = std::min(1.0, std::max(0.0, ep));
  double num = std::max(1.0 - p, DBL_MIN);
  double denom = 1.0 - std::pow(1.0 - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = std::log(num); denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

}  // namespace

// KAT helpers
extern "C" double vdo_oracle_cbrt_exact(double x) { return cbrt_exact(x); }
extern "C" double vdo_oracle_cubic3_largest_root(double P, double Q) { return cubic3_largest_root(P, Q); }
extern "C" int vdo_oracle_quartic(double a4, double a3, double a2, double a1, double a0, double* roots) {
  return solve_quartic_monic(a3 / a4, a2 / a4, a1 / a4, a0 / a4, roots);
}
extern "C" int vdo_oracle_p3p(const double* f9, const double* P9, double* R_out /*[4][9]*/, double* t_out /*[4][3]*/) {
  Pose sol[4];
  const int n = p3p(f9, f9 + 3, f9 + 6, P9, P9 + 3, P9 + 6, sol);
  for (int s = 0; s < n; ++s) { std::memcpy(R_out + 9 * s, sol[s].R, 72); std::memcpy(t_out + 3 * s, sol[s].t, 24); }
  return n;
}
// the index subsets the sequential RANSAC would draw: [max_iters][4]
extern "C" void vdo_oracle_ransac_subsets(int n, int max_iters, int32_t* idx) {
  RNG rng((uint64_t)-1);
  for (int it = 0; it < max_iters; ++it)
    for (int i = 0; i < 4; ++i)
      for (;;) {
        const int c = rng.uniform(0, n);
        bool dup = false;
        for (int j = 0; j < i; ++j) dup |= (idx[4 * it + j] == c);
        if (!dup) { idx[4 * it + i] = c; break; }
      }
}

// solvePnPRansac (AP3P flavour) up to the final refit.  X [n][3] world points, uv [n][2] pixels, K4.  Returns the number
// of inliers (0: no model); T_out 4x4 row-major camera-from-world; inlier_out [n]; stats: iterations actually run.
extern "C" int vdo_oracle_p3p_ransac(int n, const double* X, const double* uv, const double* K4, int max_iters, double thr, double confidence,
                                     double* T_out, uint8_t* inlier_out, int32_t* iters_run, int32_t* best_iter) {
  for (int i = 0; i < 16; ++i) T_out[i] = (i % 5 == 0) ? 1.0 : 0.0;
  if (inlier_out) std::memset(inlier_out, 0, (size_t)std::max(n, 0));
  if (iters_run) *iters_run = 0;
  if (best_iter) *best_iter = -1;
  if (n < 4) return 0;
  std::vector<int32_t> idx(4 * (size_t)max_iters);
  vdo_oracle_ransac_subsets(n, max_iters, idx.data());
  const double t2 = thr * thr;
  int niters = max_iters, max_good = 0, it = 0, bi = -1;
  Pose best{};
  for (; it < niters; ++it) {
    Pose h;
    if (!hypothesis(X, uv, K4, idx.data() + 4 * it, &h)) continue;
    int good = 0;
    for (int i = 0; i < n; ++i) good += reproj2(h, K4, X + 3 * i, uv + 2 * i) <= t2;
    if (good > std::max(max_good, 3)) {
      best = h; max_good = good; bi = it;
      niters = update_num_iters(confidence, (double)(n - good) / n, 4, niters);
    }
  }
  if (iters_run) *iters_run = it;
  if (best_iter) *best_iter = bi;
  if (max_good == 0) return 0;
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T_out[4 * i + j] = best.R[3 * i + j]; T_out[4 * i + 3] = best.t[i]; }
  if (inlier_out)
    for (int i = 0; i < n; ++i) inlier_out[i] = reproj2(best, K4, X + 3 * i, uv + 2 * i) <= t2;
  return max_good;
}

// KAT hook: EPnP on all given points
extern "C" double vdo_oracle_epnp(int n, const double* X, const double* uv, const double* K4, double* T_out) {
  const ref_epnp::Result r = ref_epnp::solve(n, X, uv, K4);
  for (int i = 0; i < 16; ++i) T_out[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T_out[4 * i + j] = r.R[3 * i + j]; T_out[4 * i + 3] = r.t[i]; }
  return r.err;
}

// solvePnPRansac as a whole: the RANSAC above, then - refit != 0 - the winning model re-estimated on its inliers by EPnP (the pose
// OpenCV 3.4 returns; the inlier set stays the RANSAC one)
extern "C" int vdo_oracle_ap3p_lf_ransac(int n, const double* X, const double* uv, const double* K4, int max_iters, double thr, double confidence,
                                         double* T_out, uint8_t* inlier_out, int32_t* iters_run, int32_t* best_iter);
// refit bit 0: OpenCV's final EPnP re-estimation; bit 1: Grunert's P3P as the minimal solver instead of AP3P (= vdo_pnp_problem.refit of the product).
// The default - AP3P, what the reference's calls name (src/Tracking.cc:1652-1657) - is the libm-free form of ap3p_oracle.cpp (round 5; rounds 2-4: Grunert).
extern "C" int vdo_oracle_pnp_ransac_refit(int n, const double* X, const double* uv, const double* K4, int max_iters, double thr, double confidence, int refit,
                                           double* T_out, uint8_t* inlier_out, int32_t* iters_run, int32_t* best_iter) {
  std::vector<uint8_t> inl((size_t)std::max(n, 1), 0);
  const int good = (refit & 2) ? vdo_oracle_p3p_ransac(n, X, uv, K4, max_iters, thr, confidence, T_out, inl.data(), iters_run, best_iter)
                               : vdo_oracle_ap3p_lf_ransac(n, X, uv, K4, max_iters, thr, confidence, T_out, inl.data(), iters_run, best_iter);
  if (inlier_out && n > 0) std::memcpy(inlier_out, inl.data(), (size_t)n);
  if (good >= 4 && (refit & 1)) {
    std::vector<double> Xi, ui;
    for (int i = 0; i < n; ++i) if (inl[i]) { Xi.insert(Xi.end(), X + 3 * i, X + 3 * i + 3); ui.insert(ui.end(), uv + 2 * i, uv + 2 * i + 2); }
    const ref_epnp::Result r = ref_epnp::solve((int)(ui.size() / 2), Xi.data(), ui.data(), K4);
    if (r.err >= 0.0)                                  // (always, since round 5: coplanar inliers go through EPnP like any others; only a non-finite result leaves the hypothesis)
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T_out[4 * i + j] = r.R[3 * i + j]; T_out[4 * i + 3] = r.t[i]; }
  }
  return good;
}
