// TEST INFRASTRUCTURE — CPU oracle (see vdo_oracle.h).  The minimal solver the reference NAMES in its calls:
// cv::solvePnPRansac(..., cv::SOLVEPNP_AP3P) (reference src/Tracking.cc:1652-1657, 1755-1760) = Ke & Roumeliotis, "An Efficient
// Algebraic Solution to the Perspective-Three-Point Problem" (CVPR 2017), as OpenCV 3.4's modules/calib3d/src/ap3p.cpp lays it out:
// the quartic in cos(theta1') with the coefficients g1..g7, its roots by Ferrari's formulas in complex arithmetic (solveQuartic),
// two Newton steps on every root (polishQuarticRoots), roots with |cos| > 1 dropped, solutions in root order, the fourth point of a
// RANSAC sample picks among them by reprojection error.  OpenCV is not in this image: this is a restatement from the paper and the
// published layout of that file - PARITY UNPINNED (order and rounding of the solutions cannot be checked against OpenCV here).
// What IS checked (tests/test_oracle_ap3p.py): every solution is a rotation that maps the three world points onto their bearings, the
// true pose is among them, and the solutions from real roots in front of the camera equal Grunert's (oracle/p3p_oracle.cpp).
// Two forms live here.  (1) the complex-arithmetic form above (std::complex, libm pow / cbrt / sqrt): vdo_oracle_ap3p_ransac - shares
// only the algebra with the product, the independent check.  (2) since round 5 the LIBM-FREE form (`lf`: Ferrari in real arithmetic,
// correctly rounded operations only, the same sequence as csrc/ransac.hip k_ap3p_hyp): vdo_oracle_ap3p_lf*, what
// vdo_oracle_pnp_ransac_refit (p3p_oracle.cpp) runs by default - the product's default minimal solver, bit for bit.  The two forms are
// compared in tests/test_oracle_ap3p.py::test_libm_free_form_equals_the_complex_form.  Hypotheses leave through a polar
// orthogonalisation, as cv::Rodrigues (matrix -> vector -> matrix, an SVD inside) does on the way out of solvePnPRansac.
#include <cfloat>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <vector>

#include "vdo_oracle.h"

namespace {

inline void cross(const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
inline double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double norm(const double* a) { return std::sqrt(dot(a, a)); }
inline void mat_mult(const double a[3][3], const double b[3][3], double o[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

// real parts of the four roots of a4 x^4 + a3 x^3 + a2 x^2 + a1 x + a0 (Ferrari through the resolvent cubic, complex square roots)
void solve_quartic(const double* f, double* roots) {
  const double a4 = f[0], a3 = f[1], a2 = f[2], a1 = f[3], a0 = f[4];
  const double a4_2 = a4 * a4, a3_2 = a3 * a3, a4_3 = a4_2 * a4, a2a4 = a2 * a4;
  const double p4 = (8 * a2a4 - 3 * a3_2) / (8 * a4_2);
  const double q4 = (a3_2 * a3 - 4 * a2a4 * a3 + 8 * a1 * a4_2) / (8 * a4_3);
  const double r4 = (256 * a0 * a4_3 - 3 * (a3_2 * a3_2) - 64 * a1 * a3 * a4_2 + 16 * a2a4 * a3_2) / (256 * (a4_3 * a4));
  const double p3 = ((p4 * p4) / 12 + r4) / 3;
  const double q3 = (72 * r4 * p4 - 2 * p4 * p4 * p4 - 27 * q4 * q4) / 432;
  double t;
  std::complex<double> w;
  if (q3 >= 0) w = -std::sqrt(std::complex<double>(q3 * q3 - p3 * p3 * p3)) - q3;
  else w = std::sqrt(std::complex<double>(q3 * q3 - p3 * p3 * p3)) - q3;
  if (w.imag() == 0.0) {
    w.real(std::cbrt(w.real()));
    t = 2.0 * (w.real() + p3 / w.real());
  } else {
    w = std::pow(w, 1.0 / 3);
    t = 4.0 * w.real();
  }
  const std::complex<double> sqrt_2m = std::sqrt(std::complex<double>(-2 * p4 / 3 + t));
  const double B_4A = -a3 / (4 * a4);
  const double complex1 = 4 * p4 / 3 + t;
  const std::complex<double> complex2 = 2 * q4 / sqrt_2m;
  const double sqrt_2m_rh = sqrt_2m.real() / 2;
  const double sqrt1 = std::sqrt(-(complex1 + complex2)).real() / 2;
  roots[0] = B_4A + sqrt_2m_rh + sqrt1;
  roots[1] = B_4A + sqrt_2m_rh - sqrt1;
  const double sqrt2 = std::sqrt(-(complex1 - complex2)).real() / 2;
  roots[2] = B_4A - sqrt_2m_rh + sqrt2;
  roots[3] = B_4A - sqrt_2m_rh - sqrt2;
}

// The same formulas in REAL arithmetic with +, -, *, / and sqrt only - what the product's k_ap3p_hyp runs (vdo_slam_amd/csrc/ransac.hip ap3p_quartic),
// operation for operation, so that both sides produce the same bits (libm's and the device library's pow / cbrt / complex sqrt do not round alike):
// the complex cube root enters only through 4 Re(w^(1/3)) = twice the largest root of x^3 - 3 p3 x + 2 q3 (|w|^2 = p3^3, Re w = -q3: three real roots,
// the monotone Newton iteration of p3p_oracle.cpp), complex square roots only through their real parts.
extern "C" double vdo_oracle_cbrt_exact(double x);
extern "C" double vdo_oracle_cubic3_largest_root(double P, double Q);
inline double re_csqrt(double a, double b) {      // real part of the principal square root of a + b i
  if (b == 0.0) return a >= 0.0 ? std::sqrt(a) : 0.0;
  const double m = std::sqrt(a * a + b * b);
  if (a > 0.0) return std::sqrt(2.0 * (m + a)) / 2.0;
  return std::fabs(b) / std::sqrt(2.0 * (m - a));
}
void solve_quartic_lf(const double* f, double* roots) {
  const double a4 = f[0], a3 = f[1], a2 = f[2], a1 = f[3], a0 = f[4];
  const double a4_2 = a4 * a4, a3_2 = a3 * a3, a4_3 = a4_2 * a4, a2a4 = a2 * a4;
  const double p4 = (8 * a2a4 - 3 * a3_2) / (8 * a4_2);
  const double q4 = (a3_2 * a3 - 4 * a2a4 * a3 + 8 * a1 * a4_2) / (8 * a4_3);
  const double r4 = (256 * a0 * a4_3 - 3 * (a3_2 * a3_2) - 64 * a1 * a3 * a4_2 + 16 * a2a4 * a3_2) / (256 * (a4_3 * a4));
  const double p3 = ((p4 * p4) / 12 + r4) / 3;
  const double q3 = (72 * r4 * p4 - 2 * p4 * p4 * p4 - 27 * q4 * q4) / 432;
  const double D = q3 * q3 - p3 * p3 * p3;
  double t;
  if (D >= 0) {
    const double sD = std::sqrt(D);
    double w = q3 >= 0 ? -sD - q3 : sD - q3;
    w = vdo_oracle_cbrt_exact(w);
    t = 2.0 * (w + p3 / w);
  } else {
    t = 2.0 * vdo_oracle_cubic3_largest_root(-3.0 * p3, 2.0 * q3);
  }
  const double m2 = -2 * p4 / 3 + t;
  const bool m2pos = m2 >= 0;
  const double s2m = std::sqrt(m2pos ? m2 : -m2);
  const double B_4A = -a3 / (4 * a4);
  const double complex1 = 4 * p4 / 3 + t;
  const double c2 = 2 * q4 / s2m;
  const double c2re = m2pos ? c2 : 0.0, c2im = m2pos ? 0.0 : -c2;
  const double sqrt_2m_rh = (m2pos ? s2m : 0.0) / 2;
  const double sqrt1 = re_csqrt(-(complex1 + c2re), -c2im) / 2;
  roots[0] = B_4A + sqrt_2m_rh + sqrt1;
  roots[1] = B_4A + sqrt_2m_rh - sqrt1;
  const double sqrt2 = re_csqrt(-(complex1 - c2re), c2im) / 2;
  roots[2] = B_4A - sqrt_2m_rh + sqrt2;
  roots[3] = B_4A - sqrt_2m_rh - sqrt2;
}

void polish_quartic_roots(const double* c, double* roots) {
  for (int it = 0; it < 2; ++it)
    for (int j = 0; j < 4; ++j) {
      const double x = roots[j];
      const double err = (((c[0] * x + c[1]) * x + c[2]) * x + c[3]) * x + c[4];
      const double der = ((4 * c[0] * x + 3 * c[1]) * x + 2 * c[2]) * x + c[3];
      roots[j] -= err / der;
    }
}

struct Pose { double R[9], t[3]; };

// up to 4 poses (camera from world: x_cam = R x_world + t) from 3 unit bearings b and 3 world points w
int ap3p(const double* b1, const double* b2, const double* b3, const double* w1, const double* w2, const double* w3, Pose* out, bool lf = false) {
  double u0[3] = {w1[0] - w2[0], w1[1] - w2[1], w1[2] - w2[2]};
  const double nu0 = norm(u0);
  if (!(nu0 > 0)) return 0;
  const double k1[3] = {u0[0] / nu0, u0[1] / nu0, u0[2] / nu0};
  double k3[3];
  cross(b1, b2, k3);
  const double nk3 = norm(k3);
  if (!(nk3 > 0)) return 0;
  for (int i = 0; i < 3; ++i) k3[i] /= nk3;
  double tz[3], v1[3], v2[3];
  cross(b1, k3, tz);
  cross(b1, b3, v1);
  cross(b2, b3, v2);
  const double u1[3] = {w1[0] - w3[0], w1[1] - w3[1], w1[2] - w3[2]};
  const double u1k1 = dot(u1, k1), k3b3 = dot(k3, b3);
  double f11 = k3b3, f13 = dot(k3, v1);
  const double f15 = -u1k1 * f11;
  double nl[3];
  cross(u1, k1, nl);
  const double delta = norm(nl);
  if (!(delta > 0) || k3b3 == 0.0) return 0;
  for (int i = 0; i < 3; ++i) nl[i] /= delta;
  f11 *= delta; f13 *= delta;
  const double u2k1 = u1k1 - nu0;
  double f21 = dot(tz, v2), f22 = nk3 * k3b3, f23 = dot(k3, v2);
  const double f24 = u2k1 * f22, f25 = -u2k1 * f21;
  f21 *= delta; f22 *= delta; f23 *= delta;
  const double g1 = f13 * f22, g2 = f13 * f25 - f15 * f23, g3 = f11 * f23 - f13 * f21, g4 = -f13 * f24, g5 = f11 * f22, g6 = f11 * f25 - f15 * f21, g7 = -f15 * f24;
  const double coeffs[5] = {g5 * g5 + g1 * g1 + g3 * g3, 2 * (g5 * g6 + g1 * g2 + g3 * g4), g6 * g6 + 2 * g5 * g7 + g2 * g2 + g4 * g4 - g1 * g1 - g3 * g3,
                            2 * (g6 * g7 - g1 * g2 - g3 * g4), g7 * g7 - g2 * g2 - g4 * g4};
  if (!(std::fabs(coeffs[0]) > 0)) return 0;
  double s[4];
  if (lf) solve_quartic_lf(coeffs, s); else solve_quartic(coeffs, s);
  polish_quartic_roots(coeffs, s);
  double temp[3];
  cross(k1, nl, temp);
  const double Ck1nl[3][3] = {{k1[0], nl[0], temp[0]}, {k1[1], nl[1], temp[1]}, {k1[2], nl[2], temp[2]}};
  const double Cb1k3tzT[3][3] = {{b1[0], b1[1], b1[2]}, {k3[0], k3[1], k3[2]}, {tz[0], tz[1], tz[2]}};
  const double b3p[3] = {(delta / k3b3) * b3[0], (delta / k3b3) * b3[1], (delta / k3b3) * b3[2]};
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    const double ctheta1p = s[i];
    if (!(std::fabs(ctheta1p) <= 1)) continue;
    double stheta1p = std::sqrt(1 - ctheta1p * ctheta1p);
    stheta1p = (k3b3 > 0) ? stheta1p : -stheta1p;
    double ctheta3 = g1 * ctheta1p + g2, stheta3 = g3 * ctheta1p + g4;
    const double ntheta3 = stheta1p / ((g5 * ctheta1p + g6) * ctheta1p + g7);
    ctheta3 *= ntheta3; stheta3 *= ntheta3;
    const double C13[3][3] = {{ctheta3, 0, -stheta3}, {stheta1p * stheta3, ctheta1p, stheta1p * ctheta3}, {ctheta1p * stheta3, -stheta1p, ctheta1p * ctheta3}};
    double tm[3][3], R[3][3];
    mat_mult(Ck1nl, C13, tm);
    mat_mult(tm, Cb1k3tzT, R);                  // world from camera
    const double rp3[3] = {w3[0] * R[0][0] + w3[1] * R[1][0] + w3[2] * R[2][0], w3[0] * R[0][1] + w3[1] * R[1][1] + w3[2] * R[2][1],
                           w3[0] * R[0][2] + w3[1] * R[1][2] + w3[2] * R[2][2]};      // R^T w3
    Pose& o = out[n];
    for (int a = 0; a < 3; ++a) {
      o.t[a] = stheta1p * b3p[a] - rp3[a];
      for (int b = 0; b < 3; ++b) o.R[3 * a + b] = R[b][a];
    }
    bool finite = true;
    for (int a = 0; a < 9; ++a) finite &= std::isfinite(o.R[a]);
    for (int a = 0; a < 3; ++a) finite &= std::isfinite(o.t[a]);
    if (finite) ++n;
  }
  return n;
}

inline double reproj2(const Pose& T, const double* K4, const double* X, const double* uv) {
  const double x = T.R[0] * X[0] + T.R[1] * X[1] + T.R[2] * X[2] + T.t[0], y = T.R[3] * X[0] + T.R[4] * X[1] + T.R[5] * X[2] + T.t[1],
               z = T.R[6] * X[0] + T.R[7] * X[1] + T.R[8] * X[2] + T.t[2];
  const double du = K4[0] * x / z + K4[2] - uv[0], dv = K4[1] * y / z + K4[3] - uv[1];
  return du * du + dv * dv;
}

// cv::Rodrigues on the way out of the RANSAC callback (matrix -> vector starts with R = U V^T of the SVD: the nearest orthogonal matrix) - see
// vdo_slam_amd/csrc/ransac.hip polar_orthogonalise: Higham's iteration, the same operations in the same order
void polar_orthogonalise(double* X) {
  for (int it = 0; it < 8; ++it) {
    const double c0 = X[4] * X[8] - X[5] * X[7], c1 = X[5] * X[6] - X[3] * X[8], c2 = X[3] * X[7] - X[4] * X[6];
    const double c3 = X[2] * X[7] - X[1] * X[8], c4 = X[0] * X[8] - X[2] * X[6], c5 = X[1] * X[6] - X[0] * X[7];
    const double c6 = X[1] * X[5] - X[2] * X[4], c7 = X[2] * X[3] - X[0] * X[5], c8 = X[0] * X[4] - X[1] * X[3];
    const double det = X[0] * c0 + X[1] * c1 + X[2] * c2;
    X[0] = 0.5 * (X[0] + c0 / det); X[1] = 0.5 * (X[1] + c1 / det); X[2] = 0.5 * (X[2] + c2 / det);
    X[3] = 0.5 * (X[3] + c3 / det); X[4] = 0.5 * (X[4] + c4 / det); X[5] = 0.5 * (X[5] + c5 / det);
    X[6] = 0.5 * (X[6] + c6 / det); X[7] = 0.5 * (X[7] + c7 / det); X[8] = 0.5 * (X[8] + c8 / det);
  }
}

// ap3p::solve on four points: pose from the first three, the fourth picks the solution (first one wins a tie)
bool hypothesis(const double* X, const double* uv, const double* K4, const int32_t* idx, Pose* out, bool lf = false) {
  double f[3][3];
  for (int k = 0; k < 3; ++k) {
    f[k][0] = (uv[2 * idx[k]] - K4[2]) / K4[0]; f[k][1] = (uv[2 * idx[k] + 1] - K4[3]) / K4[1]; f[k][2] = 1.0;
    const double nrm = norm(f[k]);
    for (int i = 0; i < 3; ++i) f[k][i] /= nrm;
  }
  Pose sol[4];
  const int ns = ap3p(f[0], f[1], f[2], X + 3 * idx[0], X + 3 * idx[1], X + 3 * idx[2], sol, lf);
  if (ns == 0) return false;
  int best = 0;
  double be = 0;
  for (int s = 0; s < ns; ++s) {
    const double e = reproj2(sol[s], K4, X + 3 * idx[3], uv + 2 * idx[3]);
    if (s == 0 || be > e) { be = e; best = s; }
  }
  *out = sol[best];
  polar_orthogonalise(out->R);
  for (int a = 0; a < 9; ++a) if (!std::isfinite(out->R[a])) return false;
  return true;
}

int update_num_iters(double p, double ep, int model_points, int max_iters) {
  p = std::fmin(1.0, std::fmax(0.0, p)); ep = std::fmin(1.0, std::fmax(0.0, ep));
  double num = std::fmax(1.0 - p, DBL_MIN);
  double denom = 1.0 - std::pow(1.0 - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = std::log(num); denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

}  // namespace

extern "C" int vdo_oracle_ap3p(const double* f9, const double* P9, double* R_out /*[4][9]*/, double* t_out /*[4][3]*/) {
  Pose sol[4];
  const int n = ap3p(f9, f9 + 3, f9 + 6, P9, P9 + 3, P9 + 6, sol);
  for (int s = 0; s < n; ++s) { std::memcpy(R_out + 9 * s, sol[s].R, 72); std::memcpy(t_out + 3 * s, sol[s].t, 24); }
  return n;
}

extern "C" int vdo_oracle_ap3p_quartic(const double* coeffs5, double* roots4) {
  solve_quartic(coeffs5, roots4);
  polish_quartic_roots(coeffs5, roots4);
  return 4;
}

// RANSACPointSetRegistrator::run with the AP3P callback (same subsets - vdo_oracle_ransac_subsets -, same vote and budget rule as
// vdo_oracle_p3p_ransac), up to the final refit
static int ap3p_ransac(bool lf, int n, const double* X, const double* uv, const double* K4, int max_iters, double thr, double confidence,
                       double* T_out, uint8_t* inlier_out, int32_t* iters_run, int32_t* best_iter) {
  for (int i = 0; i < 16; ++i) T_out[i] = (i % 5 == 0) ? 1.0 : 0.0;
  if (inlier_out) std::memset(inlier_out, 0, (size_t)(n > 0 ? n : 0));
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
    if (!hypothesis(X, uv, K4, idx.data() + 4 * it, &h, lf)) continue;
    int good = 0;
    for (int i = 0; i < n; ++i) good += reproj2(h, K4, X + 3 * i, uv + 2 * i) <= t2;
    if (good > (max_good > 3 ? max_good : 3)) {
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

extern "C" int vdo_oracle_ap3p_ransac(int n, const double* X, const double* uv, const double* K4, int max_iters, double thr, double confidence,
                                      double* T_out, uint8_t* inlier_out, int32_t* iters_run, int32_t* best_iter) {
  return ap3p_ransac(false, n, X, uv, K4, max_iters, thr, confidence, T_out, inlier_out, iters_run, best_iter);
}
// the same with Ferrari's formulas in real, libm-free arithmetic: the form the product runs and is compared with bit for bit
extern "C" int vdo_oracle_ap3p_lf_ransac(int n, const double* X, const double* uv, const double* K4, int max_iters, double thr, double confidence,
                                         double* T_out, uint8_t* inlier_out, int32_t* iters_run, int32_t* best_iter) {
  return ap3p_ransac(true, n, X, uv, K4, max_iters, thr, confidence, T_out, inlier_out, iters_run, best_iter);
}
extern "C" int vdo_oracle_ap3p_lf(const double* f9, const double* P9, double* R_out /*[4][9]*/, double* t_out /*[4][3]*/) {
  Pose sol[4];
  const int n = ap3p(f9, f9 + 3, f9 + 6, P9, P9 + 3, P9 + 6, sol, true);
  for (int s = 0; s < n; ++s) { std::memcpy(R_out + 9 * s, sol[s].R, 72); std::memcpy(t_out + 3 * s, sol[s].t, 24); }
  return n;
}
extern "C" int vdo_oracle_ap3p_quartic_lf(const double* coeffs5, double* roots4) {
  solve_quartic_lf(coeffs5, roots4);
  polish_quartic_roots(coeffs5, roots4);
  return 4;
}
