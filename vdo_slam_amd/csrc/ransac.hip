// RANSAC initialiser of the per-frame pose problems on gfx950: what
// Tracking::GetInitModelCam / GetInitModelObj obtain from cv::solvePnPRansac(pre_3d, cur_2d, K, 0, ..., 500 iterations,
// 0.4 px, confidence 0.98, SOLVEPNP_AP3P) (reference src/Tracking.cc:1652-1655, 1755-1758), up to OpenCV's final
// EPnP refit, which is host code (epnp_refit.hpp) applied to the winner's inliers when vdo_pnp_problem.refit is set.
//
// The sequential algorithm is kept — same subsets (cv::RNG stream, drawn on the host: 2 000 integers), same
// acceptance and budget rule — but its two data-parallel parts run at once for ALL hypotheses:
//   k_p3p_hyp      one thread per hypothesis: minimal solver on 3 points (quartic in d3/d1 + absolute orientation),
//                  4th point picks among the <= 4 solutions
//   k_ransac_vote  one workgroup per hypothesis: squared reprojection error of every correspondence, inlier count
//                  and inlier bit-mask
// and the host then replays the loop over the 500 counts (a better model shrinks the iteration budget), which
// yields exactly the model, inlier set and iteration count the sequential run would have produced.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "arena.hpp"
#include "ctx.hpp"
#include "epnp_refit.hpp"
#include "host_pool.hpp"

namespace vdo {

struct PnpDev {          // one problem inside the batch arrays
  int n, n_hyp, pt_off, hyp_off, mask_off, mask_words, pad0, pad1;
  double K[4];
  double thr2;
};

__device__ __forceinline__ double d3dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void d3cross(const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
__device__ __forceinline__ bool d3unit(double* a) { const double n = sqrt(d3dot(a, a)); if (!(n > 1e-300)) return false; a[0] /= n; a[1] /= n; a[2] /= n; return true; }

// Cube root and the largest root of a depressed cubic from +, -, *, / and sqrt only (all correctly rounded on gfx950 and on
// the host) plus exponent-field integer arithmetic: no libm/ocml transcendental, so that the CPU oracle
// (oracle/p3p_oracle.cpp, same sequence of operations) gets the SAME BITS.  cbrt: a = m * 8^k with m in [1, 8), linear
// seed, 6 Newton steps (the first brings the relative error below 1e-3, then it squares), exact scaling by 2^k.
__device__ __forceinline__ double cbrt_exact(double x) {
  if (x == 0 || x != x) return x;
  unsigned long long bits = (unsigned long long)__double_as_longlong(x);
  const unsigned long long sign = bits & 0x8000000000000000ULL;
  bits &= 0x7fffffffffffffffULL;
  int bexp = (int)(bits >> 52), adj = 0;
  if (bexp == 0x7ff) return x;
  if (bexp == 0) {                       // subnormal: scale by 2^54 (exact), take 2^-18 off the result
    bits = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)bits) * 18014398509481984.0);
    bexp = (int)(bits >> 52); adj = -18;
  }
  const int e = bexp - 1023;
  const int k = (e >= 0 ? e : e - 2) / 3, r = e - 3 * k;          // floor division: r in {0, 1, 2}
  const unsigned long long frac = bits & 0x000fffffffffffffULL;
  const double f = __longlong_as_double((long long)(frac | (1023ULL << 52)));            // [1, 2)
  const double m = __longlong_as_double((long long)(frac | ((unsigned long long)(1023 + r) << 52)));   // [1, 8)
  double y = (1.0 + (f - 1.0) * 0.26) * (r == 0 ? 1.0 : r == 1 ? 1.2599210498948732 : 1.5874010519681994);
#pragma unroll
  for (int it = 0; it < 6; ++it) y = y - (y * y * y - m) / (3.0 * (y * y));
  const double s = __longlong_as_double((long long)((unsigned long long)(1023 + k + adj) << 52));
  return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(y * s) | sign));
}

// largest root of t^3 + P t + Q when it has three real roots (P < 0, Q^2/4 + P^3/27 < 0): it lies in [m/2, m], m = 2 sqrt(-P/3);
// f is positive and convex on (root, m], so Newton from m decreases monotonically onto it; stop at the first step that does
// not decrease (rounding noise) - a deterministic rule, the same on both sides.
__device__ __forceinline__ double cubic3_largest_root(double P, double Q) {
  double t = 2 * sqrt(-P / 3);
  for (int it = 0; it < 64; ++it) {
    const double f = (t * t + P) * t + Q, fp = 3 * (t * t) + P;
    if (!(fp > 0)) break;
    const double tn = t - f / fp;
    if (!(tn < t)) break;
    t = tn;
  }
  return t;
}

// real roots of x^4 + b x^3 + c x^2 + d x + e: depressed quartic, positive root of the resolvent cubic, two quadratics; Newton polish
__device__ int quartic_real_roots(double b, double c, double d, double e, double* roots) {
  const double p = c - 3 * b * b / 8, q = d - b * c / 2 + b * b * b / 8, r = e - b * d / 4 + b * b * c / 16 - 3 * b * b * b * b / 256;
  double y[4];
  int n = 0;
  const double scale = fabs(p) + fabs(r) + 1e-300;
  if (fabs(q) < 1e-14 * scale) {
    const double disc = p * p - 4 * r;
    if (disc >= 0) {
      const double s = sqrt(disc);
      const double w0 = (-p + s) / 2, w1 = (-p - s) / 2;
      if (w0 >= 0) { y[n++] = sqrt(w0); y[n++] = -sqrt(w0); }
      if (w1 >= 0) { y[n++] = sqrt(w1); y[n++] = -sqrt(w1); }
    }
  } else {
    const double A = 2 * p, B = p * p - 4 * r, C = -q * q;
    const double P = B - A * A / 3, Q = 2 * A * A * A / 27 - A * B / 3 + C;
    const double disc = Q * Q / 4 + P * P * P / 27;
    double t;
    if (disc >= 0) {
      const double s = sqrt(disc);
      t = cbrt_exact(-Q / 2 + s) + cbrt_exact(-Q / 2 - s);
    } else {
      t = cubic3_largest_root(P, Q);
    }
    double z = t - A / 3;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const double f = ((z + A) * z + B) * z + C, fp = (3 * z + 2 * A) * z + B;
      if (fp != 0) z -= f / fp;
    }
    if (z > 0) {
      const double s = sqrt(z);
      const double h1 = (p + z - q / s) / 2, h2 = (p + z + q / s) / 2;
      const double d1 = s * s - 4 * h1, d2 = s * s - 4 * h2;
      if (d1 >= 0) { const double w = sqrt(d1); y[n++] = (-s + w) / 2; y[n++] = (-s - w) / 2; }
      if (d2 >= 0) { const double w = sqrt(d2); y[n++] = (s + w) / 2; y[n++] = (s - w) / 2; }
    }
  }
  for (int i = 0; i < n; ++i) {
    double x = y[i] - b / 4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const double f = (((x + b) * x + c) * x + d) * x + e, fp = ((4 * x + 3 * b) * x + 2 * c) * x + d;
      if (fp != 0) x -= f / fp;
    }
    roots[i] = x;
  }
  return n;
}

// orthonormal frame (columns e1 e2 e3) of a triangle: e1 along P2-P1, e3 its normal
__device__ bool tri_frame(const double* P1, const double* P2, const double* P3, double* B) {
  double e1[3] = {P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2]}, w[3] = {P3[0] - P1[0], P3[1] - P1[1], P3[2] - P1[2]}, e3[3], e2[3];
  if (!d3unit(e1)) return false;
  d3cross(e1, w, e3);
  if (!d3unit(e3)) return false;
  d3cross(e3, e1, e2);
#pragma unroll
  for (int i = 0; i < 3; ++i) { B[3 * i] = e1[i]; B[3 * i + 1] = e2[i]; B[3 * i + 2] = e3[i]; }
  return true;
}

__device__ __forceinline__ double reproj_err2(const double* R, const double* t, const double* K4, const double* X, double u, double v) {
  const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1],
               z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  const double du = K4[0] * x / z + K4[2] - u, dv = K4[1] * y / z + K4[3] - v;
  return du * du + dv * dv;
}

// hyp_pose [n_hyp_total][12] (R row-major | t), hyp_ok [n_hyp_total]
__global__ __launch_bounds__(64) void k_p3p_hyp(const PnpDev* __restrict__ probs, const double* __restrict__ X, const double* __restrict__ uv,
                                                const int32_t* __restrict__ subsets, double* __restrict__ hyp_pose, int32_t* __restrict__ hyp_ok) {
  const PnpDev P = probs[blockIdx.y];
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= P.n_hyp) return;
  const int32_t* idx = subsets + 4 * (size_t)(P.hyp_off + h);
  const double* Xp = X + 3 * (size_t)P.pt_off;
  const double* up = uv + 2 * (size_t)P.pt_off;
  double f[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f[k][0] = (up[2 * idx[k]] - P.K[2]) / P.K[0]; f[k][1] = (up[2 * idx[k] + 1] - P.K[3]) / P.K[1]; f[k][2] = 1.0;
    d3unit(f[k]);
  }
  const double *P1 = Xp + 3 * idx[0], *P2 = Xp + 3 * idx[1], *P3 = Xp + 3 * idx[2], *P4 = Xp + 3 * idx[3];
  const double u4 = up[2 * idx[3]], v4 = up[2 * idx[3] + 1];
  double* out = hyp_pose + 12 * (size_t)(P.hyp_off + h);
  int ok = 0;
  double d23[3] = {P2[0] - P3[0], P2[1] - P3[1], P2[2] - P3[2]}, d13[3] = {P1[0] - P3[0], P1[1] - P3[1], P1[2] - P3[2]}, d12[3] = {P1[0] - P2[0], P1[1] - P2[1], P1[2] - P2[2]};
  const double a2 = d3dot(d23, d23), b2 = d3dot(d13, d13), c2 = d3dot(d12, d12);
  const double ca = d3dot(f[1], f[2]), cb = d3dot(f[0], f[2]), cg = d3dot(f[0], f[1]);
  double Bw[9];
  if (a2 > 0 && b2 > 0 && c2 > 0 && tri_frame(P1, P2, P3, Bw)) {
    // Grunert's quartic in v = d3/d1 (coefficients: resultant of the three cosine-law equations, see tests/test_oracle_p3p.py)
    const double A4 = a2 * a2 - 2 * a2 * b2 - 2 * a2 * c2 + b2 * b2 - 4 * b2 * c2 * ca * ca + 2 * b2 * c2 + c2 * c2;
    const double A3 = -4 * (a2 * a2 * cb - a2 * b2 * ca * cg - a2 * b2 * cb - 2 * a2 * c2 * cb + b2 * b2 * ca * cg - 2 * b2 * c2 * ca * ca * cb - b2 * c2 * ca * cg + b2 * c2 * cb + c2 * c2 * cb);
    const double A2 = 2 * (2 * a2 * a2 * cb * cb + a2 * a2 - 4 * a2 * b2 * ca * cb * cg - 2 * a2 * b2 * cg * cg - 4 * a2 * c2 * cb * cb - 2 * a2 * c2 + 2 * b2 * b2 * ca * ca +
                           2 * b2 * b2 * cg * cg - b2 * b2 - 2 * b2 * c2 * ca * ca - 4 * b2 * c2 * ca * cb * cg + 2 * c2 * c2 * cb * cb + c2 * c2);
    const double A1 = -4 * (a2 * a2 * cb - a2 * b2 * ca * cg - 2 * a2 * b2 * cb * cg * cg + a2 * b2 * cb - 2 * a2 * c2 * cb + b2 * b2 * ca * cg - b2 * c2 * ca * cg - b2 * c2 * cb + c2 * c2 * cb);
    const double A0 = a2 * a2 - 4 * a2 * b2 * cg * cg + 2 * a2 * b2 - 2 * a2 * c2 + b2 * b2 - 2 * b2 * c2 + c2 * c2;
    if (fabs(A4) > 1e-300) {
      double vr[4];
      const int nr = quartic_real_roots(A3 / A4, A2 / A4, A1 / A4, A0 / A4, vr);
      double best = DBL_MAX;
      for (int k = 0; k < nr; ++k) {
        const double v = vr[k];
        if (!(v > 0)) continue;
        const double den = 2 * b2 * (ca * v - cg);
        if (!(fabs(den) > 1e-300)) continue;
        const double u = -(-2 * a2 * cb * v + a2 * v * v + a2 - b2 * v * v + b2 + 2 * c2 * cb * v - c2 * v * v - c2) / den;
        if (!(u > 0)) continue;
        const double w = 1 + v * v - 2 * v * cb;
        if (!(w > 0)) continue;
        const double d1 = sqrt(b2 / w), d2 = u * d1, d3 = v * d1;
        const double X1[3] = {d1 * f[0][0], d1 * f[0][1], d1 * f[0][2]}, X2[3] = {d2 * f[1][0], d2 * f[1][1], d2 * f[1][2]}, X3[3] = {d3 * f[2][0], d3 * f[2][1], d3 * f[2][2]};
        double Bc[9], R[9], t[3];
        if (!tri_frame(X1, X2, X3, Bc)) continue;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) R[3 * i + j] = Bc[3 * i] * Bw[3 * j] + Bc[3 * i + 1] * Bw[3 * j + 1] + Bc[3 * i + 2] * Bw[3 * j + 2];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = X1[i] - (R[3 * i] * P1[0] + R[3 * i + 1] * P1[1] + R[3 * i + 2] * P1[2]);
        const double e4 = reproj_err2(R, t, P.K, P4, u4, v4);
        if (e4 < best) {        // first smallest wins, like the sequential scan over the solutions
          best = e4; ok = 1;
#pragma unroll
          for (int i = 0; i < 9; ++i) out[i] = R[i];
          out[9] = t[0]; out[10] = t[1]; out[11] = t[2];
        }
      }
    }
  }
  hyp_ok[P.hyp_off + h] = ok;
}

// ---- AP3P: the minimal solver the reference's calls NAME - cv::solvePnPRansac(..., SOLVEPNP_AP3P), src/Tracking.cc:1652-1657, :1755-1760 -
// Ke & Roumeliotis, "An Efficient Algebraic Solution to the Perspective-Three-Point Problem" (CVPR 2017) in the layout of OpenCV 3.4's
// modules/calib3d/src/ap3p.cpp: the quartic in cos(theta1') with the coefficients g1 .. g7, its roots by Ferrari's formulas (the REAL PARTS of the four
// roots, as solveQuartic returns them), two Newton steps per root, roots with |cos| > 1 dropped, solutions in root order, the fourth point of the sample
// picks (first smallest reprojection error).  OpenCV evaluates Ferrari's formulas in std::complex with pow / sqrt of libm; here - and, operation for
// operation, in oracle/ap3p_oracle.cpp ap3p_lf - they are written out in real arithmetic with +, -, *, / and sqrt only (correctly rounded on both sides:
// same bits): the complex cube root appears only through 4 Re(w^(1/3)) = twice the largest root of x^3 - 3 p3 x + 2 q3 (three real roots: the monotone
// Newton iteration cubic3_largest_root), square roots of complex numbers only through their real parts.  The std::complex form stays in the oracle
// (vdo_oracle_ap3p) and the two are compared on the CPU (tests/test_oracle_ap3p.py).
__device__ __forceinline__ double re_csqrt(double a, double b) {      // real part of the principal square root of a + b i
  if (b == 0.0) return a >= 0.0 ? sqrt(a) : 0.0;
  const double m = sqrt(a * a + b * b);
  if (a > 0.0) return sqrt(2.0 * (m + a)) / 2.0;
  return fabs(b) / sqrt(2.0 * (m - a));
}
__device__ void ap3p_quartic(const double* f, double* roots) {
  const double a4 = f[0], a3 = f[1], a2 = f[2], a1 = f[3], a0 = f[4];
  const double a4_2 = a4 * a4, a3_2 = a3 * a3, a4_3 = a4_2 * a4, a2a4 = a2 * a4;
  const double p4 = (8 * a2a4 - 3 * a3_2) / (8 * a4_2);
  const double q4 = (a3_2 * a3 - 4 * a2a4 * a3 + 8 * a1 * a4_2) / (8 * a4_3);
  const double r4 = (256 * a0 * a4_3 - 3 * (a3_2 * a3_2) - 64 * a1 * a3 * a4_2 + 16 * a2a4 * a3_2) / (256 * (a4_3 * a4));
  const double p3 = ((p4 * p4) / 12 + r4) / 3;
  const double q3 = (72 * r4 * p4 - 2 * p4 * p4 * p4 - 27 * q4 * q4) / 432;
  const double D = q3 * q3 - p3 * p3 * p3;
  double t;
  if (D >= 0) {                                  // w real
    const double sD = sqrt(D);
    double w = q3 >= 0 ? -sD - q3 : sD - q3;
    w = cbrt_exact(w);
    t = 2.0 * (w + p3 / w);
  } else {                                       // w = -q3 -+ i sqrt(-D), |w|^2 = p3^3: 4 Re(w^(1/3)) = 2 x, x the largest root of x^3 - 3 p3 x + 2 q3
    t = 2.0 * cubic3_largest_root(-3.0 * p3, 2.0 * q3);
  }
  // sqrt_2m = sqrt(complex(-2 p4 / 3 + t)): real (s, 0) or imaginary (0, s)
  const double m2 = -2 * p4 / 3 + t;
  const bool m2pos = m2 >= 0;
  const double s2m = sqrt(m2pos ? m2 : -m2);
  const double B_4A = -a3 / (4 * a4);
  const double complex1 = 4 * p4 / 3 + t;
  // complex2 = 2 q4 / sqrt_2m: (2 q4 / s, 0) or (0, -2 q4 / s)
  const double c2 = 2 * q4 / s2m;
  const double c2re = m2pos ? c2 : 0.0, c2im = m2pos ? 0.0 : -c2;
  const double sqrt_2m_rh = (m2pos ? s2m : 0.0) / 2;
  const double sqrt1 = re_csqrt(-(complex1 + c2re), -c2im) / 2;
  roots[0] = B_4A + sqrt_2m_rh + sqrt1;
  roots[1] = B_4A + sqrt_2m_rh - sqrt1;
  const double sqrt2 = re_csqrt(-(complex1 - c2re), c2im) / 2;
  roots[2] = B_4A - sqrt_2m_rh + sqrt2;
  roots[3] = B_4A - sqrt_2m_rh - sqrt2;
#pragma unroll
  for (int it = 0; it < 2; ++it)                 // polishQuarticRoots
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double x = roots[j];
      const double err = (((f[0] * x + f[1]) * x + f[2]) * x + f[3]) * x + f[4];
      const double der = ((4 * f[0] * x + 3 * f[1]) * x + 2 * f[2]) * x + f[3];
      roots[j] -= err / der;
    }
}
// What cv::Rodrigues does to the solver's matrix on its way out of the RANSAC callback (PnPRansacCallback::runKernel: solvePnP -> Rodrigues(R, rvec); the vote
// then projects with Rodrigues(rvec)): matrix -> vector starts with R = U V^T of the SVD, the nearest orthogonal matrix.  For a proper solution that is the
// identity up to rounding; for the "solutions" AP3P builds from the real part of a COMPLEX pair of roots (3.4 keeps them: only |cos| <= 1 is tested) it is
// not - near a double root such a matrix is a few 1e-3 off a rotation and wins the vote.  The polar factor by Higham's iteration X <- (X + X^-T) / 2
// (cofactors and one division per entry: libm-free, the same bits in oracle/ap3p_oracle.cpp); the angle-axis round trip itself is not restated.
__device__ __forceinline__ void polar_orthogonalise(double* X) {
#pragma unroll 1
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
__device__ __forceinline__ bool d_finite(double x) { return fabs(x) <= 1.7976931348623157e308; }     // (false for NaN and infinities)

// hyp_pose [n_hyp_total][12] (R row-major | t), hyp_ok [n_hyp_total]
__global__ __launch_bounds__(64) void k_ap3p_hyp(const PnpDev* __restrict__ probs, const double* __restrict__ X, const double* __restrict__ uv,
                                                 const int32_t* __restrict__ subsets, double* __restrict__ hyp_pose, int32_t* __restrict__ hyp_ok) {
  const PnpDev P = probs[blockIdx.y];
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= P.n_hyp) return;
  const int32_t* idx = subsets + 4 * (size_t)(P.hyp_off + h);
  const double* Xp = X + 3 * (size_t)P.pt_off;
  const double* up = uv + 2 * (size_t)P.pt_off;
  double f[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f[k][0] = (up[2 * idx[k]] - P.K[2]) / P.K[0]; f[k][1] = (up[2 * idx[k] + 1] - P.K[3]) / P.K[1]; f[k][2] = 1.0;
    const double nrm = sqrt(d3dot(f[k], f[k]));
    f[k][0] /= nrm; f[k][1] /= nrm; f[k][2] /= nrm;
  }
  const double *b1 = f[0], *b2 = f[1], *b3 = f[2];
  const double *w1 = Xp + 3 * idx[0], *w2 = Xp + 3 * idx[1], *w3 = Xp + 3 * idx[2], *P4 = Xp + 3 * idx[3];
  const double u4 = up[2 * idx[3]], v4 = up[2 * idx[3] + 1];
  double* out = hyp_pose + 12 * (size_t)(P.hyp_off + h);
  int ok = 0;
  do {
    const double u0[3] = {w1[0] - w2[0], w1[1] - w2[1], w1[2] - w2[2]};
    const double nu0 = sqrt(d3dot(u0, u0));
    if (!(nu0 > 0)) break;
    const double k1[3] = {u0[0] / nu0, u0[1] / nu0, u0[2] / nu0};
    double k3[3];
    d3cross(b1, b2, k3);
    const double nk3 = sqrt(d3dot(k3, k3));
    if (!(nk3 > 0)) break;
    k3[0] /= nk3; k3[1] /= nk3; k3[2] /= nk3;
    double tz[3], v1[3], v2[3];
    d3cross(b1, k3, tz);
    d3cross(b1, b3, v1);
    d3cross(b2, b3, v2);
    const double u1[3] = {w1[0] - w3[0], w1[1] - w3[1], w1[2] - w3[2]};
    const double u1k1 = d3dot(u1, k1), k3b3 = d3dot(k3, b3);
    double f11 = k3b3, f13 = d3dot(k3, v1);
    const double f15 = -u1k1 * f11;
    double nl[3];
    d3cross(u1, k1, nl);
    const double delta = sqrt(d3dot(nl, nl));
    if (!(delta > 0) || k3b3 == 0.0) break;
    nl[0] /= delta; nl[1] /= delta; nl[2] /= delta;
    f11 *= delta; f13 *= delta;
    const double u2k1 = u1k1 - nu0;
    double f21 = d3dot(tz, v2), f22 = nk3 * k3b3, f23 = d3dot(k3, v2);
    const double f24 = u2k1 * f22, f25 = -u2k1 * f21;
    f21 *= delta; f22 *= delta; f23 *= delta;
    const double g1 = f13 * f22, g2 = f13 * f25 - f15 * f23, g3 = f11 * f23 - f13 * f21, g4 = -f13 * f24, g5 = f11 * f22, g6 = f11 * f25 - f15 * f21, g7 = -f15 * f24;
    const double coeffs[5] = {g5 * g5 + g1 * g1 + g3 * g3, 2 * (g5 * g6 + g1 * g2 + g3 * g4), g6 * g6 + 2 * g5 * g7 + g2 * g2 + g4 * g4 - g1 * g1 - g3 * g3,
                              2 * (g6 * g7 - g1 * g2 - g3 * g4), g7 * g7 - g2 * g2 - g4 * g4};
    if (!(fabs(coeffs[0]) > 0)) break;
    double sr[4];
    ap3p_quartic(coeffs, sr);
    double temp[3];
    d3cross(k1, nl, temp);
    const double Ck[9] = {k1[0], nl[0], temp[0], k1[1], nl[1], temp[1], k1[2], nl[2], temp[2]};          // Ck1nl, row-major
    const double Cb[9] = {b1[0], b1[1], b1[2], k3[0], k3[1], k3[2], tz[0], tz[1], tz[2]};                  // Cb1k3tzT
    const double sc = delta / k3b3;
    const double b3p[3] = {sc * b3[0], sc * b3[1], sc * b3[2]};
    double best = 0.0;
    for (int i = 0; i < 4; ++i) {
      const double ct1 = sr[i];
      if (!(fabs(ct1) <= 1)) continue;
      double st1 = sqrt(1 - ct1 * ct1);
      st1 = (k3b3 > 0) ? st1 : -st1;
      double ct3 = g1 * ct1 + g2, st3 = g3 * ct1 + g4;
      const double nt3 = st1 / ((g5 * ct1 + g6) * ct1 + g7);
      ct3 *= nt3; st3 *= nt3;
      const double C13[9] = {ct3, 0, -st3, st1 * st3, ct1, st1 * ct3, ct1 * st3, -st1, ct1 * ct3};
      double tm[9], Rw[9];                       // Rw: world from camera
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) tm[3 * a + b] = Ck[3 * a] * C13[b] + Ck[3 * a + 1] * C13[3 + b] + Ck[3 * a + 2] * C13[6 + b];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Rw[3 * a + b] = tm[3 * a] * Cb[b] + tm[3 * a + 1] * Cb[3 + b] + tm[3 * a + 2] * Cb[6 + b];
      const double rp3[3] = {w3[0] * Rw[0] + w3[1] * Rw[3] + w3[2] * Rw[6], w3[0] * Rw[1] + w3[1] * Rw[4] + w3[2] * Rw[7], w3[0] * Rw[2] + w3[1] * Rw[5] + w3[2] * Rw[8]};   // R^T w3
      double R[9], t[3];
      bool fin = true;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        t[a] = st1 * b3p[a] - rp3[a];
        fin = fin && d_finite(t[a]);
#pragma unroll
        for (int b = 0; b < 3; ++b) { R[3 * a + b] = Rw[3 * b + a]; fin = fin && d_finite(R[3 * a + b]); }
      }
      if (!fin) continue;
      const double e4 = reproj_err2(R, t, P.K, P4, u4, v4);
      if (!ok || best > e4) {                    // the first solution, then every strictly better one (ap3p::solve on four points)
        best = e4; ok = 1;
#pragma unroll
        for (int q = 0; q < 9; ++q) out[q] = R[q];
        out[9] = t[0]; out[10] = t[1]; out[11] = t[2];
      }
    }
  } while (false);
  if (ok) {
    double R[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = out[q];
    polar_orthogonalise(R);
#pragma unroll
    for (int q = 0; q < 9; ++q) { out[q] = R[q]; ok = ok && d_finite(R[q]); }
  }
  hyp_ok[P.hyp_off + h] = ok;
}

// one workgroup per (hypothesis, problem): inlier count + bit-mask of squared reprojection error <= thr^2
__global__ __launch_bounds__(256) void k_ransac_vote(const PnpDev* __restrict__ probs, const double* __restrict__ X, const double* __restrict__ uv,
                                                     const double* __restrict__ hyp_pose, const int32_t* __restrict__ hyp_ok,
                                                     int32_t* __restrict__ count, uint32_t* __restrict__ mask) {
  const PnpDev P = probs[blockIdx.y];
  const int h = blockIdx.x;
  if (h >= P.n_hyp) return;
  __shared__ int s_cnt[4];
  const int gh = P.hyp_off + h;
  uint32_t* mrow = mask + (size_t)P.mask_off + (size_t)h * P.mask_words;
  if (!hyp_ok[gh]) {
    for (int w = threadIdx.x; w < P.mask_words; w += 256) mrow[w] = 0;
    if (threadIdx.x == 0) count[gh] = 0;
    return;
  }
  double R[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = hyp_pose[12 * (size_t)gh + i];
  t[0] = hyp_pose[12 * (size_t)gh + 9]; t[1] = hyp_pose[12 * (size_t)gh + 10]; t[2] = hyp_pose[12 * (size_t)gh + 11];
  const double* Xp = X + 3 * (size_t)P.pt_off;
  const double* up = uv + 2 * (size_t)P.pt_off;
  int c = 0;
  const int n_pad = (P.n + 63) & ~63;
  for (int i = threadIdx.x; i < n_pad; i += 256) {
    bool in = false;
    if (i < P.n) in = reproj_err2(R, t, P.K, Xp + 3 * i, up[2 * i], up[2 * i + 1]) <= P.thr2;
    const unsigned long long b = __ballot(in);
    c += in;
    if ((threadIdx.x & 63) == 0) { mrow[i >> 5] = (uint32_t)b; if ((i >> 5) + 1 < P.mask_words) mrow[(i >> 5) + 1] = (uint32_t)(b >> 32); }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) count[gh] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// cv::RNG
struct CvRng {
  uint64_t state;
  explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
  unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

static int ransac_update_iters(double p, double ep, int model_points, int max_iters) {     // RANSACUpdateNumIters
  p = std::min(1.0, std::max(0.0, p)); ep = std::min(1.0, std::max(0.0, ep));
  double num = std::max(1.0 - p, DBL_MIN);
  double denom = 1.0 - std::pow(1.0 - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = std::log(num); denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

}  // namespace vdo

using namespace vdo;

namespace {
struct PnpTrace {            // (VDO_PNP_TRACE=1, debug; the pipelines of several sequences call in concurrently: updated under a mutex)
  std::mutex mu;
  double t[5] = {0, 0, 0, 0, 0}; long n = 0; bool on = std::getenv("VDO_PNP_TRACE") != nullptr;
  ~PnpTrace() { if (on && n) std::fprintf(stderr, "[pnp trace] calls %ld: setup %.1f us, gpu %.1f us, replay %.1f us, refit %.1f us (per call)\n", n, t[0] / n, t[1] / n, t[2] / n, t[3] / n); }
} g_pnp_trace;
inline double pnp_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

extern "C" int vdo_pnp_ransac_batch_overlap(vdo_ctx* ctx, int n_problems, const vdo_pnp_problem* probs, vdo_pnp_result* results, uint8_t** inlier_out,
                                            void (*host_work)(void*), void* host_arg) {
  if (!ctx || !probs || !results || n_problems <= 0) return set_error(VDO_ERR_INVALID, "vdo_pnp_ransac_batch: bad argument");
  const double tr0 = g_pnp_trace.on ? pnp_now_us() : 0.0;
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  static thread_local std::vector<PnpDev> hp;          // (per-thread scratch: no allocation in steady state)
  static thread_local std::vector<int32_t> subsets;
  static thread_local std::vector<double> X, uv;
  hp.assign(n_problems, PnpDev{});
  size_t tot_pts = 0, tot_hyp = 0, tot_words = 0;
  int max_hyp = 0;
  for (int k = 0; k < n_problems; ++k) {
    const vdo_pnp_problem& p = probs[k];
    if (p.n < 0 || p.max_iterations < 0 || (p.n > 0 && (!p.X || !p.uv))) return set_error(VDO_ERR_INVALID, "pnp problem %d: bad fields", k);
    PnpDev& d = hp[k];
    d.n = p.n; d.n_hyp = p.n >= 4 ? p.max_iterations : 0;
    d.pt_off = (int)tot_pts; d.hyp_off = (int)tot_hyp; d.mask_off = (int)tot_words; d.mask_words = (p.n + 63) / 64 * 2;
    std::memcpy(d.K, p.K, sizeof d.K);
    d.thr2 = p.reproj_threshold * p.reproj_threshold;
    tot_pts += p.n; tot_hyp += d.n_hyp; tot_words += (size_t)d.n_hyp * d.mask_words;
    max_hyp = std::max(max_hyp, d.n_hyp);
  }
  for (int k = 0; k < n_problems; ++k) {
    vdo_pnp_result& r = results[k];
    for (int i = 0; i < 16; ++i) r.T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    r.n_inliers = 0; r.iterations_run = 0; r.best_iteration = -1;
    if (inlier_out && inlier_out[k] && probs[k].n) std::memset(inlier_out[k], 0, (size_t)probs[k].n);
  }
  if (tot_hyp == 0) { if (host_work) host_work(host_arg); return VDO_OK; }
  // the subsets the sequential loop would draw (getSubset: 4 distinct indices by rejection, RNG seeded with (uint64)-1 per call):
  // a function of (point count, hypotheses) alone - the draws of a frame's problems cost ~30 us of the object chain, and the same
  // counts come back every few frames, so the tables are kept (8 KB per distinct count)
  subsets.resize(4 * tot_hyp);
  X.resize(3 * tot_pts); uv.resize(2 * tot_pts);
  {
    static std::mutex cache_mu;
    static std::unordered_map<uint64_t, std::vector<int32_t>> cache;
    for (int k = 0; k < n_problems; ++k) {
      const PnpDev& d = hp[k];
      if (d.n) { std::memcpy(X.data() + 3 * (size_t)d.pt_off, probs[k].X, sizeof(double) * 3 * d.n); std::memcpy(uv.data() + 2 * (size_t)d.pt_off, probs[k].uv, sizeof(double) * 2 * d.n); }
      if (!d.n_hyp) continue;
      int32_t* dst = subsets.data() + 4 * (size_t)d.hyp_off;
      const uint64_t key = ((uint64_t)(uint32_t)d.n << 32) | (uint32_t)d.n_hyp;
      {
        std::lock_guard<std::mutex> g(cache_mu);
        auto it = cache.find(key);
        if (it != cache.end()) { std::memcpy(dst, it->second.data(), sizeof(int32_t) * 4 * (size_t)d.n_hyp); continue; }
      }
      CvRng rng((uint64_t)-1);
      for (int it = 0; it < d.n_hyp; ++it) {
        int32_t* s = dst + 4 * (size_t)it;
        for (int i = 0; i < 4; ++i)
          for (;;) {
            const int c = rng.uniform(0, d.n);
            bool dup = false;
            for (int j = 0; j < i; ++j) dup |= (s[j] == c);
            if (!dup) { s[i] = c; break; }
          }
      }
      std::lock_guard<std::mutex> g(cache_mu);
      if (cache.size() < 4096) cache.emplace(key, std::vector<int32_t>(dst, dst + 4 * (size_t)d.n_hyp));
    }
  }
  const double tr1 = g_pnp_trace.on ? pnp_now_us() : 0.0;
  Arena S(ctx);
  if (!S.reserve(40 * tot_pts + 128 * tot_hyp + 4 * tot_words + sizeof(PnpDev) * (size_t)n_problems + 16 * 256 + 8192))
    return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  PnpDev* dprob = S.up(hp.data(), (size_t)n_problems);
  double *dX = S.up(X.data(), X.size()), *duv = S.up(uv.data(), uv.size());
  int32_t* dsub = S.up(subsets.data(), subsets.size());
  double* dpose = S.up<double>(nullptr, 12 * tot_hyp);
  int32_t *dok = S.up<int32_t>(nullptr, tot_hyp), *dcnt = S.up<int32_t>(nullptr, tot_hyp);
  uint32_t* dmask = S.up<uint32_t>(nullptr, tot_words);
  if (!dprob || !dX || !duv || !dsub || !dpose || !dok || !dcnt || !dmask) return set_error(VDO_ERR_OOM, "scratch arena exhausted");
  // minimal solver of the whole batch: AP3P (what the reference's calls name) unless every problem asks for Grunert's P3P (vdo_pnp_problem.refit bit 1)
  bool grunert = true;
  for (int k = 0; k < n_problems; ++k) grunert = grunert && (probs[k].refit & 2);
  for (int k = 0; k < n_problems; ++k) if (((probs[k].refit & 2) != 0) != grunert) return set_error(VDO_ERR_INVALID, "vdo_pnp_ransac_batch: the problems of a batch must name the same minimal solver");
  if (grunert) hipLaunchKernelGGL(k_p3p_hyp, dim3((max_hyp + 63) / 64, n_problems), dim3(64), 0, S.stream(), (const PnpDev*)dprob, (const double*)dX, (const double*)duv, (const int32_t*)dsub, dpose, dok);
  else hipLaunchKernelGGL(k_ap3p_hyp, dim3((max_hyp + 63) / 64, n_problems), dim3(64), 0, S.stream(), (const PnpDev*)dprob, (const double*)dX, (const double*)duv, (const int32_t*)dsub, dpose, dok);
  hipLaunchKernelGGL(k_ransac_vote, dim3(max_hyp, n_problems), dim3(256), 0, S.stream(), (const PnpDev*)dprob, (const double*)dX, (const double*)duv, (const double*)dpose, (const int32_t*)dok, dcnt, dmask);
  // (votes, poses and inlier masks of all hypotheses are read where they land in the pinned block: the replay touches a few rows)
  const int32_t *cnt = S.down_view(dcnt, tot_hyp), *okv = S.down_view(dok, tot_hyp);
  const double* pose = S.down_view(dpose, 12 * tot_hyp);
  const uint32_t* mask = S.down_view(dmask, tot_words);
  S.queue_downloads();
  if (host_work) host_work(host_arg);                      // (the caller's own host work, under the two kernels and the copy back)
  rc = S.finish("vdo_pnp_ransac_batch");
  if (rc != VDO_OK) return rc;
  if (!cnt || !okv || !pose || !mask) return set_error(VDO_ERR_OOM, "scratch arena exhausted");
  const double tr2 = g_pnp_trace.on ? pnp_now_us() : 0.0;
  // replay of RANSACPointSetRegistrator::run over the precomputed votes
  for (int k = 0; k < n_problems; ++k) {
    const PnpDev& d = hp[k];
    if (!d.n_hyp) continue;
    int niters = d.n_hyp, max_good = 0, it = 0, bi = -1;
    for (; it < niters; ++it) {
      const int g = d.hyp_off + it;
      if (!okv[g]) continue;
      if (cnt[g] > std::max(max_good, 3)) {
        max_good = cnt[g]; bi = it;
        niters = ransac_update_iters(probs[k].confidence, (double)(d.n - cnt[g]) / d.n, 4, niters);
      }
    }
    vdo_pnp_result& r = results[k];
    r.iterations_run = it; r.best_iteration = bi; r.n_inliers = max_good;
    if (bi < 0) continue;
    const double* Pz = pose + 12 * ((size_t)d.hyp_off + bi);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) r.T[4 * i + j] = Pz[3 * i + j]; r.T[4 * i + 3] = Pz[9 + i]; }
    if (inlier_out && inlier_out[k]) {
      const uint32_t* row = mask + (size_t)d.mask_off + (size_t)bi * d.mask_words;
      for (int i = 0; i < d.n; ++i) inlier_out[k][i] = (row[i >> 5] >> (i & 31)) & 1u;
    }
  }
  // OpenCV's last step: the winning model re-estimated on its inliers by EPnP (solvePnP(inliers, SOLVEPNP_EPNP)); problems are
  // independent: one per pool task (the objects of a frame in parallel)
  const double tr3 = g_pnp_trace.on ? pnp_now_us() : 0.0;
  std::vector<int> todo;
  for (int k = 0; k < n_problems; ++k) if ((probs[k].refit & 1) && results[k].n_inliers >= 4 && results[k].best_iteration >= 0) todo.push_back(k);
  if (!todo.empty()) {
    const PnpDev* hp_main = hp.data();               // (hp is thread_local: a pool thread naming it would see ITS OWN, empty, vector)
    auto refit_one = [&, hp_main](int q) {
      const int k = todo[q];
      const PnpDev& d = hp_main[k];
      const uint32_t* row = mask + (size_t)d.mask_off + (size_t)results[k].best_iteration * d.mask_words;
      thread_local std::vector<double> Xi, ui;
      thread_local epnp::Scratch scr;
      Xi.clear(); ui.clear();
      for (int i = 0; i < d.n; ++i)
        if ((row[i >> 5] >> (i & 31)) & 1u) {
          Xi.insert(Xi.end(), probs[k].X + 3 * (size_t)i, probs[k].X + 3 * (size_t)i + 3);
          ui.insert(ui.end(), probs[k].uv + 2 * (size_t)i, probs[k].uv + 2 * (size_t)i + 2);
        }
      const epnp::Result r = epnp::solve((int)(ui.size() / 2), Xi.data(), ui.data(), probs[k].K, scr);
      if (r.err >= 0.0)                              // (always, since round 5: coplanar inliers go through EPnP like any others; only a non-finite result leaves the hypothesis)
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) results[k].T[4 * i + j] = r.R[3 * i + j]; results[k].T[4 * i + 3] = r.t[i]; }
    };
    static LevelPool* pool = [] {
      const char* e = std::getenv("VDO_PNP_THREADS");
      const int nw = e ? std::atoi(e) : 3;
      return nw > 0 ? new LevelPool(std::min(nw, 15)) : nullptr;
    }();
    static std::mutex pool_mu;                       // (one batch at a time through the shared helpers; a second caller refits inline)
    if (todo.size() > 1 && pool && pool_mu.try_lock()) {
      pool->run((int)todo.size(), refit_one);
      pool_mu.unlock();
    } else {
      for (int q = 0; q < (int)todo.size(); ++q) refit_one(q);
    }
  }
  if (g_pnp_trace.on && n_problems > 1) {
    const double tr4 = pnp_now_us();
    std::lock_guard<std::mutex> lock(g_pnp_trace.mu);
    g_pnp_trace.t[0] += tr1 - tr0; g_pnp_trace.t[1] += tr2 - tr1; g_pnp_trace.t[2] += tr3 - tr2; g_pnp_trace.t[3] += tr4 - tr3; ++g_pnp_trace.n;
  }
  return VDO_OK;
}

extern "C" int vdo_pnp_ransac_batch(vdo_ctx* ctx, int n_problems, const vdo_pnp_problem* probs, vdo_pnp_result* results, uint8_t** inlier_out) {
  return vdo_pnp_ransac_batch_overlap(ctx, n_problems, probs, results, inlier_out, nullptr, nullptr);
}
extern "C" int vdo_pnp_ransac(vdo_ctx* ctx, const vdo_pnp_problem* p, vdo_pnp_result* result, uint8_t* inlier_out) {
  return vdo_pnp_ransac_batch(ctx, 1, p, result, &inlier_out);
}
