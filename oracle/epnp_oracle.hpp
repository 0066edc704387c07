// TEST INFRASTRUCTURE - CPU oracle (see vdo_oracle.h).  EPnP: the re-estimation cv::solvePnPRansac(..., SOLVEPNP_AP3P) performs
// on the inliers of the winning model since OpenCV 3.3 (solvePnP(inliers, SOLVEPNP_EPNP)); the reference pins OpenCV 3.4.0
// (/root/reference/Dockerfile:40-63) and receives that pose in Tracking::GetInitModelCam / GetInitModelObj
// (/root/reference/src/Tracking.cc:1652-1660, 1755-1763).  OpenCV is not vendored under /root/reference, so this restates the
// published algorithm (Lepetit, Moreno-Noguer, Fua: "EPnP: an accurate O(n) solution to the PnP problem", IJCV 2009) in the
// structure and WITH THE NUMERICAL TOOLS OF OpenCV's calib3d/epnp.cpp - parity unpinned until run against OpenCV 3.4.0
// (tools/pin_reference/):
//   choose_control_points            centroid + principal directions from the SVD of PW0^T PW0, scaled by sqrt(sigma / n)
//   compute_barycentric_coordinates  inverse of the 3x3 control-point matrix THROUGH ITS SVD (cvInvert(CV_SVD))
//   fill_M / compute_pose            the dense 2n x 12 matrix M, M^T M by full products, SVD of M^T M (svd_opencv below), the four vectors of the smallest singular values
//   find_betas_approx_1/2/3          cvSolve(..., CV_SVD): x = V diag(1/w) U^T b, singular values under 2 eps sum(w) dropped
//   gauss_newton                     5 iterations on the 6x4 system, solved by an orthogonal (Givens) triangularisation
//   estimate_R_and_t                 SVD of sum (pc - pc0)(pw - pw0)^T, R = U V^T, third ROW of R negated when det R < 0
// This file is deliberately NOT the product's routine (vdo_slam_amd/csrc/epnp_refit.hpp: running sums for M^T M, tridiagonal QL for its
// eigenvectors, sums shared by the three candidates, Householder QR, fixed-size stack arrays): written separately, compared to 1e-9
// (1e-6 on near-planar sets; tests/test_epnp_independent.py, tests/test_ransac_gpu.py).  Since round 5 BOTH follow OpenCV's one-sided
// Jacobi SVD wherever its conventions decide the result (signs of the principal directions, pseudo-inverse thresholds, R = U V^T with the
// third-row flip); (near-)coplanar point sets go through like any other (rounds 2-4 returned err < 0 for them on both sides - a liberty).
#pragma once
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

namespace ref_epnp {

struct Result { double R[9], t[3], err; };

// cv::SVD as OpenCV 3.4.0 computes it when built without LAPACK (the reference's Dockerfile installs none before building OpenCV):
// one-sided Jacobi (Hestenes) on the ROWS of A^T (modules/core/src/lapack.cpp, JacobiSVDImpl_ - restated from the published
// source, not vendored here).  Everything that fixes the SIGN and ORDER of the singular vectors is kept, because EPnP's control
// points - and with noisy data its result - depend on the signs of the principal directions: V starts as the identity; pairs
// (i, j), i < j, in row-major order; a pair is rotated unless |<a_i, a_j>| <= 10 eps sqrt(|a_i|^2 |a_j|^2); the rotation (c, s)
// is the one below (c >= 0 when |a_i| >= |a_j|, else s >= 0); at most max(m, 30) sweeps; singular values = row norms, sorted
// descending by selection with swaps; left vectors = rows / singular value.  (OpenCV completes the left vectors of zero singular
// values with pseudo-random vectors; here such a row stays zero - no caller reads one.)
// A is m x n row-major.  Outputs: w [n]; Ut n x m (ROW j = j-th left singular vector); Vt n x n (ROW j = j-th right singular vector).
inline void svd_opencv(int m, int n, const double* A, double* w, double* Ut, double* Vt) {
  std::vector<double> At((size_t)n * m), W(n);
  for (int i = 0; i < n; ++i) for (int k = 0; k < m; ++k) At[(size_t)i * m + k] = A[(size_t)k * n + i];
  const double eps = 10 * 2.220446049250313e-16;
  for (int i = 0; i < n; ++i) {
    double sd = 0; for (int k = 0; k < m; ++k) sd += At[(size_t)i * m + k] * At[(size_t)i * m + k];
    W[i] = sd;
    for (int k = 0; k < n; ++k) Vt[i * n + k] = i == k ? 1.0 : 0.0;
  }
  const int max_iter = m > 30 ? m : 30;
  for (int iter = 0; iter < max_iter; ++iter) {
    bool changed = false;
    for (int i = 0; i < n - 1; ++i)
      for (int j = i + 1; j < n; ++j) {
        double* Ai = &At[(size_t)i * m]; double* Aj = &At[(size_t)j * m];
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; ++k) p += Ai[k] * Aj[k];
        if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = std::hypot(p, beta);
        double c, s;
        if (beta < 0) { const double delta = (gamma - beta) * 0.5; s = std::sqrt(delta / gamma); c = p / (gamma * s * 2); }
        else { c = std::sqrt((gamma + beta) / (gamma * 2)); s = p / (gamma * c * 2); }
        a = b = 0;
        for (int k = 0; k < m; ++k) { const double t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k]; Ai[k] = t0; Aj[k] = t1; a += t0 * t0; b += t1 * t1; }
        W[i] = a; W[j] = b;
        changed = true;
        double* Vi = Vt + i * n; double* Vj = Vt + j * n;
        for (int k = 0; k < n; ++k) { const double t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k]; Vi[k] = t0; Vj[k] = t1; }
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; ++i) { double sd = 0; for (int k = 0; k < m; ++k) sd += At[(size_t)i * m + k] * At[(size_t)i * m + k]; W[i] = std::sqrt(sd); }
  for (int i = 0; i < n - 1; ++i) {
    int j = i;
    for (int k = i + 1; k < n; ++k) if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      for (int k = 0; k < m; ++k) std::swap(At[(size_t)i * m + k], At[(size_t)j * m + k]);
      for (int k = 0; k < n; ++k) std::swap(Vt[i * n + k], Vt[j * n + k]);
    }
  }
  for (int i = 0; i < n; ++i) {
    w[i] = W[i];
    const double sc = W[i] > 2.2250738585072014e-308 ? 1 / W[i] : 0.0;
    for (int k = 0; k < m; ++k) Ut[(size_t)i * m + k] = At[(size_t)i * m + k] * sc;
  }
}

// cvSolve(A, b, x, CV_SVD): minimum-norm least squares through the SVD; singular values not above 2 eps sum(w) do not contribute
inline void solve_svd(int m, int n, const double* A, const double* b, double* x) {
  std::vector<double> w(n), Ut((size_t)m * n), Vt((size_t)n * n);
  svd_opencv(m, n, A, w.data(), Ut.data(), Vt.data());
  double thr = 0; for (int j = 0; j < n; ++j) thr += w[j];
  thr *= 2.0 * 2.220446049250313e-16;
  for (int i = 0; i < n; ++i) x[i] = 0;
  for (int j = 0; j < n; ++j) {
    if (!(w[j] > thr)) continue;
    double ub = 0; for (int r = 0; r < m; ++r) ub += Ut[(size_t)j * m + r] * b[r];
    ub /= w[j];
    for (int i = 0; i < n; ++i) x[i] += Vt[j * n + i] * ub;
  }
}

// least squares of the 6 x 4 Gauss-Newton system: A is brought to triangular form by Givens rotations (applied to b as well),
// then back substitution; false when a diagonal entry of the triangle vanishes
inline bool solve_givens_6x4(double A[24], double b[6], double x[4]) {
  for (int c = 0; c < 4; ++c)
    for (int r = 5; r > c; --r) {
      const double lo = A[4 * r + c];
      if (lo == 0) continue;
      const double hi = A[4 * (r - 1) + c], h = std::hypot(hi, lo), cs = hi / h, sn = lo / h;
      for (int k = c; k < 4; ++k) { const double u = A[4 * (r - 1) + k], v = A[4 * r + k]; A[4 * (r - 1) + k] = cs * u + sn * v; A[4 * r + k] = cs * v - sn * u; }
      const double u = b[r - 1], v = b[r]; b[r - 1] = cs * u + sn * v; b[r] = cs * v - sn * u;
    }
  for (int c = 3; c >= 0; --c) {
    if (!(std::fabs(A[4 * c + c]) > 0)) return false;
    double s = b[c]; for (int k = c + 1; k < 4; ++k) s -= A[4 * c + k] * x[k];
    x[c] = s / A[4 * c + c];
  }
  return true;
}

// X [n][3] world points, uv [n][2] pixels, K4 = fx, fy, cx, cy; n >= 4
inline Result solve(int n, const double* X, const double* uv, const double* K4) {
  const double fu = K4[0], fv = K4[1], uc = K4[2], vc = K4[3];
  Result none{}; none.err = -1.0;
  // ---- choose_control_points
  double cw[4][3] = {};
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) cw[0][k] += X[3 * i + k];
  for (int k = 0; k < 3; ++k) cw[0][k] /= n;
  std::vector<double> PW0(3 * (size_t)n);
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) PW0[3 * (size_t)i + k] = X[3 * i + k] - cw[0][k];
  double PtP[9];
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double s = 0; for (int i = 0; i < n; ++i) s += PW0[3 * (size_t)i + a] * PW0[3 * (size_t)i + b]; PtP[3 * a + b] = s; }
  double dc[3], UCt[9], VCt[9];
  svd_opencv(3, 3, PtP, dc, UCt, VCt);
  for (int i = 1; i < 4; ++i) { const double k = std::sqrt(dc[i - 1] / n); for (int j = 0; j < 3; ++j) cw[i][j] = cw[0][j] + k * UCt[3 * (i - 1) + j]; }
  // ---- compute_barycentric_coordinates: CC^-1 = V diag(1/w) U^T
  double CC[9], wi[3], Ui[9], Vi[9], CCi[9];
  for (int r = 0; r < 3; ++r) for (int j = 1; j < 4; ++j) CC[3 * r + j - 1] = cw[j][r] - cw[0][r];
  svd_opencv(3, 3, CC, wi, Ui, Vi);                     // (Ui, Vi: rows = vectors)
  // (cv::SVD::backSubst: singular values not above 2 eps sum(w) do not contribute - a PSEUDO-inverse: for (near-)coplanar points, whose fourth control
  // point falls onto the centroid, every point gets a zero fourth barycentric coordinate and the algorithm goes on)
  const double thr_cc = 2.0 * 2.220446049250313e-16 * (wi[0] + wi[1] + wi[2]);
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double s = 0; for (int k = 0; k < 3; ++k) if (wi[k] > thr_cc) s += Vi[3 * k + a] * (1.0 / wi[k]) * Ui[3 * k + b]; CCi[3 * a + b] = s; }
  std::vector<double> al(4 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    double* a = &al[4 * (size_t)i];
    for (int j = 0; j < 3; ++j) a[1 + j] = CCi[3 * j] * PW0[3 * (size_t)i] + CCi[3 * j + 1] * PW0[3 * (size_t)i + 1] + CCi[3 * j + 2] * PW0[3 * (size_t)i + 2];
    a[0] = 1.0 - a[1] - a[2] - a[3];
  }
  // ---- fill_M (dense) and M^T M
  std::vector<double> M(24 * (size_t)n, 0.0);
  for (int i = 0; i < n; ++i) {
    double* m1 = &M[24 * (size_t)i]; double* m2 = m1 + 12;
    const double* a = &al[4 * (size_t)i];
    for (int j = 0; j < 4; ++j) {
      m1[3 * j] = a[j] * fu; m1[3 * j + 1] = 0.0; m1[3 * j + 2] = a[j] * (uc - uv[2 * i]);
      m2[3 * j] = 0.0; m2[3 * j + 1] = a[j] * fv; m2[3 * j + 2] = a[j] * (vc - uv[2 * i + 1]);
    }
  }
  double MtM[144];
  for (int a = 0; a < 12; ++a) for (int b = a; b < 12; ++b) { double s = 0; for (int r = 0; r < 2 * n; ++r) s += M[12 * (size_t)r + a] * M[12 * (size_t)r + b]; MtM[12 * a + b] = MtM[12 * b + a] = s; }
  double w12[12], U12[144], V12[144];
  svd_opencv(12, 12, MtM, w12, U12, V12);
  // v[0] .. v[3]: the singular vectors of the four smallest singular values, smallest first - rows 11, 10, 9, 8 of U^T in OpenCV.
  // Read from V^T here: for the symmetric positive semi-definite M^T M the two coincide in exact arithmetic (same signs), but a
  // left vector is a row of A V^T divided by its norm and carries a relative error of eps sigma_max / sigma_i (1e-8 for typical
  // noisy data, unbounded for noise-free data), a right vector does not.  DEVIATION from OpenCV at that level, on purpose.
  double v[4][12];
  for (int e = 0; e < 4; ++e) for (int k = 0; k < 12; ++k) v[e][k] = V12[12 * (11 - e) + k];
  // ---- compute_L_6x10, compute_rho
  static const int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {1, 2, 3, 2, 3, 3};
  double L[6][10], rho[6];
  for (int i = 0; i < 6; ++i) {
    double d[4][3];
    for (int e = 0; e < 4; ++e) for (int k = 0; k < 3; ++k) d[e][k] = v[e][3 * PA[i] + k] - v[e][3 * PB[i] + k];
    auto dt = [&](int a, int b) { return d[a][0] * d[b][0] + d[a][1] * d[b][1] + d[a][2] * d[b][2]; };
    L[i][0] = dt(0, 0); L[i][1] = 2 * dt(0, 1); L[i][2] = dt(1, 1); L[i][3] = 2 * dt(0, 2); L[i][4] = 2 * dt(1, 2);
    L[i][5] = dt(2, 2); L[i][6] = 2 * dt(0, 3); L[i][7] = 2 * dt(1, 3); L[i][8] = 2 * dt(2, 3); L[i][9] = dt(3, 3);
    rho[i] = 0;
    for (int k = 0; k < 3; ++k) { const double q = cw[PA[i]][k] - cw[PB[i]][k]; rho[i] += q * q; }
  }
  // ---- the three initialisations of beta, Gauss-Newton, pose; the smallest mean reprojection error wins (ties: the earlier one)
  Result best = none;
  std::vector<double> pc(3 * (size_t)n);
  for (int N = 1; N <= 3; ++N) {
    double be[4] = {0, 0, 0, 0};
    if (N == 1) {                                       // [B11 B12 B13 B14]: columns 0, 1, 3, 6
      double A[24], s[4];
      for (int i = 0; i < 6; ++i) { A[4 * i] = L[i][0]; A[4 * i + 1] = L[i][1]; A[4 * i + 2] = L[i][3]; A[4 * i + 3] = L[i][6]; }
      solve_svd(6, 4, A, rho, s);
      const double sg = s[0] < 0 ? -1.0 : 1.0;
      be[0] = std::sqrt(sg * s[0]);
      for (int k = 1; k < 4; ++k) be[k] = sg * s[k] / be[0];
    } else if (N == 2) {                                // [B11 B12 B22]: columns 0, 1, 2
      double A[18], s[3];
      for (int i = 0; i < 6; ++i) for (int k = 0; k < 3; ++k) A[3 * i + k] = L[i][k];
      solve_svd(6, 3, A, rho, s);
      if (s[0] < 0) { be[0] = std::sqrt(-s[0]); be[1] = s[2] < 0 ? std::sqrt(-s[2]) : 0.0; }
      else { be[0] = std::sqrt(s[0]); be[1] = s[2] > 0 ? std::sqrt(s[2]) : 0.0; }
      if (s[1] < 0) be[0] = -be[0];
    } else {                                            // [B11 B12 B22 B13 B23]: columns 0 .. 4
      double A[30], s[5];
      for (int i = 0; i < 6; ++i) for (int k = 0; k < 5; ++k) A[5 * i + k] = L[i][k];
      solve_svd(6, 5, A, rho, s);
      if (s[0] < 0) { be[0] = std::sqrt(-s[0]); be[1] = s[2] < 0 ? std::sqrt(-s[2]) : 0.0; }
      else { be[0] = std::sqrt(s[0]); be[1] = s[2] > 0 ? std::sqrt(s[2]) : 0.0; }
      if (s[1] < 0) be[0] = -be[0];
      be[2] = s[3] / be[0];
    }
    for (int it = 0; it < 5; ++it) {                    // gauss_newton on  rho_i = sum_{a <= b} L_i[ab] beta_a beta_b
      double A[24], r[6], dx[4];
      for (int i = 0; i < 6; ++i) {
        const double* l = L[i];
        A[4 * i + 0] = 2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3];
        A[4 * i + 1] = l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3];
        A[4 * i + 2] = l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3];
        A[4 * i + 3] = l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3];
        r[i] = rho[i] - (l[0] * be[0] * be[0] + l[1] * be[0] * be[1] + l[2] * be[1] * be[1] + l[3] * be[0] * be[2] + l[4] * be[1] * be[2] +
                         l[5] * be[2] * be[2] + l[6] * be[0] * be[3] + l[7] * be[1] * be[3] + l[8] * be[2] * be[3] + l[9] * be[3] * be[3]);
      }
      if (!solve_givens_6x4(A, r, dx)) break;
      for (int k = 0; k < 4; ++k) be[k] += dx[k];
    }
    // ---- compute_R_and_t: control points and points in the camera frame, solve_for_sign, estimate_R_and_t, reprojection_error
    double cc[4][3] = {};
    for (int e = 0; e < 4; ++e) for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) cc[j][k] += be[e] * v[e][3 * j + k];
    for (int i = 0; i < n; ++i) { const double* a = &al[4 * (size_t)i]; for (int k = 0; k < 3; ++k) pc[3 * (size_t)i + k] = a[0] * cc[0][k] + a[1] * cc[1][k] + a[2] * cc[2][k] + a[3] * cc[3][k]; }
    if (pc[2] < 0) for (double& q : pc) q = -q;
    double pc0[3] = {}, pw0[3] = {};
    for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { pc0[k] += pc[3 * (size_t)i + k]; pw0[k] += X[3 * i + k]; }
    for (int k = 0; k < 3; ++k) { pc0[k] /= n; pw0[k] /= n; }
    double ABt[9] = {};
    for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) ABt[3 * a + b] += (pc[3 * (size_t)i + a] - pc0[a]) * (X[3 * i + b] - pw0[b]);
    double wa[3], Ua[9], Va[9];
    svd_opencv(3, 3, ABt, wa, Ua, Va);                  // (rows = vectors)
    if (!(wa[2] > 1e-14 * wa[0])) {                     // rank 2: the third left vector completes a right-handed frame
      Ua[6] = Ua[1] * Ua[5] - Ua[2] * Ua[4]; Ua[7] = Ua[2] * Ua[3] - Ua[0] * Ua[5]; Ua[8] = Ua[0] * Ua[4] - Ua[1] * Ua[3];
    }
    Result cur;                                         // R = U V^T
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cur.R[3 * a + b] = Ua[a] * Va[b] + Ua[3 + a] * Va[3 + b] + Ua[6 + a] * Va[6 + b];
    const double det = cur.R[0] * (cur.R[4] * cur.R[8] - cur.R[5] * cur.R[7]) - cur.R[1] * (cur.R[3] * cur.R[8] - cur.R[5] * cur.R[6]) + cur.R[2] * (cur.R[3] * cur.R[7] - cur.R[4] * cur.R[6]);
    if (det < 0) { cur.R[6] = -cur.R[6]; cur.R[7] = -cur.R[7]; cur.R[8] = -cur.R[8]; }
    for (int k = 0; k < 3; ++k) cur.t[k] = pc0[k] - (cur.R[3 * k] * pw0[0] + cur.R[3 * k + 1] * pw0[1] + cur.R[3 * k + 2] * pw0[2]);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
      const double* p = X + 3 * i;
      const double xc = cur.R[0] * p[0] + cur.R[1] * p[1] + cur.R[2] * p[2] + cur.t[0], yc = cur.R[3] * p[0] + cur.R[4] * p[1] + cur.R[5] * p[2] + cur.t[1];
      const double zi = 1.0 / (cur.R[6] * p[0] + cur.R[7] * p[1] + cur.R[8] * p[2] + cur.t[2]);
      sum += std::hypot(uv[2 * i] - (uc + fu * xc * zi), uv[2 * i + 1] - (vc + fv * yc * zi));
    }
    cur.err = sum / n;
    if (best.err < 0 || cur.err < best.err) best = cur;
  }
  return best;
}

}  // namespace ref_epnp
