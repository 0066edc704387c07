// EPnP re-estimation of the winning RANSAC model on its inliers - the last step of cv::solvePnPRansac(..., SOLVEPNP_AP3P) in
// OpenCV 3.4 (solvePnP(inliers, SOLVEPNP_EPNP)), which Tracking::GetInitModelCam / GetInitModelObj receive as their initial model
// (reference src/Tracking.cc:1652-1660, 1755-1763).  Host code of the C-ABI (vdo_pnp_ransac_batch): the data-parallel part of the
// initialiser - 500 hypotheses x all correspondences - runs on the GPU (ransac.hip); what is left here is O(n) sums over a few
// hundred inliers and 12x12 / 6x5 / 4x4 dense algebra, strictly sequential and far below a kernel launch in size.
// The 4-control-point formulation of Lepetit, Moreno-Noguer & Fua (IJCV 2009) as structured in OpenCV's epnp.cpp:
// control points from the PCA of the points, barycentric coordinates, M^T M and the eigenvectors of its 4 smallest eigenvalues,
// the L_6x10 / rho system, three beta initialisations each refined by 5 Gauss-Newton steps, absolute orientation, smallest mean
// reprojection error wins.  OpenCV itself is not available to pin this against (parity unpinned).
// Round 5: every step whose RESULT depends on how OpenCV computes it follows OpenCV's tool - the one-sided Jacobi SVD of
// modules/core/src/lapack.cpp (small_svd below) for the principal directions of the control points (their signs), for the inverse of
// the control-point matrix (cvInvert(CV_SVD): a pseudo-inverse, what makes (near-)coplanar point sets go through instead of dividing
// by zero), for the three beta systems (cvSolve(CV_SVD): singular values under 2 eps sum(w) dropped) and for the absolute orientation
// (R = U V^T of sum (pc - pc0)(pw - pw0)^T, THIRD ROW negated when det R < 0 - on near-planar noisy sets that is not the nearest rotation,
// and it is what the reference receives).  Rounds 2-4 used normal equations, Horn's quaternion and "coplanar: keep the hypothesis" there and
// differed from the oracle by up to 1.5 in the pose on near-planar inlier sets (profiles/HISTORY.md 7.2).  What stays the product's own: M^T M by
// running sums, its eigenvectors by tridiagonalisation + QL, the sums shared by the three candidates, Householder QR for the Gauss-Newton.
// The CPU oracle (oracle/epnp_oracle.hpp) is a separately written restatement (dense M, SVD of M^T M, Givens QR); the two
// are compared to 1e-9 / 1e-6 on near-planar sets (tests/test_epnp_independent.py on the CPU, tests/test_ransac_gpu.py through the C-ABI).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace vdo {
namespace epnp {

// Symmetric eigen-decomposition by Householder tridiagonalisation + implicit QL with Wilkinson shifts (the classical tred2 /
// tql2 pair) - ~5x fewer operations than cyclic Jacobi at n = 12.  A (row-major n x n, n <= 12) is overwritten; on return the
// COLUMNS of V are the eigenvectors, w the eigenvalues, sorted descending.
inline void sym_eig_ql(int n, double* A, double* V, double* w) {
  double d[12], e[12];
  double* z = V;
  for (int i = 0; i < n * n; ++i) z[i] = A[i];
  // ---- tred2: z -> orthogonal Q with Q^T A Q tridiagonal (d diagonal, e sub-diagonal)
  for (int i = n - 1; i > 0; --i) {
    const int l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0) {
      for (int k = 0; k <= l; ++k) scale += std::fabs(z[i * n + k]);
      if (scale == 0.0) e[i] = z[i * n + l];
      else {
        for (int k = 0; k <= l; ++k) { z[i * n + k] /= scale; h += z[i * n + k] * z[i * n + k]; }
        double f = z[i * n + l];
        double g = f >= 0.0 ? -std::sqrt(h) : std::sqrt(h);
        e[i] = scale * g;
        h -= f * g;
        z[i * n + l] = f - g;
        f = 0.0;
        for (int j = 0; j <= l; ++j) {
          z[j * n + i] = z[i * n + j] / h;
          g = 0.0;
          for (int k = 0; k <= j; ++k) g += z[j * n + k] * z[i * n + k];
          for (int k = j + 1; k <= l; ++k) g += z[k * n + j] * z[i * n + k];
          e[j] = g / h;
          f += e[j] * z[i * n + j];
        }
        const double hh = f / (h + h);
        for (int j = 0; j <= l; ++j) {
          f = z[i * n + j];
          e[j] = g = e[j] - hh * f;
          for (int k = 0; k <= j; ++k) z[j * n + k] -= f * e[k] + g * z[i * n + k];
        }
      }
    } else e[i] = z[i * n + l];
    d[i] = h;
  }
  d[0] = 0.0; e[0] = 0.0;
  for (int i = 0; i < n; ++i) {
    const int l = i - 1;
    if (d[i] != 0.0) {
      for (int j = 0; j <= l; ++j) {
        double g = 0.0;
        for (int k = 0; k <= l; ++k) g += z[i * n + k] * z[k * n + j];
        for (int k = 0; k <= l; ++k) z[k * n + j] -= g * z[k * n + i];
      }
    }
    d[i] = z[i * n + i];
    z[i * n + i] = 1.0;
    for (int j = 0; j <= l; ++j) z[j * n + i] = z[i * n + j] = 0.0;
  }
  // ---- tql2: implicit QL on (d, e), rotations accumulated into z
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 60) break;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::sqrt(g * g + 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          e[i + 1] = r = std::sqrt(f * f + g * g);
          if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
          s = f / r; c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
          for (int k = 0; k < n; ++k) {
            f = z[k * n + i + 1];
            z[k * n + i + 1] = s * z[k * n + i] + c * f;
            z[k * n + i] = c * z[k * n + i] - s * f;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p; e[l] = g; e[m] = 0.0;
      }
    } while (m != l);
  }
  for (int i = 0; i < n; ++i) w[i] = d[i];
  for (int i = 0; i < n - 1; ++i) {                   // selection sort, descending
    int m = i;
    for (int j = i + 1; j < n; ++j) if (w[j] > w[m]) m = j;
    if (m != i) {
      const double tw = w[i]; w[i] = w[m]; w[m] = tw;
      for (int k = 0; k < n; ++k) { const double tv = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = tv; }
    }
  }
}

// Principal directions of the 3 x 3 scatter matrix C with the SIGNS cv::SVD gives them.  EPnP's control points are
// centroid + sqrt(sigma_i / n) * (i-th principal direction): flipping a direction moves a control point to the other side of the
// centroid, and with noisy data the pose EPnP returns depends on that choice - so this one step follows the numerical scheme of
// OpenCV 3.4.0's SVD (one-sided Jacobi on the rows of the matrix, modules/core/src/lapack.cpp; the reference's OpenCV is built
// without LAPACK, /root/reference/Dockerfile:33-50): rows orthogonalised pairwise in the order (0,1), (0,2), (1,2), rotation with
// c >= 0 if |row i| >= |row j| else s >= 0, convergence at |<row i, row j>| <= 10 eps |row i| |row j|, at most 30 sweeps, singular
// values = row norms sorted descending (selection, swaps), directions = rows / norm.  d: the three rows (unit vectors), w: their
// singular values.
inline void principal_directions(const double C[9], double w[3], double d[9]) {
  double r[3][3] = {{C[0], C[1], C[2]}, {C[3], C[4], C[5]}, {C[6], C[7], C[8]}};
  double n2[3];
  for (int i = 0; i < 3; ++i) n2[i] = r[i][0] * r[i][0] + r[i][1] * r[i][1] + r[i][2] * r[i][2];
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool changed = false;
    for (int i = 0; i < 2; ++i)
      for (int j = i + 1; j < 3; ++j) {
        double p = r[i][0] * r[j][0] + r[i][1] * r[j][1] + r[i][2] * r[j][2];
        if (std::fabs(p) <= 2.220446049250313e-15 * std::sqrt(n2[i] * n2[j])) continue;
        p *= 2.0;
        const double beta = n2[i] - n2[j], gamma = std::hypot(p, beta);
        double c, s;
        if (beta < 0.0) { s = std::sqrt((gamma - beta) * 0.5 / gamma); c = p / (gamma * s * 2.0); }
        else { c = std::sqrt((gamma + beta) / (gamma * 2.0)); s = p / (gamma * c * 2.0); }
        double ni = 0.0, nj = 0.0;
        for (int k = 0; k < 3; ++k) {
          const double ri = c * r[i][k] + s * r[j][k], rj = c * r[j][k] - s * r[i][k];
          r[i][k] = ri; r[j][k] = rj; ni += ri * ri; nj += rj * rj;
        }
        n2[i] = ni; n2[j] = nj;
        changed = true;
      }
    if (!changed) break;
  }
  int ord[3] = {0, 1, 2};
  double nr[3];
  for (int i = 0; i < 3; ++i) nr[i] = std::sqrt(r[i][0] * r[i][0] + r[i][1] * r[i][1] + r[i][2] * r[i][2]);
  for (int i = 0; i < 2; ++i) {
    int m = i;
    for (int k = i + 1; k < 3; ++k) if (nr[ord[m]] < nr[ord[k]]) m = k;
    if (m != i) { const int t = ord[i]; ord[i] = ord[m]; ord[m] = t; }
  }
  for (int i = 0; i < 3; ++i) {
    w[i] = nr[ord[i]];
    const double inv = w[i] > 0.0 ? 1.0 / w[i] : 0.0;
    for (int k = 0; k < 3; ++k) d[3 * i + k] = r[ord[i]][k] * inv;
  }
}

// cv::SVD of a small matrix as OpenCV 3.4.0 computes it without LAPACK (JacobiSVDImpl_, modules/core/src/lapack.cpp): Hestenes' one-sided
// Jacobi on the COLUMNS of A.  The conventions that decide order and sign of the vectors are OpenCV's: column pairs (i, j), i < j, in row-major
// order; a pair is left alone when |<a_i, a_j>| <= 10 eps |a_i| |a_j|; the rotation has c >= 0 when |a_i| >= |a_j|, else s >= 0; V starts as the
// identity and takes the same rotations; at most max(m, 30) sweeps; singular values = column norms, sorted descending by selection with swaps;
// left vectors = columns / singular value (a zero singular value leaves a zero vector: OpenCV fills in a pseudo-random one, nothing here reads it).
// A: m x n row-major, m <= MR, n <= NC.  w [n]; U [n][MR]: ROW j = j-th left singular vector (first m entries); V [n][NC]: ROW j = j-th right one.
template <int MR, int NC>
inline void jacobi_svd(int m, int n, const double* A, double* w, double (*U)[MR], double (*V)[NC]) {
  double col[NC][MR], nrm2[NC];
  for (int j = 0; j < n; ++j) {
    double s = 0.0;
    for (int r = 0; r < m; ++r) { col[j][r] = A[r * n + j]; s += col[j][r] * col[j][r]; }
    nrm2[j] = s;
    for (int k = 0; k < n; ++k) V[j][k] = (j == k) ? 1.0 : 0.0;
  }
  const int sweeps = m > 30 ? m : 30;
  for (int sweep = 0; sweep < sweeps; ++sweep) {
    bool rotated = false;
    for (int i = 0; i + 1 < n; ++i)
      for (int j = i + 1; j < n; ++j) {
        double p = 0.0;
        for (int r = 0; r < m; ++r) p += col[i][r] * col[j][r];
        if (std::fabs(p) <= 2.220446049250313e-15 * std::sqrt(nrm2[i] * nrm2[j])) continue;
        p *= 2.0;
        const double beta = nrm2[i] - nrm2[j], gamma = std::hypot(p, beta);
        double c, s;
        if (beta < 0.0) { s = std::sqrt((gamma - beta) * 0.5 / gamma); c = p / (gamma * s * 2.0); }
        else { c = std::sqrt((gamma + beta) / (gamma * 2.0)); s = p / (gamma * c * 2.0); }
        double ni = 0.0, nj = 0.0;
        for (int r = 0; r < m; ++r) {
          const double xi = c * col[i][r] + s * col[j][r], xj = c * col[j][r] - s * col[i][r];
          col[i][r] = xi; col[j][r] = xj; ni += xi * xi; nj += xj * xj;
        }
        nrm2[i] = ni; nrm2[j] = nj;
        for (int k = 0; k < n; ++k) { const double vi = c * V[i][k] + s * V[j][k], vj = c * V[j][k] - s * V[i][k]; V[i][k] = vi; V[j][k] = vj; }
        rotated = true;
      }
    if (!rotated) break;
  }
  int ord[NC];
  double nr[NC];
  for (int j = 0; j < n; ++j) { double s = 0.0; for (int r = 0; r < m; ++r) s += col[j][r] * col[j][r]; nr[j] = std::sqrt(s); ord[j] = j; }
  for (int i = 0; i + 1 < n; ++i) {                   // (selection with swaps: the order of equal values is OpenCV's)
    int b = i;
    for (int k = i + 1; k < n; ++k) if (nr[ord[b]] < nr[ord[k]]) b = k;
    if (b != i) { const int t = ord[i]; ord[i] = ord[b]; ord[b] = t; }
  }
  double Vs[NC][NC];
  for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k) Vs[j][k] = V[ord[j]][k];
  for (int j = 0; j < n; ++j) {
    w[j] = nr[ord[j]];
    const double inv = w[j] > 2.2250738585072014e-308 ? 1.0 / w[j] : 0.0;
    for (int r = 0; r < m; ++r) U[j][r] = col[ord[j]][r] * inv;
    for (int k = 0; k < n; ++k) V[j][k] = Vs[j][k];
  }
}
inline void small_svd(int m, int n, const double* A, double* w, double (*U)[6], double (*V)[5]) { jacobi_svd<6, 5>(m, n, A, w, U, V); }

// cvSolve(A, b, x, CV_SVD) for the beta systems (m = 6, k <= 5): x = V diag(1/w) U^T b over the singular values above 2 eps sum(w)
inline void lstsq_svd(int m, int k, const double* A, const double* b, double* x) {
  double w[5], U[5][6], V[5][5];
  small_svd(m, k, A, w, U, V);
  double thr = 0.0;
  for (int j = 0; j < k; ++j) thr += w[j];
  thr *= 2.0 * 2.220446049250313e-16;
  for (int i = 0; i < k; ++i) x[i] = 0.0;
  for (int j = 0; j < k; ++j) {
    if (!(w[j] > thr)) continue;
    double ub = 0.0;
    for (int r = 0; r < m; ++r) ub += U[j][r] * b[r];
    ub /= w[j];
    for (int i = 0; i < k; ++i) x[i] += V[j][i] * ub;
  }
}

// least squares of the 6 x 4 Gauss-Newton system by Householder QR; false when A is (numerically) rank deficient
inline bool qr_solve_6x4(double* A /*6x4 row-major, destroyed*/, double* b /*6, destroyed*/, double* x /*4*/) {
  const int m = 6, n = 4;
  for (int k = 0; k < n; ++k) {
    double norm2 = 0.0; for (int i = k; i < m; ++i) norm2 += A[i * n + k] * A[i * n + k];
    if (!(norm2 > 0.0)) return false;
    const double alpha = (A[k * n + k] > 0.0 ? -1.0 : 1.0) * std::sqrt(norm2);
    double v[6]; for (int i = k; i < m; ++i) v[i] = A[i * n + k];
    v[k] -= alpha;
    double vtv = 0.0; for (int i = k; i < m; ++i) vtv += v[i] * v[i];
    if (!(vtv > 0.0)) return false;
    for (int j = k; j < n; ++j) {
      double dot = 0.0; for (int i = k; i < m; ++i) dot += v[i] * A[i * n + j];
      const double f = 2.0 * dot / vtv;
      for (int i = k; i < m; ++i) A[i * n + j] -= f * v[i];
    }
    { double dot = 0.0; for (int i = k; i < m; ++i) dot += v[i] * b[i]; const double f = 2.0 * dot / vtv; for (int i = k; i < m; ++i) b[i] -= f * v[i]; }
  }
  for (int k = n - 1; k >= 0; --k) {
    double s = b[k]; for (int j = k + 1; j < n; ++j) s -= A[k * n + j] * x[j];
    if (A[k * n + k] == 0.0) return false;
    x[k] = s / A[k * n + k];
  }
  return true;
}

struct Result { double R[9], t[3], err; };
struct Scratch { std::vector<double> alphas; };      // per-thread buffers, grown on demand (no allocation in steady state)

// X [n][3] world points, uv [n][2] pixels, K4 = fx, fy, cx, cy.  n >= 4.
inline Result solve(int n, const double* X, const double* uv, const double* K4, Scratch& scr) {
  const double fu = K4[0], fv = K4[1], uc = K4[2], vc = K4[3];
  // ---- choose_control_points
  double cws[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) cws[0][j] += X[3 * i + j];
  for (int j = 0; j < 3; ++j) cws[0][j] /= n;
  double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double d0 = X[3 * i] - cws[0][0], d1 = X[3 * i + 1] - cws[0][1], d2 = X[3 * i + 2] - cws[0][2];
    C[0] += d0 * d0; C[1] += d0 * d1; C[2] += d0 * d2; C[4] += d1 * d1; C[5] += d1 * d2; C[8] += d2 * d2;
  }
  C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
  double Dc[9], dc[3];
  principal_directions(C, dc, Dc);
  // ((near-)coplanar points: dc[2] -> 0, the fourth control point falls onto the centroid; OpenCV goes on - the pseudo-inverse below then gives
  // every point a zero fourth coordinate - and so does this)
  for (int i = 1; i < 4; ++i) {
    const double k = std::sqrt((dc[i - 1] > 0.0 ? dc[i - 1] : 0.0) / n);
    for (int j = 0; j < 3; ++j) cws[i][j] = cws[0][j] + k * Dc[3 * (i - 1) + j];
  }
  // ---- compute_barycentric_coordinates: cvInvert(CC, CC_inv, CV_SVD) = V diag(1/w) U^T over the singular values above 2 eps sum(w)
  double cc[9], ci[9];
  for (int i = 0; i < 3; ++i) for (int j = 1; j < 4; ++j) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
  {
    double wc[5], Uc[5][6], Vc[5][5];
    small_svd(3, 3, cc, wc, Uc, Vc);
    const double thr = 2.0 * 2.220446049250313e-16 * (wc[0] + wc[1] + wc[2]);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double acc = 0.0;
        for (int k = 0; k < 3; ++k) if (wc[k] > thr) acc += Vc[k][a] * (1.0 / wc[k]) * Uc[k][b];
        ci[3 * a + b] = acc;
      }
  }
  std::vector<double>& alphas = scr.alphas;
  if (alphas.size() < 4 * (size_t)n) alphas.resize(4 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    const double* pi = X + 3 * i; double* a = &alphas[4 * (size_t)i];
    for (int j = 0; j < 3; ++j) a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
    a[0] = 1.0 - a[1] - a[2] - a[3];
  }
  // ---- sums over the points that the three candidate solutions below share.  The camera-frame points are pc_i = sum_j a_ij ccs_j (ccs = the
  // candidate's control points), so their mean is sum_j abar_j ccs_j and Horn's matrix  S = sum_i (pw_i - pw0) (pc_i - pc0)^T  equals
  // sum_j Q_j ccs_j^T  with  Q_j = sum_i a_ij (pw_i - pw0)  (the deviations pw_i - pw0 sum to zero, pc0 drops out): abar and Q are formed ONCE,
  // a candidate costs O(1) for its absolute orientation instead of three passes over the points.
  double pw0[3] = {0, 0, 0}, abar[4] = {0, 0, 0, 0}, Qj[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) pw0[k] += X[3 * i + k];
  for (int k = 0; k < 3; ++k) pw0[k] /= n;
  for (int i = 0; i < n; ++i) {
    const double* a = &alphas[4 * (size_t)i];
    const double d0 = X[3 * i] - pw0[0], d1 = X[3 * i + 1] - pw0[1], d2 = X[3 * i + 2] - pw0[2];
    for (int j = 0; j < 4; ++j) { abar[j] += a[j]; Qj[j][0] += a[j] * d0; Qj[j][1] += a[j] * d1; Qj[j][2] += a[j] * d2; }
  }
  for (int j = 0; j < 4; ++j) abar[j] /= n;
  // ---- M^T M (12 x 12).  Rows of M: [a_j fu, 0, a_j (uc - u)] and [0, a_j fv, a_j (vc - v)] per control point j, so the 3 x 3
  // block (j, k) of M^T M is  sum_i a_j a_k [fu^2, 0, fu du; 0, fv^2, fv dv; fu du, fv dv, du^2 + dv^2]  with du = uc - u, dv = vc - v:
  // four running sums per control-point pair instead of the 78 products of the dense rows
  double s1[16], su[16], sv[16], sq[16];
  for (int i = 0; i < 16; ++i) s1[i] = su[i] = sv[i] = sq[i] = 0.0;
  for (int i = 0; i < n; ++i) {
    const double* a = &alphas[4 * (size_t)i];
    const double du = uc - uv[2 * i], dv = vc - uv[2 * i + 1], dd = du * du + dv * dv;
    for (int j = 0; j < 4; ++j)
      for (int k = j; k < 4; ++k) {
        const double ajk = a[j] * a[k];
        s1[4 * j + k] += ajk; su[4 * j + k] += ajk * du; sv[4 * j + k] += ajk * dv; sq[4 * j + k] += ajk * dd;
      }
  }
  double MtM[144];
  for (int j = 0; j < 4; ++j)
    for (int k = 0; k < 4; ++k) {
      const int jj = j <= k ? j : k, kk = j <= k ? k : j;
      const double a1 = s1[4 * jj + kk], au = su[4 * jj + kk], av = sv[4 * jj + kk], aq = sq[4 * jj + kk];
      double* Bk = MtM + (3 * j) * 12 + 3 * k;
      Bk[0] = fu * fu * a1; Bk[1] = 0.0; Bk[2] = fu * au;
      Bk[12] = 0.0; Bk[13] = fv * fv * a1; Bk[14] = fv * av;
      Bk[24] = fu * au; Bk[25] = fv * av; Bk[26] = aq;
    }
  double V12[144], w12[12], MtM_in[144];
  std::memcpy(MtM_in, MtM, sizeof MtM);
  sym_eig_ql(12, MtM, V12, w12);
  const double* vcol[4];   // v[0] = eigenvector of the smallest eigenvalue ... (ut + 12*11, 12*10, 12*9, 12*8 in OpenCV's U^T)
  double vv[4][12];
  for (int e = 0; e < 4; ++e) { for (int k = 0; k < 12; ++k) vv[e][k] = V12[k * 12 + (11 - e)]; vcol[e] = vv[e]; }
  // When the four smallest eigenvalues do not stand apart - fewer than 6 points (2n < 11 equations: several EXACT zeros), exactly coplanar points seen
  // without noise - "the eigenvectors of the four smallest" are a basis of a degenerate eigenspace and which basis comes out is the solver's doing.
  // There (only there: it costs 5x the QL) the vectors are OpenCV's: the one-sided Jacobi SVD of M^T M, right singular vectors, as the oracle reads them.
  if (n < 6 || !(w12[8] > 1e-9 * w12[0])) {
    static_assert(sizeof(double[12][12]) == 144 * sizeof(double), "");
    double ws[12], Us[12][12], Vs[12][12];
    jacobi_svd<12, 12>(12, 12, MtM_in, ws, Us, Vs);
    for (int e = 0; e < 4; ++e) for (int k = 0; k < 12; ++k) vv[e][k] = Vs[11 - e][k];
  }
  // ---- compute_L_6x10, compute_rho
  double dv[4][6][3];
  for (int i = 0; i < 4; ++i) {
    int a = 0, b = 1;
    for (int j = 0; j < 6; ++j) {
      for (int k = 0; k < 3; ++k) dv[i][j][k] = vcol[i][3 * a + k] - vcol[i][3 * b + k];
      ++b;
      if (b > 3) { ++a; b = a + 1; }
    }
  }
  auto dot3 = [](const double* p, const double* q) { return p[0] * q[0] + p[1] * q[1] + p[2] * q[2]; };
  double L[60], rho[6];
  for (int i = 0; i < 6; ++i) {
    double* row = L + 10 * i;
    row[0] = dot3(dv[0][i], dv[0][i]); row[1] = 2.0 * dot3(dv[0][i], dv[1][i]); row[2] = dot3(dv[1][i], dv[1][i]);
    row[3] = 2.0 * dot3(dv[0][i], dv[2][i]); row[4] = 2.0 * dot3(dv[1][i], dv[2][i]); row[5] = dot3(dv[2][i], dv[2][i]);
    row[6] = 2.0 * dot3(dv[0][i], dv[3][i]); row[7] = 2.0 * dot3(dv[1][i], dv[3][i]); row[8] = 2.0 * dot3(dv[2][i], dv[3][i]);
    row[9] = dot3(dv[3][i], dv[3][i]);
  }
  auto dist2 = [](const double* p, const double* q) { return (p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2]); };
  rho[0] = dist2(cws[0], cws[1]); rho[1] = dist2(cws[0], cws[2]); rho[2] = dist2(cws[0], cws[3]);
  rho[3] = dist2(cws[1], cws[2]); rho[4] = dist2(cws[1], cws[3]); rho[5] = dist2(cws[2], cws[3]);
  // ---- the three beta initialisations + Gauss-Newton + pose; smallest reprojection error wins (N = 1, then 2, then 3)
  Result best{}; best.err = -1.0;
  for (int N = 1; N <= 3; ++N) {
    double betas[4] = {0, 0, 0, 0};
    if (N == 1) {                                     // betas_approx_1 = [B11 B12 B13 B14]
      double A[24], b4[4];
      for (int i = 0; i < 6; ++i) { A[4 * i] = L[10 * i]; A[4 * i + 1] = L[10 * i + 1]; A[4 * i + 2] = L[10 * i + 3]; A[4 * i + 3] = L[10 * i + 6]; }
      lstsq_svd(6, 4, A, rho, b4);
      if (b4[0] < 0) { betas[0] = std::sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0]; }
      else { betas[0] = std::sqrt(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0]; }
    } else if (N == 2) {                              // betas_approx_2 = [B11 B12 B22]
      double A[18], b3[3];
      for (int i = 0; i < 6; ++i) { A[3 * i] = L[10 * i]; A[3 * i + 1] = L[10 * i + 1]; A[3 * i + 2] = L[10 * i + 2]; }
      lstsq_svd(6, 3, A, rho, b3);
      if (b3[0] < 0) { betas[0] = std::sqrt(-b3[0]); betas[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0; }
      else { betas[0] = std::sqrt(b3[0]); betas[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0; }
      if (b3[1] < 0) betas[0] = -betas[0];
      betas[2] = 0.0; betas[3] = 0.0;
    } else {                                          // betas_approx_3 = [B11 B12 B22 B13 B23]
      double A[30], b5[5];
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 5; ++j) A[5 * i + j] = L[10 * i + j];
      lstsq_svd(6, 5, A, rho, b5);
      if (b5[0] < 0) { betas[0] = std::sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0; }
      else { betas[0] = std::sqrt(b5[0]); betas[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0; }
      if (b5[1] < 0) betas[0] = -betas[0];
      betas[2] = b5[3] / betas[0]; betas[3] = 0.0;
    }
    for (int it = 0; it < 5; ++it) {                  // gauss_newton
      double A[24], B[6], x[4];
      for (int i = 0; i < 6; ++i) {
        const double* rl = L + 10 * i; double* ra = A + 4 * i;
        ra[0] = 2 * rl[0] * betas[0] + rl[1] * betas[1] + rl[3] * betas[2] + rl[6] * betas[3];
        ra[1] = rl[1] * betas[0] + 2 * rl[2] * betas[1] + rl[4] * betas[2] + rl[7] * betas[3];
        ra[2] = rl[3] * betas[0] + rl[4] * betas[1] + 2 * rl[5] * betas[2] + rl[8] * betas[3];
        ra[3] = rl[6] * betas[0] + rl[7] * betas[1] + rl[8] * betas[2] + 2 * rl[9] * betas[3];
        B[i] = rho[i] - (rl[0] * betas[0] * betas[0] + rl[1] * betas[0] * betas[1] + rl[2] * betas[1] * betas[1] + rl[3] * betas[0] * betas[2] +
                         rl[4] * betas[1] * betas[2] + rl[5] * betas[2] * betas[2] + rl[6] * betas[0] * betas[3] + rl[7] * betas[1] * betas[3] +
                         rl[8] * betas[2] * betas[3] + rl[9] * betas[3] * betas[3]);
      }
      if (!qr_solve_6x4(A, B, x)) break;
      for (int i = 0; i < 4; ++i) betas[i] += x[i];
    }
    // ---- compute_R_and_t: control points in the camera frame, points in the camera frame, sign, absolute orientation
    double ccs[4][3];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 3; ++k) ccs[i][k] = 0.0;
    for (int e = 0; e < 4; ++e) for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) ccs[j][k] += betas[e] * vcol[e][3 * j + k];
    // solve_for_sign: the depth of the FIRST point in the camera frame decides the sign of the whole solution (OpenCV epnp.cpp solve_for_sign)
    {
      const double* a = &alphas[0];
      const double z0 = a[0] * ccs[0][2] + a[1] * ccs[1][2] + a[2] * ccs[2][2] + a[3] * ccs[3][2];
      if (z0 < 0.0) for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) ccs[j][k] = -ccs[j][k];
    }
    double pc0[3] = {0, 0, 0};
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) pc0[k] += abar[j] * ccs[j][k];
    double Sm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // S[r][k] = sum (pw - pw0)[r] (pc - pc0)[k] = sum_j Q_j[r] ccs_j[k]   (Horn: rotation world -> camera)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Sm[3 * r + k] += Qj[j][r] * ccs[j][k];
    // estimate_R_and_t: ABt = sum (pc - pc0)(pw - pw0)^T = Sm^T;  R = U V^T;  det R < 0: the third ROW changes sign (OpenCV epnp.cpp)
    Result cur;
    {
      const double ABt[9] = {Sm[0], Sm[3], Sm[6], Sm[1], Sm[4], Sm[7], Sm[2], Sm[5], Sm[8]};
      double wa[5], Ua[5][6], Va[5][5];
      small_svd(3, 3, ABt, wa, Ua, Va);
      if (!(wa[2] > 1e-14 * wa[0])) {                 // rank 2: the third left vector completes a right-handed frame (OpenCV: a pseudo-random completion)
        Ua[2][0] = Ua[0][1] * Ua[1][2] - Ua[0][2] * Ua[1][1]; Ua[2][1] = Ua[0][2] * Ua[1][0] - Ua[0][0] * Ua[1][2]; Ua[2][2] = Ua[0][0] * Ua[1][1] - Ua[0][1] * Ua[1][0];
      }
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cur.R[3 * a + b] = Ua[0][a] * Va[0][b] + Ua[1][a] * Va[1][b] + Ua[2][a] * Va[2][b];
      const double det = cur.R[0] * (cur.R[4] * cur.R[8] - cur.R[5] * cur.R[7]) - cur.R[1] * (cur.R[3] * cur.R[8] - cur.R[5] * cur.R[6]) +
                         cur.R[2] * (cur.R[3] * cur.R[7] - cur.R[4] * cur.R[6]);
      if (det < 0.0) { cur.R[6] = -cur.R[6]; cur.R[7] = -cur.R[7]; cur.R[8] = -cur.R[8]; }
    }
    for (int k = 0; k < 3; ++k) cur.t[k] = pc0[k] - (cur.R[3 * k] * pw0[0] + cur.R[3 * k + 1] * pw0[1] + cur.R[3 * k + 2] * pw0[2]);
    // reprojection_error: mean pixel distance.  Four running sums (point i goes to sum i mod 4, added as (s0 + s1) + (s2 + s3)): the divisions and
    // square roots of four points are independent of each other and of the additions, so they pipeline (and vectorise) instead of queueing
    // behind one accumulator; the order is fixed in the source, the same bits from every compiler.
    double sl[4] = {0.0, 0.0, 0.0, 0.0};
    auto reproj = [&](int i) {
      const double* pw = X + 3 * i;
      const double Xc = cur.R[0] * pw[0] + cur.R[1] * pw[1] + cur.R[2] * pw[2] + cur.t[0];
      const double Yc = cur.R[3] * pw[0] + cur.R[4] * pw[1] + cur.R[5] * pw[2] + cur.t[1];
      const double inv_Zc = 1.0 / (cur.R[6] * pw[0] + cur.R[7] * pw[1] + cur.R[8] * pw[2] + cur.t[2]);
      const double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
      const double u = uv[2 * i], v = uv[2 * i + 1];
      return std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    };
    int i4 = 0;
    for (; i4 + 3 < n; i4 += 4) { sl[0] += reproj(i4); sl[1] += reproj(i4 + 1); sl[2] += reproj(i4 + 2); sl[3] += reproj(i4 + 3); }
    for (int r = 0; i4 < n; ++i4, ++r) sl[r] += reproj(i4);
    cur.err = ((sl[0] + sl[1]) + (sl[2] + sl[3])) / n;
    if (best.err < 0.0 || cur.err < best.err) best = cur;
  }
  return best;
}

}  // namespace epnp
}  // namespace vdo
