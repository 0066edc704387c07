// TEST INFRASTRUCTURE — CPU oracle (see vdo_oracle.h).  Per-frame joint pose + optical-flow
// optimisation: Optimizer::PoseOptimizationFlow2Cam (src/Optimizer.cc:2333-2542, camera) and
// Optimizer::PoseOptimizationFlow2 (:2755-2972, per object), i.e. g2o with
//   VertexSE3Expmap           g2o/types/types_six_dof_expmap.h:67-85, se3quat.h
//   VertexSBAFlow             g2o/types/types_sba.h:78-96 (2-DoF, marginalised)
//   EdgeSE3ProjectFlow2       g2o/types/types_six_dof_expmap.h:436-476, .cpp:805-845
//   EdgeFlowPrior             g2o/types/types_six_dof_expmap.h:414-432, .cpp:772-775
//   BlockSolver_6_3 + Schur   g2o/core/block_solver.hpp:354-486  with LinearSolverDense (LDLT)
//   Levenberg + stop rules    g2o/core/optimization_algorithm_levenberg.cpp:61-164, sparse_optimizer.cpp:354-443
//
// ref_quirks = 1 reproduces SURVEY.md F3: BlockSolver_6_3 assumes 3-DoF landmarks while
// VertexSBAFlow has 2.  The 2x2 landmark Hessian (column-major h00,h10,h01,h11) is aliased onto
// the first four doubles of a column-major 3x3 block, lambda is added on the 3x3 diagonal, the
// 3x3 is inverted, and the fixed-size 3-wide axpy/atxpy helpers run at stride-2 offsets:
//     D3 = [h00+l  h11  0 ; h10  l  0 ; h01  0  l]
//   * Schur / reduced rhs use the top-left 2x2 of D3^-1 (third column of the 6x3 Hpl block is 0),
//   * back-substitution: x[2i..2i+1] = (D3_i^-1 c_i)[0..1] and ADDITIONALLY x[2i+2] += (D3_i^-1 c_i)[2]
//     with c_i = (cl[2i], cl[2i+1], cl[2i+2]) — i.e. row 2 leaks into the next landmark.
//   * The reads one past the end for the last landmark hit uninitialised heap memory in the
//     reference; they only ever get multiplied by exact zeros or land outside the solution
//     vector, so they are modelled as 0.0 here (assumes the garbage is finite).
// The dense 6x6 solve restates Eigen::LDLT (lower triangle, diagonal pivoting, isPositive()).
// ref_quirks = 0 is the mathematically intended 2x2 Schur step (NOT parity-comparable).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "ref_math.hpp"
#include "vdo_oracle.h"

using namespace vdo_oracle;

namespace {

// Eigen::LDLT<MatrixXd, Lower>::compute + solve for n = 6 (unblocked algorithm of
// Eigen/src/Cholesky/LDLT.h; Eigen is not vendored in the reference — restated).
struct LDLT6 {
  double m[36];
  int tr[6];
  int sign;  // 0 zero, 1 possemidef, -1 negsemidef, 2 indefinite
  bool compute(const double* A) {
    std::memcpy(m, A, sizeof(m));
    const int n = 6;
    sign = 0;
    for (int k = 0; k < n; ++k) {
      int big = k;
      double bv = std::fabs(m[k * 6 + k]);
      for (int i = k + 1; i < n; ++i) if (std::fabs(m[i * 6 + i]) > bv) { bv = std::fabs(m[i * 6 + i]); big = i; }
      tr[k] = big;
      if (k != big) {
        // symmetric swap on the lower triangle
        const int s = n - big - 1;
        for (int j = 0; j < k; ++j) std::swap(m[k * 6 + j], m[big * 6 + j]);
        for (int i = 0; i < s; ++i) std::swap(m[(big + 1 + i) * 6 + k], m[(big + 1 + i) * 6 + big]);
        std::swap(m[k * 6 + k], m[big * 6 + big]);
        for (int i = k + 1; i < big; ++i) std::swap(m[i * 6 + k], m[big * 6 + i]);
      }
      const int rs = n - k - 1;
      if (k > 0) {
        double temp[6];
        for (int j = 0; j < k; ++j) temp[j] = m[j * 6 + j] * m[k * 6 + j];
        double s = 0;
        for (int j = 0; j < k; ++j) s += m[k * 6 + j] * temp[j];
        m[k * 6 + k] -= s;
        for (int i = 0; i < rs; ++i) {
          double t = 0;
          for (int j = 0; j < k; ++j) t += m[(k + 1 + i) * 6 + j] * temp[j];
          m[(k + 1 + i) * 6 + k] -= t;
        }
      }
      const double akk = m[k * 6 + k];
      const bool valid = std::fabs(akk) > 0.0;
      if (k == 0 && !valid) {  // whole matrix is zero
        sign = 0;
        for (int j = 0; j < n; ++j) tr[j] = j;
        break;
      }
      if (rs > 0 && valid) for (int i = 0; i < rs; ++i) m[(k + 1 + i) * 6 + k] /= akk;
      if (sign == 1) { if (akk < 0) sign = 2; }
      else if (sign == -1) { if (akk > 0) sign = 2; }
      else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = -1; }
    }
    return true;
  }
  bool isPositive() const { return sign == 1 || sign == 0; }
  void solve(const double* b, double* x) const {
    const int n = 6;
    for (int i = 0; i < n; ++i) x[i] = b[i];
    for (int k = 0; k < n; ++k) std::swap(x[k], x[tr[k]]);                 // P b
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) x[i] -= m[i * 6 + j] * x[j];   // L^-1
    const double tol = std::numeric_limits<double>::min();
    for (int i = 0; i < n; ++i) { if (std::fabs(m[i * 6 + i]) > tol) x[i] /= m[i * 6 + i]; else x[i] = 0; }
    for (int i = n - 1; i >= 0; --i) for (int j = i + 1; j < n; ++j) x[i] -= m[j * 6 + i] * x[j];  // L^-T
    for (int k = n - 1; k >= 0; --k) std::swap(x[k], x[tr[k]]);            // P^T
  }
};

// Eigen fixed-size 3x3 inverse (cofactors / determinant), row-major in/out
void inv3(const double* a, double* o) {
  const double c00 = a[4] * a[8] - a[5] * a[7];
  const double c10 = a[5] * a[6] - a[3] * a[8];   // cofactor(1,0) sign included below
  const double c20 = a[3] * a[7] - a[4] * a[6];
  // det = sum(cofactors_col0 .* col0) with cofactors_col0 = (C00, C10, C20), Cij = cofactor of a(i,j)
  const double C00 = c00;
  const double C10 = a[2] * a[7] - a[1] * a[8];
  const double C20 = a[1] * a[5] - a[2] * a[4];
  const double det = (C00 * a[0] + C10 * a[3]) + C20 * a[6];
  const double id = 1.0 / det;
  o[0] = C00 * id; o[1] = C10 * id; o[2] = C20 * id;
  o[3] = c10 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  o[6] = c20 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}
void inv2(const double* a, double* o) {
  const double det = a[0] * a[3] - a[1] * a[2], id = 1.0 / det;
  o[0] = a[3] * id; o[1] = -a[1] * id; o[2] = -a[2] * id; o[3] = a[0] * id;
}

struct Flow2 {
  const vdo_flow2_problem* P;
  int N;
  SE3Quat T;
  std::vector<double> f;          // flow estimates [2N]
  std::vector<V3> Xw;
  Huber hub;
  // linear system
  double Hpp[36], bp[6];
  std::vector<double> hl, bl, B2;  // hl [4N] (h00,h10,h01,h11), bl [2N], B2 [12N] (6x2 row-major)
  std::vector<double> x;           // [6 + 2N]
  std::vector<double> err;         // last computed projection-edge errors [2N]
  std::vector<double> errp;        // last computed prior-edge errors [2N]

  explicit Flow2(const vdo_flow2_problem* p) : P(p), N(p->n) {
    M3 R;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = p->T0[4 * i + j];
    T = SE3Quat::fromRt(R, v3(p->T0[3], p->T0[7], p->T0[11]));      // Converter::toSE3Quat
    f.assign(p->flow, p->flow + 2 * (size_t)N);
    Xw.resize(N);
    const double fx = p->K[0], fy = p->K[1], cx = p->K[2], cy = p->K[3];
    for (int i = 0; i < N; ++i) {
      const double d = p->depth[i];
      V3 Xc{(p->obs[2 * i] - cx) * d / fx, (p->obs[2 * i + 1] - cy) * d / fy, d};
      // Xw = Twl.block(0,0,3,3)*Xw + Twl.col(3).head(3)
      const double* W = p->Twl;
      Xw[i] = v3(W[0] * Xc.x + W[1] * Xc.y + W[2] * Xc.z + W[3], W[4] * Xc.x + W[5] * Xc.y + W[6] * Xc.z + W[7],
                 W[8] * Xc.x + W[9] * Xc.y + W[10] * Xc.z + W[11]);
    }
    hub.setDelta(p->huber_delta);
    hl.resize(4 * (size_t)N); bl.resize(2 * (size_t)N); B2.resize(12 * (size_t)N);
    x.assign(6 + 2 * (size_t)N + 1, 0.0);
    err.resize(2 * (size_t)N); errp.resize(2 * (size_t)N);
  }

  // computeActiveErrors + activeRobustChi2
  double compute_errors() {
    const double fx = P->K[0], fy = P->K[1], cx = P->K[2], cy = P->K[3];
    double rchi = 0;
    for (int i = 0; i < N; ++i) {
      V3 pc = T.map(Xw[i]);
      const double u = pc.x / pc.z * fx + cx, v = pc.y / pc.z * fy + cy;
      const double e0 = (P->obs[2 * i] + f[2 * i]) - u, e1 = (P->obs[2 * i + 1] + f[2 * i + 1]) - v;
      err[2 * i] = e0; err[2 * i + 1] = e1;
      const double c = e0 * (P->info_flow * e0) + e1 * (P->info_flow * e1);
      double r0, r1;
      hub.robustify(c, r0, r1);
      rchi += r0;
      const double p0 = f[2 * i] - P->flow[2 * i], p1 = f[2 * i + 1] - P->flow[2 * i + 1];
      errp[2 * i] = p0; errp[2 * i + 1] = p1;
      rchi += p0 * (P->info_prior * p0) + p1 * (P->info_prior * p1);
    }
    return rchi;
  }

  // buildSystem at the current estimate, using the errors of the preceding compute_errors()
  void build_system() {
    const double fx = P->K[0], fy = P->K[1];
    for (int i = 0; i < 36; ++i) Hpp[i] = 0;
    for (int i = 0; i < 6; ++i) bp[i] = 0;
    for (int i = 0; i < N; ++i) {
      V3 pc = T.map(Xw[i]);
      const double X = pc.x, Y = pc.y, Z = pc.z, Z2 = Z * Z;
      double J[12];  // 2x6
      J[0] = X * Y / Z2 * fx; J[1] = -(1 + (X * X / Z2)) * fx; J[2] = Y / Z * fx; J[3] = -1. / Z * fx; J[4] = 0; J[5] = X / Z2 * fx;
      J[6] = (1 + Y * Y / Z2) * fy; J[7] = -X * Y / Z2 * fy; J[8] = -X / Z * fy; J[9] = 0; J[10] = -1. / Z * fy; J[11] = Y / Z2 * fy;
      const double e0 = err[2 * i], e1 = err[2 * i + 1];
      const double c = e0 * (P->info_flow * e0) + e1 * (P->info_flow * e1);
      double r0, r1;
      hub.robustify(c, r0, r1);
      const double wo = r1 * P->info_flow;                                    // robustInformation = rho' * Omega
      const double or0 = -(P->info_flow * e0) * r1, or1 = -(P->info_flow * e1) * r1;   // omega_r
      // flow vertex (Xi): A = I
      double h00 = wo, h11 = wo, b0 = or0, b1 = or1;
      // Hpl^T block written as 6x2: B^T (wOmega) A
      for (int a = 0; a < 6; ++a) { B2[12 * i + 2 * a] = J[a] * wo; B2[12 * i + 2 * a + 1] = J[6 + a] * wo; }
      // pose vertex (Xj)
      for (int a = 0; a < 6; ++a) {
        bp[a] += J[a] * or0 + J[6 + a] * or1;
        for (int c2 = 0; c2 < 6; ++c2) Hpp[a * 6 + c2] += J[a] * wo * J[c2] + J[6 + a] * wo * J[6 + c2];
      }
      // EdgeFlowPrior (unary, no kernel): b -= Omega e ; A += Omega
      b0 -= P->info_prior * errp[2 * i]; b1 -= P->info_prior * errp[2 * i + 1];
      h00 += P->info_prior; h11 += P->info_prior;
      hl[4 * i] = h00; hl[4 * i + 1] = 0; hl[4 * i + 2] = 0; hl[4 * i + 3] = h11;
      bl[2 * i] = b0; bl[2 * i + 1] = b1;
    }
  }
  double max_diag() const {
    double m = 0;
    for (int j = 0; j < 6; ++j) m = std::max(std::fabs(Hpp[7 * j]), m);
    for (int i = 0; i < N; ++i) { m = std::max(std::fabs(hl[4 * i]), m); m = std::max(std::fabs(hl[4 * i + 3]), m); }
    return m;
  }

  // BlockSolver::solve with Schur.  Returns false when the 6x6 LDLT is not positive.
  bool solve(double lambda) {
    const bool Q = P->ref_quirks != 0;
    std::vector<double> Dinv(9 * (size_t)N);
    double Hs[36], coef[6];
    for (int i = 0; i < 36; ++i) Hs[i] = Hpp[i];
    for (int j = 0; j < 6; ++j) { Hs[7 * j] += lambda; coef[j] = 0; }
    for (int i = 0; i < N; ++i) {
      double* Di = &Dinv[9 * i];
      if (Q) {
        const double D3[9] = {hl[4 * i] + lambda, hl[4 * i + 3], 0, hl[4 * i + 1], lambda, 0, hl[4 * i + 2], 0, lambda};
        inv3(D3, Di);
      } else {
        const double D2[4] = {hl[4 * i] + lambda, hl[4 * i + 2], hl[4 * i + 1], hl[4 * i + 3] + lambda};
        double E[4];
        inv2(D2, E);
        Di[0] = E[0]; Di[1] = E[1]; Di[2] = 0; Di[3] = E[2]; Di[4] = E[3]; Di[5] = 0; Di[6] = 0; Di[7] = 0; Di[8] = 0;
      }
      // db = Dinv * (b0, b1, next b0 | 0)
      const double b2 = (Q && i + 1 < N) ? bl[2 * i + 2] : 0.0;
      const double db0 = (Di[0] * bl[2 * i] + Di[1] * bl[2 * i + 1]) + Di[2] * b2;
      const double db1 = (Di[3] * bl[2 * i] + Di[4] * bl[2 * i + 1]) + Di[5] * b2;
      const double* B = &B2[12 * i];
      for (int a = 0; a < 6; ++a) {
        coef[a] += B[2 * a] * db0 + B[2 * a + 1] * db1;              // Bb += Bi * db (third column of Bi is 0)
        const double bd0 = B[2 * a] * Di[0] + B[2 * a + 1] * Di[3];  // BDinv = Bi * Dinv, columns 0,1
        const double bd1 = B[2 * a] * Di[1] + B[2 * a + 1] * Di[4];
        for (int c2 = 0; c2 < 6; ++c2) Hs[a * 6 + c2] -= bd0 * B[2 * c2] + bd1 * B[2 * c2 + 1];
      }
    }
    double bs[6];
    for (int j = 0; j < 6; ++j) bs[j] = bp[j] - coef[j];
    LDLT6 ch;
    ch.compute(Hs);
    if (!ch.isPositive()) return false;
    ch.solve(bs, x.data());
    // back-substitution
    std::vector<double> cl(2 * (size_t)N + 1, 0.0);
    for (int i = 0; i < N; ++i) {
      const double* B = &B2[12 * i];
      double t0 = 0, t1 = 0;
      for (int a = 0; a < 6; ++a) { t0 += B[2 * a] * (-x[a]); t1 += B[2 * a + 1] * (-x[a]); }
      cl[2 * i] = bl[2 * i] + t0; cl[2 * i + 1] = bl[2 * i + 1] + t1;
    }
    double* xl = x.data() + 6;
    for (int i = 0; i < 2 * N + 1; ++i) xl[i] = 0;
    for (int i = 0; i < N; ++i) {
      const double* Di = &Dinv[9 * i];
      const double c0 = cl[2 * i], c1 = cl[2 * i + 1], c2 = Q ? cl[2 * i + 2] : 0.0;
      xl[2 * i] += (Di[0] * c0 + Di[1] * c1) + Di[2] * c2;
      xl[2 * i + 1] += (Di[3] * c0 + Di[4] * c1) + Di[5] * c2;
      if (Q) xl[2 * i + 2] += (Di[6] * c0 + Di[7] * c1) + Di[8] * c2;   // leaks into the next landmark / one past the end
    }
    return true;
  }
};

}  // namespace

extern "C" int vdo_oracle_flow2_optimize(const vdo_flow2_problem* p, double T_out[16], double* flow_out,
                                         uint8_t* inlier_out, vdo_lm_stats* st) {
  vdo_lm_stats local;
  if (!st) st = &local;
  std::memset(st, 0, sizeof(*st));
  const int N = p->n;
  if (N < 3) {   // nInitialCorrespondences<3 (Optimizer.cc:2449-2450, 2872-2873)
    for (int i = 0; i < 16; ++i) T_out[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return 0;
  }
  Flow2 S(p);
  double lambda = -1, ni = 2;
  int nBad = 0;
  const double tau = 1e-5, upper = 2. / 3., lower = 1. / 3.;
  const int maxTrials = 10;
  bool ok = true;
  const bool trace_rho = std::getenv("VDO_ORACLE_LM_TRACE") != nullptr;      // (debug: gain ratio of every trial on stderr)
  double chi2_check = 0, last_err_chi = S.compute_errors();
  st->initial_chi2 = last_err_chi;
  int it = 0;
  for (; it < p->max_iterations && ok; ++it) {
    last_err_chi = S.compute_errors();
    double currentChi = last_err_chi, tempChi = currentChi;
    const double iniChi = currentChi;
    S.build_system();
    if (it == 0) { lambda = tau * S.max_diag(); ni = 2; nBad = 0; }
    double rho = 0;
    int qmax = 0;
    do {
      const SE3Quat Tb = S.T;
      const std::vector<double> fb = S.f;                     // push()
      const bool ok2 = S.solve(lambda);
      S.T = SE3Quat::exp(S.x.data()).compose(S.T);            // VertexSE3Expmap::oplusImpl
      for (int i = 0; i < 2 * N; ++i) S.f[i] += S.x[6 + i];  // VertexSBAFlow::oplusImpl
      last_err_chi = tempChi = S.compute_errors();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < 6; ++j) scale += S.x[j] * (lambda * S.x[j] + S.bp[j]);
      for (int j = 0; j < 2 * N; ++j) scale += S.x[6 + j] * (lambda * S.x[6 + j] + S.bl[j]);
      scale += 1e-3;
      rho /= scale;
      if (trace_rho) std::fprintf(stderr, "[lm n=%d it=%d q=%d] rho %.6g lambda %.4g chi %.9g -> %.9g\n", N, it, qmax, rho, lambda, currentChi, tempChi);
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, upper);
        lambda *= std::max(lower, alpha); ni = 2; currentChi = tempChi;
      } else {
        lambda *= ni; ni *= 2;
        S.T = Tb; S.f = fb;                                   // pop()
      }
      ++qmax; ++st->total_trials;
    } while (rho < 0 && qmax < maxTrials);
    int result;
    if (qmax == maxTrials || rho == 0) result = 1;
    else {
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      result = nBad >= 3 ? 1 : 0;
    }
    ok = (result == 0);
    if (!ok) st->stop_reason = 1;
    if (chi2_check < last_err_chi && it > 0) { ok = false; st->stop_reason = 2; }
    chi2_check = last_err_chi;
    if (it < VDO_LM_MAX_TRACE) { st->chi2_trace[it] = last_err_chi; st->trials_trace[it] = qmax; }
  }
  st->iterations = it;
  st->final_lambda = lambda;
  st->final_chi2 = last_err_chi;
  // classification uses the edges' *stored* errors, i.e. those of the last evaluated trial
  // (Optimizer.cc:2470-2508): chi2 is cast to float and compared with the float gate.
  int nbad = 0;
  const float gate = (float)p->chi2_gate;
  for (int i = 0; i < N; ++i) {
    const double e0 = S.err[2 * i], e1 = S.err[2 * i + 1];
    const float chi2 = (float)(e0 * (p->info_flow * e0) + e1 * (p->info_flow * e1));
    const bool out = chi2 > gate;
    if (inlier_out) inlier_out[i] = out ? 0 : 1;
    nbad += out;
    if (flow_out) { flow_out[2 * i] = S.f[2 * i]; flow_out[2 * i + 1] = S.f[2 * i + 1]; }
  }
  S.T.toMatrix4(T_out);
  return N - nbad;
}

// ------------------------------------------------------------------------------------------------
// Non-joint pose refinement: Optimizer::PoseOptimizationNew (src/Optimizer.cc:2177-2331) and
// Optimizer::PoseOptimizationObjMot (:2544-2753).  One VertexSE3Expmap and n unary edges
//   EdgeSE3ProjectXYZOnlyPose       g2o/types/types_six_dof_expmap.h:151-179, .cpp:266-296
//   EdgeSE3ProjectXYZOnlyObjMotion  .h:214-245, .cpp:394-443
// No marginalised vertex, so BlockSolver::solve takes the non-Schur branch
// (g2o/core/block_solver.hpp:357-366): (Hpp + lambda I) x = b by the dense LDLT.
namespace {
struct PoseOnly {
  const vdo_pose_problem* P;
  int N;
  SE3Quat T;
  Huber hub;
  bool robust;
  double Hpp[36], bp[6], x[6];
  std::vector<double> err;

  explicit PoseOnly(const vdo_pose_problem* p) : P(p), N(p->n), robust(p->huber_delta > 0) {
    M3 R;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = p->T0[4 * i + j];
    T = SE3Quat::fromRt(R, v3(p->T0[3], p->T0[7], p->T0[11]));
    if (robust) hub.setDelta(p->huber_delta);
    err.resize(2 * (size_t)N);
    for (int i = 0; i < 6; ++i) x[i] = 0;
  }
  void project(const V3& pc, double& u, double& v) const {
    if (P->kind == 0) { u = pc.x / pc.z * P->K[0] + P->K[2]; v = pc.y / pc.z * P->K[1] + P->K[3]; }
    else {
      const double* M = P->P;
      const double m1 = M[0] * pc.x + M[1] * pc.y + M[2] * pc.z + M[3];
      const double m2 = M[4] * pc.x + M[5] * pc.y + M[6] * pc.z + M[7];
      const double m3 = M[8] * pc.x + M[9] * pc.y + M[10] * pc.z + M[11];
      const double inv = 1.0 / m3;
      u = m1 * inv; v = m2 * inv;
    }
  }
  double compute_errors() {
    double rchi = 0;
    for (int i = 0; i < N; ++i) {
      const V3 pc = T.map(v3(P->Xw[3 * i], P->Xw[3 * i + 1], P->Xw[3 * i + 2]));
      double u, v;
      project(pc, u, v);
      const double e0 = P->obs[2 * i] - u, e1 = P->obs[2 * i + 1] - v;
      err[2 * i] = e0; err[2 * i + 1] = e1;
      const double c = e0 * e0 + e1 * e1;
      double r0 = c, r1 = 1;
      if (robust) hub.robustify(c, r0, r1);
      rchi += r0;
    }
    return rchi;
  }
  void jacobian(const V3& pc, double* J) const {
    const double x_ = pc.x, y_ = pc.y, z_ = pc.z;
    if (P->kind == 0) {
      const double fx = P->K[0], fy = P->K[1];
      const double invz = 1.0 / z_, invz_2 = invz * invz;
      J[0] = x_ * y_ * invz_2 * fx; J[1] = -(1 + (x_ * x_ * invz_2)) * fx; J[2] = y_ * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = x_ * invz_2 * fx;
      J[6] = (1 + y_ * y_ * invz_2) * fy; J[7] = -x_ * y_ * invz_2 * fy; J[8] = -x_ * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = y_ * invz_2 * fy;
    } else {
      const double* M = P->P;
      const double m1 = M[0] * x_ + M[1] * y_ + M[2] * z_ + M[3];
      const double m2 = M[4] * x_ + M[5] * y_ + M[6] * z_ + M[7];
      const double m3 = M[8] * x_ + M[9] * y_ + M[10] * z_ + M[11];
      const double invm3 = 1.0 / m3, invm3_2 = invm3 * invm3;
      double t[6];
      t[0] = invm3_2 * (M[0] * m3 - M[8] * m1); t[1] = invm3_2 * (M[1] * m3 - M[9] * m1); t[2] = invm3_2 * (M[2] * m3 - M[10] * m1);
      t[3] = invm3_2 * (M[4] * m3 - M[8] * m2); t[4] = invm3_2 * (M[5] * m3 - M[9] * m2); t[5] = invm3_2 * (M[6] * m3 - M[10] * m2);
      for (int r = 0; r < 2; ++r) {
        const double* tr = t + 3 * r;
        J[6 * r + 0] = -1.0 * (y_ * tr[2] - z_ * tr[1]);
        J[6 * r + 1] = -1.0 * (z_ * tr[0] - x_ * tr[2]);
        J[6 * r + 2] = -1.0 * (x_ * tr[1] - y_ * tr[0]);
        J[6 * r + 3] = -1.0 * tr[0]; J[6 * r + 4] = -1.0 * tr[1]; J[6 * r + 5] = -1.0 * tr[2];
      }
    }
  }
  void build_system() {
    for (int i = 0; i < 36; ++i) Hpp[i] = 0;
    for (int i = 0; i < 6; ++i) bp[i] = 0;
    for (int i = 0; i < N; ++i) {
      const V3 pc = T.map(v3(P->Xw[3 * i], P->Xw[3 * i + 1], P->Xw[3 * i + 2]));
      double J[12];
      jacobian(pc, J);
      const double e0 = err[2 * i], e1 = err[2 * i + 1];
      double r0, r1 = 1;
      if (robust) hub.robustify(e0 * e0 + e1 * e1, r0, r1);
      // base_unary_edge.hpp:55-66:  b -= ((rho1 * A^T) * Omega) * e ;  A += (A^T * (rho1 Omega)) * A   (Omega = I2)
      for (int a = 0; a < 6; ++a) {
        bp[a] -= (r1 * J[a]) * e0 + (r1 * J[6 + a]) * e1;
        for (int c = 0; c < 6; ++c) Hpp[a * 6 + c] += (J[a] * r1) * J[c] + (J[6 + a] * r1) * J[6 + c];
      }
    }
  }
  bool solve(double lambda) {
    double Hs[36];
    for (int i = 0; i < 36; ++i) Hs[i] = Hpp[i];
    for (int j = 0; j < 6; ++j) Hs[7 * j] += lambda;
    LDLT6 ch;
    ch.compute(Hs);
    if (!ch.isPositive()) return false;
    ch.solve(bp, x);
    return true;
  }
};
}  // namespace

// KAT helpers for tests/test_ref_g2o.py (the reference's own g2o, compiled verbatim, on the other side): one EdgeSE3ProjectFlow2 + its EdgeFlowPrior at
// pose T16 and flow estimate flow_est - errors and Jacobians as compute_errors() / build_system() above form them (Jpose 2x6 row-major; the flow Jacobians are I2)
extern "C" void vdo_oracle_edge_flow2_jac(const double K4[4], const double Twl16[16], double depth, const double obs[2], const double flow_est[2], const double flow_meas[2],
                                          const double T16[16], double err2[2], double Jpose12[12], double errp2[2]) {
  vdo_flow2_problem q;
  std::memset(&q, 0, sizeof q);
  q.n = 1; q.obs = obs; q.flow = flow_meas; q.depth = &depth;
  for (int i = 0; i < 4; ++i) q.K[i] = K4[i];
  for (int i = 0; i < 16; ++i) { q.Twl[i] = Twl16[i]; q.T0[i] = T16[i]; }
  q.info_flow = 1.0; q.info_prior = 1.0; q.huber_delta = 1e30; q.chi2_gate = 1.0; q.max_iterations = 1; q.ref_quirks = 1;
  Flow2 S(&q);
  S.f[0] = flow_est[0]; S.f[1] = flow_est[1];
  S.compute_errors();
  err2[0] = S.err[0]; err2[1] = S.err[1]; errp2[0] = S.errp[0]; errp2[1] = S.errp[1];
  const double fx = K4[0], fy = K4[1];
  V3 pc = S.T.map(S.Xw[0]);
  const double X = pc.x, Y = pc.y, Z = pc.z, Z2 = Z * Z;
  double* J = Jpose12;                                     // (the expressions of build_system)
  J[0] = X * Y / Z2 * fx; J[1] = -(1 + (X * X / Z2)) * fx; J[2] = Y / Z * fx; J[3] = -1. / Z * fx; J[4] = 0; J[5] = X / Z2 * fx;
  J[6] = (1 + Y * Y / Z2) * fy; J[7] = -X * Y / Z2 * fy; J[8] = -X / Z * fy; J[9] = 0; J[10] = -1. / Z * fy; J[11] = Y / Z2 * fy;
}
extern "C" void vdo_oracle_huber(double delta, double e2, double rho2[2]) { Huber h; h.setDelta(delta); h.robustify(e2, rho2[0], rho2[1]); }
// VertexSE3Expmap::oplusImpl: exp(update) * estimate, the estimate given as a 4x4 through Converter::toSE3Quat's SE3Quat(R, t); u == NULL: the round trip alone
extern "C" void vdo_oracle_se3quat_oplus(const double T16[16], const double* u, double out16[16]) {
  M3 R;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = T16[4 * i + j];
  SE3Quat T = SE3Quat::fromRt(R, v3(T16[3], T16[7], T16[11]));
  if (u) T = SE3Quat::exp(u).compose(T);
  T.toMatrix4(out16);
}

extern "C" int vdo_oracle_edge_unary_jac(const vdo_pose_problem* p, const double* T16, const double* Xw, const double* obs, double* err2, double* J12) {
  // KAT helper: error and analytic Jacobian of one unary edge at pose T16 (4x4 row-major)
  vdo_pose_problem q = *p;
  q.n = 1; q.obs = obs; q.Xw = Xw;
  for (int i = 0; i < 16; ++i) q.T0[i] = T16[i];
  PoseOnly S(&q);
  S.compute_errors();
  err2[0] = S.err[0]; err2[1] = S.err[1];
  S.jacobian(S.T.map(v3(Xw[0], Xw[1], Xw[2])), J12);
  return 0;
}

extern "C" int vdo_oracle_pose_optimize(const vdo_pose_problem* p, double T_out[16], uint8_t* inlier_out, vdo_lm_stats* st) {
  vdo_lm_stats local;
  if (!st) st = &local;
  std::memset(st, 0, sizeof(*st));
  const int N = p->n;
  if (N < 3) {   // nInitialCorrespondences<3 (Optimizer.cc:2264-2265 returns 0; :2659-2660 returns eye)
    for (int i = 0; i < 16; ++i) T_out[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return 0;
  }
  PoseOnly S(p);
  double lambda = -1, ni = 2;
  int nBad = 0;
  const double tau = 1e-5, upper = 2. / 3., lower = 1. / 3.;
  const int maxTrials = 10;
  bool ok = true;
  double chi2_check = 0, last_err_chi = S.compute_errors();
  st->initial_chi2 = last_err_chi;
  int it = 0;
  for (; it < p->max_iterations && ok; ++it) {
    last_err_chi = S.compute_errors();
    double currentChi = last_err_chi, tempChi = currentChi;
    const double iniChi = currentChi;
    S.build_system();
    if (it == 0) {
      double m = 0;
      for (int j = 0; j < 6; ++j) m = std::max(std::fabs(S.Hpp[7 * j]), m);
      lambda = tau * m; ni = 2; nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const SE3Quat Tb = S.T;
      const bool ok2 = S.solve(lambda);
      S.T = SE3Quat::exp(S.x).compose(S.T);
      last_err_chi = tempChi = S.compute_errors();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < 6; ++j) scale += S.x[j] * (lambda * S.x[j] + S.bp[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, upper);
        lambda *= std::max(lower, alpha); ni = 2; currentChi = tempChi;
      } else {
        lambda *= ni; ni *= 2;
        S.T = Tb;
      }
      ++qmax; ++st->total_trials;
    } while (rho < 0 && qmax < maxTrials);
    int result;
    if (qmax == maxTrials || rho == 0) result = 1;
    else {
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      result = nBad >= 3 ? 1 : 0;
    }
    ok = (result == 0);
    if (!ok) st->stop_reason = 1;
    if (chi2_check < last_err_chi && it > 0) { ok = false; st->stop_reason = 2; }
    chi2_check = last_err_chi;
    if (it < VDO_LM_MAX_TRACE) { st->chi2_trace[it] = last_err_chi; st->trials_trace[it] = qmax; }
  }
  st->iterations = it;
  st->final_lambda = lambda;
  st->final_chi2 = last_err_chi;
  // classification on the stored errors of the last evaluated trial (Optimizer.cc:2284-2299, 2679-2694)
  int nbad = 0;
  const float gate = (float)p->chi2_gate;
  for (int i = 0; i < N; ++i) {
    const float chi2 = (float)(S.err[2 * i] * S.err[2 * i] + S.err[2 * i + 1] * S.err[2 * i + 1]);
    const bool out = chi2 > gate;
    if (inlier_out) inlier_out[i] = out ? 0 : 1;
    nbad += out;
  }
  S.T.toMatrix4(T_out);
  return N - nbad;
}
