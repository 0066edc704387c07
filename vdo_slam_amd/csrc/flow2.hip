// Per-frame joint pose + optical-flow Levenberg–Marquardt on gfx950 (K16/K17 in SURVEY.md):
// Optimizer::PoseOptimizationFlow2Cam (src/Optimizer.cc:2333-2542) and
// Optimizer::PoseOptimizationFlow2 (:2755-2972) with g2o's
//   EdgeSE3ProjectFlow2 / EdgeFlowPrior   g2o/types/types_six_dof_expmap.cpp:772-775,805-845
//   VertexSE3Expmap / SE3Quat             g2o/types/types_six_dof_expmap.h:67-85, se3quat.h:41-301
//   BlockSolver_6_3 Schur + LinearSolverDense (Eigen LDLT)   g2o/core/block_solver.hpp:354-486
//   Levenberg + modified stop rules       g2o/core/optimization_algorithm_levenberg.cpp:61-164,
//                                         g2o/core/sparse_optimizer.cpp:354-443
//
// These problems are tiny (<=~1.2k correspondences, 6 unknowns after Schur) and latency-bound
// (SURVEY.md H4), so the WHOLE LM loop runs inside ONE persistent workgroup per problem —
// one launch for the camera problem, one launch for all objects of a frame (grid = #objects).
// Correspondences are strided over the 256 threads, per-landmark data stays in L2-resident
// scratch, pose-side sums use wave shuffles + one LDS stage, the 6x6 pivoted LDLT and the SE(3)
// update run on lane 0.  ref_quirks=1 reproduces the BlockSolver_6_3 / 2-DoF aliasing (F3)
// exactly as analysed in oracle/flow_oracle.cpp (product code does not use the oracle).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"
#include "lm_dev.hpp"

namespace vdo {

struct Flow2Dev {        // one problem, device pointers into the batch arrays
  int n, max_iterations, ref_quirks, pad;
  int64_t off;           // element offset of this problem inside the per-landmark arrays
  double K[4], Twl[16], T0[16];
  double info_flow, info_prior, huber_delta, huber_dsqr, chi2_gate;
};

struct Flow2Arrays {
  const double* in;      // per problem at 5*off: key points [2][n], measured flow [2][n], depth [n]   (SoA planes)
  double *Xw, *f0, *f1, *err, *B2a, *B2b, *hla, *hlb, *bla, *blb, *xl;
  double* flow_out; unsigned char* inlier_out;
  vdo_flow2_result* results;
};

#ifdef F2_PROFILE
#define F2_TICK(slot) do { if (tid == 0) { const long long t_ = clock64(); s_prof[slot] += t_ - s_tprev; s_tprev = t_; } } while (0)
#else
#define F2_TICK(slot) do { } while (0)
#endif

// Work per LM trial: two sweeps over the correspondences and one serial section -
//   (1) Schur sums  sum_i B_i D_i^-1 B_i^T, sum_i B_i D_i^-1 b_i  for this lambda        -> 27 block sums
//   (2) lane 0: 6x6 pivoted LDLT, SE3 exp, pose part of computeScale
//   (3) per correspondence: back-substitution, flow update, edge errors AT THE TRIAL POINT and - speculatively -
//       the linearisation there (Jacobians, B blocks, 27 pose sums) into the alternate buffers   -> 29 block sums
// An accepted trial makes the alternate buffers the current ones (g2o: the next iteration's computeActiveErrors +
// buildSystem see exactly this estimate: same inputs, same code, same bits); a rejected one leaves them untouched.
__global__ __launch_bounds__(F2_THREADS) void k_flow2_lm(const Flow2Dev* __restrict__ probs, Flow2Arrays A) {
  const Flow2Dev P = probs[blockIdx.x];
  const int N = P.n, tid = threadIdx.x;
  const int64_t off = P.off;
  const double* __restrict__ obs = A.in + 5 * off; const double* __restrict__ meas = obs + 2 * (size_t)N; const double* __restrict__ depth = obs + 4 * (size_t)N;
  double* __restrict__ Xw = A.Xw + 3 * off; double* fcur = A.f0 + 2 * off; double* ftry = A.f1 + 2 * off;
  double* __restrict__ err = A.err + 2 * off; double* __restrict__ xl = A.xl + 2 * off;
  double *Bc = A.B2a + 12 * off, *Bt = A.B2b + 12 * off, *hc = A.hla + off, *ht = A.hlb + off, *bc = A.bla + 2 * off, *bt = A.blb + 2 * off;   // current / trial linearisation
  vdo_flow2_result* res = A.results + blockIdx.x;

  __shared__ double s_scr[F2_WAVES * 4], s_red[32];
  __shared__ double s_wide[29 * (F2_THREADS + 1)];
  __shared__ SE3d s_T, s_Ttry;
  __shared__ double s_Hc[27], s_xp[6];   // s_Hc: Hpp (lower triangle, packed) + bp of the current linearisation
  __shared__ double s_lambda, s_rho;
  __shared__ int s_ctrl[4];   // [2] ok2
#ifdef F2_PROFILE
  __shared__ long long s_prof[16], s_tprev;
  if (tid == 0) { for (int i = 0; i < 16; ++i) s_prof[i] = 0; s_tprev = clock64(); }
#endif

  if (N < 3) {   // nInitialCorrespondences<3 -> identity, 0 inliers (Optimizer.cc:2449-2450, 2872-2873)
    if (tid < 16) res->T[tid] = (tid % 5 == 0) ? 1.0 : 0.0;
    if (tid < N) { A.inlier_out[off + tid] = 0; A.flow_out[2 * (off + tid)] = meas[tid]; A.flow_out[2 * (off + tid) + 1] = meas[N + tid]; }   // nothing optimised: flows stay as measured
    if (tid == 0) { res->n_inliers = 0; res->iterations = 0; res->trials = 0; res->stop_reason = 0; res->initial_chi2 = res->final_chi2 = res->final_lambda = 0; }
    return;
  }
  const double fx = P.K[0], fy = P.K[1], cx = P.K[2], cy = P.K[3];
  const bool Q = P.ref_quirks != 0;
  // ---- setup: Xw, flows, initial pose (Converter::toSE3Quat)
  for (int i = tid; i < N; i += F2_THREADS) {
    const double dz = depth[i];
    const double x = (obs[i] - cx) * dz / fx, y = (obs[N + i] - cy) * dz / fy;
    const double* W = P.Twl;
    Xw[i] = W[0] * x + W[1] * y + W[2] * dz + W[3];
    Xw[N + i] = W[4] * x + W[5] * y + W[6] * dz + W[7];
    Xw[2 * N + i] = W[8] * x + W[9] * y + W[10] * dz + W[11];
    fcur[i] = meas[i]; fcur[N + i] = meas[N + i];
    xl[i] = 0.0; xl[N + i] = 0.0;
  }
  if (tid < 6) s_xp[tid] = 0.0;
  if (tid == 0) {
    const double R[9] = {P.T0[0], P.T0[1], P.T0[2], P.T0[4], P.T0[5], P.T0[6], P.T0[8], P.T0[9], P.T0[10]};
    s_T.r = q_from_R(R);
    q_normalize_pos(s_T.r);
    s_T.t[0] = P.T0[3]; s_T.t[1] = P.T0[7]; s_T.t[2] = P.T0[11];
  }
  __syncthreads();

  // D_i^-1 of landmark i for this lambda.  ref_quirks (F3): BlockSolver_6_3 treats the 2-DoF flow vertex as 3-DoF, so the
  // block is D3 = [h+l, h, 0; 0, l, 0; 0, 0, l] (h = Hll diagonal, exact zeros elsewhere) and Eigen's cofactor inverse of it is
  //   [ l*l, -(h*l), 0;  0, (h+l)*l, 0;  0, 0, (h+l)*l ] * (1 / ((l*l)*(h+l)))
  // - every other cofactor is a difference of products with a literal zero, i.e. exactly +-0 (oracle/flow_oracle.cpp runs the
  // general 3x3 formula on the same block and gets the same bits).  d0 = Dinv(0,0), d1 = Dinv(0,1), d2 = Dinv(1,1) = Dinv(2,2).
  auto dinv_q = [&](const double hh, const double lam, double& d0, double& d1, double& d2) {
    const double a0 = hh + lam;
    const double C00 = lam * lam, hl_ = hh * lam, al = a0 * lam;
    const double id = 1.0 / (C00 * a0);
    d0 = C00 * id; d1 = (-hl_) * id; d2 = al * id;
  };
  auto dinv_of = [&](const double hh, const double lam, double* Di) {      // ref_quirks == 0: the proper 2x2 block
    const double a0 = hh + lam, a1 = 0.0, a2 = 0.0, a3 = hh + lam;
    const double id = 1.0 / (a0 * a3 - a1 * a2);
    Di[0] = a3 * id; Di[1] = -a1 * id; Di[2] = 0; Di[3] = -a2 * id; Di[4] = a0 * id; Di[5] = 0; Di[6] = 0; Di[7] = 0; Di[8] = 0;
  };

  // Sweep (3): TRIAL -> finish the solve for every correspondence (reads the current linearisation Br/hr/br and x_p),
  // then evaluate + linearise the edges at (T, f) into Bw/hw/bw.  Block sums -> s_red[0..26] (Hpp lower, bp), [27] robust chi2,
  // [28] landmark part of computeScale; returns the per-thread max of the Hll diagonal (computeLambdaInit).
  auto sweep = [&](auto trial_c, const double lam, const bool ok2, const double* __restrict__ Br, const double* __restrict__ hr, const double* __restrict__ br,
                   double* __restrict__ Bw, double* __restrict__ hw, double* __restrict__ bw, const double* fin, double* fout) -> double {
    constexpr bool TRIAL = decltype(trial_c)::value;
    const SE3d T = TRIAL ? s_Ttry : s_T;
    double xp[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) xp[j] = s_xp[j];
    double acc[29];
#pragma unroll
    for (int i = 0; i < 29; ++i) acc[i] = 0.0;
    double hmax = 0.0;
    for (int i = tid; i < N; i += F2_THREADS) {
      double f0v, f1v;
      if (TRIAL) {
        // back-substitution c = b_l - B^T x_p for this landmark
        const double* B = Br + i;
        double t0 = 0, t1 = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) { t0 += B[(2 * a) * N] * (-xp[a]); t1 += B[(2 * a + 1) * N] * (-xp[a]); }
        const double b0 = br[i], b1 = br[N + i];
        const double c0 = b0 + t0, c1 = b1 + t1;
        double x0, x1;
        if (Q) {
          // x[2i..2i+2] = D_i^-1 c_i with the aliased 3x3 block (dinv_q): rows 0/1 give this landmark's flow update, row 2 of
          // landmark i-1 (= its d2 * c0 of THIS landmark: the aliased third component) was written into slot 2i first
          double d0, d1, d2;
          dinv_q(hr[i], lam, d0, d1, d2);
          x0 = d0 * c0 + d1 * c1;
          x1 = d2 * c1;
          if (i > 0) {
            double p0_, p1_, p2_;
            dinv_q(hr[i - 1], lam, p0_, p1_, p2_);
            x0 = p2_ * c0 + x0;
          }
        } else {
          double Di[9];
          dinv_of(hr[i], lam, Di);
          x0 = (Di[0] * c0 + Di[1] * c1) + Di[2] * 0.0;
          x1 = (Di[3] * c0 + Di[4] * c1) + Di[5] * 0.0;
        }
        if (!ok2) x0 = xl[i];         // failed LDLT: stale x (the reference keeps the previous content); the trial is rejected anyway
        const double x1e = ok2 ? x1 : xl[N + i];
        xl[i] = x0; xl[N + i] = x1e;
        f0v = fin[i] + x0; f1v = fin[N + i] + x1e;
        fout[i] = f0v; fout[N + i] = f1v;
        acc[28] += x0 * (lam * x0 + b0) + x1e * (lam * x1e + b1);
      } else {
        f0v = fin[i]; f1v = fin[N + i];
      }
      // computeActiveErrors at (T, f)
      double pc[3];
      const double xw[3] = {Xw[i], Xw[N + i], Xw[2 * N + i]};
      q_rotate(T.r, xw, pc);
      const double X = pc[0] + T.t[0], Y = pc[1] + T.t[1], Z = pc[2] + T.t[2], Z2 = Z * Z;
      const double u = X / Z * fx + cx, v = Y / Z * fy + cy;
      const double e0 = (obs[i] + f0v) - u, e1 = (obs[N + i] + f1v) - v;
      err[i] = e0; err[N + i] = e1;
      const double c = e0 * (P.info_flow * e0) + e1 * (P.info_flow * e1);
      double r0, r1;
      huber_f2(c, P.huber_delta, P.huber_dsqr, r0, r1);
      const double p0 = f0v - meas[i], p1 = f1v - meas[N + i];
      acc[27] += r0 + (p0 * (P.info_prior * p0) + p1 * (P.info_prior * p1));
      // buildSystem at the same point
      double J[12];
      J[0] = X * Y / Z2 * fx; J[1] = -(1 + (X * X / Z2)) * fx; J[2] = Y / Z * fx; J[3] = -1. / Z * fx; J[4] = 0; J[5] = X / Z2 * fx;
      J[6] = (1 + Y * Y / Z2) * fy; J[7] = -X * Y / Z2 * fy; J[8] = -X / Z * fy; J[9] = 0; J[10] = -1. / Z * fy; J[11] = Y / Z2 * fy;
      const double wo = r1 * P.info_flow;
      const double or0 = -(P.info_flow * e0) * r1, or1 = -(P.info_flow * e1) * r1;
#pragma unroll
      for (int a = 0; a < 6; ++a) { Bw[(2 * a) * N + i] = J[a] * wo; Bw[(2 * a + 1) * N + i] = J[6 + a] * wo; }
      int k = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        acc[21 + a] += J[a] * or0 + J[6 + a] * or1;
#pragma unroll
        for (int c2 = 0; c2 <= a; ++c2) acc[k++] += J[a] * wo * J[c2] + J[6 + a] * wo * J[6 + c2];   // lower triangle
      }
      const double h = wo + P.info_prior;
      hw[i] = h;                // Hll block = h * I2 (off-diagonals are exact zeros)
      bw[i] = or0 - P.info_prior * p0;
      bw[N + i] = or1 - P.info_prior * p1;
      hmax = fmax(hmax, h);
    }
    block_reduce_wide<29>(acc, s_wide, s_red);
    return hmax;
  };

  double lambda = -1, ni = 2;
  int nBad = 0, it = 0, total_trials = 0, stop_reason = 0;
  const double tau = 1e-5, upper = 2. / 3., lower = 1. / 3.;
  double chi2_check = 0;
  // initial computeActiveErrors + buildSystem
  double hmax = sweep(std::false_type{}, 0.0, true, nullptr, nullptr, nullptr, Bc, hc, bc, fcur, nullptr);
  double last_err_chi = s_red[27];
  const double initial_chi2 = last_err_chi;
  if (tid < 27) s_Hc[tid] = s_red[tid];
  {
    // computeLambdaInit: max |H(j,j)| over pose and flow vertices
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) hmax = fmax(hmax, __shfl_down(hmax, off2, 64));
    if ((tid & 63) == 0) s_scr[tid >> 6] = hmax;
    __syncthreads();
    if (tid == 0) {
      double mm = s_scr[0];
      for (int w = 1; w < F2_WAVES; ++w) mm = fmax(mm, s_scr[w]);
      for (int j = 0; j < 6; ++j) mm = fmax(mm, fabs(s_Hc[j * (j + 3) / 2]));
      s_lambda = tau * mm;
    }
    __syncthreads();
    lambda = s_lambda; ni = 2; nBad = 0;
  }
  F2_TICK(4);
  bool built = true;
  bool ok = true;
  for (; it < P.max_iterations && ok; ++it) {
    // computeActiveErrors + buildSystem at the current estimate: already there after an accepted trial; repeated
    // only when the previous trial was rejected without ending the iteration loop (non-finite chi2)
    if (!built) {
      sweep(std::false_type{}, 0.0, true, nullptr, nullptr, nullptr, Bc, hc, bc, fcur, nullptr);
      last_err_chi = s_red[27];
      if (tid < 27) s_Hc[tid] = s_red[tid];
      __syncthreads();
      built = true;
    }
    double currentChi = last_err_chi, tempChi = currentChi;
    const double iniChi = currentChi;
    double rho = 0;
    int qmax = 0;
    do {
      // ---- (1) Schur sums for this lambda (with the F3 aliasing)
      {
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; ++i) acc[i] = 0.0;
        const double* __restrict__ Br = Bc; const double* __restrict__ hr = hc; const double* __restrict__ br = bc;
        for (int i = tid; i < N; i += F2_THREADS) {
          const double bl0 = br[i], bl1 = br[N + i];
          const double* B = Br + i;
          double Bv[12];
#pragma unroll
          for (int a = 0; a < 12; ++a) Bv[a] = B[a * N];
          if (Q) {
            double d0, d1, d2;
            dinv_q(hr[i], lambda, d0, d1, d2);
            const double db0 = d0 * bl0 + d1 * bl1, db1 = d2 * bl1;      // (the aliased third row/column only ever meets exact zeros)
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              acc[21 + a] += Bv[2 * a] * db0 + Bv[2 * a + 1] * db1;
              const double bd0 = Bv[2 * a] * d0;
              const double bd1 = Bv[2 * a] * d1 + Bv[2 * a + 1] * d2;
#pragma unroll
              for (int c2 = 0; c2 <= a; ++c2) acc[k++] += bd0 * Bv[2 * c2] + bd1 * Bv[2 * c2 + 1];   // lower triangle (LDLT reads only it)
            }
          } else {
            double Di[9];
            dinv_of(hr[i], lambda, Di);
            const double db0 = Di[0] * bl0 + Di[1] * bl1, db1 = Di[3] * bl0 + Di[4] * bl1;
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              acc[21 + a] += Bv[2 * a] * db0 + Bv[2 * a + 1] * db1;
              const double bd0 = Bv[2 * a] * Di[0] + Bv[2 * a + 1] * Di[3];
              const double bd1 = Bv[2 * a] * Di[1] + Bv[2 * a + 1] * Di[4];
#pragma unroll
              for (int c2 = 0; c2 <= a; ++c2) acc[k++] += bd0 * Bv[2 * c2] + bd1 * Bv[2 * c2 + 1];
            }
          }
        }
        block_reduce_wide<27>(acc, s_wide, s_red);
      }
      F2_TICK(0);
      // ---- (2) reduced 6x6 system, SE3 update, pose part of computeScale
      if (tid == 0) {
        double Hs[36], bs[6], xs[6];                                   // registers: every index below is a constant after unrolling
        {
          int k = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int c2 = 0; c2 <= a; ++c2) { Hs[c2 * 6 + a] = s_Hc[k]; Hs[a * 6 + c2] = s_Hc[k] - s_red[k]; ++k; }
          }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) { Hs[7 * j] += lambda; bs[j] = s_Hc[21 + j] - s_red[21 + j]; }
        const bool ok2 = ldlt6_solve_reg(Hs, bs, xs);
        s_ctrl[2] = ok2 ? 1 : 0;
        if (ok2) {
#pragma unroll
          for (int j = 0; j < 6; ++j) s_xp[j] = xs[j];
        }
        // (failed LDLT leaves x untouched in the reference; the trial is rejected anyway)
        s_Ttry = se3_exp_compose(s_xp, s_T);
        double s = 0;
        for (int j = 0; j < 6; ++j) s += s_xp[j] * (lambda * s_xp[j] + s_Hc[21 + j]);
        s_rho = s;    // pose part of computeScale
      }
      __syncthreads();
      F2_TICK(1);
      const bool ok2 = s_ctrl[2] != 0;
      // ---- (3) finish the solve per correspondence, errors + speculative linearisation at the trial point
      sweep(std::true_type{}, lambda, ok2, Bc, hc, bc, Bt, ht, bt, fcur, ftry);
      last_err_chi = tempChi = s_red[27];
      const double scale = (s_rho + s_red[28]) + 1e-3;
      F2_TICK(2);
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, upper);
        lambda *= fmax(lower, alpha); ni = 2; currentChi = tempChi; built = true;
        { double* t_ = fcur; fcur = ftry; ftry = t_; }                       // discardTop(): accept (uniform pointer swaps)
        { double* t_ = Bc; Bc = Bt; Bt = t_; t_ = hc; hc = ht; ht = t_; t_ = bc; bc = bt; bt = t_; }
        if (tid < 27) s_Hc[tid] = s_red[tid];
        if (tid == 32) s_T = s_Ttry;
      } else {
        lambda *= ni; ni *= 2; built = false;                               // pop(): keep (s_T, fcur) and their linearisation
      }
      __syncthreads();
      F2_TICK(3);
      ++qmax; ++total_trials;
    } while (rho < 0 && qmax < 10);
    int result;
    if (qmax == 10 || rho == 0) result = 1;
    else {
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      result = nBad >= 3 ? 1 : 0;
    }
    ok = (result == 0);
    if (!ok) stop_reason = 1;
    if (chi2_check < last_err_chi && it > 0) { ok = false; stop_reason = 2; }
    chi2_check = last_err_chi;
  }
  // ---- classification on the stored errors of the last evaluated trial (Optimizer.cc:2470-2508)
  double cnt[1] = {0.0};
  const float gate = (float)P.chi2_gate;
  for (int i = tid; i < N; i += F2_THREADS) {
    const double e0 = err[i], e1 = err[N + i];
    const float chi2 = (float)(e0 * (P.info_flow * e0) + e1 * (P.info_flow * e1));
    const bool outl = chi2 > gate;
    A.inlier_out[off + i] = outl ? 0 : 1;
    cnt[0] += outl ? 0.0 : 1.0;
    A.flow_out[2 * (off + i)] = fcur[i];
    A.flow_out[2 * (off + i) + 1] = fcur[N + i];
  }
  block_reduce<1>(cnt, s_scr, s_red);
  if (tid == 0) {
    se3_to_matrix(s_T, res->T);
    res->n_inliers = (int)(s_red[0] + 0.5);
    res->iterations = it; res->trials = total_trials; res->stop_reason = stop_reason;
    res->initial_chi2 = initial_chi2; res->final_chi2 = last_err_chi; res->final_lambda = lambda;
#ifdef F2_PROFILE
    for (int i = 0; i < 8; ++i) res->T[i] = (double)s_prof[i];     // cycles per phase instead of the pose (debug build only)
#endif
  }
}

}  // namespace vdo

using namespace vdo;

struct vdo_flow2_batch {
  vdo_ctx* ctx = nullptr;
  int n_problems = 0;
  int64_t total = 0;
  std::vector<void*> allocs;
  Flow2Dev* d_probs = nullptr;
  Flow2Arrays A{};
  std::vector<int64_t> offs;
  std::vector<int> ns;
  char* h_pin = nullptr;          // pinned staging of the results: [results NP][flow_out 2T doubles][inlier_out T bytes]
  double* h_up = nullptr;         // pinned staging of in-place problem updates: 5 doubles per point of capacity
  std::vector<Flow2Dev> hp;       // host mirror of d_probs
  std::vector<int> caps;          // capacity (points) of every problem slot
  bool probs_dirty = false;       // host mirror changed since the last upload of d_probs
};

extern "C" int vdo_flow2_batch_destroy(vdo_flow2_batch* b) {
  if (!b) return VDO_OK;
  if (b->ctx) ctx_bind(b->ctx);
  for (void* p : b->allocs) hipFree(p);
  if (b->h_pin) hipHostFree(b->h_pin);
  if (b->h_up) hipHostFree(b->h_up);
  delete b;
  return VDO_OK;
}

extern "C" int vdo_flow2_batch_create(vdo_ctx* ctx, int n_problems, const vdo_flow2_problem* probs, vdo_flow2_batch** out) {
  if (!ctx || !probs || !out || n_problems <= 0) return set_error(VDO_ERR_INVALID, "vdo_flow2_batch_create: bad argument");
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  vdo_flow2_batch* b = new vdo_flow2_batch();
  b->ctx = ctx; b->n_problems = n_problems;
  std::vector<Flow2Dev> hp(n_problems);
  int64_t total = 0;
  for (int k = 0; k < n_problems; ++k) {
    const vdo_flow2_problem& p = probs[k];
    if (p.n < 0 || (p.n > 0 && (!p.obs || !p.flow || !p.depth))) { delete b; return set_error(VDO_ERR_INVALID, "flow2 problem %d: null input", k); }
    Flow2Dev& d = hp[k];
    d.n = p.n; d.max_iterations = p.max_iterations; d.ref_quirks = p.ref_quirks; d.pad = 0; d.off = total;
    std::memcpy(d.K, p.K, sizeof(d.K)); std::memcpy(d.Twl, p.Twl, sizeof(d.Twl)); std::memcpy(d.T0, p.T0, sizeof(d.T0));
    d.info_flow = p.info_flow; d.info_prior = p.info_prior; d.huber_delta = p.huber_delta;
    d.huber_dsqr = (double)(float)(p.huber_delta * p.huber_delta);     // float member, robust_kernel_impl.h:84
    d.chi2_gate = p.chi2_gate;
    b->offs.push_back(total); b->ns.push_back(p.n);
    total += p.n;
  }
  b->total = total;
  hipStream_t s = ctx->stream;
  auto dev = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (bytes == 0) bytes = 8;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    b->allocs.push_back(p);
    return p;
  };
  const size_t T = (size_t)total, NP = (size_t)n_problems;
  std::vector<double> in(5 * T);
  for (int k = 0; k < n_problems; ++k) {
    const vdo_flow2_problem& p = probs[k];
    if (p.n == 0) continue;
    // per-problem SoA planes: key points [2][n], measured flow [2][n], depth [n]: consecutive lanes read consecutive doubles
    double* po = in.data() + 5 * b->offs[k]; double* pm = po + 2 * (size_t)p.n;
    for (int i = 0; i < p.n; ++i) { po[i] = p.obs[2 * i]; po[p.n + i] = p.obs[2 * i + 1]; pm[i] = p.flow[2 * i]; pm[p.n + i] = p.flow[2 * i + 1]; }
    std::memcpy(po + 4 * (size_t)p.n, p.depth, sizeof(double) * p.n);
  }
  double* d_in = (double*)dev(40 * T);
  b->d_probs = (Flow2Dev*)dev(sizeof(Flow2Dev) * NP);
  Flow2Arrays& A = b->A;
  A.in = d_in;
  A.Xw = (double*)dev(24 * T); A.f0 = (double*)dev(16 * T); A.f1 = (double*)dev(16 * T);
  A.err = (double*)dev(16 * T); A.B2a = (double*)dev(96 * T); A.B2b = (double*)dev(96 * T);
  A.hla = (double*)dev(8 * T); A.hlb = (double*)dev(8 * T); A.bla = (double*)dev(16 * T); A.blb = (double*)dev(16 * T);
  A.xl = (double*)dev(16 * T);
  A.flow_out = (double*)dev(16 * T); A.inlier_out = (unsigned char*)dev(T);
  A.results = (vdo_flow2_result*)dev(sizeof(vdo_flow2_result) * NP);
  for (void* p : b->allocs) if (!p) { vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  if (!A.results || !d_in) { vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  hipMemcpyAsync(d_in, in.data(), 40 * T, hipMemcpyHostToDevice, s);
  hipMemcpyAsync(b->d_probs, hp.data(), sizeof(Flow2Dev) * NP, hipMemcpyHostToDevice, s);
  if (hipHostMalloc((void**)&b->h_pin, sizeof(vdo_flow2_result) * NP + 16 * T + T + 64) != hipSuccess) { b->h_pin = nullptr; vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_OOM, "hipHostMalloc failed"); }
  if (hipStreamSynchronize(s) != hipSuccess) { vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_NO_DEVICE, "flow2 upload failed"); }
  b->hp = hp; b->caps = b->ns;
  *out = b;
  return VDO_OK;
}

// Slots for problems that are (re)defined every frame: capacity[k] points each, initially empty.
extern "C" int vdo_flow2_batch_reserve(vdo_ctx* ctx, int n_problems, const int32_t* capacity, vdo_flow2_batch** out) {
  if (!ctx || !capacity || !out || n_problems <= 0) return set_error(VDO_ERR_INVALID, "vdo_flow2_batch_reserve: bad argument");
  int64_t tot = 0;
  for (int k = 0; k < n_problems; ++k) { if (capacity[k] < 0) return set_error(VDO_ERR_INVALID, "negative capacity"); tot += capacity[k]; }
  std::vector<double> zeros(2 * (size_t)std::max<int64_t>(1, *std::max_element(capacity, capacity + n_problems)), 0.0);
  std::vector<vdo_flow2_problem> dummy(n_problems);
  for (int k = 0; k < n_problems; ++k) {
    std::memset(&dummy[k], 0, sizeof(vdo_flow2_problem));
    dummy[k].n = capacity[k]; dummy[k].obs = zeros.data(); dummy[k].flow = zeros.data(); dummy[k].depth = zeros.data();
  }
  int rc = vdo_flow2_batch_create(ctx, n_problems, dummy.data(), out);
  if (rc != VDO_OK) return rc;
  vdo_flow2_batch* b = *out;
  if (hipHostMalloc((void**)&b->h_up, sizeof(double) * 5 * (size_t)std::max<int64_t>(tot, 1) + sizeof(Flow2Dev) * (size_t)n_problems) != hipSuccess) { b->h_up = nullptr; vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_OOM, "hipHostMalloc failed"); }
  for (int k = 0; k < n_problems; ++k) { b->hp[k].n = 0; b->ns[k] = 0; }
  hipMemcpyAsync(b->d_probs, b->hp.data(), sizeof(Flow2Dev) * (size_t)n_problems, hipMemcpyHostToDevice, ctx->stream);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) { vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_NO_DEVICE, "flow2 reserve failed"); }
  return VDO_OK;
}

// (Re)define problem k of a reserved batch: inputs go through pinned staging into the slot's HBM arrays,
// stream-ordered, no synchronisation (the staging of slot k is reused by the next vdo_flow2_batch_set of slot k:
// run + fetch in between, as a frame does).  p == NULL empties the slot.
extern "C" int vdo_flow2_batch_set(vdo_flow2_batch* b, int k, const vdo_flow2_problem* p) {
  if (!b || k < 0 || k >= b->n_problems || !b->h_up) return set_error(VDO_ERR_INVALID, "vdo_flow2_batch_set: bad argument / batch not created by vdo_flow2_batch_reserve");
  int rc = ctx_bind(b->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = b->ctx->stream;
  Flow2Dev& d = b->hp[k];
  const int n = p ? p->n : 0;
  if (n < 0 || n > b->caps[k] || (n > 0 && (!p->obs || !p->flow || !p->depth))) return set_error(VDO_ERR_INVALID, "vdo_flow2_batch_set: %d points do not fit slot %d (capacity %d)", n, k, b->caps[k]);
  if (n) {
    const int64_t off = b->offs[k];
    double* st = b->h_up + 5 * off;
    double *po = st, *pm = st + 2 * (size_t)n, *pd = st + 4 * (size_t)n;
    for (int i = 0; i < n; ++i) { po[i] = p->obs[2 * i]; po[n + i] = p->obs[2 * i + 1]; pm[i] = p->flow[2 * i]; pm[n + i] = p->flow[2 * i + 1]; pd[i] = p->depth[i]; }
    hipMemcpyAsync((double*)b->A.in + 5 * off, st, 40 * (size_t)n, hipMemcpyHostToDevice, s);      // one copy: the slot has the staging's layout
    d.max_iterations = p->max_iterations; d.ref_quirks = p->ref_quirks;
    std::memcpy(d.K, p->K, sizeof(d.K)); std::memcpy(d.Twl, p->Twl, sizeof(d.Twl)); std::memcpy(d.T0, p->T0, sizeof(d.T0));
    d.info_flow = p->info_flow; d.info_prior = p->info_prior; d.huber_delta = p->huber_delta;
    d.huber_dsqr = (double)(float)(p->huber_delta * p->huber_delta);
    d.chi2_gate = p->chi2_gate;
  }
  d.n = n; b->ns[k] = n;
  b->probs_dirty = true;                 // the descriptors go up in one copy at the next run
  return VDO_OK;
}

extern "C" int vdo_flow2_batch_run(vdo_flow2_batch* b) {
  if (!b) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(b->ctx);
  if (rc != VDO_OK) return rc;
  if (b->probs_dirty) {
    Flow2Dev* pst = (Flow2Dev*)(b->h_up + 5 * (size_t)std::max<int64_t>(b->total, 1));           // pinned copy of the descriptors
    std::memcpy(pst, b->hp.data(), sizeof(Flow2Dev) * (size_t)b->n_problems);
    hipMemcpyAsync(b->d_probs, pst, sizeof(Flow2Dev) * (size_t)b->n_problems, hipMemcpyHostToDevice, b->ctx->stream);
    b->probs_dirty = false;
  }
  hipLaunchKernelGGL(k_flow2_lm, dim3(b->n_problems), dim3(F2_THREADS), 0, b->ctx->stream, (const Flow2Dev*)b->d_probs, b->A);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "k_flow2_lm launch: %s", hipGetErrorString(e));
  return VDO_OK;
}

extern "C" int vdo_flow2_batch_fetch(vdo_flow2_batch* b, vdo_flow2_result* results, double** flow_out, uint8_t** inlier_out) {
  if (!b || !results) return set_error(VDO_ERR_INVALID, "null argument");
  int rc = ctx_bind(b->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = b->ctx->stream;
  // everything through the pinned block (pageable D2H copies cost ~50-100 us each): 1-3 copies, one sync, then memcpy
  const size_t NP = (size_t)b->n_problems, T = (size_t)b->total;
  vdo_flow2_result* pr = (vdo_flow2_result*)b->h_pin;
  double* pf = (double*)(b->h_pin + sizeof(vdo_flow2_result) * NP);
  uint8_t* pi = (uint8_t*)(pf + 2 * T);
  hipMemcpyAsync(pr, b->A.results, sizeof(vdo_flow2_result) * NP, hipMemcpyDeviceToHost, s);
  if (flow_out && T) hipMemcpyAsync(pf, b->A.flow_out, 16 * T, hipMemcpyDeviceToHost, s);
  if (inlier_out && T) hipMemcpyAsync(pi, b->A.inlier_out, T, hipMemcpyDeviceToHost, s);
  hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "flow2 fetch: %s", hipGetErrorString(e));
  std::memcpy(results, pr, sizeof(vdo_flow2_result) * NP);
  for (int k = 0; k < b->n_problems; ++k) {
    if (b->ns[k] == 0) continue;
    if (flow_out && flow_out[k]) std::memcpy(flow_out[k], pf + 2 * b->offs[k], sizeof(double) * 2 * b->ns[k]);
    if (inlier_out && inlier_out[k]) std::memcpy(inlier_out[k], pi + b->offs[k], (size_t)b->ns[k]);
  }
  return VDO_OK;
}

extern "C" int vdo_flow2_optimize(vdo_ctx* ctx, const vdo_flow2_problem* p, vdo_flow2_result* result, double* flow_out, uint8_t* inlier_out) {
  vdo_flow2_batch* b = nullptr;
  int rc = vdo_flow2_batch_create(ctx, 1, p, &b);
  if (rc != VDO_OK) return rc;
  rc = vdo_flow2_batch_run(b);
  if (rc == VDO_OK) rc = vdo_flow2_batch_fetch(b, result, &flow_out, &inlier_out);
  vdo_flow2_batch_destroy(b);
  return rc;
}
