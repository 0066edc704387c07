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
  const double *obs, *meas, *depth;     // [2n],[2n],[n]
  double *Xw, *fcur, *ftry, *err, *errp, *B2, *hl, *bl, *cl, *dinv, *xl;
  double* flow_out; unsigned char* inlier_out;
  vdo_flow2_result* results;
};

#ifdef F2_PROFILE
#define F2_TICK(slot) do { if (tid == 0) { const long long t_ = clock64(); s_prof[slot] += t_ - s_tprev; s_tprev = t_; } } while (0)
#else
#define F2_TICK(slot) do { } while (0)
#endif

__global__ __launch_bounds__(F2_THREADS) void k_flow2_lm(const Flow2Dev* __restrict__ probs, Flow2Arrays A) {
  const Flow2Dev P = probs[blockIdx.x];
  const int N = P.n, tid = threadIdx.x;
  const int64_t off = P.off;
  const double* __restrict__ obs = A.obs + 2 * off; const double* __restrict__ meas = A.meas + 2 * off; const double* __restrict__ depth = A.depth + off;
  double* __restrict__ Xw = A.Xw + 3 * off; double* fcur = A.fcur + 2 * off; double* ftry = A.ftry + 2 * off;
  double* __restrict__ err = A.err + 2 * off; double* __restrict__ errp = A.errp + 2 * off; double* __restrict__ B2 = A.B2 + 12 * off;
  double* __restrict__ hl = A.hl + off; double* __restrict__ bl = A.bl + 2 * off; double* __restrict__ cl = A.cl + 2 * off + blockIdx.x;
  double* __restrict__ dinv = A.dinv + 9 * off; double* __restrict__ xl = A.xl + 2 * off + blockIdx.x;
  vdo_flow2_result* res = A.results + blockIdx.x;

  __shared__ double s_scr[F2_WAVES * 28], s_red[28];
  __shared__ double s_wide[28 * (F2_THREADS + 1)];
  __shared__ SE3d s_T, s_Ttry;
  __shared__ double s_Hpp[36], s_bp[6], s_xp[6], s_Hs[36], s_bs[6], s_xs[6];
  __shared__ double s_lambda, s_rho;
  __shared__ int s_ctrl[4];   // [0] continue outer, [1] continue trial loop, [2] ok2, [3] accepted
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
  // ---- setup: Xw, flows, initial pose (Converter::toSE3Quat)
  for (int i = tid; i < N; i += F2_THREADS) {
    const double dz = depth[i];
    const double x = (obs[i] - cx) * dz / fx, y = (obs[N + i] - cy) * dz / fy;
    const double* W = P.Twl;
    Xw[i] = W[0] * x + W[1] * y + W[2] * dz + W[3];
    Xw[N + i] = W[4] * x + W[5] * y + W[6] * dz + W[7];
    Xw[2 * N + i] = W[8] * x + W[9] * y + W[10] * dz + W[11];
    fcur[i] = meas[i]; fcur[N + i] = meas[N + i];
    xl[i] = 0.0; xl[N + 1 + i] = 0.0;
  }
  if (tid < 6) s_xp[tid] = 0.0;
  if (tid == 0) {
    const double R[9] = {P.T0[0], P.T0[1], P.T0[2], P.T0[4], P.T0[5], P.T0[6], P.T0[8], P.T0[9], P.T0[10]};
    s_T.r = q_from_R(R);
    q_normalize_pos(s_T.r);
    s_T.t[0] = P.T0[3]; s_T.t[1] = P.T0[7]; s_T.t[2] = P.T0[11];
    s_ctrl[0] = 1;
  }
  __syncthreads();

  // errors at (T, f): writes err/errp, returns robust chi2 (block-wide)
  auto compute_errors = [&](const SE3d& T, const double* f) -> double {
    double part[1] = {0.0};
    for (int i = tid; i < N; i += F2_THREADS) {
      double pc[3];
      const double xw[3] = {Xw[i], Xw[N + i], Xw[2 * N + i]};
      q_rotate(T.r, xw, pc);
      pc[0] += T.t[0]; pc[1] += T.t[1]; pc[2] += T.t[2];
      const double u = pc[0] / pc[2] * fx + cx, v = pc[1] / pc[2] * fy + cy;
      const double e0 = (obs[i] + f[i]) - u, e1 = (obs[N + i] + f[N + i]) - v;
      err[i] = e0; err[N + i] = e1;
      const double c = e0 * (P.info_flow * e0) + e1 * (P.info_flow * e1);
      double r0, r1;
      huber_f2(c, P.huber_delta, P.huber_dsqr, r0, r1);
      const double p0 = f[i] - meas[i], p1 = f[N + i] - meas[N + i];
      errp[i] = p0; errp[N + i] = p1;
      part[0] += r0 + (p0 * (P.info_prior * p0) + p1 * (P.info_prior * p1));
    }
    block_reduce<1>(part, s_scr, s_red);
    const double r = s_red[0];
    __syncthreads();
    return r;
  };

  double lambda = -1, ni = 2;
  int nBad = 0, it = 0, total_trials = 0, stop_reason = 0;
  const double tau = 1e-5, upper = 2. / 3., lower = 1. / 3.;
  double chi2_check = 0;
  double last_err_chi = compute_errors(s_T, fcur);
  const double initial_chi2 = last_err_chi;
  bool err_valid = true;
  bool ok = true;
  for (; it < P.max_iterations && ok; ++it) {
    F2_TICK(7);
    // computeActiveErrors at the current estimate: err/errp and the chi2 are already those of this estimate
    // when the previous trial was accepted (or right after the initial evaluation) - same inputs, same code,
    // same bits - so the pass is only repeated after a rejected trial.
    if (!err_valid) last_err_chi = compute_errors(s_T, fcur);
    F2_TICK(0);
    double currentChi = last_err_chi, tempChi = currentChi;
    const double iniChi = currentChi;
    // ---- buildSystem
    {
      double acc[28];
#pragma unroll
      for (int i = 0; i < 28; ++i) acc[i] = 0.0;
      const SE3d T = s_T;
      for (int i = tid; i < N; i += F2_THREADS) {
        double pc[3];
        const double xw[3] = {Xw[i], Xw[N + i], Xw[2 * N + i]};
        q_rotate(T.r, xw, pc);
        const double X = pc[0] + T.t[0], Y = pc[1] + T.t[1], Z = pc[2] + T.t[2], Z2 = Z * Z;
        double J[12];
        J[0] = X * Y / Z2 * fx; J[1] = -(1 + (X * X / Z2)) * fx; J[2] = Y / Z * fx; J[3] = -1. / Z * fx; J[4] = 0; J[5] = X / Z2 * fx;
        J[6] = (1 + Y * Y / Z2) * fy; J[7] = -X * Y / Z2 * fy; J[8] = -X / Z * fy; J[9] = 0; J[10] = -1. / Z * fy; J[11] = Y / Z2 * fy;
        const double e0 = err[i], e1 = err[N + i];
        const double c = e0 * (P.info_flow * e0) + e1 * (P.info_flow * e1);
        double r0, r1;
        huber_f2(c, P.huber_delta, P.huber_dsqr, r0, r1);
        const double wo = r1 * P.info_flow;
        const double or0 = -(P.info_flow * e0) * r1, or1 = -(P.info_flow * e1) * r1;
#pragma unroll
        for (int a = 0; a < 6; ++a) { B2[(2 * a) * N + i] = J[a] * wo; B2[(2 * a + 1) * N + i] = J[6 + a] * wo; }
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          acc[21 + a] += J[a] * or0 + J[6 + a] * or1;
#pragma unroll
          for (int c2 = 0; c2 <= a; ++c2) acc[k++] += J[a] * wo * J[c2] + J[6 + a] * wo * J[6 + c2];   // lower triangle
        }
        const double h = wo + P.info_prior;
        hl[i] = h;            // Hll block = h * I2 (off-diagonals are exact zeros)
        bl[i] = or0 - P.info_prior * errp[i];
        bl[N + i] = or1 - P.info_prior * errp[N + i];
        acc[27] = fmax(acc[27], h);
      }
      // max needs its own reduction: do it through the same tree with fmax on slot 27
      double mx[1] = {acc[27]};
      acc[27] = 0;
      block_reduce_wide<28>(acc, s_wide, s_red);
      if (tid == 0) {
        int k = 0;
        for (int a = 0; a < 6; ++a) for (int c2 = 0; c2 <= a; ++c2) { s_Hpp[a * 6 + c2] = s_red[k]; s_Hpp[c2 * 6 + a] = s_red[k]; ++k; }
        for (int a = 0; a < 6; ++a) s_bp[a] = s_red[21 + a];
      }
      __syncthreads();
      if (it == 0) {
        // computeLambdaInit: max |H(j,j)| over pose and flow vertices
        double m = mx[0];
#pragma unroll
        for (int off2 = 32; off2 > 0; off2 >>= 1) m = fmax(m, __shfl_down(m, off2, 64));
        if ((tid & 63) == 0) s_scr[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
          double mm = s_scr[0];
          for (int w = 1; w < F2_WAVES; ++w) mm = fmax(mm, s_scr[w]);
          for (int j = 0; j < 6; ++j) mm = fmax(mm, fabs(s_Hpp[7 * j]));
          s_lambda = tau * mm;
        }
        __syncthreads();
        lambda = s_lambda; ni = 2; nBad = 0;
        __syncthreads();
      }
    }
    F2_TICK(1);
    double rho = 0;
    int qmax = 0;
    do {
      // ---- solve (Schur with the F3 aliasing)
      const bool Q = P.ref_quirks != 0;
      double acc[28];
#pragma unroll
      for (int i = 0; i < 28; ++i) acc[i] = 0.0;
      for (int i = tid; i < N; i += F2_THREADS) {
        double Di[9];
        if (Q) {
          const double hh = hl[i];
          const double D3[9] = {hh + lambda, hh, 0, 0.0, lambda, 0, 0.0, 0, lambda};
          inv3_dev(D3, Di);
        } else {
          const double a0 = hl[i] + lambda, a1 = 0.0, a2 = 0.0, a3 = hl[i] + lambda;
          const double id = 1.0 / (a0 * a3 - a1 * a2);
          Di[0] = a3 * id; Di[1] = -a1 * id; Di[2] = 0; Di[3] = -a2 * id; Di[4] = a0 * id; Di[5] = 0; Di[6] = 0; Di[7] = 0; Di[8] = 0;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) dinv[k * N + i] = Di[k];
        const double b2 = (Q && i + 1 < N) ? bl[i + 1] : 0.0;
        const double db0 = (Di[0] * bl[i] + Di[1] * bl[N + i]) + Di[2] * b2;
        const double db1 = (Di[3] * bl[i] + Di[4] * bl[N + i]) + Di[5] * b2;
        const double* B = B2 + i;
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          acc[21 + a] += B[(2 * a) * N] * db0 + B[(2 * a + 1) * N] * db1;
          const double bd0 = B[(2 * a) * N] * Di[0] + B[(2 * a + 1) * N] * Di[3];
          const double bd1 = B[(2 * a) * N] * Di[1] + B[(2 * a + 1) * N] * Di[4];
#pragma unroll
          for (int c2 = 0; c2 <= a; ++c2) acc[k++] += bd0 * B[(2 * c2) * N] + bd1 * B[(2 * c2 + 1) * N];   // lower triangle (LDLT reads only it)
        }
      }
      block_reduce_wide<28>(acc, s_wide, s_red);
      F2_TICK(2);
      if (tid == 0) {
        double* Hs = s_Hs; double* bs = s_bs; double* xs = s_xs;      // LDS, not scratch: the pivoted LDLT indexes dynamically
        for (int i = 0; i < 36; ++i) Hs[i] = s_Hpp[i];
        int k = 0;
        for (int a = 0; a < 6; ++a) for (int c2 = 0; c2 <= a; ++c2) { Hs[a * 6 + c2] -= s_red[k]; ++k; }
        for (int j = 0; j < 6; ++j) { Hs[7 * j] += lambda; bs[j] = s_bp[j] - s_red[21 + j]; }
        const bool ok2 = ldlt6_solve(Hs, bs, xs);
        s_ctrl[2] = ok2 ? 1 : 0;
        if (ok2) for (int j = 0; j < 6; ++j) s_xp[j] = xs[j];
        // (failed LDLT leaves x untouched in the reference; the trial is rejected anyway)
      }
      __syncthreads();
      F2_TICK(3);
      const bool ok2 = s_ctrl[2] != 0;
      double xp[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) xp[j] = s_xp[j];
      // ---- back-substitution: cl = bl - B^T xp
      for (int i = tid; i < N; i += F2_THREADS) {
        const double* B = B2 + i;
        double t0 = 0, t1 = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) { t0 += B[(2 * a) * N] * (-xp[a]); t1 += B[(2 * a + 1) * N] * (-xp[a]); }
        cl[i] = bl[i] + t0; cl[N + 1 + i] = bl[N + i] + t1;
      }
      if (tid == 0) cl[N] = 0.0;
      __syncthreads();
      F2_TICK(4);
      // xl[2i..2i+1] = (Dinv_i c_i)[0..1] (+ row 2 of landmark i-1), update, scale
      double sc[1] = {0.0};
      for (int i = tid; i < N; i += F2_THREADS) {
        const double* Di = dinv + i;
        const double c0 = cl[i], c1 = cl[N + 1 + i], c2 = Q ? cl[i + 1] : 0.0;
        double x0 = (Di[0] * c0 + Di[N] * c1) + Di[2 * N] * c2;
        const double x1 = (Di[3 * N] * c0 + Di[4 * N] * c1) + Di[5 * N] * c2;
        if (Q && i > 0) {
          const double* Dp = dinv + (i - 1);
          const double leak = (Dp[6 * N] * cl[i - 1] + Dp[7 * N] * cl[N + i]) + Dp[8 * N] * c0;
          x0 = leak + x0;            // landmark i-1 wrote first, then landmark i added its row 0
        }
        if (!ok2) { x0 = xl[i]; }   // stale x (reference keeps the previous content)
        const double x1e = ok2 ? x1 : xl[N + 1 + i];
        xl[i] = x0; xl[N + 1 + i] = x1e;
        ftry[i] = fcur[i] + x0; ftry[N + i] = fcur[N + i] + x1e;
        sc[0] += x0 * (lambda * x0 + bl[i]) + x1e * (lambda * x1e + bl[N + i]);
      }
      if (tid == 0) {
        s_Ttry = se3_exp_compose(s_xp, s_T);
        double s = 0;
        for (int j = 0; j < 6; ++j) s += s_xp[j] * (lambda * s_xp[j] + s_bp[j]);
        s_rho = s;    // pose part of computeScale
      }
      block_reduce<1>(sc, s_scr, s_red);
      const double scale = (s_rho + s_red[0]) + 1e-3;
      __syncthreads();
      F2_TICK(5);
      last_err_chi = tempChi = compute_errors(s_Ttry, ftry);
      F2_TICK(6);
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, upper);
        lambda *= fmax(lower, alpha); ni = 2; currentChi = tempChi; err_valid = true;
        { double* t_ = fcur; fcur = ftry; ftry = t_; }                       // discardTop(): accept (uniform pointer swap)
        if (tid == 0) s_T = s_Ttry;
      } else {
        lambda *= ni; ni *= 2; err_valid = false;                         // pop(): keep (s_T, fcur)
      }
      __syncthreads();
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
    // SE3Quat::to_homogeneous_matrix
    const Q4 q = s_T.r;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    double* T = res->T;
    T[0] = 1 - (tyy + tzz); T[1] = txy - twz; T[2] = txz + twy; T[3] = s_T.t[0];
    T[4] = txy + twz; T[5] = 1 - (txx + tzz); T[6] = tyz - twx; T[7] = s_T.t[1];
    T[8] = txz - twy; T[9] = tyz + twx; T[10] = 1 - (txx + tyy); T[11] = s_T.t[2];
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
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
  std::vector<double> obs(2 * T), meas(2 * T), dep(T);
  for (int k = 0; k < n_problems; ++k) {
    const vdo_flow2_problem& p = probs[k];
    if (p.n == 0) continue;
    // per-problem SoA planes [2][n]: consecutive lanes read consecutive doubles (512 B per wave load)
    double* po = obs.data() + 2 * b->offs[k]; double* pm = meas.data() + 2 * b->offs[k];
    for (int i = 0; i < p.n; ++i) { po[i] = p.obs[2 * i]; po[p.n + i] = p.obs[2 * i + 1]; pm[i] = p.flow[2 * i]; pm[p.n + i] = p.flow[2 * i + 1]; }
    std::memcpy(dep.data() + b->offs[k], p.depth, sizeof(double) * p.n);
  }
  double* d_obs = (double*)dev(16 * T); double* d_meas = (double*)dev(16 * T); double* d_dep = (double*)dev(8 * T);
  b->d_probs = (Flow2Dev*)dev(sizeof(Flow2Dev) * NP);
  Flow2Arrays& A = b->A;
  A.obs = d_obs; A.meas = d_meas; A.depth = d_dep;
  A.Xw = (double*)dev(24 * T); A.fcur = (double*)dev(16 * T); A.ftry = (double*)dev(16 * T);
  A.err = (double*)dev(16 * T); A.errp = (double*)dev(16 * T); A.B2 = (double*)dev(96 * T);
  A.hl = (double*)dev(8 * T); A.bl = (double*)dev(16 * T + 8); A.cl = (double*)dev(16 * T + 8 * NP + 8);
  A.dinv = (double*)dev(72 * T); A.xl = (double*)dev(16 * T + 8 * NP + 8);
  A.flow_out = (double*)dev(16 * T); A.inlier_out = (unsigned char*)dev(T);
  A.results = (vdo_flow2_result*)dev(sizeof(vdo_flow2_result) * NP);
  for (void* p : b->allocs) if (!p) { vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  if (!A.results || !d_obs) { vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  hipMemcpyAsync(d_obs, obs.data(), 16 * T, hipMemcpyHostToDevice, s);
  hipMemcpyAsync(d_meas, meas.data(), 16 * T, hipMemcpyHostToDevice, s);
  hipMemcpyAsync(d_dep, dep.data(), 8 * T, hipMemcpyHostToDevice, s);
  hipMemcpyAsync(b->d_probs, hp.data(), sizeof(Flow2Dev) * NP, hipMemcpyHostToDevice, s);
  hipMemsetAsync(A.xl, 0, 16 * T + 8 * NP + 8, s);
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
    hipMemcpyAsync((double*)b->A.obs + 2 * off, po, 16 * (size_t)n, hipMemcpyHostToDevice, s);
    hipMemcpyAsync((double*)b->A.meas + 2 * off, pm, 16 * (size_t)n, hipMemcpyHostToDevice, s);
    hipMemcpyAsync((double*)b->A.depth + off, pd, 8 * (size_t)n, hipMemcpyHostToDevice, s);
    d.max_iterations = p->max_iterations; d.ref_quirks = p->ref_quirks;
    std::memcpy(d.K, p->K, sizeof(d.K)); std::memcpy(d.Twl, p->Twl, sizeof(d.Twl)); std::memcpy(d.T0, p->T0, sizeof(d.T0));
    d.info_flow = p->info_flow; d.info_prior = p->info_prior; d.huber_delta = p->huber_delta;
    d.huber_dsqr = (double)(float)(p->huber_delta * p->huber_delta);
    d.chi2_gate = p->chi2_gate;
  }
  d.n = n; b->ns[k] = n;
  Flow2Dev* pst = (Flow2Dev*)(b->h_up + 5 * (size_t)std::max<int64_t>(b->total, 1)) + k;       // pinned copy of the descriptor
  *pst = d;
  hipMemcpyAsync(b->d_probs + k, pst, sizeof(Flow2Dev), hipMemcpyHostToDevice, s);
  return VDO_OK;
}

extern "C" int vdo_flow2_batch_run(vdo_flow2_batch* b) {
  if (!b) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(b->ctx);
  if (rc != VDO_OK) return rc;
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
