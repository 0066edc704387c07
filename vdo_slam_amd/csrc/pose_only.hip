// Non-joint per-frame pose refinement on gfx950: Optimizer::PoseOptimizationNew
// (reference src/Optimizer.cc:2177-2331) and Optimizer::PoseOptimizationObjMot (:2544-2753) with
//   EdgeSE3ProjectXYZOnlyPose        g2o/types/types_six_dof_expmap.h:151-179, .cpp:266-296
//   EdgeSE3ProjectXYZOnlyObjMotion   .h:214-245, .cpp:394-443
//   BaseUnaryEdge::constructQuadraticForm   g2o/core/base_unary_edge.hpp:43-72
//   BlockSolver::solve, non-Schur branch    g2o/core/block_solver.hpp:357-366 (dense pivoted LDLT)
//   Levenberg + the modified stop rules     as in flow2.hip
// Same execution model as flow2.hip: the whole LM loop of one problem inside one persistent
// 256-thread workgroup, a batch of problems (all objects of a frame) per launch.  27 running
// sums per linearisation (21 Hpp lower-triangle + 6 b), one block reduction, LDLT on lane 0.
#include <cstring>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"
#include "lm_dev.hpp"

namespace vdo {

struct PoseDev {
  int n, kind, max_iterations, robust;
  int64_t off;
  double K[4], P[12], T0[16];
  double huber_delta, huber_dsqr, chi2_gate;
};

struct PoseArrays {
  const double *obs, *Xw;          // [2n], [3n]
  double* err;                     // [2n]
  unsigned char* inlier_out;
  vdo_flow2_result* results;
};

__device__ __forceinline__ void pose_project(const PoseDev& P, const double* pc, double& u, double& v) {
  if (P.kind == 0) { u = pc[0] / pc[2] * P.K[0] + P.K[2]; v = pc[1] / pc[2] * P.K[1] + P.K[3]; }
  else {
    const double* M = P.P;
    const double m1 = M[0] * pc[0] + M[1] * pc[1] + M[2] * pc[2] + M[3];
    const double m2 = M[4] * pc[0] + M[5] * pc[1] + M[6] * pc[2] + M[7];
    const double m3 = M[8] * pc[0] + M[9] * pc[1] + M[10] * pc[2] + M[11];
    const double inv = 1.0 / m3;
    u = m1 * inv; v = m2 * inv;
  }
}

__device__ __forceinline__ void pose_jacobian(const PoseDev& P, const double* pc, double* J) {
  const double x = pc[0], y = pc[1], z = pc[2];
  if (P.kind == 0) {
    const double fx = P.K[0], fy = P.K[1];
    const double invz = 1.0 / z, invz_2 = invz * invz;
    J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
    J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
  } else {
    const double* M = P.P;
    const double m1 = M[0] * x + M[1] * y + M[2] * z + M[3];
    const double m2 = M[4] * x + M[5] * y + M[6] * z + M[7];
    const double m3 = M[8] * x + M[9] * y + M[10] * z + M[11];
    const double invm3 = 1.0 / m3, invm3_2 = invm3 * invm3;
    double t[6];
    t[0] = invm3_2 * (M[0] * m3 - M[8] * m1); t[1] = invm3_2 * (M[1] * m3 - M[9] * m1); t[2] = invm3_2 * (M[2] * m3 - M[10] * m1);
    t[3] = invm3_2 * (M[4] * m3 - M[8] * m2); t[4] = invm3_2 * (M[5] * m3 - M[9] * m2); t[5] = invm3_2 * (M[6] * m3 - M[10] * m2);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double* tr = t + 3 * r;
      J[6 * r + 0] = -1.0 * (y * tr[2] - z * tr[1]);
      J[6 * r + 1] = -1.0 * (z * tr[0] - x * tr[2]);
      J[6 * r + 2] = -1.0 * (x * tr[1] - y * tr[0]);
      J[6 * r + 3] = -1.0 * tr[0]; J[6 * r + 4] = -1.0 * tr[1]; J[6 * r + 5] = -1.0 * tr[2];
    }
  }
}

__global__ __launch_bounds__(F2_THREADS) void k_pose_lm(const PoseDev* __restrict__ probs, PoseArrays A) {
  const PoseDev P = probs[blockIdx.x];
  const int N = P.n, tid = threadIdx.x;
  const double* obs = A.obs + 2 * P.off; const double* Xw = A.Xw + 3 * P.off;
  double* err = A.err + 2 * P.off;
  vdo_flow2_result* res = A.results + blockIdx.x;

  __shared__ double s_scr[F2_WAVES * 27], s_red[27];
  __shared__ double s_wpart[F2_WAVES * 32];
  __shared__ SE3d s_T, s_Ttry;
  __shared__ double s_Hpp[36], s_bp[6], s_xp[6];
  __shared__ double s_lambda, s_scale;
  __shared__ int s_ok2;

  if (N < 3) {   // nInitialCorrespondences<3 (Optimizer.cc:2264-2265, 2659-2660)
    if (tid < 16) res->T[tid] = (tid % 5 == 0) ? 1.0 : 0.0;
    if (tid < N) A.inlier_out[P.off + tid] = 0;
    if (tid == 0) { res->n_inliers = 0; res->iterations = 0; res->trials = 0; res->stop_reason = 0; res->initial_chi2 = res->final_chi2 = res->final_lambda = 0; }
    return;
  }
  if (tid < 6) s_xp[tid] = 0.0;
  if (tid == 0) {
    const double R[9] = {P.T0[0], P.T0[1], P.T0[2], P.T0[4], P.T0[5], P.T0[6], P.T0[8], P.T0[9], P.T0[10]};
    s_T.r = q_from_R(R);
    q_normalize_pos(s_T.r);
    s_T.t[0] = P.T0[3]; s_T.t[1] = P.T0[7]; s_T.t[2] = P.T0[11];
  }
  __syncthreads();

  auto compute_errors = [&](const SE3d& T) -> double {
    double part[1] = {0.0};
    for (int i = tid; i < N; i += F2_THREADS) {
      double pc[3];
      q_rotate(T.r, Xw + 3 * i, pc);
      pc[0] += T.t[0]; pc[1] += T.t[1]; pc[2] += T.t[2];
      double u, v;
      pose_project(P, pc, u, v);
      const double e0 = obs[2 * i] - u, e1 = obs[2 * i + 1] - v;
      err[2 * i] = e0; err[2 * i + 1] = e1;
      const double c = e0 * e0 + e1 * e1;
      double r0 = c, r1 = 1.0;
      if (P.robust) huber_f2(c, P.huber_delta, P.huber_dsqr, r0, r1);
      part[0] += r0;
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
  double last_err_chi = compute_errors(s_T);
  const double initial_chi2 = last_err_chi;
  bool ok = true, err_valid = true;
  for (; it < P.max_iterations && ok; ++it) {
    if (!err_valid) last_err_chi = compute_errors(s_T);     // stored errors already belong to this estimate after an accepted trial
    double currentChi = last_err_chi, tempChi = currentChi;
    const double iniChi = currentChi;
    {   // ---- buildSystem
      double acc[27];
#pragma unroll
      for (int i = 0; i < 27; ++i) acc[i] = 0.0;
      const SE3d T = s_T;
      for (int i = tid; i < N; i += F2_THREADS) {
        double pc[3], J[12];
        q_rotate(T.r, Xw + 3 * i, pc);
        pc[0] += T.t[0]; pc[1] += T.t[1]; pc[2] += T.t[2];
        pose_jacobian(P, pc, J);
        const double e0 = err[2 * i], e1 = err[2 * i + 1];
        double r0, r1 = 1.0;
        if (P.robust) huber_f2(e0 * e0 + e1 * e1, P.huber_delta, P.huber_dsqr, r0, r1);
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          acc[21 + a] -= (r1 * J[a]) * e0 + (r1 * J[6 + a]) * e1;
#pragma unroll
          for (int c2 = 0; c2 <= a; ++c2) acc[k++] += (J[a] * r1) * J[c2] + (J[6 + a] * r1) * J[6 + c2];
        }
      }
      block_reduce_bfly<27>(acc, s_wpart, s_red);
      if (tid == 0) {
        int k = 0;
        double mm = 0;
        for (int a = 0; a < 6; ++a) for (int c2 = 0; c2 <= a; ++c2) { s_Hpp[a * 6 + c2] = s_red[k]; s_Hpp[c2 * 6 + a] = s_red[k]; ++k; }
        for (int a = 0; a < 6; ++a) { s_bp[a] = s_red[21 + a]; mm = fmax(mm, fabs(s_Hpp[7 * a])); }
        s_lambda = tau * mm;
      }
      __syncthreads();
      if (it == 0) { lambda = s_lambda; ni = 2; nBad = 0; }
    }
    double rho = 0;
    int qmax = 0;
    do {
      if (tid == 0) {
        double Hs[36], xs[6], bs[6];
        for (int i = 0; i < 36; ++i) Hs[i] = s_Hpp[i];
        for (int j = 0; j < 6; ++j) { Hs[7 * j] += lambda; bs[j] = s_bp[j]; }
        const bool ok2 = ldlt6_solve(Hs, bs, xs);
        s_ok2 = ok2 ? 1 : 0;
        if (ok2) for (int j = 0; j < 6; ++j) s_xp[j] = xs[j];       // a failed LDLT leaves x untouched
        s_Ttry = se3_exp_compose(s_xp, s_T);
        double s = 0;
        for (int j = 0; j < 6; ++j) s += s_xp[j] * (lambda * s_xp[j] + s_bp[j]);
        s_scale = s + 1e-3;
      }
      __syncthreads();
      const bool ok2 = s_ok2 != 0;
      const double scale = s_scale;
      last_err_chi = tempChi = compute_errors(s_Ttry);
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - cube_rn(2 * rho - 1);
        alpha = fmin(alpha, upper);
        lambda *= fmax(lower, alpha); ni = 2; currentChi = tempChi; err_valid = true;
        if (tid == 0) s_T = s_Ttry;
      } else {
        lambda *= ni; ni *= 2; err_valid = false;
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
  // ---- classification on the stored errors of the last evaluated trial (Optimizer.cc:2284-2299, 2679-2694)
  double cnt[1] = {0.0};
  const float gate = (float)P.chi2_gate;
  for (int i = tid; i < N; i += F2_THREADS) {
    const float chi2 = (float)(err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]);
    const bool outl = chi2 > gate;
    A.inlier_out[P.off + i] = outl ? 0 : 1;
    cnt[0] += outl ? 0.0 : 1.0;
  }
  block_reduce<1>(cnt, s_scr, s_red);
  if (tid == 0) {
    se3_to_matrix(s_T, res->T);
    res->n_inliers = (int)(s_red[0] + 0.5);
    res->iterations = it; res->trials = total_trials; res->stop_reason = stop_reason;
    res->initial_chi2 = initial_chi2; res->final_chi2 = last_err_chi; res->final_lambda = lambda;
  }
}

}  // namespace vdo

using namespace vdo;

struct vdo_pose_batch {
  vdo_ctx* ctx = nullptr;
  int n_problems = 0;
  int64_t total = 0;
  std::vector<int64_t> off;
  std::vector<int> n;
  PoseDev* d_probs = nullptr;
  PoseArrays A{};
  std::vector<void*> allocs;
};

extern "C" int vdo_pose_batch_destroy(vdo_pose_batch* b) {
  if (!b) return VDO_OK;
  if (b->ctx) ctx_bind(b->ctx);
  for (void* p : b->allocs) hipFree(p);
  delete b;
  return VDO_OK;
}

extern "C" int vdo_pose_batch_create(vdo_ctx* ctx, int n_problems, const vdo_pose_problem* probs, vdo_pose_batch** out) {
  if (!ctx || !probs || !out || n_problems <= 0) return set_error(VDO_ERR_INVALID, "vdo_pose_batch_create: bad argument");
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  vdo_pose_batch* b = new vdo_pose_batch();
  b->ctx = ctx; b->n_problems = n_problems;
  std::vector<PoseDev> hp(n_problems);
  int64_t tot = 0;
  for (int k = 0; k < n_problems; ++k) {
    const vdo_pose_problem& p = probs[k];
    if (p.n < 0 || (p.kind != 0 && p.kind != 1) || (p.n > 0 && (!p.obs || !p.Xw))) { delete b; return set_error(VDO_ERR_INVALID, "pose problem %d: bad fields", k); }
    PoseDev& d = hp[k];
    d.n = p.n; d.kind = p.kind; d.max_iterations = p.max_iterations; d.robust = p.huber_delta > 0 ? 1 : 0;
    d.off = tot;
    std::memcpy(d.K, p.K, sizeof d.K); std::memcpy(d.P, p.P, sizeof d.P); std::memcpy(d.T0, p.T0, sizeof d.T0);
    d.huber_delta = p.huber_delta;
    d.huber_dsqr = (double)(float)(p.huber_delta * p.huber_delta);     // RobustKernel::_delta^2 kept in a float member (robust_kernel.h)
    d.chi2_gate = p.chi2_gate;
    b->off.push_back(tot); b->n.push_back(p.n);
    tot += p.n;
  }
  b->total = tot;
  hipStream_t s = ctx->stream;
  std::vector<double> obs(2 * (size_t)tot + 2), xw(3 * (size_t)tot + 3);
  for (int k = 0; k < n_problems; ++k) {
    if (!probs[k].n) continue;
    std::memcpy(obs.data() + 2 * b->off[k], probs[k].obs, sizeof(double) * 2 * probs[k].n);
    std::memcpy(xw.data() + 3 * b->off[k], probs[k].Xw, sizeof(double) * 3 * probs[k].n);
  }
  auto alloc = [&](void** p, size_t bytes) -> bool {
    if (hipMalloc(p, bytes ? bytes : 8) != hipSuccess) return false;
    b->allocs.push_back(*p);
    return true;
  };
  double *d_obs = nullptr, *d_xw = nullptr;
  if (!alloc((void**)&d_obs, obs.size() * 8) || !alloc((void**)&d_xw, xw.size() * 8) || !alloc((void**)&b->A.err, (2 * (size_t)tot + 2) * 8) ||
      !alloc((void**)&b->A.inlier_out, (size_t)tot + 8) || !alloc((void**)&b->A.results, sizeof(vdo_flow2_result) * n_problems) ||
      !alloc((void**)&b->d_probs, sizeof(PoseDev) * n_problems)) {
    vdo_pose_batch_destroy(b);
    return set_error(VDO_ERR_OOM, "hipMalloc failed");
  }
  hipMemcpyAsync(d_obs, obs.data(), obs.size() * 8, hipMemcpyHostToDevice, s);
  hipMemcpyAsync(d_xw, xw.data(), xw.size() * 8, hipMemcpyHostToDevice, s);
  hipMemcpyAsync(b->d_probs, hp.data(), sizeof(PoseDev) * n_problems, hipMemcpyHostToDevice, s);
  b->A.obs = d_obs; b->A.Xw = d_xw;
  if (hipStreamSynchronize(s) != hipSuccess) { vdo_pose_batch_destroy(b); return set_error(VDO_ERR_NO_DEVICE, "upload failed"); }
  *out = b;
  return VDO_OK;
}

extern "C" int vdo_pose_batch_run(vdo_pose_batch* b) {
  if (!b) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(b->ctx);
  if (rc != VDO_OK) return rc;
  hipLaunchKernelGGL(k_pose_lm, dim3(b->n_problems), dim3(F2_THREADS), 0, b->ctx->stream, (const PoseDev*)b->d_probs, b->A);
  return VDO_OK;
}

extern "C" int vdo_pose_batch_fetch(vdo_pose_batch* b, vdo_flow2_result* results, uint8_t** inlier_out) {
  if (!b || !results) return set_error(VDO_ERR_INVALID, "null argument");
  int rc = ctx_bind(b->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = b->ctx->stream;
  hipMemcpyAsync(results, b->A.results, sizeof(vdo_flow2_result) * b->n_problems, hipMemcpyDeviceToHost, s);
  if (inlier_out)
    for (int k = 0; k < b->n_problems; ++k)
      if (inlier_out[k] && b->n[k]) hipMemcpyAsync(inlier_out[k], b->A.inlier_out + b->off[k], (size_t)b->n[k], hipMemcpyDeviceToHost, s);
  hipError_t e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "vdo_pose_batch_fetch: %s", hipGetErrorString(e));
  return VDO_OK;
}

extern "C" int vdo_pose_optimize(vdo_ctx* ctx, const vdo_pose_problem* p, vdo_flow2_result* result, uint8_t* inlier_out) {
  vdo_pose_batch* b = nullptr;
  int rc = vdo_pose_batch_create(ctx, 1, p, &b);
  if (rc != VDO_OK) return rc;
  rc = vdo_pose_batch_run(b);
  if (rc == VDO_OK) rc = vdo_pose_batch_fetch(b, result, &inlier_out);
  vdo_pose_batch_destroy(b);
  return rc;
}
