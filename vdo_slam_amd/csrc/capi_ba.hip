// C-ABI entry points (include/vdo_slam_hip.h) for the batch-BA path + host-side
// Levenberg–Marquardt driver.  The control flow restates the *modified* g2o of the reference:
//   SparseOptimizer::optimize            g2o/core/sparse_optimizer.cpp:354-443 (incl. :393-396 chi2 abort)
//   OptimizationAlgorithmLevenberg::solve g2o/core/optimization_algorithm_levenberg.cpp:61-164 (incl. _nBad rule :154-161)
//   SparseOptimizerTerminateAction        g2o/core/sparse_optimizer_terminate_action.cpp:49-85
// All heavy work is HIP kernels (ba_sweep.hip, ba_solve.hip); the host only sequences
// launches and reads back a handful of scalars per Levenberg trial.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ba_dev.hpp"
#include "ctx.hpp"

using namespace vdo;

namespace {

template <class T>
int upload(T** dst, const T* src, size_t n, hipStream_t s) {
  if (n == 0) { *dst = nullptr; return VDO_OK; }
  if (hipMalloc((void**)dst, n * sizeof(T)) != hipSuccess) return set_error(VDO_ERR_OOM, "hipMalloc(%zu) failed", n * sizeof(T));
  if (src) {
    if (hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, s) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "H2D copy failed");
  } else {
    hipMemsetAsync(*dst, 0, n * sizeof(T), s);
  }
  return VDO_OK;
}

// runs of equal pose index, split at VDO_CHUNK
std::vector<Chunk> make_chunks(const int32_t* pose, int n) {
  std::vector<Chunk> out;
  int b = 0;
  while (b < n) {
    int e = b + 1;
    while (e < n && pose[e] == pose[b] && e - b < VDO_CHUNK) ++e;
    out.push_back(Chunk{pose[b], b, e, 0});
    b = e;
  }
  return out;
}

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct vdo_ba {
  vdo_ctx* ctx = nullptr;
  BADev d;
  std::vector<void*> allocs;
  vdo_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int oplus_calls = 0;            // VertexSE3::_numOplusCalls (same value on every vertex)
  double* h_scal = nullptr;       // pinned
  int32_t* h_flags = nullptr;     // pinned
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

#define UP(field, src, n)                                                          \
  do {                                                                             \
    int rc_ = upload(&ba->d.field, src, (size_t)(n), s);                           \
    if (rc_ != VDO_OK) { vdo_ba_destroy(ba); return rc_; }                          \
    if (ba->d.field) ba->allocs.push_back((void*)ba->d.field);                     \
  } while (0)

extern "C" int vdo_ba_create(vdo_ctx* ctx, const vdo_ba_graph* g, vdo_ba** out) {
  if (!ctx || !g || !out) return set_error(VDO_ERR_INVALID, "vdo_ba_create: null argument");
  if (g->n_pose <= 0 || g->n_point < 0 || g->n_eb < 0 || g->n_et < 0 || g->n_ep < 0 || g->n_prior < 0)
    return set_error(VDO_ERR_INVALID, "vdo_ba_create: negative/empty sizes");
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  const int P = g->n_pose, L = g->n_point, Eb = g->n_eb, Et = g->n_et, Ep = g->n_ep, Npr = g->n_prior;
  // ---- validate indices
  for (int e = 0; e < Eb; ++e)
    if ((unsigned)g->eb_pose[e] >= (unsigned)P || (unsigned)g->eb_point[e] >= (unsigned)L)
      return set_error(VDO_ERR_INVALID, "binary edge %d: index out of range", e);
  for (int e = 0; e < Et; ++e)
    if ((unsigned)g->et_pose[e] >= (unsigned)P || (unsigned)g->et_p1[e] >= (unsigned)L || (unsigned)g->et_p2[e] >= (unsigned)L || g->et_p1[e] == g->et_p2[e])
      return set_error(VDO_ERR_INVALID, "ternary edge %d: index out of range", e);
  for (int e = 0; e < Ep; ++e)
    if ((unsigned)g->ep_i[e] >= (unsigned)P || (unsigned)g->ep_j[e] >= (unsigned)P || g->ep_i[e] == g->ep_j[e])
      return set_error(VDO_ERR_INVALID, "pose-pose edge %d: index out of range", e);
  for (int e = 0; e < Npr; ++e)
    if ((unsigned)g->pr_pose[e] >= (unsigned)P) return set_error(VDO_ERR_INVALID, "prior %d: index out of range", e);
  // ---- chains: next/prev ternary edge per point
  std::vector<int32_t> next_e(L, -1), prev_e(L, -1);
  for (int e = 0; e < Et; ++e) {
    if (next_e[g->et_p1[e]] != -1 || prev_e[g->et_p2[e]] != -1)
      return set_error(VDO_ERR_UNSUPPORTED, "ternary edge %d: landmark tracks must be simple chains", e);
    next_e[g->et_p1[e]] = e;
    prev_e[g->et_p2[e]] = e;
  }
  std::vector<int32_t> chain_off{0}, chain_pt, chain_edge;
  chain_pt.reserve(L); chain_edge.reserve(L);
  int visited = 0;
  for (int l = 0; l < L; ++l) {
    if (prev_e[l] != -1) continue;
    int cur = l;
    for (;;) {
      chain_pt.push_back(cur); ++visited;
      int e = next_e[cur];
      if (e == -1) { chain_edge.push_back(-1); break; }
      chain_edge.push_back(e);
      cur = g->et_p2[e];
    }
    chain_off.push_back((int32_t)chain_pt.size());
  }
  if (visited != L) return set_error(VDO_ERR_UNSUPPORTED, "ternary edges form a cycle");
  // ---- chunks
  std::vector<Chunk> cb = make_chunks(g->eb_pose, Eb), ct = make_chunks(g->et_pose, Et);
  const int ncb = (int)cb.size(), nct = (int)ct.size();
  std::vector<Chunk> cinc(cb);
  for (int rep = 0; rep < 2; ++rep)
    for (const Chunk& c : ct) cinc.push_back(Chunk{c.pose, Eb + rep * Et + c.begin, Eb + rep * Et + c.end, 0});
  std::vector<int32_t> pc_off(P + 1, 0), pc_idx(ncb + nct);
  for (int i = 0; i < ncb; ++i) pc_off[cb[i].pose + 1]++;
  for (int i = 0; i < nct; ++i) pc_off[ct[i].pose + 1]++;
  for (int p = 0; p < P; ++p) pc_off[p + 1] += pc_off[p];
  {
    std::vector<int32_t> fill(pc_off.begin(), pc_off.end() - 1);
    for (int i = 0; i < ncb; ++i) pc_idx[fill[cb[i].pose]++] = i;
    for (int i = 0; i < nct; ++i) pc_idx[fill[ct[i].pose]++] = ncb + i;
  }
  std::vector<int32_t> inc_pose((size_t)Eb + 2 * (size_t)Et), inc_point((size_t)Eb + 2 * (size_t)Et);
  for (int e = 0; e < Eb; ++e) { inc_pose[e] = g->eb_pose[e]; inc_point[e] = g->eb_point[e]; }
  for (int e = 0; e < Et; ++e) {
    inc_pose[Eb + e] = g->et_pose[e]; inc_point[Eb + e] = g->et_p1[e];
    inc_pose[Eb + Et + e] = g->et_pose[e]; inc_point[Eb + Et + e] = g->et_p2[e];
  }

  vdo_ba* ba = new vdo_ba();
  ba->ctx = ctx;
  hipStream_t s = ctx->stream;
  BADev& d = ba->d;
  d.P = P; d.L = L; d.Eb = Eb; d.Et = Et; d.Ep = Ep; d.Npr = Npr; d.Ninc = Eb + 2 * Et;
  d.huber_eb = g->huber_eb; d.huber_et = g->huber_et; d.huber_ep = g->huber_ep;
  d.dsqr_eb = (double)(float)(g->huber_eb * g->huber_eb);   // float member, robust_kernel_impl.h:84
  d.dsqr_et = (double)(float)(g->huber_et * g->huber_et);
  d.dsqr_ep = (double)(float)(g->huber_ep * g->huber_ep);
  d.n_chunks_b = ncb; d.n_chunks_t = nct; d.n_chunks_inc = ncb + 2 * nct;
  d.n_chains = (int)chain_off.size() - 1;
  UP(pose[0], g->pose, 12 * (size_t)P); UP(pose[1], g->pose, 12 * (size_t)P);
  UP(point[0], g->point, 3 * (size_t)L); UP(point[1], g->point, 3 * (size_t)L);
  UP(eb_pose, g->eb_pose, Eb); UP(eb_point, g->eb_point, Eb); UP(eb_z, g->eb_z, 3 * (size_t)Eb); UP(eb_w, g->eb_w, Eb);
  UP(et_p1, g->et_p1, Et); UP(et_p2, g->et_p2, Et); UP(et_pose, g->et_pose, Et); UP(et_z, g->et_z, 3 * (size_t)Et); UP(et_w, g->et_w, Et);
  UP(ep_i, g->ep_i, Ep); UP(ep_j, g->ep_j, Ep); UP(ep_z, g->ep_z, 12 * (size_t)Ep); UP(ep_info, g->ep_info, 36 * (size_t)Ep);
  UP(pr_pose, g->pr_pose, Npr); UP(pr_z, g->pr_z, 12 * (size_t)Npr); UP(pr_info, g->pr_info, 36 * (size_t)Npr);
  UP(chunks_b, cb.data(), ncb); UP(chunks_t, ct.data(), nct); UP(chunks_inc, cinc.data(), cinc.size());
  UP(pc_off, pc_off.data(), P + 1); UP(pc_idx, pc_idx.data(), pc_idx.size());
  UP(inc_pose, inc_pose.data(), inc_pose.size()); UP(inc_point, inc_point.data(), inc_point.size());
  UP(chain_off, chain_off.data(), chain_off.size()); UP(chain_pt, chain_pt.data(), chain_pt.size());
  UP(chain_edge, chain_edge.data(), chain_edge.size());
  const double* Z = nullptr;
  UP(Hpp, Z, 36 * (size_t)P); UP(bp, Z, 6 * (size_t)P); UP(Hll, Z, 9 * (size_t)L); UP(bl, Z, 3 * (size_t)L);
  UP(Binc, Z, 18 * (size_t)d.Ninc); UP(Oll, Z, 9 * (size_t)Et); UP(Hpp_ep, Z, 36 * (size_t)Ep);
  UP(chunk_sums, Z, 18 * (size_t)(ncb + nct));
  UP(chunk_chi, Z, 2 * (size_t)(ncb + nct + 1) + 2 * (size_t)(Ep + Npr));
  UP(Dinv, Z, 9 * (size_t)L); UP(Gl, Z, 9 * (size_t)L);
  UP(ul, Z, 3 * (size_t)L); UP(wl, Z, 3 * (size_t)L); UP(xl, Z, 3 * (size_t)L);
  UP(Minv, Z, 36 * (size_t)P);
  UP(xp, Z, 6 * (size_t)P); UP(rp, Z, 6 * (size_t)P); UP(zp, Z, 6 * (size_t)P); UP(pp, Z, 6 * (size_t)P);
  UP(qp, Z, 6 * (size_t)P); UP(bs, Z, 6 * (size_t)P); UP(qs, Z, 6 * (size_t)P);
  UP(chunk_q, Z, 6 * (size_t)d.n_chunks_inc);
  UP(scal, Z, S_COUNT);
  const int32_t* ZI = nullptr;
  UP(flags, ZI, 4);
  if (hipHostMalloc((void**)&ba->h_scal, S_COUNT * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&ba->h_flags, 4 * sizeof(int32_t)) != hipSuccess) {
    vdo_ba_destroy(ba);
    return set_error(VDO_ERR_OOM, "hipHostMalloc failed");
  }
  hipEventCreate(&ba->ev0); hipEventCreate(&ba->ev1);
  if (hipStreamSynchronize(s) != hipSuccess) { vdo_ba_destroy(ba); return set_error(VDO_ERR_NO_DEVICE, "upload failed: %s", hipGetErrorString(hipGetLastError())); }
  *out = ba;
  return VDO_OK;
}

extern "C" int vdo_ba_destroy(vdo_ba* ba) {
  if (!ba) return VDO_OK;
  if (ba->ctx) ctx_bind(ba->ctx);
  for (void* p : ba->allocs) hipFree(p);
  if (ba->h_scal) hipHostFree(ba->h_scal);
  if (ba->h_flags) hipHostFree(ba->h_flags);
  if (ba->ev0) hipEventDestroy(ba->ev0);
  if (ba->ev1) hipEventDestroy(ba->ev1);
  delete ba;
  return VDO_OK;
}

extern "C" int vdo_ba_set_allreduce(vdo_ba* ba, vdo_allreduce_fn fn, void* user) {
  if (!ba) return set_error(VDO_ERR_INVALID, "null handle");
  ba->allreduce = fn; ba->allreduce_user = user;
  return VDO_OK;
}

static int sync_check(vdo_ba* ba, const char* what) {
  hipError_t e = hipStreamSynchronize(ba->ctx->stream);
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "%s: %s", what, hipGetErrorString(e));
  e = hipGetLastError();
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "%s: %s", what, hipGetErrorString(e));
  return VDO_OK;
}

extern "C" int vdo_ba_linearize(vdo_ba* ba, int repeat, float* ms_sweep) {
  if (!ba) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ba->ctx->stream;
  if (repeat < 1) repeat = 1;
  if (ms_sweep) {
    launch_sweep_eb_only(ba->d, s);   // warm-up
    hipEventRecord(ba->ev0, s);
    for (int i = 0; i < repeat; ++i) launch_sweep_eb_only(ba->d, s);
    hipEventRecord(ba->ev1, s);
    hipEventSynchronize(ba->ev1);
    float ms = 0;
    hipEventElapsedTime(&ms, ba->ev0, ba->ev1);
    *ms_sweep = ms / repeat;
  }
  for (int i = 0; i < (ms_sweep ? 1 : repeat); ++i) launch_linearize(ba->d, s);
  return sync_check(ba, "vdo_ba_linearize");
}

extern "C" int vdo_ba_download_system(vdo_ba* ba, vdo_ba_system* out) {
  if (!ba || !out) return set_error(VDO_ERR_INVALID, "null argument");
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  const BADev& d = ba->d;
  hipStream_t s = ba->ctx->stream;
  auto D2H = [&](void* dst, const void* src, size_t bytes) { if (dst && bytes) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s); };
  D2H(out->Hpp, d.Hpp, sizeof(double) * 36 * (size_t)d.P);
  D2H(out->bp, d.bp, sizeof(double) * 6 * (size_t)d.P);
  D2H(out->Hll, d.Hll, sizeof(double) * 9 * (size_t)d.L);
  D2H(out->bl, d.bl, sizeof(double) * 3 * (size_t)d.L);
  D2H(out->Hll_et, d.Oll, sizeof(double) * 9 * (size_t)d.Et);
  D2H(out->Hpp_ep, d.Hpp_ep, sizeof(double) * 36 * (size_t)d.Ep);
  std::vector<double> binc;
  if (out->Hpl_eb || out->Hlp1_et || out->Hlp2_et) {
    binc.resize(18 * (size_t)d.Ninc);
    D2H(binc.data(), d.Binc, sizeof(double) * binc.size());
  }
  D2H(ba->h_scal, d.scal, sizeof(double) * S_COUNT);
  rc = sync_check(ba, "vdo_ba_download_system");
  if (rc != VDO_OK) return rc;
  const size_t N = d.Ninc, Eb = d.Eb, Et = d.Et;
  if (out->Hpl_eb)
    for (int i = 0; i < 18; ++i) std::memcpy(out->Hpl_eb + i * Eb, binc.data() + i * N, sizeof(double) * Eb);
  for (int rep = 0; rep < 2; ++rep) {
    double* dst = rep == 0 ? out->Hlp1_et : out->Hlp2_et;
    if (!dst) continue;
    for (int r = 0; r < 3; ++r)          // dst: 3x6 (point x pose) = transpose of the stored 6x3
      for (int c = 0; c < 6; ++c)
        std::memcpy(dst + (size_t)(r * 6 + c) * Et, binc.data() + (size_t)(c * 3 + r) * N + Eb + rep * Et, sizeof(double) * Et);
  }
  out->chi2 = ba->h_scal[S_CHI2];
  out->robust_chi2 = ba->h_scal[S_RCHI2];
  return VDO_OK;
}

extern "C" int vdo_ba_get_estimates(vdo_ba* ba, double* pose_out, double* point_out) {
  if (!ba) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ba->ctx->stream;
  if (pose_out) hipMemcpyAsync(pose_out, ba->d.pose[0], sizeof(double) * 12 * (size_t)ba->d.P, hipMemcpyDeviceToHost, s);
  if (point_out && ba->d.L) hipMemcpyAsync(point_out, ba->d.point[0], sizeof(double) * 3 * (size_t)ba->d.L, hipMemcpyDeviceToHost, s);
  return sync_check(ba, "vdo_ba_get_estimates");
}

extern "C" int vdo_ba_set_estimates(vdo_ba* ba, const double* pose, const double* point) {
  if (!ba) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ba->ctx->stream;
  if (pose) hipMemcpyAsync(ba->d.pose[0], pose, sizeof(double) * 12 * (size_t)ba->d.P, hipMemcpyHostToDevice, s);
  if (point && ba->d.L) hipMemcpyAsync(ba->d.point[0], point, sizeof(double) * 3 * (size_t)ba->d.L, hipMemcpyHostToDevice, s);
  ba->oplus_calls = 0;
  return sync_check(ba, "vdo_ba_set_estimates");
}

namespace {

// read back device scalars + flags (one sync)
int fetch(vdo_ba* ba) {
  hipStream_t s = ba->ctx->stream;
  hipMemcpyAsync(ba->h_scal, ba->d.scal, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, s);
  hipMemcpyAsync(ba->h_flags, ba->d.flags, sizeof(int32_t) * 4, hipMemcpyDeviceToHost, s);
  return sync_check(ba, "LM scalar readback");
}

// computeActiveErrors + activeRobustChi2 at estimate[which]
int robust_chi2(vdo_ba* ba, int which, double* out) {
  launch_errors(ba->d, which, ba->ctx->stream);
  int rc = fetch(ba);
  if (rc != VDO_OK) return rc;
  *out = ba->h_scal[S_RCHI2];
  return VDO_OK;
}

// (H + lambda I) x = b  ->  xp/xl on device.  ok=false mirrors a failed Cholesky.
int solve_trial(vdo_ba* ba, double lambda, const vdo_lm_options* opt, bool* ok, int* pcg_iters) {
  const BADev& d = ba->d;
  hipStream_t s = ba->ctx->stream;
  launch_factor(d, lambda, s);
  launch_reduced_rhs(d, s);
  launch_pcg_init(d, s);
  double tol = opt->pcg_tolerance > 0 ? opt->pcg_tolerance : 1e-10;
  int maxit = opt->pcg_max_iterations > 0 ? opt->pcg_max_iterations : std::min(20000, 24 * d.P + 200);
  const double tol2 = tol * tol;
  int it = 0;
  *ok = true;
  while (it < maxit) {
    const int batch = std::min(16, maxit - it);
    for (int k = 0; k < batch; ++k) launch_pcg_iter_tol(d, lambda, tol2, d.qs, s);
    it += batch;
    int rc = fetch(ba);
    if (rc != VDO_OK) return rc;
    if (ba->h_flags[0]) { *ok = false; break; }
    if (ba->h_flags[1] == 1) break;
    if (ba->h_flags[1] == 2) { *ok = false; break; }
  }
  *pcg_iters = ba->h_flags[2];
  return VDO_OK;
}

}  // namespace

extern "C" int vdo_ba_optimize(vdo_ba* ba, const vdo_lm_options* opt, vdo_lm_stats* st) {
  if (!ba || !opt) return set_error(VDO_ERR_INVALID, "null argument");
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  vdo_lm_stats local;
  if (!st) st = &local;
  std::memset(st, 0, sizeof(*st));
  BADev& d = ba->d;
  hipStream_t s = ba->ctx->stream;
  const double t_begin = now_ms();
  double lambda = -1, ni = 2;
  int nBad = 0;
  const double tau = 1e-5, upper = 2. / 3., lower = 1. / 3.;
  const int maxTrials = 10;
  bool forceStop = false, ok = true;
  double action_lastChi = 0, chi2_check = 0, last_err_chi = 0;
#define CK(x) do { rc = (x); if (rc != VDO_OK) return rc; } while (0)
  CK(robust_chi2(ba, 0, &last_err_chi));
  st->initial_chi2 = last_err_chi;
  int it = 0;
  for (; it < opt->max_iterations && !forceStop && ok; ++it) {
    double t0 = now_ms();
    launch_linearize(d, s);                 // errors + buildSystem in one sweep (same estimate)
    if (it == 0) launch_max_diag(d, s);
    CK(fetch(ba));
    st->ms_linearize += now_ms() - t0;
    last_err_chi = ba->h_scal[S_RCHI2];
    double currentChi = last_err_chi, tempChi = currentChi;
    const double iniChi = currentChi;
    if (it == 0) { lambda = tau * ba->h_scal[S_MAXDIAG]; ni = 2; nBad = 0; }
    double rho = 0;
    int qmax = 0;
    do {
      t0 = now_ms();
      bool ok2 = true;
      int pcg_it = 0;
      CK(solve_trial(ba, lambda, opt, &ok2, &pcg_it));
      const bool ortho = (++ba->oplus_calls > 1000);
      if (ortho) ba->oplus_calls = 0;
      launch_backsub_update(d, lambda, ortho, s);      // update() into the trial buffers (push/pop = keep [0])
      launch_errors(d, 1, s);
      CK(fetch(ba));
      st->ms_solve += now_ms() - t0;
      last_err_chi = tempChi = ba->h_scal[S_RCHI2];
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = ba->h_scal[S_SCALE] + 1e-3;
      rho /= scale;
      if (opt->verbose > 1)
        std::fprintf(stderr, "  trial %d lambda=%.4g pcg=%d chi2 %.9g -> %.9g rho=%.4g\n", qmax, lambda, pcg_it, currentChi, tempChi, rho);
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, upper);
        const double sf = std::max(lower, alpha);
        lambda *= sf; ni = 2; currentChi = tempChi;
        std::swap(d.pose[0], d.pose[1]);               // discardTop(): accept the trial
        std::swap(d.point[0], d.point[1]);
      } else {
        lambda *= ni; ni *= 2;                          // pop(): estimate[0] untouched
      }
      ++qmax;
      ++st->total_trials;
    } while (rho < 0 && qmax < maxTrials && !forceStop);
    int result;
    if (qmax == maxTrials || rho == 0) result = 1;
    else {
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      result = nBad >= 3 ? 1 : 0;
    }
    ok = (result == 0);
    if (!ok && st->stop_reason == 0) st->stop_reason = 1;
    if (chi2_check < last_err_chi && it > 0) { ok = false; st->stop_reason = 2; }
    chi2_check = last_err_chi;
    if (opt->verbose || opt->gain_threshold >= 0) CK(robust_chi2(ba, 0, &last_err_chi));
    if (opt->verbose)
      std::fprintf(stderr, "iteration= %d\t chi2= %.6f\t lambda= %.6g\t levenbergIter= %d\n", it, last_err_chi, lambda, qmax);
    if (it < VDO_LM_MAX_TRACE) { st->chi2_trace[it] = last_err_chi; st->trials_trace[it] = qmax; }
    if (opt->gain_threshold >= 0) {
      if (it == 0) action_lastChi = last_err_chi;
      else {
        const double gain = (action_lastChi - last_err_chi) / last_err_chi;
        action_lastChi = last_err_chi;
        if (gain >= 0 && gain < opt->gain_threshold) { forceStop = true; if (ok) st->stop_reason = 3; }
      }
    }
  }
  st->iterations = it;
  st->final_lambda = lambda;
  CK(robust_chi2(ba, 0, &st->final_chi2));
  st->ms_total = now_ms() - t_begin;
#undef CK
  return VDO_OK;
}
