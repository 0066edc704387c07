// Host-side Levenberg–Marquardt driver of the batch-BA path (C-ABI: vdo_ba_optimize).
// The control flow restates the *modified* g2o of the reference:
//   SparseOptimizer::optimize            g2o/core/sparse_optimizer.cpp:354-443 (incl. :393-396 chi2 abort)
//   OptimizationAlgorithmLevenberg::solve g2o/core/optimization_algorithm_levenberg.cpp:61-164 (incl. _nBad rule :154-161)
//   SparseOptimizerTerminateAction        g2o/core/sparse_optimizer_terminate_action.cpp:49-85
// All heavy work is HIP kernels (ba_sweep.hip, ba_solve.hip); the host only sequences
// launches and reads back a handful of scalars per Levenberg trial.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "ba_host.hpp"

using namespace vdo;

namespace {
double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
}  // namespace

namespace {

// read back device scalars + flags (one sync): a one-wave kernel writes them into the mapped pinned block - the two D2H copies this replaces were 13 us
// of blit kernel each, 2-3 times per LM iteration (profiles/r05_ba_large_kernel_stats.txt: __amd_rocclr_copyBuffer 6 % of the stream)
// (round 6) the kernel's last act is a ticket in the same pinned block and the host POLLS it: hipStreamSynchronize returns ~5 us after the kernel's end (completion signal,
// barrier packet; tools/sync_latency_probe.hip: launch -> host knows 11.7 us, 7.0 with the ticket) - once per Levenberg trial.  The data is fenced system-wide before the ticket;
// a ticket that does not come within 50 ms (a faulted kernel never writes it; a trial of the largest graphs takes ~1 ms) falls back to the stream wait, which reports the error.
int fetch(vdo_ba* ba) {
  static const bool no_poll = std::getenv("VDO_BA_NO_TICKET") != nullptr;
  if (no_poll || ba->d.sharded) {                        // (sharded: the exchanges of the all-reduce hook may run on the host's side of the stream)
    launch_publish_scalars(ba->d, ba->d_hscal, ba->ctx->stream);
    return sync_check(ba, "LM scalar readback");
  }
  const uint32_t want = ++ba->ticket ? ba->ticket : ++ba->ticket;      // (never 0)
  launch_publish_scalars(ba->d, ba->d_hscal, ba->ctx->stream, want);
  volatile uint32_t* word = reinterpret_cast<volatile uint32_t*>(ba->h_flags) + 4;
  const double t0 = now_ms();
  for (uint32_t spins = 0; *word != want; ++spins) {
    if ((spins & 1023u) == 1023u && now_ms() - t0 > 50.0) return sync_check(ba, "LM scalar readback");
    __builtin_ia32_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "LM scalar readback: %s", hipGetErrorString(e));
  return VDO_OK;
}

// the second set of linearisation outputs (ba_dev.hpp lin_swap), allocated on the first single-GPU optimisation of the handle; false: no memory for it (the caller keeps the
// error-evaluation pass of rounds 1-5)
bool ensure_alt(vdo_ba* ba) {
  BADev& d = ba->d;
  if (d.Hpp_alt) return true;
  if (ba->alt_failed) return false;
  hipStream_t s = ba->ctx->stream;
  auto get = [&](double*& p, size_t n) {
    if (n == 0) { p = nullptr; return true; }
    p = (double*)ba_device_alloc(ba, n * sizeof(double));
    if (!p) return false;
    hipMemsetAsync(p, 0, n * sizeof(double), s);
    return true;
  };
  const size_t P = (size_t)d.P, L = (size_t)d.L, Et = (size_t)d.Et;
  const bool ok = get(d.Finc_alt, std::max<size_t>((size_t)d.Eb, VDO_TILE_THREADS) + Et + 1) && get(d.Hll_alt, L) && get(d.bl_alt, 3 * L) && get(d.Oll_alt, 9 * Et) &&
                  get(d.part_sums_alt, (size_t)d.ps_stride * (size_t)std::max(d.NPS, 1)) && get(d.ep_blk_alt, 84 * (size_t)std::max(d.Ep + d.Npr, 1)) && get(d.Hpp_ep_alt, 36 * (size_t)d.Ep) &&
                  get(d.hub_we_alt, (size_t)d.n_hub_edges) && get(d.Hpp_alt, 69 * P + 6);
  if (!ok) {
    (void)hipGetLastError();
    d.Finc_alt = d.Hll_alt = d.bl_alt = d.Oll_alt = d.part_sums_alt = d.ep_blk_alt = d.Hpp_ep_alt = d.hub_we_alt = d.Hpp_alt = nullptr;      // (what was allocated goes with the handle)
    ba->alt_failed = true;
  }
  return ok;
}

// computeActiveErrors + activeRobustChi2 at estimate[which]
int robust_chi2(vdo_ba* ba, int which, double* out) {
  launch_errors(ba->d, which, ba->ctx->stream, ba->red);
  int rc = fetch(ba);
  if (rc != VDO_OK) return rc;
  *out = ba->h_scal[S_RCHI2];
  return VDO_OK;
}

constexpr int kPcgSlowTiny = 12;                 // ... on a reduced system of at most kDenseTinyUnknowns unknowns
constexpr int64_t kDenseTinyUnknowns = 384;
constexpr int kPcgSlow = 60;                     // PCG iterations per solve above which auto mode switches to the dense solver
constexpr int64_t kDenseMaxUnknowns = 8192;     // 6P above this: S no longer "small" (512 MB at 8192) - PCG only

int dense_prepare(vdo_ba* ba) {
  if (ba->dense_S) return VDO_OK;
  const int64_t ld = (6 * (int64_t)ba->d.P + 63) / 64 * 64;
  void *pS = ba_device_alloc(ba, sizeof(double) * (size_t)ld * (size_t)ld), *pW = ba_device_alloc(ba, sizeof(double) * (size_t)ld * 64), *pr = ba_device_alloc(ba, sizeof(double) * 2 * (size_t)ld);
  if (!pS || !pW || !pr)      // (whatever was allocated is released with the handle)
    return set_error(VDO_ERR_OOM, "dense reduced-camera solver: hipMalloc(%lld x %lld doubles) failed", (long long)ld, (long long)ld);
  ba->dense_S = (double*)pS; ba->dense_W = (double*)pW; ba->dense_rhs = (double*)pr; ba->dense_ld = ld;
  return VDO_OK;
}

// (H + lambda I) x = b  ->  xp/xl on device.  ok=false mirrors a failed Cholesky.
// solver: 2 = Schur + chain-preconditioned PCG; 3 = Schur + dense MFMA Cholesky of the reduced-camera matrix; 0 = auto: dense when
// the pose graph is not a set of paths (loop closures, branches: the chain preconditioner then misses edges) or once a PCG solve
// needed more than kPcgSlow iterations, as long as 6P <= kDenseMaxUnknowns.
// PCG: the first batch of iterations is only ENQUEUED (*pending = true): the caller enqueues the update and the error evaluation of the trial
// behind it and reads everything back with ONE synchronisation; solve_trial_finish then looks at the flags and - in the rare case that the
// first batch did not converge - goes on iterating (the caller then repeats update + errors).  Same kernels in the same order as a look
// after every batch: same bits, two host round trips less per trial (each ~25 us: a tenth of an iteration on the 60-frame graph).
int solve_trial(vdo_ba* ba, double lambda, const vdo_lm_options* opt, bool* ok, int* pcg_iters, bool* pending) {
  const BADev& d = ba->d;
  hipStream_t s = ba->ctx->stream;
  const bool small = 6 * (int64_t)d.P <= kDenseMaxUnknowns;
  bool dense = opt->solver == 3 || (opt->solver == 0 && small && ba->dense_tiles_ok && (!ba->pose_graph_is_paths || ba->last_solver == 3));
  // a TINY reduced system (<= 384 unknowns: the 20-frame windows of PartialBatchOptimization) is assembled and factored (two MFMA panels) in less time than the handful of
  // mat-vec round trips a PCG solve takes: 0.24 against 0.30 ms per LM iteration on such a window (VDO_BA_TINY_PCG: A/B switch back to the PCG-first rule)
  static const bool tiny_pcg = std::getenv("VDO_BA_TINY_PCG") != nullptr;
  if (!tiny_pcg && opt->solver == 0 && ba->dense_tiles_ok && 6 * (int64_t)d.P <= kDenseTinyUnknowns) dense = true;
  // (dense_tiles_ok: the dense assembly's workgroup fits - padded incidences per thread, LDS at the graph's largest tile; capi_ba.hip)
  if (dense && !ba->dense_tiles_ok)
    return set_error(VDO_ERR_UNSUPPORTED, d.n_hubs ? "dense solver: this graph has %d hub landmark(s) (static points beyond a tile's capacity, ba_hub.hip) - the dense assembly walks tiles only; use the PCG solver"
                                                   : "dense solver: a tile of this graph does not fit the dense assembly (more than %d pose slots of LDS); use the PCG solver", d.n_hubs ? d.n_hubs : 200);
  if (dense && !small && opt->solver == 3) return set_error(VDO_ERR_UNSUPPORTED, "dense solver: %lld unknowns exceed %lld", 6LL * d.P, (long long)kDenseMaxUnknowns);
  launch_factor_and_rhs(d, lambda, s, ba->red, ba->side, ba->ev_fork, ba->ev_join, !dense, ba->lin_exchange_pending);
  ba->lin_exchange_pending = false;
  if (dense) {
    int rc = dense_prepare(ba);
    if (rc != VDO_OK) return rc;
    // (round 6) a reduced system of at most 128 unknowns - the 20-frame windows - is finished by ONE workgroup in LDS (k_dense_small: 8 launches less per trial)
    static const bool no_small = std::getenv("VDO_BA_NO_DENSE_SMALL") != nullptr;
    const bool one_wg = !no_small && 6 * (int64_t)d.P <= 128;
    launch_dense_assemble(d, ba->dense_S, ba->dense_ld, lambda, s, ba->red, !one_wg, one_wg && ba->dense_S_clean);
    if (one_wg) { launch_dense_small(d, ba->dense_S, ba->dense_ld, lambda, s); ba->dense_S_clean = true; }
    else {
      launch_dense_rhs(d, ba->dense_rhs, ba->dense_ld, s);
      launch_dense_solve(d, ba->dense_S, ba->dense_ld, ba->dense_W, ba->dense_rhs, s);
    }
    // (no read-back here: like the PCG batch, the factorisation's failure flag comes back with the trial's scalars - the caller enqueues the update and the error
    //  evaluation behind the solve and reads everything with ONE synchronisation; solve_trial_finish looks at the flag.  A failed factorisation leaves garbage in x:
    //  the trial is then rejected whatever its errors are, as after a read-back in between.)
    *ok = true;
    *pcg_iters = 0;
    *pending = true;
    ba->last_solver = 3;
    ba->dense_pending = true;
    return VDO_OK;
  }
  ba->last_solver = 2;
  launch_pcg_init(d, s);
  // Default tolerance of the PCG (relative preconditioned residual): 1e-8 since the end of round 6, 1e-10 before (g2o's own LinearSolverPCG stops at 1e-6; the reference uses its
  // direct solvers).  profiles/r06_pcg_tolerance_ab.txt: at 1e-9 the final chi2 of a 5-iteration Levenberg run is the 1e-10 run's to 12 digits on all four bench graphs, at 1e-8 to
  // 10-11 digits (the parity bars: chi2 trace 1e-6, estimates 1e-4), same iterations and trials; one CG iteration less in most solves: -7 % per LM iteration on the
  // OMD-shaped graph, -8 % on the 1 M-point one.  VDO_BA_PCG_TOL overrides the default (vdo_lm_options.pcg_tolerance > 0 overrides both).
  static const double tol_env = std::getenv("VDO_BA_PCG_TOL") ? std::atof(std::getenv("VDO_BA_PCG_TOL")) : 0.0;
  double tol = opt->pcg_tolerance > 0 ? opt->pcg_tolerance : (tol_env > 0 ? tol_env : 1e-8);
  int maxit = opt->pcg_max_iterations > 0 ? opt->pcg_max_iterations : std::min(20000, 24 * d.P + 200);
  const double tol2 = tol * tol;
  *ok = true;
  ba->pcg_it = 0; ba->pcg_parity = 0; ba->pcg_maxit = maxit; ba->pcg_tol2 = tol2;
  // The chain preconditioner converges in a handful of iterations and the count drifts slowly from trial to trial (3, 3, 4, 4, 5 on the bench's
  // graph): the first batch is what the previous solve needed + 1, at most 6 - an iteration that finds the convergence flag set still costs its
  // three (empty) launches, ~13 us.  (6 for the first solve of a run; a batch that turns out too short takes the second-look path of the caller.)
  // (round 6: no cap of 6 any more - the gauge-free windows of PartialBatchOptimization need 7 .. 35 iterations per solve, rising by one or two from trial to trial, and
  //  EVERY trial of theirs took the second-look path: a host round trip, a batch of 12, the update and the error evaluation twice.  The margin grows with the count.)
  const int batch = std::min(std::min(64, maxit), ba->pcg_last > 0 ? std::max(2, ba->pcg_last + 1 + ba->pcg_last / 6) : 6);
  for (int k = 0; k < batch; ++k, ba->pcg_parity ^= 1) launch_pcg_iter(d, lambda, tol2, ba->pcg_parity, s, ba->red);
  ba->pcg_it = batch;
  *pending = true;
  return VDO_OK;
}

// After a read-back that followed solve_trial's first batch: done (*again = false), or more iterations were needed and ran (*again = true: the
// caller's update + error evaluation used an unconverged x and must be repeated).
int solve_trial_finish(vdo_ba* ba, double lambda, const vdo_lm_options* opt, bool* ok, int* pcg_iters, bool* again) {
  const BADev& d = ba->d;
  hipStream_t s = ba->ctx->stream;
  *again = false;
  if (ba->dense_pending) {                 // the trial was solved by the dense factorisation: only its failure flag is of interest
    ba->dense_pending = false;
    *ok = ba->h_flags[0] == 0;
    *pcg_iters = 0;
    return VDO_OK;
  }
  for (;;) {
    if (ba->h_flags[0]) { *ok = false; break; }
    if (ba->h_flags[1] == 1) break;
    if (ba->h_flags[1] == 2) { *ok = false; break; }
    if (ba->pcg_it >= ba->pcg_maxit) break;
    const int batch = std::min(12, ba->pcg_maxit - ba->pcg_it);
    for (int k = 0; k < batch; ++k, ba->pcg_parity ^= 1) launch_pcg_iter(d, lambda, ba->pcg_tol2, ba->pcg_parity, s, ba->red);
    ba->pcg_it += batch;
    *again = true;
    int rc = fetch(ba);
    if (rc != VDO_OK) return rc;
  }
  *pcg_iters = ba->h_flags[2];
  ba->pcg_last = *ok ? *pcg_iters : 0;
  const bool small = 6 * (int64_t)d.P <= kDenseMaxUnknowns;
  // the next trials go to the dense solver: a slow PCG - or a TINY reduced system (a 20-frame window: 120 unknowns, two MFMA panels) on which a dozen mat-vec
  // round trips already cost more than assembling and factoring it
  const bool tiny = 6 * (int64_t)d.P <= kDenseTinyUnknowns;
  if (opt->solver == 0 && small && *ok && (*pcg_iters > kPcgSlow || (tiny && ba->dense_tiles_ok && *pcg_iters > kPcgSlowTiny))) ba->last_solver = 3;
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
  ba->lin_current = false;        // (the accepted steps move estimate[0] away from the last linearisation)
  ba->lin_exchange_pending = false;
  ba->dense_pending = false;
  ba->pcg_last = 0;
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
  // The chi2 of the start estimate is the chi2 of the first linearisation (the same kernels on the same estimate: the same bits, what the loop relies on for its accepted
  // trials): it is read with the first iteration's scalars instead of by an error evaluation + host round trip of its own (round 6; VDO_BA_LM_RECHECK=1: as before).
  static const bool chi_passes = std::getenv("VDO_BA_LM_RECHECK") != nullptr;
  if (chi_passes || opt->max_iterations <= 0) { CK(robust_chi2(ba, 0, &last_err_chi)); st->initial_chi2 = last_err_chi; }
  bool last_accepted = false;      // the last trial that ran was accepted: estimate[0] is its estimate, last_err_chi its robust chi2
  static const bool spec_off = std::getenv("VDO_BA_NO_SPEC_LIN") != nullptr;      // (A/B switch: the error-evaluation pass of rounds 1-5)
  const bool use_spec = !d.sharded && !spec_off && ensure_alt(ba);
  bool spec_lin = false;          // the system at estimate[0] is already on the device (left there by the accepted trial of the last iteration)
  int it = 0;
  for (; it < opt->max_iterations && !forceStop && ok; ++it) {
    double t0 = now_ms();
    static const bool sync_each = std::getenv("VDO_BA_LM_SYNC_EACH") != nullptr;      // (A/B: a host round trip after the linearisation and after the PCG batch, as before)
    // sharded, lambda known (every iteration but the first): the linearisation's all-reduce is left to the first trial's, which follows at once (one exchange instead of two)
    static const bool merge_off = std::getenv("VDO_BA_NO_EXCHANGE_MERGE") != nullptr;    // (A/B switch)
    const bool defer = d.sharded && it > 0 && !sync_each && !merge_off;
    // (spec_lin: the accepted trial of the last iteration was evaluated by a LINEARISATION at its estimate into the second set, and the sets changed places: the system is there)
    if (!spec_lin) launch_linearize(d, s, ba->red, defer);          // errors + buildSystem in one sweep (same estimate); its chi2 stays in S_LIN_RCHI2
    ba->lin_exchange_pending = defer && !spec_lin;
    // The chi2 of the linearisation is first NEEDED when the first trial is judged: it is read back with that trial's scalars (one host round
    // trip less per iteration).  Only the first iteration needs something before its first trial: the largest diagonal entry (lambda).
    bool have_lin = false;
    double currentChi = 0, tempChi = 0, iniChi = 0;
    if (spec_lin) { currentChi = tempChi = iniChi = last_err_chi; have_lin = true; spec_lin = false; }      // (the chi2 of that linearisation = the accepted trial's, same kernels on the same estimate)
    if (it == 0) {
      launch_max_diag(d, s, ba->red);
      CK(fetch(ba));
      lambda = tau * ba->h_scal[S_MAXDIAG]; ni = 2; nBad = 0;
      last_err_chi = currentChi = tempChi = iniChi = ba->h_scal[S_LIN_RCHI2];
      if (!(chi_passes || opt->max_iterations <= 0)) st->initial_chi2 = last_err_chi;
      have_lin = true;
    }
    if (sync_each && !have_lin) { CK(fetch(ba)); last_err_chi = currentChi = tempChi = iniChi = ba->h_scal[S_LIN_RCHI2]; have_lin = true; }
    st->ms_linearize += now_ms() - t0;
    double rho = 0;
    int qmax = 0;
    bool accepted = false;          // the last trial of this iteration was accepted: estimate[0] IS that trial's estimate
    do {
      t0 = now_ms();
      bool ok2 = true, pending = false, again = false;
      int pcg_it = 0;
      CK(solve_trial(ba, lambda, opt, &ok2, &pcg_it, &pending));
      if (sync_each && pending) { CK(fetch(ba)); CK(solve_trial_finish(ba, lambda, opt, &ok2, &pcg_it, &again)); pending = false; again = false; }
      const bool ortho = (++ba->oplus_calls > 1000);
      if (ortho) ba->oplus_calls = 0;
      launch_backsub_update(d, lambda, ortho, s);      // update() into the trial buffers (push/pop = keep [0])
      // The trial's errors.  Single GPU (round 6): by a full linearisation AT THE TRIAL ESTIMATE into the second set of buffers - its chi2 is the error evaluation's, bit for
      // bit (same kernels, same estimate), and an accepted trial - the rule on these graphs - has the next iteration's system already: the separate error pass (a sweep over
      // every edge + the pose-pose edges + a reduction per trial) is gone from every accepted iteration; a rejected trial paid a linearisation for an error evaluation.
      // (not in the last iteration the caller asked for: nothing would read that system)
      const bool spec_now = use_spec && it + 1 < opt->max_iterations;
      auto evaluate_trial = [&]() {
        if (spec_now) { BADev da = d; lin_swap(da); launch_linearize(da, s, ba->red, false, 1); }
        else launch_errors(d, 1, s, ba->red);
      };
      evaluate_trial();
      CK(fetch(ba));
      if (!have_lin) { currentChi = iniChi = ba->h_scal[S_LIN_RCHI2]; have_lin = true; }
      if (pending) {
        CK(solve_trial_finish(ba, lambda, opt, &ok2, &pcg_it, &again));
        if (again) {                                   // (the first batch of PCG iterations had not converged: update and errors once more, from the converged x)
          launch_backsub_update(d, lambda, ortho, s);
          evaluate_trial();
          CK(fetch(ba));
        }
      }
      st->ms_solve += now_ms() - t0;
      last_err_chi = tempChi = ba->h_scal[S_RCHI2];
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = ba->h_scal[S_SCALE] + 1e-3;
      rho /= scale;
      if (opt->verbose > 1)
        std::fprintf(stderr, "  trial %d lambda=%.4g pcg=%d%s chi2 %.9g -> %.9g rho=%.4g\n", qmax, lambda, pcg_it, again ? " (second look)" : "", currentChi, tempChi, rho);
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, upper);
        const double sf = std::max(lower, alpha);
        lambda *= sf; ni = 2; currentChi = tempChi;
        std::swap(d.pose[0], d.pose[1]);               // discardTop(): accept the trial
        std::swap(d.point[0], d.point[1]);
        if (spec_now) { lin_swap(d); spec_lin = true; }   // ... and its linearisation becomes the current one
        accepted = true;
      } else {
        lambda *= ni; ni *= 2;                          // pop(): estimate[0] untouched
        accepted = false;
        spec_lin = false;
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
    // The errors of estimate[0] for the trace and the gain rule (g2o evaluates them again after the iteration): when the iteration ended on an
    // accepted trial they are what that trial's evaluation just produced - the same kernels on the same estimate, the same bits (last_err_chi
    // holds them) - so the pass and its host round trip are spent only after an iteration whose last trial was rejected
    // (VDO_BA_LM_RECHECK=1: always, as before).
    const bool recheck = std::getenv("VDO_BA_LM_RECHECK") != nullptr;
    if ((opt->verbose || opt->gain_threshold >= 0) && (recheck || !accepted)) { CK(robust_chi2(ba, 0, &last_err_chi)); last_accepted = true; }      // (last_err_chi is estimate[0]'s again)
    else last_accepted = accepted;
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
  // ... and the chi2 of the final estimate is the last accepted trial's, when the run ended on one
  if (last_accepted && it > 0 && !chi_passes) st->final_chi2 = last_err_chi;
  else CK(robust_chi2(ba, 0, &st->final_chi2));
  st->ms_total = now_ms() - t_begin;
#undef CK
  return VDO_OK;
}
