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
#include <cstdlib>

#include "lm_dev.hpp"

namespace vdo {

struct Flow2Dev {        // one problem, device pointers into the batch arrays
  int n, max_iterations, ref_quirks, pad;
  int64_t off;           // element offset of this problem inside the per-landmark arrays
  int64_t out_off, out_total;   // outputs are packed by the ACTUAL sizes (one D2H copy of exactly what was computed): offset of this problem, sum over the batch
  double K[4], Twl[16], T0[16];
  double info_flow, info_prior, huber_delta, huber_dsqr, chi2_gate;
};

struct Flow2Arrays {
  const double* in;      // per problem at 5*off: key points [2][n], measured flow [2][n], depth [n]   (SoA planes); HBM, or - batches
                         // made by vdo_flow2_batch_reserve - the mapped pinned block vdo_flow2_batch_set writes (read once, in the setup)
  double* om;            // per problem at 4*off: HBM copy of key points + measured flow (what the sweeps read)
  double *Xw, *f0, *f1, *err, *B2a, *B2b, *hla, *hlb, *bla, *blb, *xl;
  char* out;             // [results n_problems][refined flows 2*out_total doubles, (x, y) per point][inlier flags out_total bytes]
  int n_problems;
  struct Flow2Comm* comm;   // [n_problems]: exchange area of the workgroup cluster of every problem
  unsigned int tag_base;    // launch number << 16: exchange tags of earlier launches never match, the area needs no clearing
  int max_cluster;          // workgroups per problem in this launch (<= F2_CLUSTER): lowered by the launch for very large batches
};

// One problem is spread over a CLUSTER of up to F2_CLUSTER workgroups (one CU each; a single wave needs ~8k cycles per
// correspondence and trial, so one workgroup with 5 correspondences per thread is 4x slower than five with one each).
// The workgroups exchange their block sums through this area (double-buffered by phase parity) and meet at a counter
// barrier; every workgroup then adds the partial sums in the same order, runs the same 6x6 solve and takes the same
// decisions - same bits everywhere, so the control flow (and the number of barriers) is identical across the cluster.
constexpr int F2_CLUSTER = 8;
static_assert(F2_THREADS >= 32 * F2_CLUSTER, "cluster exchange: one thread per (workgroup, value)");
// Exchange slot: one double as two {32-bit half, 32-bit phase tag} words.  8-byte stores are single-copy atomic, so a reader
// that finds both tags equal to the phase it waits for has the value of that phase - no fence, no separate flag, one round
// trip through L2 (the LL idea of RCCL's low-latency protocol).
struct Flow2Slot { unsigned int lo, tag_lo, hi, tag_hi; };
struct Flow2Comm {
  Flow2Slot slot[2][F2_CLUSTER][32];    // [phase & 1][workgroup][0..28 sums, 29 max, 30 Hll diagonal of the chunk's last landmark]
};
typedef unsigned int f2_u32x4 __attribute__((ext_vector_type(4)));
// 16-byte slot accesses of the exchange.  F2_XSCOPE: 3 = system scope (sc0 sc1: past every cache), 2 = agent scope (sc1: coherent
// across the XCDs' L2s), 1 = workgroup scope bits (sc0: misses the CU's L1, served by the XCD's L2 - valid only between
// workgroups of ONE XCD).
#ifndef F2_XSCOPE
#define F2_XSCOPE 2
#endif
#if F2_XSCOPE == 3
#define F2_SC "sc0 sc1"
#elif F2_XSCOPE == 2
#define F2_SC "sc1"
#else
#define F2_SC "sc0"
#endif
__device__ __forceinline__ void f2_slot_store(Flow2Slot* p, const f2_u32x4 w) { asm volatile("global_store_dwordx4 %0, %1, off " F2_SC :: "v"(p), "v"(w) : "memory"); }
__device__ __forceinline__ f2_u32x4 f2_slot_load(const Flow2Slot* p) {
  f2_u32x4 w;
  asm volatile("global_load_dwordx4 %0, %1, off " F2_SC "\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  return w;
}

// what the Schur sums of a trial leave in registers for the solve sweep of the same trial (first correspondence of a thread)
struct F2Pre { double B[12], b0, b1, d0, d1, d2, p2; int i; };

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
  // Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own L2: the workgroups of a cluster take
  // ids that are equal mod 8, so that the block sums they exchange stay inside one L2.
#ifdef F2_LINEAR_MAP
  const int prob = blockIdx.x / F2_CLUSTER, wg = blockIdx.x % F2_CLUSTER;
#else
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int wg = rest % F2_CLUSTER, prob = (rest / F2_CLUSTER) * 8 + xcd;
#endif
  if (prob >= A.n_problems) return;
  const Flow2Dev P = probs[prob];
  const int N = P.n, tid = threadIdx.x;
  const int Gp = min(A.max_cluster, max(1, (N + F2_THREADS - 1) / F2_THREADS));      // workgroups that share this problem
  if (wg >= Gp) return;
  __builtin_amdgcn_s_setprio(3);        // latency-bound persistent workgroups: issue ahead of the throughput kernels sharing the CU
  const int chunk = (N + Gp - 1) / Gp, c_lo = wg * chunk, c_hi = min(N, c_lo + chunk);   // this workgroup's correspondences [c_lo, c_hi): nothing but
  const int first = c_lo + tid, stride = F2_THREADS;                                      // the block sums (and one Hll entry) crosses workgroups
  Flow2Comm* comm = A.comm + prob;
  const int64_t off = P.off;
  const double* __restrict__ in_obs = A.in + 5 * off; const double* __restrict__ in_meas = in_obs + 2 * (size_t)N; const double* __restrict__ depth = in_obs + 4 * (size_t)N;
  double* __restrict__ obs = A.om + 4 * off; double* __restrict__ meas = obs + 2 * (size_t)N;
  double* __restrict__ Xw = A.Xw + 3 * off; double* fcur = A.f0 + 2 * off; double* ftry = A.f1 + 2 * off;
  double* __restrict__ err = A.err + 2 * off; double* __restrict__ xl = A.xl + 2 * off;
  double *Bc = A.B2a + 12 * off, *Bt = A.B2b + 12 * off, *hc = A.hla + off, *ht = A.hlb + off, *bc = A.bla + 2 * off, *bt = A.blb + 2 * off;   // current / trial linearisation
  vdo_flow2_result* res = (vdo_flow2_result*)A.out + prob;
  double* __restrict__ flow_out = (double*)(A.out + sizeof(vdo_flow2_result) * (size_t)A.n_problems) + 2 * P.out_off;
  unsigned char* __restrict__ inlier_out = (unsigned char*)((double*)(A.out + sizeof(vdo_flow2_result) * (size_t)A.n_problems) + 2 * P.out_total) + P.out_off;

  __shared__ double s_scr[F2_WAVES * 4], s_red[32];
  __shared__ double s_wpart[F2_WAVES * 32];   // wave totals of the block sums (block_reduce_bfly)
  __shared__ SE3d s_T, s_Ttry;
  __shared__ double s_Hc[27], s_xp[6], s_Hs[36], s_bs[6], s_xs[6];   // s_Hc: Hpp (lower triangle, packed) + bp of the current linearisation
  __shared__ double s_rho;
  __shared__ int s_ctrl[4];   // [1] failed solve: trial skipped, [2] ok2, [3] cluster exchange timed out
  __shared__ double s_hlast;  // Hll diagonal of the last landmark of this workgroup's chunk (last sweep)
  if (tid == 0) s_ctrl[3] = 0;
#ifdef F2_PROFILE
  __shared__ long long s_prof[16], s_tprev;
  if (tid == 0) { for (int i = 0; i < 16; ++i) s_prof[i] = 0; s_tprev = clock64(); }
#endif

  if (N < 3) {   // nInitialCorrespondences<3 -> identity, 0 inliers (Optimizer.cc:2449-2450, 2872-2873)
    if (tid < 16) res->T[tid] = (tid % 5 == 0) ? 1.0 : 0.0;
    if (tid < N) { inlier_out[tid] = 0; flow_out[2 * tid] = in_meas[tid]; flow_out[2 * tid + 1] = in_meas[N + tid]; }   // nothing optimised: flows stay as measured
    if (tid == 0) { res->n_inliers = 0; res->iterations = 0; res->trials = 0; res->stop_reason = 0; res->initial_chi2 = res->final_chi2 = res->final_lambda = 0; }
    return;
  }
  const double fx = P.K[0], fy = P.K[1], cx = P.K[2], cy = P.K[3];
  const bool Q = P.ref_quirks != 0;
  // ---- setup: Xw, flows, initial pose (Converter::toSE3Quat)
  for (int i = first; i < c_hi; i += stride) {
    const double dz = depth[i];
    const double o0 = in_obs[i], o1 = in_obs[N + i], m0 = in_meas[i], m1 = in_meas[N + i];
    obs[i] = o0; obs[N + i] = o1; meas[i] = m0; meas[N + i] = m1;
    const double x = (o0 - cx) * dz / fx, y = (o1 - cy) * dz / fy;
    const double* W = P.Twl;
    Xw[i] = W[0] * x + W[1] * y + W[2] * dz + W[3];
    Xw[N + i] = W[4] * x + W[5] * y + W[6] * dz + W[7];
    Xw[2 * N + i] = W[8] * x + W[9] * y + W[10] * dz + W[11];
    fcur[i] = m0; fcur[N + i] = m1;
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

  // Cluster-wide sums: s_red[0..K) holds this workgroup's block sums on entry, the sums over the cluster on return (added in
  // workgroup order by everybody); mx: block maximum in, cluster maximum out; hb: Hll diagonal of this chunk's last landmark
  // in, of the PREVIOUS chunk's last landmark out (the F3 aliasing couples neighbours).  One exchange = one barrier.
  int phase = 0;
  __shared__ double s_part[F2_CLUSTER][32];
  auto cluster_sum = [&](const int K, double& mx, double& hb) -> bool {
    if (Gp == 1) return true;
    const unsigned tag = A.tag_base + (unsigned)phase + 1u;
    if (tid < 31) {
      const double v = tid < K ? s_red[tid] : (tid == 29 ? mx : (tid == 30 ? hb : 0.0));
      const unsigned long long u = (unsigned long long)__double_as_longlong(v);
      const f2_u32x4 w = {(unsigned)u, tag, (unsigned)(u >> 32), tag};
      f2_slot_store(&comm->slot[phase & 1][wg][tid], w);
    }
    {
      const int g2 = tid >> 5, q = tid & 31;                // one thread per (workgroup, value)
      double v = 0.0;
      int bad = 0;
      if (g2 < Gp && q < 31) {
        const Flow2Slot* src = &comm->slot[phase & 1][g2][q];
        f2_u32x4 w = f2_slot_load(src);
        int spins = 0;
        while (w.y != tag || w.w != tag) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 20)) { bad = 1; break; }      // never hang the GPU: the launch fails instead (vdo_flow2_batch_fetch reports it)
          w = f2_slot_load(src);
        }
        v = __longlong_as_double((long long)(((unsigned long long)w.z << 32) | (unsigned long long)w.x));
      }
      if (g2 < F2_CLUSTER) s_part[g2][q] = v;
      if (bad) s_ctrl[3] = 1;
    }
    __syncthreads();
    if (s_ctrl[3]) return false;
    if (tid < K) {
      double a = 0.0;
      for (int g2 = 0; g2 < Gp; ++g2) a += s_part[g2][tid];
      s_red[tid] = a;
    }
    if (tid == 32) {
      double m = 0.0;
      for (int g2 = 0; g2 < Gp; ++g2) m = fmax(m, s_part[g2][29]);
      s_red[31] = m;
      s_red[30] = wg > 0 ? s_part[wg - 1][30] : 0.0;
    }
    __syncthreads();
    mx = s_red[31]; hb = s_red[30];
    ++phase;
    return true;
  };
#define F2_CLUSTER_SUM(K, mx, hb) do { if (!cluster_sum(K, mx, hb)) { if (wg == 0 && tid == 0) { res->stop_reason = -1; res->iterations = -1; } return; } } while (0)

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
                   double* __restrict__ Bw, double* __restrict__ hw, double* __restrict__ bw, const double* fin, double* fout, const double hb_prev, const F2Pre& pre) -> double {
    constexpr bool TRIAL = decltype(trial_c)::value;
    const SE3d T = TRIAL ? s_Ttry : s_T;
    double xp[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) xp[j] = s_xp[j];
    double acc[29];
#pragma unroll
    for (int i = 0; i < 29; ++i) acc[i] = 0.0;
    double hmax = 0.0;
    for (int i = first; i < c_hi; i += stride) {
      double f0v, f1v;
      if (TRIAL) {
        // back-substitution c = b_l - B^T x_p for this landmark (first correspondence of the thread: B, b_l and D^-1 are
        // still in registers from the Schur sums of this trial - same loads, same lambda, same divisions)
        const bool hp = Q && i == pre.i;
        double Bl[12], b0, b1;
        if (hp) {
#pragma unroll
          for (int a = 0; a < 12; ++a) Bl[a] = pre.B[a];
          b0 = pre.b0; b1 = pre.b1;
        } else {
          const double* B = Br + i;
#pragma unroll
          for (int a = 0; a < 12; ++a) Bl[a] = B[a * N];
          b0 = br[i]; b1 = br[N + i];
        }
        double t0 = 0, t1 = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) { t0 += Bl[2 * a] * (-xp[a]); t1 += Bl[2 * a + 1] * (-xp[a]); }
        const double c0 = b0 + t0, c1 = b1 + t1;
        double x0, x1;
        if (Q) {
          // x[2i..2i+2] = D_i^-1 c_i with the aliased 3x3 block (dinv_q): rows 0/1 give this landmark's flow update, row 2 of
          // landmark i-1 (= its d2 * c0 of THIS landmark: the aliased third component) was written into slot 2i first
          double d0, d1, d2, p2_ = 0;
          if (hp) { d0 = pre.d0; d1 = pre.d1; d2 = pre.d2; p2_ = pre.p2; }
          else {
            dinv_q(hr[i], lam, d0, d1, d2);
            if (i > 0) { double p0_, p1_; dinv_q(i > c_lo ? hr[i - 1] : hb_prev, lam, p0_, p1_, p2_); }      // (the previous chunk's last landmark belongs to another workgroup)
          }
          x0 = d0 * c0 + d1 * c1;
          x1 = d2 * c1;
          if (i > 0) x0 = p2_ * c0 + x0;
        } else {
          double Di[9];
          dinv_of(hr[i], lam, Di);
          x0 = (Di[0] * c0 + Di[1] * c1) + Di[2] * 0.0;
          x1 = (Di[3] * c0 + Di[4] * c1) + Di[5] * 0.0;
        }
        F2_TICK(8);
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
      if (TRIAL) F2_TICK(9);
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
      if (i == c_hi - 1) s_hlast = h;
      if (TRIAL) F2_TICK(10);
    }
    block_reduce_bfly<29>(acc, s_wpart, s_red);
    if (TRIAL) F2_TICK(11);
    return hmax;
  };

  double lambda = -1, ni = 2;
  int nBad = 0, it = 0, total_trials = 0, stop_reason = 0;
  const double tau = 1e-5, upper = 2. / 3., lower = 1. / 3.;
  double chi2_check = 0;
  // initial computeActiveErrors + buildSystem
  double hmax = sweep(std::false_type{}, 0.0, true, nullptr, nullptr, nullptr, Bc, hc, bc, fcur, nullptr, 0.0, F2Pre{});
#pragma unroll
  for (int off2 = 32; off2 > 0; off2 >>= 1) hmax = fmax(hmax, __shfl_down(hmax, off2, 64));
  if ((tid & 63) == 0) s_scr[tid >> 6] = hmax;
  __syncthreads();
  double hmx = s_scr[0];
  for (int w = 1; w < F2_WAVES; ++w) hmx = fmax(hmx, s_scr[w]);
  double hb_cur = s_hlast, hb_try = 0.0;     // -> Hll diagonal of the landmark just before this chunk (current / trial linearisation)
  F2_CLUSTER_SUM(29, hmx, hb_cur);
  double last_err_chi = s_red[27];
  const double initial_chi2 = last_err_chi;
  if (tid < 27) s_Hc[tid] = s_red[tid];
  __syncthreads();
  {
    // computeLambdaInit: max |H(j,j)| over pose and flow vertices
    double mm = hmx;
    for (int j = 0; j < 6; ++j) mm = fmax(mm, fabs(s_Hc[j * (j + 3) / 2]));
    lambda = tau * mm; ni = 2; nBad = 0;
  }
  F2_TICK(4);
  bool built = true;
  bool ok = true;
  for (; it < P.max_iterations && ok; ++it) {
    // computeActiveErrors + buildSystem at the current estimate: already there after an accepted trial; repeated
    // only when the previous trial was rejected without ending the iteration loop (non-finite chi2)
    if (!built) {
      sweep(std::false_type{}, 0.0, true, nullptr, nullptr, nullptr, Bc, hc, bc, fcur, nullptr, 0.0, F2Pre{});
      { double d_ = 0; hb_cur = s_hlast; F2_CLUSTER_SUM(29, d_, hb_cur); }
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
      F2Pre pre;
      pre.i = -1;
      {
        double acc[28];      // 27 Schur sums + [27] the landmark part of computeScale for the STALE x (what a failed solve leaves behind)
#pragma unroll
        for (int i = 0; i < 28; ++i) acc[i] = 0.0;
        const double* __restrict__ Br = Bc; const double* __restrict__ hr = hc; const double* __restrict__ br = bc;
        for (int i = first; i < c_hi; i += stride) {
          const double bl0 = br[i], bl1 = br[N + i];
          { const double xs0 = xl[i], xs1 = xl[N + i]; acc[27] += xs0 * (lambda * xs0 + bl0) + xs1 * (lambda * xs1 + bl1); }
          const double* B = Br + i;
          double Bv[12];
#pragma unroll
          for (int a = 0; a < 12; ++a) Bv[a] = B[a * N];
          if (Q) {
            double d0, d1, d2;
            dinv_q(hr[i], lambda, d0, d1, d2);
            const double db0 = d0 * bl0 + d1 * bl1, db1 = d2 * bl1;      // (the aliased third row/column only ever meets exact zeros)
            if (i == first) {
#pragma unroll
              for (int a = 0; a < 12; ++a) pre.B[a] = Bv[a];
              pre.b0 = bl0; pre.b1 = bl1; pre.d0 = d0; pre.d1 = d1; pre.d2 = d2; pre.p2 = 0.0; pre.i = i;
              if (i > 0) { double p0_, p1_; dinv_q(i > c_lo ? hr[i - 1] : hb_cur, lambda, p0_, p1_, pre.p2); }
            }
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
        F2_TICK(12);
        block_reduce_bfly<28>(acc, s_wpart, s_red);
        F2_TICK(13);
        { double d_ = 0, e_ = 0; F2_CLUSTER_SUM(28, d_, e_); }
      }
      F2_TICK(0);
      // ---- (2) reduced 6x6 system, SE3 update, pose part of computeScale
      // reduced system: lower triangle Hpp - (Schur sums) + lambda I, rhs bp - (Schur sums); one thread per entry
      if (tid < 36) {
        const int a = tid / 6, c2 = tid - 6 * a;
        if (c2 <= a) {
          const int k = a * (a + 1) / 2 + c2;
          double v = s_Hc[k] - s_red[k];
          if (c2 == a) v += lambda;
          s_Hs[tid] = v;
        } else {
          s_Hs[tid] = s_Hc[c2 * (c2 + 1) / 2 + a];
        }
      } else if (tid < 42) {
        const int j = tid - 36;
        s_bs[j] = s_Hc[21 + j] - s_red[21 + j];
      }
      __syncthreads();
      F2_TICK(5);
      bool ok2w = false;
      if (tid < 64) ok2w = ldlt6_solve_lanes(s_Hs, s_bs, s_xs);      // wave 0: one row of the 6x6 system per lane
      if (tid == 0) {
        const bool ok2 = ok2w;
        F2_TICK(6);
        s_ctrl[2] = ok2 ? 1 : 0;
        if (ok2) {
#pragma unroll
          for (int j = 0; j < 6; ++j) s_xp[j] = s_xs[j];
        }
        // (failed LDLT leaves x untouched in the reference; the trial is rejected anyway)
        double s = 0;
        for (int j = 0; j < 6; ++j) s += s_xp[j] * (lambda * s_xp[j] + s_Hc[21 + j]);
        s_rho = s;    // pose part of computeScale
        // A failed solve rejects the trial whatever its errors are (tempChi = DBL_MAX) as long as computeScale - known here: the
        // stale x against the current gradient - is positive, and everything the evaluation would leave behind (errors, chi2,
        // the trial linearisation) is overwritten by the trial that follows: skip the SE3 update and the sweep.  Not for the
        // last trial of an iteration (its errors are the ones classified).
        bool skip = false;
        if (!ok2 && qmax + 1 < 10) skip = (currentChi - 1.7976931348623157e308) / ((s + s_red[27]) + 1e-3) < 0;
        s_ctrl[1] = skip ? 1 : 0;
        if (!skip) s_Ttry = se3_exp_compose(s_xp, s_T);
        F2_TICK(7);
      }
      __syncthreads();
      F2_TICK(1);
      const bool ok2 = s_ctrl[2] != 0;
      if (s_ctrl[1]) {
        rho = (currentChi - 1.7976931348623157e308) / ((s_rho + s_red[27]) + 1e-3);
        lambda *= ni; ni *= 2; built = false;
        ++qmax; ++total_trials;
        continue;
      }
      // ---- (3) finish the solve per correspondence, errors + speculative linearisation at the trial point
      sweep(std::true_type{}, lambda, ok2, Bc, hc, bc, Bt, ht, bt, fcur, ftry, hb_cur, pre);
      { double d_ = 0; hb_try = s_hlast; F2_CLUSTER_SUM(29, d_, hb_try); }
      last_err_chi = tempChi = s_red[27];
      const double scale = (s_rho + s_red[28]) + 1e-3;
      F2_TICK(2);
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - cube_rn(2 * rho - 1);
        alpha = fmin(alpha, upper);
        lambda *= fmax(lower, alpha); ni = 2; currentChi = tempChi; built = true;
        { double* t_ = fcur; fcur = ftry; ftry = t_; }                       // discardTop(): accept (uniform pointer swaps)
        { double* t_ = Bc; Bc = Bt; Bt = t_; t_ = hc; hc = ht; ht = t_; t_ = bc; bc = bt; bt = t_; hb_cur = hb_try; }
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
  for (int i = first; i < c_hi; i += stride) {
    const double e0 = err[i], e1 = err[N + i];
    const float chi2 = (float)(e0 * (P.info_flow * e0) + e1 * (P.info_flow * e1));
    const bool outl = chi2 > gate;
    inlier_out[i] = outl ? 0 : 1;
    cnt[0] += outl ? 0.0 : 1.0;
    flow_out[2 * i] = fcur[i];
    flow_out[2 * i + 1] = fcur[N + i];
  }
  block_reduce<1>(cnt, s_scr, s_red);
  { double d_ = 0, e_ = 0; F2_CLUSTER_SUM(1, d_, e_); }
  if (tid == 0 && wg == 0) {
    se3_to_matrix(s_T, res->T);
    res->n_inliers = (int)(s_red[0] + 0.5);
    res->iterations = it; res->trials = total_trials; res->stop_reason = stop_reason;
    res->initial_chi2 = initial_chi2; res->final_chi2 = last_err_chi; res->final_lambda = lambda;
#ifdef F2_PROFILE
    for (int i = 0; i < 14; ++i) res->T[i] = (double)s_prof[i];     // cycles per phase instead of the pose (debug build only)
#endif
  }
#ifdef F2_PROFILE
  __syncthreads();
  if (tid == 0 && wg < 2) res->T[14 + wg] = (double)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf) + 100.0 * blockIdx.x;   // XCC_ID of workgroups 0 / 1 (+ 100 x block id)
#endif
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
  unsigned run_seq = 0;           // launches so far (tags of the cluster exchange)
  const Flow2Dev* d_probs_run = nullptr;   // descriptors the kernel reads: d_probs, or the mapped pinned copy (reserved batches)
};

extern "C" int vdo_flow2_batch_destroy(vdo_flow2_batch* b) {
  if (!b) return VDO_OK;
  if (b->ctx) { ctx_bind(b->ctx); hipStreamSynchronize(b->ctx->stream); }      // (a launch may still be in flight: the camera stage runs one frame ahead)
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
    d.out_off = total;
    b->offs.push_back(total); b->ns.push_back(p.n);
    total += p.n;
  }
  for (Flow2Dev& d : hp) d.out_total = total;
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
  A.om = (double*)dev(32 * T);
  A.Xw = (double*)dev(24 * T); A.f0 = (double*)dev(16 * T); A.f1 = (double*)dev(16 * T);
  A.err = (double*)dev(16 * T); A.B2a = (double*)dev(96 * T); A.B2b = (double*)dev(96 * T);
  A.hla = (double*)dev(8 * T); A.hlb = (double*)dev(8 * T); A.bla = (double*)dev(16 * T); A.blb = (double*)dev(16 * T);
  A.xl = (double*)dev(16 * T);
  A.n_problems = n_problems;
  A.comm = (Flow2Comm*)dev(sizeof(Flow2Comm) * NP);
  if (A.comm) hipMemsetAsync(A.comm, 0, sizeof(Flow2Comm) * NP, s);
  for (void* p : b->allocs) if (!p) { vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  if (!d_in) { vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  hipMemcpyAsync(d_in, in.data(), 40 * T, hipMemcpyHostToDevice, s);
  hipMemcpyAsync(b->d_probs, hp.data(), sizeof(Flow2Dev) * NP, hipMemcpyHostToDevice, s);
  // The outputs (results, refined flows, inlier flags: ~17 B per correspondence) are written by the kernel straight into
  // pinned, device-mapped host memory: fetching them is a stream synchronisation, not a copy - a D2H copy queued behind a
  // running LM kernel would also hold up the copies of the other streams on the same SDMA queue for the kernel's duration.
  if (hipHostMalloc((void**)&b->h_pin, sizeof(vdo_flow2_result) * NP + 16 * T + T + 64, hipHostMallocMapped) != hipSuccess) { b->h_pin = nullptr; vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_OOM, "hipHostMalloc failed"); }
  {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, b->h_pin, 0) != hipSuccess || !dp) { vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_NO_DEVICE, "hipHostGetDevicePointer failed"); }
    A.out = (char*)dp;
  }
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
  // Inputs and descriptors of a reserved batch live in MAPPED pinned memory: vdo_flow2_batch_set writes them, the kernel reads
  // them through the mapping (once, in its setup: 40 B per correspondence) - defining the problems of a frame costs no copy.
  if (hipHostMalloc((void**)&b->h_up, sizeof(double) * 5 * (size_t)std::max<int64_t>(tot, 1) + sizeof(Flow2Dev) * (size_t)n_problems, hipHostMallocMapped) != hipSuccess) { b->h_up = nullptr; vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_OOM, "hipHostMalloc failed"); }
  void* dp = nullptr;
  if (hipHostGetDevicePointer(&dp, b->h_up, 0) != hipSuccess || !dp) { vdo_flow2_batch_destroy(b); return set_error(VDO_ERR_NO_DEVICE, "hipHostGetDevicePointer failed"); }
  b->A.in = (const double*)dp;
  b->d_probs_run = (const Flow2Dev*)((const double*)dp + 5 * (size_t)std::max<int64_t>(tot, 1));
  for (int k = 0; k < n_problems; ++k) { b->hp[k].n = 0; b->ns[k] = 0; }
  std::memcpy(b->h_up + 5 * (size_t)std::max<int64_t>(tot, 1), b->hp.data(), sizeof(Flow2Dev) * (size_t)n_problems);
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
    (void)s;                               // (the kernel reads the slot through the mapping)
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
    int64_t used = 0;
    for (int k = 0; k < b->n_problems; ++k) { b->hp[k].out_off = used; used += b->hp[k].n; }
    for (int k = 0; k < b->n_problems; ++k) b->hp[k].out_total = used;
    std::memcpy(b->h_up + 5 * (size_t)std::max<int64_t>(b->total, 1), b->hp.data(), sizeof(Flow2Dev) * (size_t)b->n_problems);   // mapped: no copy
    b->probs_dirty = false;
  }
  b->A.tag_base = (++b->run_seq) << 16;
  // The workgroups of a cluster wait for each other, so all of them must be resident at once: keep their number below what
  // the device can hold (2 workgroups per CU by LDS; half of that is left to whatever else is running) by lowering the
  // cluster size for very large batches - a single-workgroup "cluster" waits for nobody.
  {
    int budget = 256;
    if (const char* e = std::getenv("VDO_LM_CLUSTER_BUDGET")) budget = std::max(1, std::atoi(e));
    int mc = F2_CLUSTER;
    for (; mc > 1; mc >>= 1) {
      int64_t wgs = 0;
      for (int k = 0; k < b->n_problems; ++k) if (b->ns[k] >= 3) wgs += std::min(mc, std::max(1, (b->ns[k] + F2_THREADS - 1) / F2_THREADS));
      if (wgs <= budget) break;
    }
    b->A.max_cluster = mc;
  }
  hipLaunchKernelGGL(k_flow2_lm, dim3(((b->n_problems + 7) / 8) * 8 * F2_CLUSTER), dim3(F2_THREADS), 0, b->ctx->stream, b->d_probs_run ? b->d_probs_run : (const Flow2Dev*)b->d_probs, b->A);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "k_flow2_lm launch: %s", hipGetErrorString(e));
  return VDO_OK;
}

extern "C" int vdo_flow2_batch_fetch(vdo_flow2_batch* b, vdo_flow2_result* results, double** flow_out, uint8_t** inlier_out) {
  if (!b || !results) return set_error(VDO_ERR_INVALID, "null argument");
  int rc = ctx_bind(b->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = b->ctx->stream;
  // results, refined flows and inlier flags are packed by the actual problem sizes in the pinned block the kernel wrote
  const size_t NP = (size_t)b->n_problems;
  size_t used = 0;
  for (int k = 0; k < b->n_problems; ++k) used += (size_t)b->ns[k];
  const bool want_pts = (flow_out || inlier_out) && used;
  const size_t bytes = sizeof(vdo_flow2_result) * NP + (want_pts ? 17 * used : 0);
  (void)bytes;                                             // (already in host memory: the kernel wrote through the mapping)
  hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "flow2 fetch: %s", hipGetErrorString(e));
  const vdo_flow2_result* pr = (const vdo_flow2_result*)b->h_pin;
  const double* pf = (const double*)(b->h_pin + sizeof(vdo_flow2_result) * NP);
  const uint8_t* pi = (const uint8_t*)(pf + 2 * used);
  std::memcpy(results, pr, sizeof(vdo_flow2_result) * NP);
  for (size_t k = 0; k < NP; ++k) if (pr[k].iterations < 0) return set_error(VDO_ERR_NO_DEVICE, "k_flow2_lm: the workgroup cluster of problem %d lost its barrier", (int)k);
  size_t o = 0;
  for (int k = 0; k < b->n_problems; ++k) {
    const size_t n = (size_t)b->ns[k];
    if (n && flow_out && flow_out[k]) std::memcpy(flow_out[k], pf + 2 * o, sizeof(double) * 2 * n);
    if (n && inlier_out && inlier_out[k]) std::memcpy(inlier_out[k], pi + o, n);
    o += n;
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
