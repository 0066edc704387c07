// TEST INFRASTRUCTURE — CPU oracle (see vdo_oracle.h).  Batch dynamic-SLAM
// bundle adjustment: the arithmetic g2o performs for the graphs built by
// Optimizer::FullBatchOptimization (src/Optimizer.cc:1232-2175) and
// Optimizer::PartialBatchOptimization (:42-1230).
//
//  * linearisation + quadratic form   g2o/core/block_solver.hpp:502-560,
//        base_unary_edge.hpp:43-72, base_binary_edge.hpp:55-120, base_multi_edge.hpp:36-48,171-222
//  * robust weighting (rho' * Omega)    g2o/core/base_edge.h:96-102, robust_kernel_impl.cpp:65-91
//  * Levenberg–Marquardt incl. ORB-SLAM2 stop rule   g2o/core/optimization_algorithm_levenberg.cpp:61-189
//  * outer loop incl. "chi2 increased" abort         g2o/core/sparse_optimizer.cpp:354-443
//  * terminate action (gain threshold)                g2o/core/sparse_optimizer_terminate_action.cpp:49-85
//  * linear solve: sparse Cholesky on the WHOLE system (no Schur: BlockSolverX with
//    _doSchur=false because no vertex is marginalised)  — sparse_chol.hpp
//
// Summation order: edges are visited class by class (prior, EdgeSE3, binary, ternary)
// instead of by g2o edge id; this changes results only at rounding level.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <limits>
#include <unordered_map>
#include <vector>

#include "ref_edges.hpp"
#include "sparse_chol.hpp"
#include "vdo_oracle.h"

using namespace vdo_oracle;

namespace {

struct BlockRef { int64_t pos; bool transposed; };  // pos: offset inside a scalar column; see Sys

struct BA {
  const vdo_ba_graph* g;
  int L, P, n;
  std::vector<Iso> pose;
  std::vector<V3> point;
  std::vector<int> oplusCalls;
  Huber hub_eb, hub_et, hub_ep;
  bool use_eb, use_et, use_ep;

  // scalar CSC (upper block triangle, full diagonal blocks)
  std::vector<int64_t> Ap;
  std::vector<int> Ai;
  std::vector<double> Ax, b, x, diag_backup;
  // block bookkeeping
  std::vector<int> bdim, boff;  // per block column
  struct Blk { int rb, cb; int64_t rowpos; };
  std::vector<Blk> blocks;
  std::unordered_map<uint64_t, int> blkmap;
  std::vector<int> eb_blk, et_blk12, et_blk1h, et_blk2h, ep_blk;
  SparseChol chol;
  bool analyzed = false;

  int var_point(int l) const { return l; }
  int var_pose(int p) const { return L + p; }

  explicit BA(const vdo_ba_graph* gg) : g(gg) {
    L = g->n_point; P = g->n_pose;
    pose.resize(P); point.resize(L); oplusCalls.assign(P, 0);
    for (int p = 0; p < P; ++p) pose[p] = iso_from12(g->pose + 12 * p);
    for (int l = 0; l < L; ++l) point[l] = v3(g->point[3 * l], g->point[3 * l + 1], g->point[3 * l + 2]);
    use_eb = g->huber_eb > 0; use_et = g->huber_et > 0; use_ep = g->huber_ep > 0;
    if (use_eb) hub_eb.setDelta(g->huber_eb);
    if (use_et) hub_et.setDelta(g->huber_et);
    if (use_ep) hub_ep.setDelta(g->huber_ep);
    build_structure();
  }

  int get_block(int r, int c) {
    if (r > c) std::swap(r, c);
    uint64_t key = (uint64_t)(uint32_t)r << 32 | (uint32_t)c;
    auto it = blkmap.find(key);
    if (it != blkmap.end()) return it->second;
    int id = (int)blocks.size();
    blocks.push_back({r, c, 0});
    blkmap.emplace(key, id);
    return id;
  }

  void build_structure() {
    int nb = L + P;
    bdim.resize(nb); boff.resize(nb + 1);
    for (int i = 0; i < nb; ++i) bdim[i] = i < L ? 3 : 6;
    boff[0] = 0;
    for (int i = 0; i < nb; ++i) boff[i + 1] = boff[i] + bdim[i];
    n = boff[nb];
    for (int i = 0; i < nb; ++i) get_block(i, i);
    eb_blk.resize(g->n_eb);
    for (int e = 0; e < g->n_eb; ++e) eb_blk[e] = get_block(var_point(g->eb_point[e]), var_pose(g->eb_pose[e]));
    et_blk12.resize(g->n_et); et_blk1h.resize(g->n_et); et_blk2h.resize(g->n_et);
    for (int e = 0; e < g->n_et; ++e) {
      et_blk12[e] = get_block(var_point(g->et_p1[e]), var_point(g->et_p2[e]));
      et_blk1h[e] = get_block(var_point(g->et_p1[e]), var_pose(g->et_pose[e]));
      et_blk2h[e] = get_block(var_point(g->et_p2[e]), var_pose(g->et_pose[e]));
    }
    ep_blk.resize(g->n_ep);
    for (int e = 0; e < g->n_ep; ++e) ep_blk[e] = get_block(var_pose(g->ep_i[e]), var_pose(g->ep_j[e]));
    // column structure
    std::vector<std::vector<int>> colblocks(nb);
    for (int id = 0; id < (int)blocks.size(); ++id) colblocks[blocks[id].cb].push_back(id);
    Ap.assign(n + 1, 0);
    std::vector<int64_t> colnnz(nb, 0);
    for (int c = 0; c < nb; ++c) {
      auto& v = colblocks[c];
      std::sort(v.begin(), v.end(), [&](int a, int b2) { return blocks[a].rb < blocks[b2].rb; });
      int64_t pos = 0;
      for (int id : v) { blocks[id].rowpos = pos; pos += bdim[blocks[id].rb]; }
      colnnz[c] = pos;
    }
    for (int c = 0; c < nb; ++c)
      for (int j = 0; j < bdim[c]; ++j) Ap[boff[c] + j + 1] = Ap[boff[c] + j] + colnnz[c];
    Ai.resize(Ap[n]);
    for (int c = 0; c < nb; ++c)
      for (int j = 0; j < bdim[c]; ++j) {
        int64_t base = Ap[boff[c] + j];
        for (int id : colblocks[c])
          for (int i = 0; i < bdim[blocks[id].rb]; ++i) Ai[base + blocks[id].rowpos + i] = boff[blocks[id].rb] + i;
      }
    Ax.assign(Ap[n], 0.0);
    b.assign(n, 0.0); x.assign(n, 0.0);
  }

  // add M (dr x dc row-major, for block (rvar, cvar)) into the system; handles transposition
  inline void add_block(int blk, int rvar, int cvar, const double* M, int dr, int dc) {
    const Blk& B = blocks[blk];
    if (B.rb == rvar && (rvar != cvar || true) && B.cb == cvar) {
      for (int j = 0; j < dc; ++j) {
        double* col = &Ax[Ap[boff[cvar] + j] + B.rowpos];
        for (int i = 0; i < dr; ++i) col[i] += M[i * dc + j];
      }
    } else {  // stored as (cvar, rvar): add M^T
      for (int j = 0; j < dr; ++j) {
        double* col = &Ax[Ap[boff[rvar] + j] + B.rowpos];
        for (int i = 0; i < dc; ++i) col[i] += M[j * dc + i];
      }
    }
  }
  inline void add_diag(int var, const double* M, int d) { add_block(blkmap[(uint64_t)(uint32_t)var << 32 | (uint32_t)var], var, var, M, d, d); }

  // ---- errors ---------------------------------------------------------------------
  // computeActiveErrors + activeRobustChi2 (sparse_optimizer.cpp:61-114)
  double compute_errors(double* plain_chi2 = nullptr) const {
    double rchi = 0, chi = 0;
    double e[6];
    for (int k = 0; k < g->n_prior; ++k) {
      edge_prior(iso_from12(g->pr_z + 12 * k), pose[g->pr_pose[k]], e, nullptr);
      double c = chi2_6(e, g->pr_info + 36 * k);
      chi += c; rchi += c;
    }
    for (int k = 0; k < g->n_ep; ++k) {
      edge_se3(iso_from12(g->ep_z + 12 * k), pose[g->ep_i[k]], pose[g->ep_j[k]], e, nullptr, nullptr);
      double c = chi2_6(e, g->ep_info + 36 * k);
      chi += c;
      if (use_ep) { double r0, r1; hub_ep.robustify(c, r0, r1); rchi += r0; } else rchi += c;
    }
    for (int k = 0; k < g->n_eb; ++k) {
      V3 z{g->eb_z[k], g->eb_z[g->n_eb + k], g->eb_z[2 * (int64_t)g->n_eb + k]};
      edge_eb(pose[g->eb_pose[k]], point[g->eb_point[k]], z, e, nullptr, nullptr);
      double c = chi2_3(e, g->eb_w[k]);
      chi += c;
      if (use_eb) { double r0, r1; hub_eb.robustify(c, r0, r1); rchi += r0; } else rchi += c;
    }
    for (int k = 0; k < g->n_et; ++k) {
      V3 z{g->et_z[k], g->et_z[g->n_et + k], g->et_z[2 * (int64_t)g->n_et + k]};
      edge_et(pose[g->et_pose[k]], point[g->et_p1[k]], point[g->et_p2[k]], z, e, nullptr, nullptr, nullptr);
      double c = chi2_3(e, g->et_w[k]);
      chi += c;
      if (use_et) { double r0, r1; hub_et.robustify(c, r0, r1); rchi += r0; } else rchi += c;
    }
    if (plain_chi2) *plain_chi2 = chi;
    return rchi;
  }
  // BaseEdge::chi2: e . (Omega e)
  static double chi2_6(const double e[6], const double* info) {
    double s = 0;
    for (int i = 0; i < 6; ++i) { double t = 0; for (int j = 0; j < 6; ++j) t += info[i * 6 + j] * e[j]; s += e[i] * t; }
    return s;
  }
  static double chi2_3(const double e[3], double w) { return e[0] * (w * e[0]) + e[1] * (w * e[1]) + e[2] * (w * e[2]); }

  // ---- buildSystem ----------------------------------------------------------------
  // A^T W B for small row-major blocks: out(da x db) = Ja^T (w*I or Omega) Jb
  template <int D, int DA, int DB>
  static void JtWJ(const double* Ja, const double* W /*DxD or null*/, double w, const double* Jb, double* out) {
    double WJb[D * DB];
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < DB; ++j) {
        if (W) { double s = 0; for (int k = 0; k < D; ++k) s += W[i * D + k] * Jb[k * DB + j]; WJb[i * DB + j] = w * s; }
        else WJb[i * DB + j] = w * Jb[i * DB + j];
      }
    for (int a = 0; a < DA; ++a)
      for (int c = 0; c < DB; ++c) { double s = 0; for (int i = 0; i < D; ++i) s += Ja[i * DA + a] * WJb[i * DB + c]; out[a * DB + c] = s; }
  }
  template <int D, int DA>
  static void Jtr(const double* Ja, const double* r, double* out) {
    for (int a = 0; a < DA; ++a) { double s = 0; for (int i = 0; i < D; ++i) s += Ja[i * DA + a] * r[i]; out[a] = s; }
  }

  void build_system(vdo_ba_system* out = nullptr) {
    std::fill(Ax.begin(), Ax.end(), 0.0);
    std::fill(b.begin(), b.end(), 0.0);
    double e[6], Ji[36], Jj[36], H[36], r[6], g6[6];
    // EdgeSE3Prior: no robust kernel (BaseUnaryEdge::constructQuadraticForm)
    for (int k = 0; k < g->n_prior; ++k) {
      int v = g->pr_pose[k];
      const double* info = g->pr_info + 36 * k;
      edge_prior(iso_from12(g->pr_z + 12 * k), pose[v], e, Ji);
      for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += info[i * 6 + j] * e[j]; r[i] = -s; }
      JtWJ<6, 6, 6>(Ji, info, 1.0, Ji, H);
      Jtr<6, 6>(Ji, r, g6);
      add_diag(var_pose(v), H, 6);
      for (int i = 0; i < 6; ++i) b[boff[var_pose(v)] + i] += g6[i];
    }
    for (int k = 0; k < g->n_ep; ++k) {
      int vi = g->ep_i[k], vj = g->ep_j[k];
      const double* info = g->ep_info + 36 * k;
      edge_se3(iso_from12(g->ep_z + 12 * k), pose[vi], pose[vj], e, Ji, Jj);
      double rho1 = 1.0;
      if (use_ep) { double r0; hub_ep.robustify(chi2_6(e, info), r0, rho1); }
      for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += info[i * 6 + j] * e[j]; r[i] = -s * rho1; }
      JtWJ<6, 6, 6>(Ji, info, rho1, Ji, H); add_diag(var_pose(vi), H, 6);
      JtWJ<6, 6, 6>(Jj, info, rho1, Jj, H); add_diag(var_pose(vj), H, 6);
      JtWJ<6, 6, 6>(Ji, info, rho1, Jj, H); add_block(ep_blk[k], var_pose(vi), var_pose(vj), H, 6, 6);
      if (out && out->Hpp_ep) for (int i = 0; i < 36; ++i) out->Hpp_ep[36 * k + i] = H[i];
      Jtr<6, 6>(Ji, r, g6); for (int i = 0; i < 6; ++i) b[boff[var_pose(vi)] + i] += g6[i];
      Jtr<6, 6>(Jj, r, g6); for (int i = 0; i < 6; ++i) b[boff[var_pose(vj)] + i] += g6[i];
    }
    double Jp[18], Jl[9], Hpl[18], Hll[9], Hpp[36];
    const int64_t nb_ = g->n_eb;
    for (int k = 0; k < g->n_eb; ++k) {
      int vp = g->eb_pose[k], vl = g->eb_point[k];
      V3 z{g->eb_z[k], g->eb_z[nb_ + k], g->eb_z[2 * nb_ + k]};
      edge_eb(pose[vp], point[vl], z, e, Jp, Jl);
      double w = g->eb_w[k], rho1 = 1.0;
      if (use_eb) { double r0; hub_eb.robustify(chi2_3(e, w), r0, rho1); }
      double we = w * rho1;
      for (int i = 0; i < 3; ++i) r[i] = -(w * e[i]) * rho1;
      JtWJ<3, 6, 6>(Jp, nullptr, we, Jp, Hpp); add_diag(var_pose(vp), Hpp, 6);
      JtWJ<3, 3, 3>(Jl, nullptr, we, Jl, Hll); add_diag(var_point(vl), Hll, 3);
      JtWJ<3, 6, 3>(Jp, nullptr, we, Jl, Hpl);
      add_block(eb_blk[k], var_pose(vp), var_point(vl), Hpl, 6, 3);
      if (out && out->Hpl_eb) for (int i = 0; i < 18; ++i) out->Hpl_eb[i * nb_ + k] = Hpl[i];
      Jtr<3, 6>(Jp, r, g6); for (int i = 0; i < 6; ++i) b[boff[var_pose(vp)] + i] += g6[i];
      Jtr<3, 3>(Jl, r, g6); for (int i = 0; i < 3; ++i) b[boff[var_point(vl)] + i] += g6[i];
    }
    double J1[9], J2[9], Jh[18], H12[9], H1h[18], H2h[18];
    const int64_t nt_ = g->n_et;
    for (int k = 0; k < g->n_et; ++k) {
      int v1 = g->et_p1[k], v2 = g->et_p2[k], vh = g->et_pose[k];
      V3 z{g->et_z[k], g->et_z[nt_ + k], g->et_z[2 * nt_ + k]};
      edge_et(pose[vh], point[v1], point[v2], z, e, J1, J2, Jh);
      double w = g->et_w[k], rho1 = 1.0;
      if (use_et) { double r0; hub_et.robustify(chi2_3(e, w), r0, rho1); }
      double we = w * rho1;
      for (int i = 0; i < 3; ++i) r[i] = -(w * e[i]) * rho1;
      JtWJ<3, 3, 3>(J1, nullptr, we, J1, Hll); add_diag(var_point(v1), Hll, 3);
      JtWJ<3, 3, 3>(J2, nullptr, we, J2, Hll); add_diag(var_point(v2), Hll, 3);
      JtWJ<3, 6, 6>(Jh, nullptr, we, Jh, Hpp); add_diag(var_pose(vh), Hpp, 6);
      JtWJ<3, 3, 3>(J1, nullptr, we, J2, H12); add_block(et_blk12[k], var_point(v1), var_point(v2), H12, 3, 3);
      JtWJ<3, 3, 6>(J1, nullptr, we, Jh, H1h); add_block(et_blk1h[k], var_point(v1), var_pose(vh), H1h, 3, 6);
      JtWJ<3, 3, 6>(J2, nullptr, we, Jh, H2h); add_block(et_blk2h[k], var_point(v2), var_pose(vh), H2h, 3, 6);
      if (out) {
        if (out->Hll_et) for (int i = 0; i < 9; ++i) out->Hll_et[i * nt_ + k] = H12[i];
        if (out->Hlp1_et) for (int i = 0; i < 18; ++i) out->Hlp1_et[i * nt_ + k] = H1h[i];
        if (out->Hlp2_et) for (int i = 0; i < 18; ++i) out->Hlp2_et[i * nt_ + k] = H2h[i];
      }
      Jtr<3, 3>(J1, r, g6); for (int i = 0; i < 3; ++i) b[boff[var_point(v1)] + i] += g6[i];
      Jtr<3, 3>(J2, r, g6); for (int i = 0; i < 3; ++i) b[boff[var_point(v2)] + i] += g6[i];
      Jtr<3, 6>(Jh, r, g6); for (int i = 0; i < 6; ++i) b[boff[var_pose(vh)] + i] += g6[i];
    }
    if (out) {
      for (int p = 0; p < P; ++p) {
        int v = var_pose(p);
        const Blk& B = blocks[blkmap[(uint64_t)(uint32_t)v << 32 | (uint32_t)v]];
        for (int i = 0; i < 6; ++i) {
          for (int j = 0; j < 6; ++j) if (out->Hpp) out->Hpp[36 * p + i * 6 + j] = Ax[Ap[boff[v] + j] + B.rowpos + i];
          if (out->bp) out->bp[6 * p + i] = b[boff[v] + i];
        }
      }
      for (int l = 0; l < L; ++l) {
        int v = var_point(l);
        const Blk& B = blocks[blkmap[(uint64_t)(uint32_t)v << 32 | (uint32_t)v]];
        for (int i = 0; i < 3; ++i) {
          for (int j = 0; j < 3; ++j) if (out->Hll) out->Hll[9 * l + i * 3 + j] = Ax[Ap[boff[v] + j] + B.rowpos + i];
          if (out->bl) out->bl[3 * l + i] = b[boff[v] + i];
        }
      }
    }
  }

  int64_t diag_index(int k) const {  // position of A(k,k) in Ax
    for (int64_t p = Ap[k]; p < Ap[k + 1]; ++p) if (Ai[p] == k) return p;
    return -1;
  }
  std::vector<int64_t> diagpos;
  void ensure_diagpos() {
    if (!diagpos.empty()) return;
    diagpos.resize(n);
    for (int k = 0; k < n; ++k) diagpos[k] = diag_index(k);
  }
  double max_diagonal() {  // computeLambdaInit (levenberg.cpp:166-180)
    ensure_diagpos();
    double m = 0;
    for (int k = 0; k < n; ++k) m = std::max(std::fabs(Ax[diagpos[k]]), m);
    return m;
  }
  bool solve_lambda(double lambda) {
    ensure_diagpos();
    diag_backup.resize(n);
    for (int k = 0; k < n; ++k) { diag_backup[k] = Ax[diagpos[k]]; Ax[diagpos[k]] += lambda; }
    if (!analyzed) { chol.analyze(n, Ap, Ai); analyzed = true; }
    bool ok = chol.factor(Ap, Ai, Ax);
    x = b;
    if (ok) chol.solve(x.data());
    for (int k = 0; k < n; ++k) Ax[diagpos[k]] = diag_backup[k];  // restoreDiagonal
    return ok;
  }
  void apply_update() {  // SparseOptimizer::update
    for (int l = 0; l < L; ++l) {
      const double* d = &x[boff[var_point(l)]];
      point[l] = point[l] + v3(d[0], d[1], d[2]);
    }
    for (int p = 0; p < P; ++p) iso_oplus(pose[p], &x[boff[var_pose(p)]], oplusCalls[p]);
  }
};

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" int vdo_oracle_ba_linearize(const vdo_ba_graph* g, vdo_ba_system* out) {
  BA ba(g);
  double chi;
  out->robust_chi2 = ba.compute_errors(&chi);
  out->chi2 = chi;
  ba.build_system(out);
  return 0;
}

extern "C" int64_t vdo_oracle_ba_normal_equations(const vdo_ba_graph* g, int32_t* rows, int32_t* cols,
                                                  double* vals, int64_t cap, double* rhs) {
  BA ba(g);
  ba.compute_errors();
  ba.build_system();
  int64_t nnz = 0;
  for (int c = 0; c < ba.n; ++c)
    for (int64_t p = ba.Ap[c]; p < ba.Ap[c + 1]; ++p) {
      if (ba.Ai[p] > c) continue;
      if (rows && nnz < cap) { rows[nnz] = ba.Ai[p]; cols[nnz] = c; vals[nnz] = ba.Ax[p]; }
      ++nnz;
    }
  if (rhs) for (int i = 0; i < ba.n; ++i) rhs[i] = ba.b[i];
  return nnz;
}

extern "C" int vdo_oracle_ba_solve(const vdo_ba_graph* g, double lambda, double* x) {
  BA ba(g);
  ba.compute_errors();
  ba.build_system();
  bool ok = ba.solve_lambda(lambda);
  for (int i = 0; i < ba.n; ++i) x[i] = ba.x[i];
  return ok ? 0 : -1;
}

extern "C" int vdo_oracle_ba_optimize(const vdo_ba_graph* g, const vdo_lm_options* opt,
                                      double* pose_out, double* point_out, vdo_lm_stats* st) {
  BA ba(g);
  vdo_lm_stats local;
  if (!st) st = &local;
  std::memset(st, 0, sizeof(*st));
  const double t_begin = now_ms();
  // OptimizationAlgorithmLevenberg state (levenberg.cpp:42-54)
  double lambda = -1, ni = 2;
  int nBad = 0;
  const double tau = 1e-5, upper = 2. / 3., lower = 1. / 3.;
  const int maxTrials = 10;
  bool forceStop = false;         // SparseOptimizerTerminateAction flag
  double action_lastChi = 0;
  double chi2_check = 0.0;
  bool ok = true;
  int result = 0;                  // 0 OK, 1 Terminate, 2 Fail
  double last_err_chi = ba.compute_errors();
  st->initial_chi2 = last_err_chi;
  int it = 0;
  st->stop_reason = 0;
  for (; it < opt->max_iterations && !forceStop && ok; ++it) {
    // ---- OptimizationAlgorithmLevenberg::solve(it) ----
    last_err_chi = ba.compute_errors();
    double currentChi = last_err_chi, tempChi = currentChi, iniChi = currentChi;
    double t0 = now_ms();
    ba.build_system();
    st->ms_linearize += now_ms() - t0;
    if (it == 0) { lambda = tau * ba.max_diagonal(); ni = 2; nBad = 0; }
    double rho = 0;
    int qmax = 0;
    std::vector<Iso> pose_bak;
    std::vector<V3> point_bak;
    do {
      pose_bak = ba.pose; point_bak = ba.point;           // push()
      t0 = now_ms();
      bool ok2 = ba.solve_lambda(lambda);
      st->ms_solve += now_ms() - t0;
      ba.apply_update();
      last_err_chi = tempChi = ba.compute_errors();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;                                    // computeScale
      for (int j = 0; j < ba.n; ++j) scale += ba.x[j] * (lambda * ba.x[j] + ba.b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, upper);
        double sf = std::max(lower, alpha);
        lambda *= sf; ni = 2; currentChi = tempChi;        // discardTop()
      } else {
        lambda *= ni; ni *= 2;
        ba.pose = pose_bak; ba.point = point_bak;          // pop()
      }
      ++qmax;
      ++st->total_trials;
    } while (rho < 0 && qmax < maxTrials && !forceStop);
    if (qmax == maxTrials || rho == 0) result = 1;
    else {
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      result = nBad >= 3 ? 1 : 0;
    }
    // ---- back in SparseOptimizer::optimize ----
    ok = (result == 0);
    if (!ok && st->stop_reason == 0) st->stop_reason = 1;
    if (chi2_check < last_err_chi && it > 0) { ok = false; st->stop_reason = 2; }
    chi2_check = last_err_chi;
    // verbose()/terminate action both recompute the active errors at the current estimate
    if (opt->verbose || opt->gain_threshold >= 0) last_err_chi = ba.compute_errors();
    if (opt->verbose)
      std::fprintf(stderr, "iteration= %d\t chi2= %.6f\t lambda= %.6g\t levenbergIter= %d\n", it, last_err_chi, lambda, qmax);
    if (it < VDO_LM_MAX_TRACE) { st->chi2_trace[it] = last_err_chi; st->trials_trace[it] = qmax; }
    if (opt->gain_threshold >= 0) {                        // postIteration: terminate action
      if (it == 0) action_lastChi = last_err_chi;
      else {
        double gain = (action_lastChi - last_err_chi) / last_err_chi;
        action_lastChi = last_err_chi;
        if (gain >= 0 && gain < opt->gain_threshold) { forceStop = true; if (ok) st->stop_reason = 3; }
      }
    }
  }
  st->iterations = it;
  st->final_lambda = lambda;
  st->final_chi2 = ba.compute_errors();
  st->ms_total = now_ms() - t_begin;
  for (int p = 0; p < ba.P; ++p) iso_to12(ba.pose[p], pose_out + 12 * p);
  for (int l = 0; l < ba.L; ++l) { point_out[3 * l] = ba.point[l].x; point_out[3 * l + 1] = ba.point[l].y; point_out[3 * l + 2] = ba.point[l].z; }
  return 0;
}

// ---- KAT helpers ---------------------------------------------------------------------
extern "C" void vdo_oracle_se3_exp(const double u[6], double T16[16]) { SE3Quat::exp(u).toMatrix4(T16); }
extern "C" void vdo_oracle_iso_oplus(const double T12[12], const double d[6], double out12[12]) {
  Iso X = iso_from12(T12); int calls = 0; iso_oplus(X, d, calls); iso_to12(X, out12);
}
extern "C" void vdo_oracle_iso_to_mqt(const double T12[12], double e[6]) { toVectorMQT(iso_from12(T12), e); }
extern "C" void vdo_oracle_edge_se3_jac(const double Z12[12], const double Xi12[12], const double Xj12[12],
                                        double e[6], double Ji[36], double Jj[36]) {
  edge_se3(iso_from12(Z12), iso_from12(Xi12), iso_from12(Xj12), e, Ji, Jj);
}
extern "C" void vdo_oracle_edge_prior_jac(const double Z12[12], const double X12[12], double e[6], double J[36]) {
  edge_prior(iso_from12(Z12), iso_from12(X12), e, J);
}
extern "C" void vdo_oracle_edge_eb_jac(const double X12[12], const double p[3], const double z[3],
                                       double e[3], double Jpose[18], double Jpoint[9]) {
  edge_eb(iso_from12(X12), v3(p[0], p[1], p[2]), v3(z[0], z[1], z[2]), e, Jpose, Jpoint);
}
extern "C" void vdo_oracle_edge_et_jac(const double H12[12], const double p1[3], const double p2[3], const double z[3],
                                       double e[3], double Jp1[9], double Jp2[9], double Jh[18]) {
  edge_et(iso_from12(H12), v3(p1[0], p1[1], p1[2]), v3(p2[0], p2[1], p2[2]), v3(z[0], z[1], z[2]), e, Jp1, Jp2, Jh);
}
