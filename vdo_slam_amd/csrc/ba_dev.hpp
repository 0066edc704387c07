// Device-resident state of one batch-BA problem (SoA in HBM) + kernel launch prototypes.
// Layout notes (DESIGN.md §"Data layout in HBM"):
//   pose   [P][12]  AoS (one gather = 96 contiguous bytes, shared by ~10^3 consecutive edges)
//   point  [L][3]   AoS (24 B gather)
//   edges  SoA, camera-major: idx int32, z [3][E], w [E]  -> fully coalesced 40 B/edge
//   Binc   [18][Ninc] SoA 6x3 pose-x-point block per incidence (Ninc = Eb + 2 Et), written once per sweep
//   chunks: runs of <= VDO_CHUNK incidences that share one pose vertex -> block-level reduction,
//           no atomics on the pose side, deterministic summation order
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VDO_CHUNK 1024       // incidences per workgroup chunk
#define VDO_SWEEP_THREADS 256

namespace vdo {

struct Chunk { int32_t pose, begin, end, pad; };

struct BADev {
  int P = 0, L = 0, Eb = 0, Et = 0, Ep = 0, Npr = 0, Ninc = 0;
  double huber_eb = 0, huber_et = 0, huber_ep = 0, dsqr_eb = 0, dsqr_et = 0, dsqr_ep = 0;
  // estimates: [0] current, [1] trial
  double* pose[2] = {nullptr, nullptr};
  double* point[2] = {nullptr, nullptr};
  // edges
  int32_t *eb_pose = nullptr, *eb_point = nullptr; double *eb_z = nullptr, *eb_w = nullptr;
  int32_t *et_p1 = nullptr, *et_p2 = nullptr, *et_pose = nullptr; double *et_z = nullptr, *et_w = nullptr;
  int32_t *ep_i = nullptr, *ep_j = nullptr; double *ep_z = nullptr, *ep_info = nullptr;
  int32_t* pr_pose = nullptr; double *pr_z = nullptr, *pr_info = nullptr;
  // chunk lists
  Chunk* chunks_b = nullptr; int n_chunks_b = 0;   // binary edges, index space [0,Eb)
  Chunk* chunks_t = nullptr; int n_chunks_t = 0;   // ternary edges, index space [0,Et)
  // pose -> chunk CSR (over the unified incidence chunk list: binary chunks, then ternary chunks twice)
  int32_t *pc_off = nullptr, *pc_idx = nullptr;
  int n_chunks_inc = 0;                              // n_chunks_b + 2*n_chunks_t
  Chunk* chunks_inc = nullptr;                       // unified, index space [0,Ninc)
  // linear system
  double *Hpp = nullptr, *bp = nullptr;              // [P][36], [P][6]
  double *Hll = nullptr, *bl = nullptr;              // [L][9],  [L][3]
  double* Binc = nullptr;                            // [18][Ninc]
  int32_t *inc_pose = nullptr, *inc_point = nullptr; // [Ninc]
  double* Oll = nullptr;                             // [9][Et]  p1 x p2 blocks
  double* Hpp_ep = nullptr;                          // [Ep][36]
  double* chunk_sums = nullptr;                      // [18][n_chunks_b + n_chunks_t] sweep partials
  double* chunk_chi = nullptr;                       // [2][n_chunks_b + n_chunks_t + 1] error partials
  // chains
  int32_t *chain_off = nullptr, *chain_pt = nullptr, *chain_edge = nullptr; int n_chains = 0;
  int32_t* chain_of_static = nullptr;
  // solver workspaces
  double *Dinv = nullptr, *Gl = nullptr;             // [L][9] each
  double *ul = nullptr, *wl = nullptr, *xl = nullptr; // [L][3]
  double *Minv = nullptr;                            // [P][36]
  double *xp = nullptr, *rp = nullptr, *zp = nullptr, *pp = nullptr, *qp = nullptr, *bs = nullptr, *qs = nullptr;  // [6P]
  double* chunk_q = nullptr;                         // [6][n_chunks_inc]
  double* scal = nullptr;                            // device scalars (see enum below)
  int32_t* flags = nullptr;                          // [0] factor failure
};

enum Scal { S_CHI2 = 0, S_RCHI2, S_MAXDIAG, S_RZ, S_PQ, S_RZ0, S_SCALE, S_RZNEW, S_BNORM, S_COUNT = 16 };

// ---- launchers (ba_sweep.hip) -----------------------------------------------------------
void launch_errors(const BADev& d, int which, hipStream_t s);          // chi2 of estimate[which] -> scal[S_CHI2..]
void launch_linearize(const BADev& d, hipStream_t s);                  // build system at estimate[0]
void launch_sweep_eb_only(const BADev& d, hipStream_t s);              // just the K18 binary-edge sweep (bench)
// ---- launchers (ba_solve.hip) -----------------------------------------------------------
void launch_max_diag(const BADev& d, hipStream_t s);
void launch_factor(const BADev& d, double lambda, hipStream_t s);      // chain LDL^T, preconditioner
void launch_reduced_rhs(const BADev& d, hipStream_t s);
void launch_pcg_init(const BADev& d, hipStream_t s);
void launch_pcg_iter_tol(const BADev& d, double lambda, double tol2, double* qs, hipStream_t s);
void launch_backsub_update(const BADev& d, double lambda, bool ortho, hipStream_t s);

}  // namespace vdo
