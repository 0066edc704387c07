// Host-side handle of one batch-BA problem (shared by capi_ba.hip and ba_lm.hip).
#pragma once
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ba_dev.hpp"
#include "ctx.hpp"

struct vdo_ba {
  vdo_ctx* ctx = nullptr;
  vdo::BADev d;
  std::vector<void*> allocs;
  bool pooled = false;            // device arrays (as far as they fit), pinned block, side stream and events belong to the context's pool (ctx.hpp), not to this handle
  vdo::Reducer red;               // cross-rank sum/max hook (vdo_ba_set_allreduce); unset = single GPU
  int oplus_calls = 0;            // VertexSE3::_numOplusCalls (same value on every vertex)
  double* h_scal = nullptr;       // pinned, device-mapped: [S_COUNT doubles][4 int32 flags] in ONE block; a one-workgroup kernel publishes the device scalars and
  uint32_t ticket = 0;            // read-backs so far: k_publish_scalars writes it behind the scalars (word 4 of the flags' row) and fetch() polls it instead of waiting for the stream
  int32_t* h_flags = nullptr;     // flags into it (ba_lm.hip fetch: two D2H copies of 13 us each on the stream before round 6); h_flags points behind the scalars
  double* d_hscal = nullptr;      // the device-side address of that block
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t side = nullptr;     // second stream of a solve: the reduced right-hand side beside the pose-chain factorisation (ba_solve.hip)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // permutations between the caller's numbering and the tile-major device numbering
  std::vector<int32_t> pt_old_of_new, pt_new_of_old;
  double* hub_binc = nullptr;                               // [n_hub_edges][18] explicit blocks of the hub edges (vdo_ba_download_system), allocated on first use
  std::vector<int32_t> hub_eb_old;                          // original EdgeSE3PointXYZ id of every hub edge (ba_hub.hip)
  std::vector<int32_t> eb_old_of_new, et_old_of_new;        // (eb: by padded device entry, -1 where a block entry holds no edge)
  int n_eb = 0;                                             // EdgeSE3PointXYZ edges of the graph (the device's d.Eb counts padded block entries)
  std::vector<int32_t> inc_of_eb, inc1_of_et, inc2_of_et;   // by NEW edge index
  std::vector<double> h_tmp;
  // dense reduced-camera solver (ba_dense.hip), allocated on first use: S [ld][ld], W [ld/64][64][64], rhs [ld]
  double *dense_S = nullptr, *dense_W = nullptr, *dense_rhs = nullptr;
  int64_t dense_ld = 0;
  bool dense_S_clean = false;        // k_dense_small zeroes what it read of S: the next assembly needs no memset
  bool dense_tiles_ok = true;        // every tile's padded incidence count (256 * ept + 2 * ternary edges) fits VDO_TILE_INC: what k_schur_dense_tile holds per thread
  bool pose_graph_is_paths = true;   // every EdgeSE3 lies on a simple path (the chain preconditioner covers them all)
  int last_solver = 0;            // 2 PCG, 3 dense: what the last trial used
  int pcg_it = 0, pcg_parity = 0, pcg_maxit = 0, pcg_last = 0;      // (pcg_last: iterations the previous solve of this run needed)      // state of the PCG solve of the trial in flight (ba_lm.hip solve_trial / solve_trial_finish)
  double pcg_tol2 = 0;
  bool alt_failed = false;             // the second set of linearisation buffers could not be allocated (ba_lm.hip ensure_alt)
  bool dense_pending = false;          // the trial in flight was solved by the dense factorisation (solve_trial_finish reads its failure flag)
  bool lin_exchange_pending = false;   // sharded: the linearisation in front of the next trial deferred its all-reduce to that trial's launch_factor_and_rhs
  bool lin_current = false;       // the blocks on the device (Hpp, bp, Hll, bl, Finc) are the linearisation AT estimate[0]: set by vdo_ba_linearize,
                                  // cleared by vdo_ba_optimize / vdo_ba_set_estimates (vdo_ba_download_system re-linearises when it is not)
  int compact_edges = 0;          // bit 0: uniform eb_w, 1: fp32 eb_z, 2: uniform et_w, 3: et_z all zero (ba_dev.hpp)
};

namespace vdo {
void* ba_device_alloc(vdo_ba* ba, size_t bytes);      // capi_ba.hip: from the context's slab (pooled handles) or hipMalloc; released by vdo_ba_destroy
inline int sync_check(vdo_ba* ba, const char* what) {
  hipError_t e = hipStreamSynchronize(ba->ctx->stream);
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "%s: %s", what, hipGetErrorString(e));
  e = hipGetLastError();
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "%s: %s", what, hipGetErrorString(e));
  if (ba->red.err) return set_error(VDO_ERR_INVALID, "%s: all-reduce hook failed (%d)", what, ba->red.err);
  return VDO_OK;
}
}  // namespace vdo
