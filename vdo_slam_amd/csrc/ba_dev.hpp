// Device-resident state of one batch-BA problem + kernel launch prototypes.
//
// HBM layout.  The graph is re-ordered once, on the host, into TILES:
//   tile  = up to VDO_TILE_PTS landmark points (whole tracks/chains, grouped by first observing
//           frame) together with ALL edges incident to them (<= VDO_TILE_INC incidences).
//   points are renumbered tile-major (chains contiguous, in chain order);
//   binary / ternary edges are stored tile-major and, inside a tile, sorted by pose vertex.
// Consequences: every landmark-side accumulation (3x3+3 diagonal, B^T v) is local to one
// workgroup and lives in LDS (ds_add_f64) — no global atomics; pose-side accumulation is a
// wave-level segmented reduction into per-(tile,pose) partials, summed later in fixed order.
//   pose    [P][12]      AoS (R row-major | t); few, L2-resident
//   point   [L][3]       AoS, tile-major: read once per tile, coalesced
//   edges   SoA, tile-major: key int32 (pose-slot<<16 | local point), z [3][E], w [E]; the EdgeSE3PointXYZ edges of a tile are a padded,
//           thread-transposed block (Tile::ept): Eb counts block entries, not edges
//   Finc    [Eb+Et]      FACTORED pose-x-point blocks, written once per sweep.  Every 6x3 block of
//           Hpl is  s*we*[ I ; k[c]x ] * (R^T or I)  with c = the point in the pose's frame (zc resp.
//           v = H^-1 p2) and R the pose rotation; c is a function of the point and the pose, both staged in LDS
//           by every consumer anyway, so ONLY we (the Huber-weighted information scalar) is stored:
//           8 B/edge instead of 144 B, both for the sweep's write and for every PCG mat-vec.
//   part_q / part_m / part_sums   per-(tile,pose-slot) partial sums of the solver and of the sweep (NPS = total slots): pose-major rows, see below
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"

#define VDO_TILE_PTS 256        // max points per tile
#define VDO_TILE_THREADS 256
#ifndef VDO_TILE_EPT
#define VDO_TILE_EPT 6          // EdgeSE3PointXYZ edges per thread of the tile kernels: what a tile costs apart from its edges (two HBM round trips at its head,
#endif                          // barriers, write-back: ~9 k of 16 k cycles at 3 edges per thread, phase probe) is spread over this many
#define VDO_TILE_INC (VDO_TILE_THREADS * VDO_TILE_EPT)        // max incidences per tile

namespace vdo {

struct Tile {
  int32_t pt_begin, pt_end;       // points [pt_begin, pt_end)
  int32_t eb_begin, eb_end;       // binary edges
  int32_t et_begin, et_end;       // ternary edges
  int32_t inc_begin;              // incidences: [inc_begin, +nb) binary, then nt (p1), then nt (p2)
  int32_t slot_begin, slot_end;   // pose slots [slot_begin, slot_end) into tile_pose / part_* arrays
  int32_t chain_begin, chain_end; // chains [chain_begin, chain_end) into chain_off
  int32_t ept;                    // EdgeSE3PointXYZ entries per thread of this tile's edge block: [eb_begin, eb_end) = 256 * ept entries, entry j * 256 + t = j-th edge of thread t (key -1: none)
};

struct BADev {
  int P = 0, L = 0, Eb = 0, Et = 0, Ep = 0, Npr = 0, Ninc = 0, n_tiles = 0, NPS = 0, n_chains = 0, max_slots = 0;
  int n_dyn_tiles = 0;                   // the first n_dyn_tiles tiles of the launch order hold a multi-point landmark chain (dynamic track); the others static points only
  double huber_eb = 0, huber_et = 0, huber_ep = 0, dsqr_eb = 0, dsqr_et = 0, dsqr_ep = 0;
  // estimates: [0] current, [1] trial
  double* pose[2] = {nullptr, nullptr};
  double* point[2] = {nullptr, nullptr};
  // tiles
  Tile* tiles = nullptr;                 // [n_tiles] in LAUNCH order (capi_ba.hip): workgroup b works on tiles[b]
  int32_t* tile_pose = nullptr;          // [NPS] global pose id of each slot
  int32_t* chain_off = nullptr;          // [n_chains+1] point ranges (points of a chain are contiguous)
  int32_t* pt_prev_edge = nullptr;       // [L] ternary edge linking point l-1 -> l (or -1: chain head)
  // edges (tile-major)
  int32_t* eb_key = nullptr; double *eb_z = nullptr, *eb_w = nullptr;         // key = slot<<16 | local point
  int32_t *et_key = nullptr, *et_slot = nullptr; double *et_z = nullptr, *et_w = nullptr;  // key = lp1 | lp2<<16
  // compact edge inputs, chosen at vdo_ba_create when they lose nothing:
  //   eb_w == nullptr : every EdgeSE3PointXYZ carries the same information scalar (eb_w_uni) - true for every graph the reference
  //                     builds (one sigma per edge class, src/Optimizer.cc:1330-1335)
  //   eb_zf != nullptr: every measurement component is exactly a float (the reference's are: Get3DinCamera returns CV_32F) -
  //                     stored as fp32 planes [3][Eb], widened exactly in the kernel; eb_z is then not allocated
  //   et_w == nullptr / et_z == nullptr: uniform weight / all-zero measurements of the ternary edges (always, in the reference)
  double eb_w_uni = 0, et_w_uni = 0;
  float* eb_zf = nullptr;
  int32_t* inc_key = nullptr;            // [Ninc] slot<<16 | local point
  int32_t *ep_i = nullptr, *ep_j = nullptr; double *ep_z = nullptr, *ep_info = nullptr;
  int32_t* pr_pose = nullptr; double *pr_z = nullptr, *pr_info = nullptr;
  // pose -> slots CSR, pose -> pose-pose edges CSR
  int32_t *ps_off = nullptr, *ps_idx = nullptr;
  // sweep partials, POSE-MAJOR: the (tile,slot) partial of global slot s is row slot_dst[s] of part_sums, and the rows of pose p are
  // [ps_off[p], ps_off[p+1]) - contiguous, so k_finalize_pose streams them.  A row holds ps_stride running sums: 16 when no pose
  // vertex carries both EdgeSE3PointXYZ and ternary edges (never, in the reference's graphs: cameras see points, motions link them) -
  // the row then holds whichever kind the pose has (pose_kind: 0 binary, 1 ternary) - else 32 (binary | ternary).
  int32_t* slot_dst = nullptr;           // [NPS]
  int32_t* pose_kind = nullptr;          // [P]
  int ps_stride = 16;
  int32_t *pe_off = nullptr, *pe_idx = nullptr;   // entry = edge<<1 | side (0: pose is i, 1: pose is j)
  // pose chains (paths of the EdgeSE3 graph) for the block-tridiagonal preconditioner, in path order
  int n_pchains = 0;
  int pc_nwave = 1;                               // waves of a chain's workgroup = segments of the partitioned substitutions (ba_solve.hip)
  int pc_closed = 0;                              // 1 (VDO_BA_PCHAIN_CLOSED): the closed-form block inverse in k_pchain_factor (a measured dead end, ba_solve.hip)
  int pc_maxlen = 0, pc_lds = 0;                  // longest chain; 1: a chain's strip [len][6] fits the LDS of its workgroup (k_pcg_chain), 0: global-memory path
  int32_t *pc_off = nullptr, *pc_pose = nullptr;  // [n_pchains+1], [P]
  int32_t* pc_edge = nullptr;                     // [P] edge<<1|side linking position k-1 -> k (side 0: previous pose is the edge's i), -1 at a chain head
  // TWISTED chains (capi_ba.hip): first half of the path, second half backwards (its first position carries pc_edge = -1), the middle pose last.  The last
  // position - the joint - has its ordinary link to the position before it and ONE far link to position pc_far_pos[c] (the end of the first half).
  int32_t *pc_far_pos = nullptr, *pc_far_edge = nullptr;   // [n_pchains] global chain position of the far predecessor / its link (edge<<1|side); -1: not twisted
  // linear system
  double *Hpp = nullptr, *bp = nullptr;              // [P][36], [P][6]
  double *Hll = nullptr, *bl = nullptr;              // [L] (the landmark diagonal block is Hll[l] * I3, see ba_sweep.hip),  [L][3]
  double* Finc = nullptr;                            // [Eb+Et] factored pose-landmark blocks: we (c is recomputed, ba_solve.hip make_f)
  double* Binc = nullptr;                            // [18][Ninc] explicit 6x3 blocks — only materialised for vdo_ba_download_system
  double* Oll = nullptr;                             // [Et][9]  p1 x p2 blocks, one record per edge (round 6; [9][Et] planes before: a chain step fetched nine cache lines for 72 bytes)
  double* Hpp_ep = nullptr;                          // [Ep][36]
  double* ep_blk = nullptr;                          // [Ep+Npr][84] per pose-pose edge: Hii | Hjj | bi | bj (added to the pose blocks by k_finalize_pose)
  int32_t *pr_off = nullptr, *pr_idx = nullptr;      // pose -> priors CSR
  double* part_sums = nullptr;                       // [NPS][ps_stride] sweep partials, pose-major rows (see slot_dst)
  double* part_red = nullptr;                        // [256] per-block partials of k_update / k_max_diag
  double* part_chi = nullptr;                        // [2][n_tiles] + [2][Ep+Npr]
  // solver workspaces
  double *Dinv = nullptr, *Gl = nullptr;             // [L][9]: forward pivots^-1, G_k = Delta_{k-1}^-1 O_{k-1}
  double *Gdiag = nullptr, *Goff = nullptr;          // [L][9]: [Hll^-1]_{kk}, [Hll^-1]_{k-1,k}
  // A point that is a chain of its own (every static landmark: most of the graph) has Hll + lambda I = (Hll[l] + lambda) I3: its factor,
  // its pivot inverse and its block of Hll^-1 are ONE scalar - dscal[l] = 1 / (Hll[l] + lambda) - and none of the four [L][9] arrays is
  // written or read for it (pt_single[l] = 1).
  double* dscal = nullptr;                           // [L]
  uint8_t* pt_single = nullptr;                      // [L]
  double* xl = nullptr;                              // [L][3]
  double* Minv = nullptr;                            // [P][36] chain position k: Delta_k^-1 of the block LDL^T (chains of length 1: plain block-Jacobi)
  double* Adg = nullptr;                             // [P][36] S_pp + lambda I by pose id (input of the chain factorisation)
  double* Lc = nullptr;                              // [P][36] chain position k: L_k = E_{k-1,k}^T Delta_{k-1}^-1
  double* Lfar = nullptr;                            // [n_pchains][36] twisted chains: L of the joint's far link = E_{far,joint}^T Delta_far^-1
  double *Pf = nullptr, *Qb = nullptr;               // [P][36] chain position k: products of -L / -L^T from the segment's end (k_pchain_prefix)
  double *xp = nullptr, *rp = nullptr, *zp = nullptr, *pp = nullptr, *qp = nullptr, *bs = nullptr, *qs = nullptr;  // [6P]
  double* pp2 = nullptr;                             // [6P] second search-direction buffer (the PCG iterations ping-pong between pp and pp2)
  double *part_pq = nullptr, *part_rz = nullptr;     // [(P+3)/4] p.q per workgroup of k_pcg_q; [n_pchains] r.z per chain of k_pcg_chain
  double* part_q = nullptr;                          // [NPS][8] pose-major rows (row slot_dst[s] of slot s), 6 used: Schur mat-vec partials
  double *part_m = nullptr, *part_m8 = nullptr;      // [NPS][16] + [NPS][8] pose-major rows: the 21 preconditioner partials of a slot (16 + 5)
  double* scal = nullptr;
  int32_t* flags = nullptr;                          // [0] factor failure [1] pcg state [2] pcg iterations [3] arrival counter of k_pcg_chain
  // multi-GPU shards: poses replicated, points + their edges owned by one rank.
  // Hpp | bp | red_chi are ONE allocation so that a linearisation needs a single all-reduce.
  // SECOND SET of everything a linearisation writes (round 6, ba_lm.hip): the error evaluation of a Levenberg trial is a full linearisation at the trial estimate into this set -
  // an accepted trial swaps the sets (lin_swap) and the next iteration has its system already; a rejected one leaves the current set alone.  NULL: not allocated (sharded handles).
  double *Finc_alt = nullptr, *Hll_alt = nullptr, *bl_alt = nullptr, *Oll_alt = nullptr, *part_sums_alt = nullptr, *ep_blk_alt = nullptr, *Hpp_ep_alt = nullptr, *hub_we_alt = nullptr;
  double* Hpp_alt = nullptr;             // the whole block Hpp | bp | red_chi | msum | qs of the second set
  int sharded = 0, shard_rank = 0;
  // HUB landmarks (round 6, ba_hub.hip): a STATIC point whose observations do not fit one tile (more than 256 distinct pose vertices / 256 per-pose pieces / 1 536 edges - a
  // point seen in 300+ frames) belongs to no tile.  One workgroup per hub walks its edges; every hub edge owns one pose-major partial row (its own "slot": tile_pose /
  // slot_dst / ps_off count it), so the pose side reaches k_finalize_pose, k_gather_q / k_pcg_q and k_precond_finalize like the rows of a (tile, slot) pair.
  int n_hubs = 0, n_hub_edges = 0;
  int32_t* hub_off = nullptr;            // [n_hubs + 1] edge ranges
  int32_t* hub_point = nullptr;          // [n_hubs] device point id (hubs come after every tile's points; each is a chain of its own)
  int32_t* hub_pose = nullptr;           // [n_hub_edges] pose vertex
  int32_t* hub_row = nullptr;            // [n_hub_edges] pose-major row of the edge's partials (part_sums / part_q / part_m)
  double* hub_z = nullptr;               // [3][n_hub_edges] measurements
  double* hub_w = nullptr;               // [n_hub_edges] information scalars
  double* hub_we = nullptr;              // [n_hub_edges] Huber-weighted information of the last linearisation (the Finc of these edges)
  double* hub_chi = nullptr;             // [2][n_hubs] chi2 / robust chi2 partials
  double* red_chi = nullptr;                         // [4] chi2, robust chi2, scale partials (follows bp)
  double* msum = nullptr;                            // [P][21] block-Jacobi partials awaiting the all-reduce
};

// Cross-rank reduction hook (vdo_ba_set_allreduce); a sticky error is picked up at the next readback.
struct Reducer {
  int (*fn)(void*, void*, int64_t, int) = nullptr;
  void* user = nullptr;
  mutable int err = 0;
  void operator()(void* buf, int64_t n, int op = 0) const { if (fn && !err) err = fn(user, buf, n, op); }
};

enum Scal { S_CHI2 = 0, S_RCHI2, S_MAXDIAG, S_RZ, S_PQ, S_RZ0, S_SCALE, S_RZNEW, S_BETA, S_LIN_CHI2, S_LIN_RCHI2, S_COUNT = 16 };   // S_LIN_*: chi2 of the last LINEARISATION (kept while the trials' error evaluations overwrite S_CHI2 / S_RCHI2)

// ---- ba_sweep.hip
void launch_errors(const BADev& d, int which, hipStream_t s, const Reducer& R);      // chi2 of estimate[which] -> scal
void launch_schur_matvec_only(const BADev& d, hipStream_t s);                     // k_schur_tile<0> alone (vdo_ba_profile_schur)
void launch_linearize(const BADev& d, hipStream_t s, const Reducer& R, bool defer_exchange = false, int which = 0);      // build system at estimate[which] (+chi2)
// the two sets of linearisation outputs change places (host copy of the descriptor: the kernels take it by value)
inline void lin_swap(BADev& d) {
  auto sw = [](double*& a, double*& b) { double* t = a; a = b; b = t; };
  sw(d.Finc, d.Finc_alt); sw(d.Hll, d.Hll_alt); sw(d.bl, d.bl_alt); sw(d.Oll, d.Oll_alt); sw(d.part_sums, d.part_sums_alt); sw(d.ep_blk, d.ep_blk_alt); sw(d.Hpp_ep, d.Hpp_ep_alt);
  sw(d.hub_we, d.hub_we_alt);
  const int64_t P = d.P;
  sw(d.Hpp, d.Hpp_alt);
  d.bp = d.Hpp + 36 * P; d.red_chi = d.bp + 6 * P; d.msum = d.red_chi + 4; d.qs = d.msum + 21 * P + 1;
}
void launch_linearize_finish(const BADev& d, hipStream_t s);                                               // the chi2 of a linearisation whose exchange was deferred
void launch_sweep_only(const BADev& d, hipStream_t s, int which = 0);             // just the K18 tile sweep kernel (bench)
// ---- ba_solve.hip
void launch_max_diag(const BADev& d, hipStream_t s, const Reducer& R);
// landmark-chain LDL^T + inverse blocks + block-Jacobi + pose-chain factorisation, and the reduced right-hand side qs (beside it on `side` if given)
void launch_factor_and_rhs(const BADev& d, double lambda, hipStream_t s, const Reducer& R, hipStream_t side, hipEvent_t fork, hipEvent_t join,
                           bool precond = true, bool lin_pending = false);   // precond = false: without the PCG's preconditioner (block-Jacobi sums, pose-chain factorisation) - the dense solver's trials
void launch_pcg_init(const BADev& d, hipStream_t s);
void launch_pcg_iter(const BADev& d, double lambda, double tol2, int parity, hipStream_t s, const Reducer& R);   // parity: 0, 1, 0, ... from the first iteration after launch_pcg_init
void launch_backsub_update(const BADev& d, double lambda, bool ortho, hipStream_t s);
void launch_expand_binc(const BADev& d, hipStream_t s);            // Finc -> explicit Binc (download/debug only)
void launch_dense_assemble(const BADev& d, double* S, int64_t ld, double lambda, hipStream_t s, const Reducer& R, bool init = true, bool clean = false);   // explicit reduced-camera matrix (init = false: only - sum B Hll^-1 B^T; clean: S is zero already)
void launch_dense_small(const BADev& d, double* S, int64_t ld, double lambda, hipStream_t s);      // 6P <= 128: Hpp + lambda, right-hand side, Cholesky and both substitutions in ONE workgroup -> xp (ba_dense.hip)
void launch_dense_rhs(const BADev& d, double* rhs, int64_t ld, hipStream_t s);
// ---- ba_dense.hip
// ba_hub.hip: the hub landmarks' share of the tile kernels' work (no-ops without hubs)
void launch_hub_sweep(const BADev& d, int which, bool build, hipStream_t s);
void launch_hub_precond(const BADev& d, hipStream_t s);
void launch_hub_schur(const BADev& d, int mode, const double* v, const double* v2, hipStream_t s);
void launch_hub_expand_binc(const BADev& d, double* binc18, hipStream_t s);      // [n_hub_edges][18] explicit 6x3 blocks (vdo_ba_download_system)
void launch_publish_scalars(const BADev& d, double* h_block_dev, hipStream_t s, uint32_t ticket = 0);      // ba_solve.hip (ticket != 0: written behind the data, system-scope fence in between)
void launch_dense_solve(const BADev& d, double* S, int64_t ld, double* Winv, double* rhs, hipStream_t s);    // MFMA Cholesky + substitutions -> xp

size_t dense_tile_lds(const BADev& d);      // dynamic LDS of k_schur_dense_tile (ba_solve.hip)
// A tile kernel's dynamic LDS grows with the pose slots of the graph's largest tile (a landmark seen from 150 frames needs 150 slots): past the runtime's
// default limit the launch has to say so.  The attribute is per (kernel, device): the size already granted is remembered per kernel instantiation and device
// (one runtime call per growth, none per launch - ADVICE r5); a refused request is reported through set_error and fails the launch's caller at its next sync_check
// with that message instead of a generic asynchronous launch failure.
template <typename Kernel>
inline size_t raise_lds(Kernel kernel, size_t bytes) {
  static std::atomic<size_t> granted[64];
  if (bytes > (size_t)(48 * 1024)) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (bytes > granted[dev].load(std::memory_order_relaxed)) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e == hipSuccess) granted[dev].store(bytes, std::memory_order_relaxed);
      else set_error(VDO_ERR_UNSUPPORTED, "a tile kernel needs %zu bytes of LDS per workgroup and the device refused it (%s)", bytes, hipGetErrorString(e));
    }
  }
  return bytes;
}
#define VDO_LDS_MAX_BYTES (160 * 1024)

}  // namespace vdo
