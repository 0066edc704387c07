// Batch-BA linearisation sweep for gfx950 (K18 in SURVEY.md §2.1): per-edge SE(3) residual +
// Jacobian, Huber weight and block accumulation — the work of g2o's
// SparseOptimizer::computeActiveErrors (g2o/core/sparse_optimizer.cpp:61-114) and
// BlockSolver::buildSystem (g2o/core/block_solver.hpp:502-560) for the edge classes
//   EdgeSE3PointXYZ            g2o/types/edge_se3_pointxyz.cpp:99-140
//   LandmarkMotionTernaryEdge  g2o/types/types_dyn_slam3d.cpp:53-85   (F4 kept: no factor 2)
// (EdgeSE3 / EdgeSE3Prior live in ba_posepose.hip.)
//
// One workgroup per TILE (see ba_dev.hpp).  Per tile:
//   1. the tile's points (<=256 x 24 B, contiguous) and the inverse poses of its pose slots are
//      staged in LDS;
//   2. edges stream in fully coalesced (the tile's EdgeSE3PointXYZ edges are a thread-transposed block: entry j * 256 + t is the j-th edge of
//      thread t, every load one contiguous row - ba_dev.hpp Tile::ept) - 16 B per EdgeSE3PointXYZ in the compact form (key 4 B + fp32 measurement 12 B, one
//      information scalar per edge class), 36 B in the general one; of the 6x3 pose-landmark block only the Huber-weighted
//      information scalar we is written (8 B): the block is we * [I ; k[c]x] * (R^T | I) with c a function of the point and
//      the pose, which every consumer has in LDS anyway (ba_solve.hip make_f);
//   3. landmark sums accumulate in LDS with ds_add_f64 (all edges of a point are in the tile): the 3x3 block is
//      (sum of we) * I - J_point^T J_point = R R^T = I for both edge classes - so 1 + 3 sums per point, written once (32 B);
//   4. the pose 6x6+6 contribution of an edge depends on 16 running sums only
//      (J_pose = [-I | 2[zc]x]  resp. [I | -[v]x]): Σw, Σw·zc, Σw·zc zcᵀ, Σw·e, Σw·zc×e.
//      Every thread sums them in registers over its <= VDO_TILE_EPT consecutive EdgeSE3PointXYZ edges, which belong to ONE pose slot (edges are
//      pose-sorted inside the tile; the tile builder - capi_ba.hip close_tile - cuts every slot's run into pieces of <= VDO_TILE_EPT, one per thread), the
//      threads' totals go through a segmented DPP scan into per-slot LDS accumulators and leave as one 128-byte row per
//      (tile, slot) of the POSE-MAJOR partial array; k_finalize_pose streams a pose's rows, expands them to the 6x6 block + rhs
//      and adds the blocks of the pose's EdgeSE3 / prior edges (k_posepose), all in fixed order.
// No global atomics.  HBM bytes of a linearisation with this layout: vdo_slam_amd/ba.py linearize_byte_model;
// what bounds the kernel (measured): VALU issue (fp64) at 4 workgroups per CU; 109-130 us for 13.3 M edges = 510 MB by counters
// = 0.49-0.54 of the HBM peak (round 3: 210 us - the head of every tile was a chain of four dependent loads).
#include <cstdlib>
#include "ba_dev.hpp"
#include "ba_tile.hpp"
#include "se3_dev.hpp"

namespace vdo {

void launch_posepose(const BADev& d, int which, bool build, double* ep_chi, hipStream_t s);

// -DSWEEP_PROF (tools/build_variant.sh, debug builds only): shader-clock cycles of every wave of k_sweep_tile<true> per phase
#ifdef SWEEP_PROF
__device__ unsigned long long g_sweep_prof[16];
#define SW_TICK(slot) do { if ((threadIdx.x & 63) == 0) { const long long t_ = clock64(); sw_t[slot] = t_ - sw_prev; sw_prev = t_; } } while (0)
#else
#define SW_TICK(slot) do { } while (0)
#endif

// LDS carve-up (doubles): pts[3][VDO_SWEEP_PLANE] (planes x | y | z) | accpt[4][TP] | slotW[12*S] | accpose[ps_stride*S] | red[40] | sdst[S] (int32: where the slots' rows go)
// (ps_stride = 16: a slot carries binary OR ternary sums, both kinds share its 16 accumulators; 14 KB + 224 B per pose slot of the
// largest tile: 4 workgroups per CU at 81 slots)
// (point planes VDO_SWEEP_PLANE doubles apart: more than the 255 x 8 bytes a ds_read2_b64 spans and not a multiple of 64, so that the three coordinate reads
//  of an edge stay three ds_read_b64 - 2 LDS cycles each, banks mod 64 - instead of being paired into a ds_read2[st64]_b64: 8 cycles, banks mod 32)
#define VDO_SWEEP_PLANE (VDO_TILE_PTS + 2)
// (the 16 accumulators of a slot sit ps_stride + 2 doubles apart in LDS: the tails of the segmented scan - a few lanes per 16-lane group, each adding its 16
//  totals to ITS slot - would all meet on one bank per component with rows of 16 doubles = 128 bytes = the 32 banks of a 64-bit LDS atomic)
#define VDO_SWEEP_APAD 2
__host__ __device__ inline size_t sweep_lds_doubles(int max_slots, bool build, int ps_stride) {
  return 3 * VDO_SWEEP_PLANE + (build ? 4 * VDO_TILE_PTS : 0) + 12 * (size_t)max_slots + (build ? (size_t)(ps_stride + VDO_SWEEP_APAD) * (size_t)max_slots + ((size_t)max_slots + 1) / 2 : 0) + 40;
}

// Per-thread running sums of the TERNARY edges (the EdgeSE3PointXYZ edges of a thread share a slot by construction: acc_terms): a thread owns
// a few CONSECUTIVE edges of the slot-sorted list, so they mostly share a slot and their 16 sums add up in registers; a change of slot inside
// the chunk is flushed to the slot's LDS accumulators at once, and only the final (slot, sums) of every thread goes through the segmented DPP
// scan - one scan per thread instead of one per edge.
__device__ __forceinline__ void acc_edge(double (&acc)[16], int& cur, int slot, double we, D3 c, D3 er, double* accpose_base, int arow) {
  if (slot != cur) {
    if (cur >= 0) {
      double* dst = accpose_base + arow * cur;
#pragma unroll
      for (int i = 0; i < 16; ++i) atomicAdd(dst + i, acc[i]);
    }
    cur = slot;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  }
  // (the running sums take their terms by fused multiply-add: VALU issue is what the kernel has least of, and the order of these sums - hence
  // their last bits - already differs from the oracle's sequential order; the residual and the cross products stay unfused)
  const double wx = we * c.x, wy = we * c.y, wz = we * c.z;
  acc[0] += we; acc[1] += wx; acc[2] += wy; acc[3] += wz;
  acc[4] = __builtin_fma(wx, c.x, acc[4]); acc[5] = __builtin_fma(wx, c.y, acc[5]); acc[6] = __builtin_fma(wx, c.z, acc[6]); acc[7] = __builtin_fma(wy, c.y, acc[7]);
  acc[8] = __builtin_fma(wy, c.z, acc[8]); acc[9] = __builtin_fma(wz, c.z, acc[9]);
  acc[10] = __builtin_fma(we, er.x, acc[10]); acc[11] = __builtin_fma(we, er.y, acc[11]); acc[12] = __builtin_fma(we, er.z, acc[12]);
  acc[13] = __builtin_fma(we, c.y * er.z - c.z * er.y, acc[13]); acc[14] = __builtin_fma(we, c.z * er.x - c.x * er.z, acc[14]); acc[15] = __builtin_fma(we, c.x * er.y - c.y * er.x, acc[15]);
}
// the 16 running sums of one edge (see acc_edge), for a thread whose edges share a pose slot
__device__ __forceinline__ void acc_terms(double (&acc)[16], double we, D3 c, D3 er) {
  const double wx = we * c.x, wy = we * c.y, wz = we * c.z;
  acc[0] += we; acc[1] += wx; acc[2] += wy; acc[3] += wz;
  acc[4] = __builtin_fma(wx, c.x, acc[4]); acc[5] = __builtin_fma(wx, c.y, acc[5]); acc[6] = __builtin_fma(wx, c.z, acc[6]); acc[7] = __builtin_fma(wy, c.y, acc[7]);
  acc[8] = __builtin_fma(wy, c.z, acc[8]); acc[9] = __builtin_fma(wz, c.z, acc[9]);
  acc[10] = __builtin_fma(we, er.x, acc[10]); acc[11] = __builtin_fma(we, er.y, acc[11]); acc[12] = __builtin_fma(we, er.z, acc[12]);
  acc[13] = __builtin_fma(we, __builtin_fma(c.y, er.z, -(c.z * er.y)), acc[13]); acc[14] = __builtin_fma(we, __builtin_fma(c.z, er.x, -(c.x * er.z)), acc[14]);
  acc[15] = __builtin_fma(we, __builtin_fma(c.x, er.y, -(c.y * er.x)), acc[15]);
}
__device__ __forceinline__ void acc_finish(double (&acc)[16], int cur, double* accpose_base, int arow) {
  const SegCtl16 sc = seg_ctl16(cur);
  const SegFlags sf = seg_flags(sc);
  double* dst = accpose_base + arow * (cur >= 0 ? cur : 0);
  { double g[4] = {acc[0], acc[1], acc[2], acc[3]}; seg_apply16<4>(g, sc, sf, dst); }
  { double g[4] = {acc[4], acc[5], acc[6], acc[7]}; seg_apply16<4>(g, sc, sf, dst + 4); }
  { double g[4] = {acc[8], acc[9], acc[10], acc[11]}; seg_apply16<4>(g, sc, sf, dst + 8); }
  { double g[4] = {acc[12], acc[13], acc[14], acc[15]}; seg_apply16<4>(g, sc, sf, dst + 12); }
}

// What a thread holds of a tile before the tile's staging barrier: everything the head of the tile requests from HBM (the persistent form of the
// kernel requests the NEXT tile's while this one computes).
// The sweep streams ~510 MB through an L2 of 8 x 4 MB per launch: what is read or written ONCE (edge inputs, we, landmark sums, the partial rows) carries the
// non-temporal hint, so that it does not push out what IS used again - the pose array, the descriptors and slot tables of the tiles (shared cache lines
// between neighbouring tiles) - and streams past the resident lines instead of through them: 0.104 -> 0.093 ms on the 13.3 M-edge graph (A/B on one box,
// profiles/r05_sweep_ab.txt).  A software prefetch of the next tile's descriptor / slot tables / points into the XCD's L2 on top of it: no gain, removed.
// -DVDO_SWEEP_NO_NT: plain accesses (A/B).
#ifdef VDO_SWEEP_NO_NT
#define VDO_NT_LOAD(p) (*(p))
#define VDO_NT_STORE(v, p) (*(p) = (v))
#else
#define VDO_NT_LOAD(p) __builtin_nontemporal_load(p)
#define VDO_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

template <bool COMPACT>
struct SweepHead {
  int my_pose, my_dst;            // pose id and partial-row id of slot min(thread, slots - 1)
  double pv[3];                   // the tile's points: <= 768 doubles, three per thread
  int ekey[VDO_TILE_EPT];         // this thread's <= VDO_TILE_EPT EdgeSE3PointXYZ edges (one pose slot): key, measurement (+ information scalar)
  float ezf[COMPACT ? VDO_TILE_EPT : 1][3];
  double ezd[COMPACT ? 1 : VDO_TILE_EPT][3], ew[COMPACT ? 1 : VDO_TILE_EPT];
};
// Every load of the head is UNCONDITIONAL, from a clamped index (a load under a branch is waited for at the end of that branch: the three edges of
// a thread were three round trips in a row in the round-3 form), and everything hangs on the descriptor alone (one scalar load):
//   descriptor -> slot pose ids, row ids, points (-> poses) ;  descriptor -> this thread's edges: entry j * 256 + thread of the tile's edge block,
//   i.e. every load is one contiguous row of 256 entries (ba_dev.hpp Tile::ept; no thread table).
// Values stay as loaded (fp32) until they are used behind the staging barrier.
template <bool BUILD, bool COMPACT>
__device__ __forceinline__ void sweep_request(const BADev& d, const Tile& T, int which, int tid, SweepHead<COMPACT>& h) {
  const int npts = T.pt_end - T.pt_begin, nslot = T.slot_end - T.slot_begin;
  const double* __restrict__ point = d.point[which] + 3 * (int64_t)T.pt_begin;
  const int64_t Eb = d.Eb;
  const int my_slot = min(tid, max(nslot - 1, 0));        // (tile_pose / slot_dst carry one entry of padding)
  h.my_pose = d.tile_pose[T.slot_begin + my_slot];
  h.my_dst = 0;
  if (BUILD) h.my_dst = d.slot_dst[T.slot_begin + my_slot];
#pragma unroll
  for (int k = 0; k < 3; ++k) h.pv[k] = point[min(tid + k * VDO_TILE_THREADS, 3 * npts - 1)];
  const int ebase = (T.ept ? T.eb_begin : 0) + tid, jmax = max(T.ept - 1, 0);        // (rows past the tile's last one repeat it; a tile without edges reads entry `tid` of the first block)
  if (COMPACT) {                                          // 16 B per edge: one information scalar per edge class, fp32 measurements (ba_dev.hpp); Eb > 0
#pragma unroll
    for (int j = 0; j < VDO_TILE_EPT; ++j) {
      const int e = ebase + min(j, jmax) * VDO_TILE_THREADS;
      h.ekey[j] = VDO_NT_LOAD(d.eb_key + e);
#pragma unroll
      for (int k = 0; k < 3; ++k) h.ezf[j][k] = VDO_NT_LOAD(d.eb_zf + k * Eb + e);
    }
  } else {
#pragma unroll
    for (int j = 0; j < VDO_TILE_EPT; ++j) { h.ekey[j] = -1; h.ew[j] = 0.0; h.ezd[j][0] = h.ezd[j][1] = h.ezd[j][2] = 0.0; }
    if (T.ept > 0) {                                      // (uniform)
#pragma unroll
      for (int j = 0; j < VDO_TILE_EPT; ++j) {
        const int e = ebase + min(j, jmax) * VDO_TILE_THREADS;
        h.ekey[j] = d.eb_key[e];
#pragma unroll
        for (int k = 0; k < 3; ++k) h.ezd[j][k] = d.eb_zf ? (double)d.eb_zf[k * Eb + e] : d.eb_z[k * Eb + e];
        h.ew[j] = d.eb_w ? d.eb_w[e] : d.eb_w_uni;
      }
    }
  }
}

// COMPACT: the edge inputs are in the 16-byte form (eb_zf set, eb_w not: every graph the reference builds).
// (A persistent form - a few workgroups per CU walking tiles b, b + grid, ..., the head of the next tile requested behind the compute phase of
// this one so that its registers are free - was built on this very body in round 4 and is SLOWER: 0.228 ms against 0.140 on the 13.3 M-edge graph.
// The loop makes the compiler keep ~30 loop-invariant addresses and 35 scalars alive across all phases; at 128 registers it spills, and every
// spill reload sits in the same in-order queue as the requests in flight.  DESIGN.md 4.1.)
template <bool BUILD, bool COMPACT>
__global__ __launch_bounds__(VDO_TILE_THREADS, 4) void k_sweep_tile(BADev d, int which) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
#ifdef SWEEP_PROF
  long long sw_t[10], sw_prev = clock64();
#pragma unroll
  for (int i = 0; i < 10; ++i) sw_t[i] = 0;
#undef SW_TICK
#define SW_TICK(slot) do { if ((threadIdx.x & 63) == 0) { const long long t_ = clock64(); sw_t[slot] += t_ - sw_prev; sw_prev = t_; } } while (0)
#endif
  const int ti = blockIdx.x;                               // (descriptors are stored in launch order: tiles with dynamic tracks first - ternary edges, ~1.5x the work - not in the tail)
  const int tid = threadIdx.x;
  double* pts = smem;
  double* accpt = pts + 3 * VDO_SWEEP_PLANE;                 // [4][TP] SoA: sum of we | b.x | b.y | b.z  (lanes hit 16 bank pairs by point id)
  double* slotW = accpt + (BUILD ? 4 * VDO_TILE_PTS : 0);
  double* accpose = slotW + 12 * d.max_slots;
  const int arow = d.ps_stride + VDO_SWEEP_APAD, tofs = d.ps_stride == 32 ? 16 : 0;      // row of a slot's accumulators in LDS; where its ternary sums start
  double* red = accpose + (BUILD ? arow * d.max_slots : 0);
  int* sdst = reinterpret_cast<int*>(red + 40);
  const double* __restrict__ pose = d.pose[which];
  const int64_t Eb = d.Eb, Et = d.Et;
  // ---- The head of a tile is a chain of dependent loads, each an HBM round trip of 2-3 k cycles (phase probe, DESIGN.md 4.1: head + staging +
  // barrier were half of a tile's 21 k cycles in the round-3 form): sweep_request.
  const Tile T = d.tiles[ti];
  SweepHead<COMPACT> h;
  sweep_request<BUILD, COMPACT>(d, T, which, tid, h);
  {
    const int npts = T.pt_end - T.pt_begin, nslot = T.slot_end - T.slot_begin;
    SW_TICK(0);
    // ---- inverse poses of the slots; points -> LDS; zero accumulators
    auto stage_slot = [&](int sidx, int pid) {
      const IsoD W = iso_inv(iso_load(pose + 12 * (int64_t)pid));
      double* o = slotW + 12 * sidx;
#pragma unroll
      for (int i = 0; i < 9; ++i) o[i] = W.r[i];
      o[9] = W.t.x; o[10] = W.t.y; o[11] = W.t.z;
    };
    asm volatile("" : "+v"(h.my_pose));                  // (keeps the request of the pose id where it was made - the compiler would sink it into the branch below, behind a wait for every other request)
    if (tid < nslot) stage_slot(tid, h.my_pose);
    if (BUILD) sdst[min(tid, max(nslot - 1, 0))] = h.my_dst;     // (by every thread - the clamped ones repeat the last slot: keeps the request out of the branch above)
    for (int sidx = tid + VDO_TILE_THREADS; sidx < nslot; sidx += VDO_TILE_THREADS) {       // (more than 256 slots in a tile: not in any graph of the bench)
      stage_slot(sidx, d.tile_pose[T.slot_begin + sidx]);
      if (BUILD) sdst[sidx] = d.slot_dst[T.slot_begin + sidx];
    }
    // points -> LDS as three planes x | y | z of VDO_TILE_PTS: a coordinate of the 64 points of an edge row is ONE ds_read_b64 (2 LDS cycles, banks by
    // point id mod 32 - what the tile builder's placement keeps apart, capi_ba.hip close_tile) instead of a ds_read2_b64 + ds_read_b64 over 24-byte records (10)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = tid + k * VDO_TILE_THREADS;             // flat index into the tile's [npts][3] block (coalesced request), < 768
      const int q = (i * 0xAAAB) >> 17;                     // i / 3
      if (i < 3 * npts) pts[(i - 3 * q) * VDO_SWEEP_PLANE + q] = h.pv[k];
    }
    if (BUILD) {
      for (int i = tid; i < 4 * VDO_TILE_PTS; i += VDO_TILE_THREADS) accpt[i] = 0.0;
      for (int i = tid; i < arow * nslot; i += VDO_TILE_THREADS) accpose[i] = 0.0;
    }
    SW_TICK(1);
    __syncthreads();
    SW_TICK(2);
    int ekey[VDO_TILE_EPT];
    float ezf[VDO_TILE_EPT][3];
#pragma unroll
    for (int j = 0; j < VDO_TILE_EPT; ++j) {
      ekey[j] = h.ekey[j];
#pragma unroll
      for (int k = 0; k < 3; ++k) { ezf[j][k] = COMPACT ? h.ezf[COMPACT ? j : 0][k] : 0.f; if (COMPACT) asm volatile("" : "+v"(ezf[j][k])); }      // (opaque: the widening to fp64 happens from here on)
    }
    int ecnt = 0;                                         // (a thread's edges are rows 0 .. ecnt - 1 of its column of the block)
#pragma unroll
    for (int j = 0; j < VDO_TILE_EPT; ++j) ecnt += (j < T.ept && ekey[j] >= 0) ? 1 : 0;
    double ezd[VDO_TILE_EPT][3], ew[VDO_TILE_EPT];
    if (!COMPACT) {
#pragma unroll
      for (int j = 0; j < VDO_TILE_EPT; ++j) { ew[j] = h.ew[COMPACT ? 0 : j]; for (int k = 0; k < 3; ++k) ezd[j][k] = h.ezd[COMPACT ? 0 : j][k]; }
    }
    double chi = 0.0, rchi = 0.0;
    // ------------------------------------------------------------------ EdgeSE3PointXYZ
    {
      double acc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0;
      const int slot = ecnt ? (ekey[0] >> 16) : -1;
      double Wp[12];                                 // W.r = R^T = Jl (row-major), W.t of the thread's slot
      {
        const double* Ws = slotW + 12 * (slot >= 0 ? slot : 0);
#pragma unroll
        for (int i = 0; i < 12; ++i) Wp[i] = Ws[i];
      }
#pragma unroll
      for (int j = 0; j < VDO_TILE_EPT; ++j) {
        if (j < ecnt) {
          const int e = T.eb_begin + j * VDO_TILE_THREADS + tid;
          const int lp = ekey[j] & 0xffff;
          const double w = COMPACT ? d.eb_w_uni : ew[j];
          const D3 z = COMPACT ? D3{(double)ezf[j][0], (double)ezf[j][1], (double)ezf[j][2]} : D3{ezd[j][0], ezd[j][1], ezd[j][2]};
          const D3 p{pts[lp], pts[VDO_SWEEP_PLANE + lp], pts[2 * VDO_SWEEP_PLANE + lp]};
          const D3 zc = cam_point(Wp, p);
          const D3 er = zc - z;
          const double c2 = chi2_w3(w, er);
          double rho0, rho1;
          huber_dev(c2, d.huber_eb, d.dsqr_eb, rho0, rho1);
          chi += c2; rchi += rho0;
          if (BUILD) {
            const double we = w * rho1;
            // 6x3 block Hpl = -we * [ I ; 2[zc]x ] * Jl  -> only we is stored (8 B instead of 144 B): the consumers recompute zc from
            // the point and the pose exactly as above (ba_solve.hip make_f)
            VDO_NT_STORE(we, d.Finc + e);
            // landmark side: Hll += we * Jl^T Jl = we * R R^T = we * I (R is a rotation: g2o's product differs from I by a few
            // 1e-16, far inside the 1e-12 parity bar) -> ONE running sum per point; bl += -we * R e   (R e = Jl^T e)
            // (R e and the cross product zc x e by fused multiply-adds: no cancellation follows them, the blocks move by ~1e-16 of their size -
            // unlike zc itself, whose last bit the subtraction zc - z amplifies past the 1e-12 parity bar)
            const D3 Re{__builtin_fma(Wp[6], er.z, __builtin_fma(Wp[3], er.y, Wp[0] * er.x)), __builtin_fma(Wp[7], er.z, __builtin_fma(Wp[4], er.y, Wp[1] * er.x)),
                        __builtin_fma(Wp[8], er.z, __builtin_fma(Wp[5], er.y, Wp[2] * er.x))};
            atomicAdd(accpt + lp, we);
            atomicAdd(accpt + VDO_TILE_PTS + lp, -we * Re.x); atomicAdd(accpt + 2 * VDO_TILE_PTS + lp, -we * Re.y); atomicAdd(accpt + 3 * VDO_TILE_PTS + lp, -we * Re.z);
            acc_terms(acc, we, zc, er);
          }
        }
      }
      SW_TICK(3);
      if (BUILD) acc_finish(acc, slot, accpose, arow);
      SW_TICK(4);
    }
    // ------------------------------------------------------------ LandmarkMotionTernaryEdge
    {
      const int nte = T.et_end - T.et_begin;
      const int per = (nte + VDO_TILE_THREADS - 1) / VDO_TILE_THREADS;
      double acc[16];
      int cur = -1;
      for (int j = 0; j < per; ++j) {
        const int e = T.et_begin + tid * per + j;
        if (e < T.et_end) {
          const int key = d.et_key[e];
          const int slot = d.et_slot[e];
          const int l1 = key & 0xffff, l2 = key >> 16;
          const double w = d.et_w ? d.et_w[e] : d.et_w_uni;
          const D3 z = d.et_z ? D3{d.et_z[e], d.et_z[Et + e], d.et_z[2 * Et + e]} : D3{0.0, 0.0, 0.0};
          const double* Hi = slotW + 12 * slot;   // Hi.r = R_H^T, Hi.t ; J2 = -Hi.r
          const D3 p1{pts[l1], pts[VDO_SWEEP_PLANE + l1], pts[2 * VDO_SWEEP_PLANE + l1]};
          const D3 p2{pts[l2], pts[VDO_SWEEP_PLANE + l2], pts[2 * VDO_SWEEP_PLANE + l2]};
          const D3 v = cam_point(Hi, p2);
          const D3 er = p1 - v - z;
          const double c2 = chi2_w3(w, er);
          double rho0, rho1;
          huber_dev(c2, d.huber_et, d.dsqr_et, rho0, rho1);
          chi += c2; rchi += rho0;
          if (BUILD) {
            const double we = w * rho1;
            double* O = d.Oll + 9 * (int64_t)e;  // O = we * J1^T J2 = -we * Hi.r (p1 x p2); one 72-byte record per edge: the chain kernels read a block per step
#pragma unroll
            for (int i = 0; i < 9; ++i) O[i] = -we * Hi[i];
            // (H,p1): we * [I ; [v]x]   and   (H,p2): -we * [I ; [v]x] * Hi.r   -> both from (we, v)
            d.Finc[Eb + e] = we;
            // p1: Hll += we*I, b += -we*e ; p2: Hll += we*R_H R_H^T = we*I, b += we * R_H e
            atomicAdd(accpt + l1, we);
            atomicAdd(accpt + VDO_TILE_PTS + l1, -we * er.x); atomicAdd(accpt + 2 * VDO_TILE_PTS + l1, -we * er.y); atomicAdd(accpt + 3 * VDO_TILE_PTS + l1, -we * er.z);
            const D3 Re = rotT(Hi, er);
            atomicAdd(accpt + l2, we);
            atomicAdd(accpt + VDO_TILE_PTS + l2, we * Re.x); atomicAdd(accpt + 2 * VDO_TILE_PTS + l2, we * Re.y); atomicAdd(accpt + 3 * VDO_TILE_PTS + l2, we * Re.z);
            acc_edge(acc, cur, slot, we, v, er, accpose + tofs, arow);
          }
        }
      }
      if (BUILD && nte > 0) acc_finish(acc, cur, accpose + tofs, arow);      // (uniform: tiles of static points have no ternary edges - 256 scan instructions less)
    }
    // ---- write back
    SW_TICK(5);
    {   // chi2 partials of the tile: one barrier (it also orders the LDS atomics before the reads of the write-back); waves added in fixed order
      const int lane = tid & 63, wv = tid >> 6;
      chi = wave_sum(chi); rchi = wave_sum(rchi);
      if (lane == 0) { red[wv] = chi; red[16 + wv] = rchi; }
      __syncthreads();
      if (tid == 0) {
        double sa = 0, sb = 0;
        for (int w = 0; w < VDO_TILE_THREADS / 64; ++w) { sa += red[w]; sb += red[16 + w]; }
        d.part_chi[ti] = sa; d.part_chi[d.n_tiles + ti] = sb;
      }
    }
    SW_TICK(6);
    if (BUILD) {
      // landmarks: Hll = (sum of we) * I -> one double per point; bl - coalesced: consecutive lanes write consecutive doubles
      double* __restrict__ H = d.Hll + (int64_t)T.pt_begin;
      for (int i = tid; i < npts; i += VDO_TILE_THREADS) VDO_NT_STORE(accpt[i], H + i);
      double* __restrict__ b = d.bl + 3 * (int64_t)T.pt_begin;
      for (int i = tid; i < 3 * npts; i += VDO_TILE_THREADS) {
        const int l = i / 3, k = i - 3 * l;
        VDO_NT_STORE(accpt[(1 + k) * VDO_TILE_PTS + l], b + i);
      }
      // per-(tile,slot) partials -> their pose-major rows: 128 (256) contiguous bytes per slot
      if (d.ps_stride == 16) {
        for (int i = tid; i < 16 * nslot; i += VDO_TILE_THREADS) {
          const int sidx = i >> 4, k = i & 15;
          VDO_NT_STORE(accpose[arow * sidx + k], d.part_sums + 16 * (int64_t)sdst[sidx] + k);
        }
      } else {
        for (int i = tid; i < 32 * nslot; i += VDO_TILE_THREADS) {
          const int sidx = i >> 5, k = i & 31;
          VDO_NT_STORE(accpose[arow * sidx + k], d.part_sums + 32 * (int64_t)sdst[sidx] + k);
        }
      }
    }
    SW_TICK(7);
#ifdef SWEEP_PROF
    sw_t[9] += 1;
#endif
  }
#ifdef SWEEP_PROF
  if (BUILD && (threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 5) {          // (a sample: the atomics of every wave would be the kernel)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_sweep_prof[i], (unsigned long long)sw_t[i]);
    atomicAdd(&g_sweep_prof[15], 1ull);
    atomicAdd(&g_sweep_prof[14], (unsigned long long)sw_t[9]);
  }
#endif
}

// Expand the running sums of every (tile,slot) partial of a pose into its 6x6 block and rhs.
// + the contributions of the pose's EdgeSE3 / prior edges (k_posepose left them in ep_blk), in the order of the pose's edge list;
// + (chi_mode >= 0) the chi2 reduction of the whole linearisation in one extra workgroup: every partial it reads was written by an
// earlier launch.
__device__ void reduce_chi_body(const BADev& d, int mode, double* lds);
#define VDO_FIN_THREADS 512
__global__ __launch_bounds__(VDO_FIN_THREADS) void k_finalize_pose(BADev d, int add_posepose, int chi_mode) {
  __shared__ double lds[24];
  __shared__ double part[VDO_FIN_THREADS / 64][32];
  if (chi_mode >= 0 && blockIdx.x == gridDim.x - 1) {
    reduce_chi_body(d, chi_mode, lds);
    return;
  }
  // One WORKGROUP per pose: the pose's partial rows are contiguous (pose-major part_sums) and its waves stream a contiguous share each - lane l
  // always meets component l % stride (shares start at multiples of 64 doubles) - then wave 0 adds the waves' sums in wave order.  (One wave per
  // pose, round 3, left the linearisation of a graph of few cameras and many points waiting for ~200 waves that stream 150 KB each in a
  // dependent loop: 72 us on the 13.3 M-edge graph.)
  const int p = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, st = d.ps_stride;
  const int rows = d.ps_off[p + 1] - d.ps_off[p], q = 64 / st;
  const int share = ((rows + VDO_FIN_THREADS / 64 - 1) / (VDO_FIN_THREADS / 64) + q - 1) / q * q;      // rows per wave, a multiple of 64 doubles
  const double* __restrict__ base = d.part_sums + ((int64_t)d.ps_off[p] + (int64_t)wv * share) * st;
  const int64_t n = (int64_t)max(0, min(share, rows - wv * share)) * st;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int64_t i = lane;
  for (; i + 192 < n; i += 256) { a0 += base[i]; a1 += base[i + 64]; a2 += base[i + 128]; a3 += base[i + 192]; }
  for (; i < n; i += 64) a0 += base[i];
  double a = (a0 + a1) + (a2 + a3);
  a += __shfl_xor(a, 32, 64);
  if (st == 16) a += __shfl_xor(a, 16, 64);
  if (lane < 32) part[wv][lane] = a;
  __syncthreads();
  if (wv != 0) return;
  a = 0.0;
  if (lane < 32) {
#pragma unroll
    for (int w = 0; w < VDO_FIN_THREADS / 64; ++w) a += part[w][lane];
  }
  const int kind = d.pose_kind[p];
  double sb[16], stn[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double lo = __shfl(a, k, 64), hi = __shfl(a, 16 + k, 64);
    sb[k] = st == 32 ? lo : (kind == 0 ? lo : 0.0);
    stn[k] = st == 32 ? hi : (kind == 1 ? lo : 0.0);
  }
  double Hm[36], b[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) Hm[i] = 0;
  // top-left: S0 * I
  const double s0 = sb[0] + stn[0];
  Hm[0] = s0; Hm[7] = s0; Hm[14] = s0;
  // top-right = [a]x with a = -2 S1(binary) - S1(ternary);  [a]x = [0 -az ay; az 0 -ax; -ay ax 0]
  const double ax = -2.0 * sb[1] - stn[1], ay = -2.0 * sb[2] - stn[2], az = -2.0 * sb[3] - stn[3];
  Hm[0 * 6 + 4] = -az; Hm[0 * 6 + 5] = ay;
  Hm[1 * 6 + 3] = az;  Hm[1 * 6 + 5] = -ax;
  Hm[2 * 6 + 3] = -ay; Hm[2 * 6 + 4] = ax;
  // bottom-right: 4 (tr(S2) I - S2) binary + (tr(S2) I - S2) ternary
  const double trb = sb[4] + sb[7] + sb[9], trt = stn[4] + stn[7] + stn[9];
  Hm[3 * 6 + 3] = 4.0 * (trb - sb[4]) + (trt - stn[4]);
  Hm[3 * 6 + 4] = -4.0 * sb[5] - stn[5];
  Hm[3 * 6 + 5] = -4.0 * sb[6] - stn[6];
  Hm[4 * 6 + 4] = 4.0 * (trb - sb[7]) + (trt - stn[7]);
  Hm[4 * 6 + 5] = -4.0 * sb[8] - stn[8];
  Hm[5 * 6 + 5] = 4.0 * (trb - sb[9]) + (trt - stn[9]);
  b[0] = sb[10] - stn[10]; b[1] = sb[11] - stn[11]; b[2] = sb[12] - stn[12];
  b[3] = 2.0 * sb[13] - stn[13]; b[4] = 2.0 * sb[14] - stn[14]; b[5] = 2.0 * sb[15] - stn[15];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < i; ++j) Hm[i * 6 + j] = Hm[j * 6 + i];
  // every lane holds the whole block: lane i < 36 stores entry i, lanes 36..41 the right-hand side (coalesced)
  double out = 0.0;
#pragma unroll
  for (int i = 0; i < 36; ++i) out = lane == i ? Hm[i] : out;
#pragma unroll
  for (int i = 0; i < 6; ++i) out = lane == 36 + i ? b[i] : out;
  if (add_posepose && lane < 42) {
    for (int q = d.pe_off[p]; q < d.pe_off[p + 1]; ++q) {
      const int ent = d.pe_idx[q];
      const double* blk = d.ep_blk + 84 * (int64_t)(ent >> 1);
      out += lane < 36 ? blk[36 * (ent & 1) + lane] : blk[72 + 6 * (ent & 1) + (lane - 36)];
    }
    for (int q = d.pr_off[p]; q < d.pr_off[p + 1]; ++q) {
      const double* blk = d.ep_blk + 84 * (int64_t)(d.Ep + d.pr_idx[q]);
      out += lane < 36 ? blk[lane] : blk[72 + (lane - 36)];
    }
  }
  if (lane < 36) d.Hpp[36 * (int64_t)p + lane] = out;
  else if (lane < 42) d.bp[6 * (int64_t)p + (lane - 36)] = out;
}

// Fixed-order final reduction of the chi2 partials: tiles, then pose-pose edges.
// mode 0: single GPU, everything -> scal.   Shards: mode 1 = this rank's tiles -> red_chi (summed
// across ranks by the hook), mode 2/3 = red_chi + the replicated pose-pose edges -> scal
// (3 also publishes the all-reduced computeScale() partial that k_update left in red_chi[2]).
// (+ 8: the reduction belongs to a linearisation - the totals are also kept in S_LIN_CHI2 / S_LIN_RCHI2.)
__device__ void reduce_chi_body(const BADev& d, int mode_lin, double* lds) {
  const int mode = mode_lin & 7, lin = mode_lin >> 3;
  const int nt = d.n_tiles, n2 = d.Ep + d.Npr;
  const double* ep_chi = d.part_chi + 2 * (int64_t)nt;
  double a0 = 0, a1 = 0;
  // 256 lanes with stride 256 whatever the workgroup size (k_reduce_chi: 256 threads, the last workgroup of k_finalize_pose: 512): the chi2 of a
  // linearisation and of an error evaluation at the same estimate are the SAME BITS - ba_lm.hip skips the re-evaluation after an accepted trial on that
  // (ADVICE r4).  The waves beyond the fourth add zeros, in wave order.
  if (threadIdx.x < 256) {
    if (mode <= 1) {
      for (int i = threadIdx.x; i < nt; i += 256) { a0 += d.part_chi[i]; a1 += d.part_chi[nt + i]; }
      for (int i = threadIdx.x; i < d.n_hubs; i += 256) { a0 += d.hub_chi[i]; a1 += d.hub_chi[d.n_hubs + i]; }      // (hub landmarks, ba_hub.hip: rank-local like the tiles)
    }
    if (mode != 1)
      for (int i = threadIdx.x; i < n2; i += 256) { a0 += ep_chi[i]; a1 += ep_chi[n2 + i]; }
  }
  a0 = block_sum1(a0, lds);
  a1 = block_sum1(a1, lds);
  if (threadIdx.x == 0) {
    if (mode == 1) { d.red_chi[0] = a0; d.red_chi[1] = a1; }
    else {
      if (mode >= 2) { a0 += d.red_chi[0]; a1 += d.red_chi[1]; }
      d.scal[S_CHI2] = a0; d.scal[S_RCHI2] = a1;
      if (lin) { d.scal[S_LIN_CHI2] = a0; d.scal[S_LIN_RCHI2] = a1; }
      if (mode == 3) d.scal[S_SCALE] = d.red_chi[2];
    }
  }
}
__global__ __launch_bounds__(256) void k_reduce_chi(BADev d, int mode) {
  __shared__ double lds[24];
  reduce_chi_body(d, mode, lds);
}

// ---------------------------------------------------------------------------- launchers
static double* ep_chi_buf(const BADev& d) { return d.part_chi + 2 * (int64_t)d.n_tiles; }

void launch_errors(const BADev& d, int which, hipStream_t s, const Reducer& R) {
  const size_t lds = sweep_lds_doubles(d.max_slots, false, d.ps_stride) * sizeof(double);
  if (d.n_tiles) {
    if (d.eb_zf && !d.eb_w) hipLaunchKernelGGL((k_sweep_tile<false, true>), dim3(d.n_tiles), dim3(VDO_TILE_THREADS), raise_lds(k_sweep_tile<false, true>, lds), s, d, which);
    else hipLaunchKernelGGL((k_sweep_tile<false, false>), dim3(d.n_tiles), dim3(VDO_TILE_THREADS), raise_lds(k_sweep_tile<false, false>, lds), s, d, which);
  }
  launch_hub_sweep(d, which, false, s);
  launch_posepose(d, which, false, ep_chi_buf(d), s);
  if (!d.sharded) { hipLaunchKernelGGL(k_reduce_chi, dim3(1), dim3(256), 0, s, d, 0); return; }
  hipLaunchKernelGGL(k_reduce_chi, dim3(1), dim3(256), 0, s, d, 1);
  R(d.red_chi, 3);
  hipLaunchKernelGGL(k_reduce_chi, dim3(1), dim3(256), 0, s, d, which == 1 ? 3 : 2);
}

// the chi2 of a linearisation whose exchange was deferred (launch_linearize), behind that exchange
void launch_linearize_finish(const BADev& d, hipStream_t s) { hipLaunchKernelGGL(k_reduce_chi, dim3(1), dim3(256), 0, s, d, 2 + 8); }

void launch_sweep_only(const BADev& d, hipStream_t s, int which) {
  const size_t lds = sweep_lds_doubles(d.max_slots, true, d.ps_stride) * sizeof(double);
  launch_hub_sweep(d, which, true, s);
  if (!d.n_tiles) return;
  if (d.eb_zf && !d.eb_w) hipLaunchKernelGGL((k_sweep_tile<true, true>), dim3(d.n_tiles), dim3(VDO_TILE_THREADS), raise_lds(k_sweep_tile<true, true>, lds), s, d, which);
  else hipLaunchKernelGGL((k_sweep_tile<true, false>), dim3(d.n_tiles), dim3(VDO_TILE_THREADS), raise_lds(k_sweep_tile<true, false>, lds), s, d, which);
}

// defer_exchange (sharded solves): leave the partial Hpp | bp | chi2 of this rank where they are - the caller's launch_factor_and_rhs, which follows at once, sends them
// with the block-Jacobi sums and the reduced right-hand side in ONE all-reduce (its tile passes read nothing of the pose blocks) and finishes the chi2 behind it.
void launch_linearize(const BADev& d, hipStream_t s, const Reducer& R, bool defer_exchange, int which) {
  launch_posepose(d, which, true, ep_chi_buf(d), s);       // per-edge blocks -> ep_blk (independent of the sweep)
  launch_sweep_only(d, s, which);
  // pose blocks = landmark-side sums + pose-pose blocks.  Shards: the (replicated) pose-pose terms are added by rank 0 only, the
  // all-reduce of Hpp | bp | chi2 then hands every rank the same bits.
  const int add_pp = (!d.sharded || d.shard_rank == 0) ? 1 : 0;
  const int nb = d.P;                                      // one workgroup per pose
  if (!d.sharded) { hipLaunchKernelGGL(k_finalize_pose, dim3(nb + 1), dim3(VDO_FIN_THREADS), 0, s, d, add_pp, 8); return; }   // (+ the chi2 reduction in the last workgroup)
  hipLaunchKernelGGL(k_finalize_pose, dim3(nb + 1), dim3(VDO_FIN_THREADS), 0, s, d, add_pp, 1);
  if (defer_exchange) return;
  R(d.Hpp, 42 * (int64_t)d.P + 2);
  hipLaunchKernelGGL(k_reduce_chi, dim3(1), dim3(256), 0, s, d, 2 + 8);
}

}  // namespace vdo

#ifdef SWEEP_PROF
extern "C" int vdo_debug_sweep_prof(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(vdo::g_sweep_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(vdo::g_sweep_prof), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif
