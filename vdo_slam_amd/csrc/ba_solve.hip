// Linear solve of one Levenberg trial, (H + lambda I) x = b, for the batch graph on gfx950.
// The reference hands the whole (poses+motions+points) system to CSparse Cholesky
// (g2o/solvers/linear_solver_csparse.h:108-144; no Schur).  Here the landmark
// part is eliminated first — block-diagonal for static points, block-tridiagonal along each
// dynamic track (the ternary edge couples consecutive observations, src/Optimizer.cc:1704-1741)
// — and the reduced pose/motion system S = Hpp - Hpl Hll^-1 Hlp is solved matrix-free with
// conjugate gradients preconditioned by the block-tridiagonal matrix M = blockdiag(S) + EdgeSE3
// off-diagonal blocks along every pose chain (block LDL^T, k_pchain_factor; round 5: long chains in TWISTED order - two half-depth
// recurrences on two waves that meet in a joint with one far link, capi_ba.hip).  x equals the direct solve up to the PCG tolerance.
//
// One workgroup per TILE for everything that touches landmarks (ba_dev.hpp): each thread keeps
// its <= VDO_TILE_EPT incidences of ONE pose slot in registers (the Huber-weighted information scalar from HBM, the 6x3 block
// recomputed from the LDS-resident point and inverse pose: make_f), B^T v accumulates per point in
// LDS, the landmark-chain solves run there, and B w is segment-reduced per pose slot into pose-major
// partial rows -> 8 B per incidence from HBM per CG iteration, no global atomics.  One WAVE per
// pose and one workgroup per pose chain for the vector phases of the CG (k_pcg_q, k_pcg_chain).
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "ba_dev.hpp"
#include "ba_tile.hpp"
#include "se3_dev.hpp"

namespace vdo {

// computeLambdaInit (g2o/core/optimization_algorithm_levenberg.cpp:166-180): max |H(j,j)|
__global__ __launch_bounds__(1024) void k_max_diag(BADev d) {
  __shared__ double lds[17];
  double m = 0;
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = gtid; i < 6 * (int64_t)d.P; i += gsz) m = fmax(m, fabs(d.Hpp[36 * (i / 6) + 7 * (i % 6)]));
  for (int64_t i = gtid; i < (int64_t)d.L; i += gsz) m = fmax(m, fabs(d.Hll[i]));        // (the landmark block is Hll[l] * I3)
  m = block_max1(m, lds);
  if (threadIdx.x == 0) d.part_red[blockIdx.x] = m;
}

__device__ __forceinline__ bool spd3_inv(const double* a, double* o) {
  const double det = sym3_inv(a, o);
  const double m2 = a[0] * a[4] - a[1] * a[3];
  return (a[0] > 0) && (m2 > 0) && (det > 0);
}

// Per landmark chain: block LDL^T (Delta_k = D_k + lambda I - O^T Delta_{k-1}^-1 O) and the
// diagonal / first off-diagonal blocks of Hll^-1 needed by the exact block-Jacobi preconditioner:
//   G_kk = Delta_k^-1 + Gl_{k+1} G_{k+1,k+1} Gl_{k+1}^T ,  G_{k,k+1} = -Gl_{k+1} G_{k+1,k+1}
// steps of the forward / backward walk whose inputs are requested together.  Round 6 (profiles/r06_landmark_chain_walk_ab.txt): 8 / 2 instead of 4 / 2 - ms per LM iteration
// 1.27 -> 1.24 on the OMD-shaped graph, 0.79 -> 0.775 on the 1 M-point one; 8 / 4, 6 / 3, 12 / 4 no better.  Dealing the tracks to more, smaller workgroups (the 11 k tracks of the
// OMD-shaped graph fill only 86 workgroups of 128) makes it SLOWER (16 per workgroup: equal; 8: +11 %; 1: +85 %): the walk is bound by its instruction stream, not by the CUs' address paths.
#ifndef VDO_FC
#define VDO_FC 8
#endif
#ifndef VDO_BC
#define VDO_BC 2
#endif
__global__ __launch_bounds__(128) void k_factor_chains(BADev d, double lambda) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.n_chains) return;
  const int64_t p0 = d.chain_off[c], p1 = d.chain_off[c + 1];
  if (p1 - p0 == 1) {                                  // a point on its own: (Hll + lambda) I3 -> one scalar (ba_dev.hpp dscal)
    const double hd = d.Hll[p0] + lambda;
    d.dscal[p0] = 1.0 / hd;
    if (!(hd > 0)) atomicOr(d.flags, 1);
    return;
  }
  double prev[9];
  bool ok = true;
  // The inputs of a step (Hll of the point, the O block through the pt_prev_edge -> Oll pointer chase) do not depend on the recursion, and a step is ~500 cycles of
  // arithmetic against ~2 500 of memory latency: with the next step's inputs requested one step ahead (rounds 2-4) the walk ran at one memory latency per step.
  // Round 5: chunks of FC steps - the next chunk's Hll and O blocks and the edge indices of the chunk after it are requested while this chunk computes (unconditional
  // loads from clamped positions: a load under a branch is waited for at its end).
  constexpr int FC = VDO_FC;
  auto cl_h = [&](int64_t l) -> int64_t { return l < p1 ? l : p1 - 1; };
  auto cl_e = [&](int64_t l) -> int64_t { return l < p1 ? (l > p0 ? l : p0 + 1) : p1 - 1; };      // (p1 - p0 >= 2 here: positions p0 + 1 .. p1 - 1 have a previous edge)
  double hl[FC], O[FC][9];
  int64_t en[FC];
  {
    int64_t e0[FC];
#pragma unroll
    for (int j = 0; j < FC; ++j) { hl[j] = d.Hll[cl_h(p0 + j)]; e0[j] = d.pt_prev_edge[cl_e(p0 + j)]; }
#pragma unroll
    for (int j = 0; j < FC; ++j) en[j] = d.pt_prev_edge[cl_e(p0 + FC + j)];
#pragma unroll
    for (int j = 0; j < FC; ++j)
#pragma unroll
      for (int i = 0; i < 9; ++i) O[j][i] = d.Oll[9 * e0[j] + i];
  }
  for (int64_t l0 = p0; l0 < p1; l0 += FC) {
    double hln[FC], On[FC][9];
    int64_t enn[FC];
#pragma unroll
    for (int j = 0; j < FC; ++j) {
      hln[j] = d.Hll[cl_h(l0 + FC + j)];
#pragma unroll
      for (int i = 0; i < 9; ++i) On[j][i] = d.Oll[9 * en[j] + i];
    }
#pragma unroll
    for (int j = 0; j < FC; ++j) enn[j] = d.pt_prev_edge[cl_e(l0 + 2 * FC + j)];
#pragma unroll
    for (int j = 0; j < FC; ++j) {
      const int64_t l = l0 + j;
      if (l < p1) {
        const double hd = hl[j] + lambda;
        double D[9] = {hd, 0.0, 0.0, 0.0, hd, 0.0, 0.0, 0.0, hd};
        if (l > p0) {
          double G[9];
          mat3_mul(prev, O[j], G);
#pragma unroll
          for (int i = 0; i < 9; ++i) d.Gl[9 * l + i] = G[i];
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) D[3 * i + q] -= O[j][i] * G[q] + O[j][3 + i] * G[3 + q] + O[j][6 + i] * G[6 + q];
        }
        ok &= spd3_inv(D, prev);
#pragma unroll
        for (int i = 0; i < 9; ++i) d.Dinv[9 * l + i] = prev[i];
      }
    }
#pragma unroll
    for (int j = 0; j < FC; ++j) {
      hl[j] = hln[j]; en[j] = enn[j];
#pragma unroll
      for (int i = 0; i < 9; ++i) O[j][i] = On[j][i];
    }
  }
  // backward: inverse blocks.  Its inputs - Gl of the step, Dinv of the one below, written by the walk above - come two steps ahead (chunks of BC = 2).
  double Gn[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { Gn[i] = prev[i]; d.Gdiag[9 * (p1 - 1) + i] = prev[i]; }
  constexpr int BC = VDO_BC;
  auto cl_b = [&](int64_t l) -> int64_t { return l > p0 ? l : p0; };
  double Gx[BC][9], Dx[BC][9];                          // step l = lb - j reads Gl[l + 1], Dinv[l]
#pragma unroll
  for (int j = 0; j < BC; ++j)
#pragma unroll
    for (int i = 0; i < 9; ++i) { Gx[j][i] = d.Gl[9 * cl_b(p1 - 1 - j) + i]; Dx[j][i] = d.Dinv[9 * cl_b(p1 - 2 - j) + i]; }
  for (int64_t lb = p1 - 2; lb >= p0; lb -= BC) {
    double Gxn[BC][9], Dxn[BC][9];
#pragma unroll
    for (int j = 0; j < BC; ++j)
#pragma unroll
      for (int i = 0; i < 9; ++i) { Gxn[j][i] = d.Gl[9 * cl_b(lb - BC - j + 1) + i]; Dxn[j][i] = d.Dinv[9 * cl_b(lb - BC - j) + i]; }
#pragma unroll
    for (int j = 0; j < BC; ++j) {
      const int64_t l = lb - j;
      if (l >= p0) {
        double T1[9];
        mat3_mul(Gx[j], Gn, T1);                       // Gl_{k+1} G_{k+1,k+1}
#pragma unroll
        for (int i = 0; i < 9; ++i) d.Goff[9 * (l + 1) + i] = -T1[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q) Gn[3 * i + q] = Dx[j][3 * i + q] + T1[3 * i] * Gx[j][3 * q] + T1[3 * i + 1] * Gx[j][3 * q + 1] + T1[3 * i + 2] * Gx[j][3 * q + 2];
#pragma unroll
        for (int i = 0; i < 9; ++i) d.Gdiag[9 * l + i] = Gn[i];
      }
    }
#pragma unroll
    for (int j = 0; j < BC; ++j)
#pragma unroll
      for (int i = 0; i < 9; ++i) { Gx[j][i] = Gxn[j][i]; Dx[j][i] = Dxn[j][i]; }
  }
  if (!ok) atomicOr(d.flags, 1);
}

// 6x6 SPD inverse via Cholesky; returns false if a pivot is not positive.
__device__ bool spd6_inv(const double* A, double* Ainv) {
  double Lm[36];
  bool ok = true;
  for (int i = 0; i < 36; ++i) Lm[i] = 0;
  for (int j = 0; j < 6; ++j) {
    double s = A[j * 6 + j];
    for (int k = 0; k < j; ++k) s -= Lm[j * 6 + k] * Lm[j * 6 + k];
    if (!(s > 0)) { ok = false; s = 1.0; }
    const double dj = sqrt(s);
    Lm[j * 6 + j] = dj;
    for (int i = j + 1; i < 6; ++i) {
      double t = A[i * 6 + j];
      for (int k = 0; k < j; ++k) t -= Lm[i * 6 + k] * Lm[j * 6 + k];
      Lm[i * 6 + j] = t / dj;
    }
  }
  double Li[36];
  for (int i = 0; i < 36; ++i) Li[i] = 0;
  for (int c = 0; c < 6; ++c) {
    Li[c * 6 + c] = 1.0 / Lm[c * 6 + c];
    for (int r = c + 1; r < 6; ++r) {
      double t = 0;
      for (int k = c; k < r; ++k) t -= Lm[r * 6 + k] * Li[k * 6 + c];
      Li[r * 6 + c] = t / Lm[r * 6 + r];
    }
  }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double t = 0;
      for (int k = (i > j ? i : j); k < 6; ++k) t += Li[k * 6 + i] * Li[k * 6 + j];
      Ainv[i * 6 + j] = t;
    }
  return ok;
}

// Factored pose-landmark block (ba_dev.hpp): F = (we, c).  kind 0: EdgeSE3PointXYZ, 1: ternary (H,p1), 2: ternary (H,p2).
//   kind 0:  B = -we [ I ; 2[c]x ] Rt        kind 1:  B = we [ I ; [c]x ]        kind 2:  B = -we [ I ; [c]x ] Rt
// (Rt = R^T of the pose vertex, row-major).
// Only `we` lives in HBM (Finc [Eb+Et]: 8 B per edge); c is recomputed from the tile's points and inverse poses staged in LDS -
// estimate [0], the linearisation point - with the very operations of the sweep (ba_sweep.hip: zc = W p + t_W, v = H^-1 p2), so
// the block is bit-for-bit the one the sweep's sums were formed from.  That is 24 B per incidence less in the sweep's write and
// in every pass of the solver over the blocks (one CG iteration reads them once).
struct FInc { double we, cx, cy, cz; };
// inverse pose (R^T | -R^T t, 12 doubles; the first 9 are the R^T the block expansion needs) of every pose slot + the points of the tile
__device__ __forceinline__ void stage_slot_w_pts(const BADev& d, const Tile& T, double* slotW, double* pts) {
  const int nslot = T.slot_end - T.slot_begin, npts = T.pt_end - T.pt_begin;
  for (int sidx = threadIdx.x; sidx < nslot; sidx += blockDim.x) {
    const IsoD W = iso_inv(iso_load(d.pose[0] + 12 * (int64_t)d.tile_pose[T.slot_begin + sidx]));
    double* o = slotW + 12 * sidx;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = W.r[i];
    o[9] = W.t.x; o[10] = W.t.y; o[11] = W.t.z;
  }
  const double* __restrict__ point = d.point[0] + 3 * (int64_t)T.pt_begin;
  for (int i = threadIdx.x; i < 3 * npts; i += blockDim.x) pts[i] = point[i];
}
// incidence li of the tile (key = slot << 16 | local point) -> its factored block (we from HBM, c from LDS)
// PLANES: the tile's points sit in LDS as three planes x | y | z, VDO_TILE_PLANE doubles apart (k_schur_tile, as in the sweep: a coordinate of the points of 64 edges is
// one conflict-free 8-byte read per half wave; [l][3] puts every third bank under double load); else [l][3]
#define VDO_TILE_PLANE (VDO_TILE_PTS + 2)
template <bool PLANES = false>
__device__ __forceinline__ FInc make_f(const BADev& d, const Tile& T, int li, int kind, int key, double we, const double* slotW, const double* pts) {
  int lp = key & 0xffff;                                   // kind 0: the observed point; kind 2: p2
  if (kind == 1) lp = d.et_key[T.et_begin + (li - (T.eb_end - T.eb_begin))] >> 16;     // (H, p1): c = H^-1 p2 as well
  const double* W = slotW + 12 * (key >> 16);
  const D3 p = PLANES ? D3{pts[lp], pts[VDO_TILE_PLANE + lp], pts[2 * VDO_TILE_PLANE + lp]} : D3{pts[3 * lp], pts[3 * lp + 1], pts[3 * lp + 2]};
  const D3 c = cam_point(W, p);
  return FInc{we, c.x, c.y, c.z};
}
// explicit 6x3 block (row-major 18) — used by the preconditioner and the debug expansion
__device__ __forceinline__ void expand_block(int kind, const FInc& f, const double* Rt, double (&B)[18]) {
  if (kind == 1) {
    B[0] = f.we; B[1] = 0; B[2] = 0; B[3] = 0; B[4] = f.we; B[5] = 0; B[6] = 0; B[7] = 0; B[8] = f.we;
    B[9] = 0; B[10] = -f.we * f.cz; B[11] = f.we * f.cy;
    B[12] = f.we * f.cz; B[13] = 0; B[14] = -f.we * f.cx;
    B[15] = -f.we * f.cy; B[16] = f.we * f.cx; B[17] = 0;
    return;
  }
  const double s2 = (kind == 0 ? -2.0 : -1.0) * f.we;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double a = Rt[j], b = Rt[3 + j], cc = Rt[6 + j];   // column j of Rt
    B[0 * 3 + j] = -f.we * a;
    B[1 * 3 + j] = -f.we * b;
    B[2 * 3 + j] = -f.we * cc;
    B[3 * 3 + j] = s2 * (f.cy * cc - f.cz * b);
    B[4 * 3 + j] = s2 * (f.cz * a - f.cx * cc);
    B[5 * 3 + j] = s2 * (f.cx * b - f.cy * a);
  }
}
// incidence li of a tile -> (kind, index into Finc)
__device__ __forceinline__ void inc_locate(const Tile& T, int li, int64_t Eb, int& kind, int64_t& fidx) {
  const int nb = T.eb_end - T.eb_begin, nt = T.et_end - T.et_begin;
  if (li < nb) { kind = 0; fidx = T.eb_begin + li; }
  else if (li < nb + nt) { kind = 1; fidx = Eb + T.et_begin + (li - nb); }
  else { kind = 2; fidx = Eb + T.et_begin + (li - nb - nt); }
}
// out(6x6 upper, 21 values) += B1 G B2^T (+ transpose if sym2) ; helper computing full 6x6 product
__device__ __forceinline__ void bgbt(const double (&B1)[18], const double* G, const double (&B2)[18], double (&out)[36]) {
  double T1[18];   // B1 G (6x3)
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) T1[3 * r + c] = B1[3 * r] * G[c] + B1[3 * r + 1] * G[3 + c] + B1[3 * r + 2] * G[6 + c];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) out[6 * r + c] = T1[3 * r] * B2[3 * c] + T1[3 * r + 1] * B2[3 * c + 1] + T1[3 * r + 2] * B2[3 * c + 2];
}

// Exact diagonal blocks of the Schur correction, per (tile,slot):  sum B G B^T  (21 upper entries)
// DYN = false: the tile holds single-point chains only (every tile behind the first d.n_dyn_tiles of the launch order: static landmarks) - the
// 6x3 blocks, the 6x6 products and the ternary edges are not compiled in, and the kernel fits twice as many waves (230 registers with them).
template <bool DYN>
__global__ __launch_bounds__(VDO_TILE_THREADS) void k_precond_tile(BADev d, int tile0) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // (head of a tile: every request unconditional and as early as its address is known - ba_sweep.hip)
  const int ti = tile0 + (int)blockIdx.x, tid = threadIdx.x;
  const Tile T = d.tiles[ti];
  const int nslot = T.slot_end - T.slot_begin, npts = T.pt_end - T.pt_begin;
  const int nb = T.eb_end - T.eb_begin, nt = T.et_end - T.et_begin;
  double* accm = smem;                       // [21 * S]
  double* slotW = accm + 21 * d.max_slots;   // [12 * S]
  double* pts = slotW + 12 * d.max_slots;    // [3][VDO_TILE_PLANE]: planes x | y | z (make_f<true>)
  int* sdst = reinterpret_cast<int*>(pts + 3 * VDO_TILE_PLANE);      // [S] rows of the slots' partials
  const int my_slot = min(tid, max(nslot - 1, 0));
  int my_pose = d.tile_pose[T.slot_begin + my_slot];
  const int my_dst = d.slot_dst[T.slot_begin + my_slot];
  double pvl[3];
  {
    const double* __restrict__ point = d.point[0] + 3 * (int64_t)T.pt_begin;
#pragma unroll
    for (int k = 0; k < 3; ++k) pvl[k] = point[min(tid + k * VDO_TILE_THREADS, 3 * npts - 1)];
  }
  // this thread's EdgeSE3PointXYZ incidences: rows of its column of the tile's edge block (ba_dev.hpp Tile::ept), one contiguous row per load
  const int ebase = (T.ept ? T.eb_begin : 0) + tid, jmax = max(T.ept - 1, 0);
  int keyb[VDO_TILE_EPT];
  double web[VDO_TILE_EPT];
#pragma unroll
  for (int q = 0; q < VDO_TILE_EPT; ++q) { const int e = ebase + min(q, jmax) * VDO_TILE_THREADS; keyb[q] = __builtin_nontemporal_load(d.eb_key + e); web[q] = __builtin_nontemporal_load(d.Finc + e);      /* streamed once per launch: past the resident lines of L2, not through them (ba_sweep.hip VDO_NT_LOAD) */ }
  for (int i = tid; i < 21 * nslot; i += VDO_TILE_THREADS) accm[i] = 0.0;
  auto stage_slot = [&](int sidx, int pid) {
    const IsoD W = iso_inv(iso_load(d.pose[0] + 12 * (int64_t)pid));
    double* o = slotW + 12 * sidx;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = W.r[i];
    o[9] = W.t.x; o[10] = W.t.y; o[11] = W.t.z;
  };
  asm volatile("" : "+v"(my_pose));         // (keeps the request where it was made: the compiler would sink it into the branch, behind a wait for every other request)
  if (tid < nslot) stage_slot(tid, my_pose);
  for (int sidx = tid + VDO_TILE_THREADS; sidx < nslot; sidx += VDO_TILE_THREADS) { stage_slot(sidx, d.tile_pose[T.slot_begin + sidx]); sdst[sidx] = d.slot_dst[T.slot_begin + sidx]; }      // (a tile of more than 256 slots: one long dynamic track)
  sdst[my_slot] = my_dst;
  // what hangs on the keys: is the point a chain of its own, and its scalar factor
  unsigned char sgl[VDO_TILE_EPT];
  double dsc[VDO_TILE_EPT];
#pragma unroll
  for (int q = 0; q < VDO_TILE_EPT; ++q) { const int64_t l = T.pt_begin + (keyb[q] >= 0 ? (keyb[q] & 0xffff) : 0); sgl[q] = d.pt_single[l]; dsc[q] = d.dscal[l]; }
  int ecnt = 0;
#pragma unroll
  for (int q = 0; q < VDO_TILE_EPT; ++q) ecnt += (q < T.ept && keyb[q] >= 0) ? 1 : 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int i = tid + k * VDO_TILE_THREADS, q3 = (i * 0xAAAB) >> 17; if (i < 3 * npts) pts[(i - 3 * q3) * VDO_TILE_PLANE + q3] = pvl[k]; }
  __syncthreads();
  {
    // EdgeSE3PointXYZ incidences: the thread's column of the tile's edge block = <= VDO_TILE_EPT consecutive edges of ONE pose slot - their contributions
    // add up in registers and go through ONE segmented scan.  A point that is a chain of its own (every static landmark) has
    // [Hll^-1]_ll = g I3, and its block B = -we [I ; 2[c]x] R^T gives  B G B^T = g we^2 [I ; 2[c]x] [I ; 2[c]x]^T  (R drops out): a function of
    // ten running sums  s, s c, s c c^T  (s = g we^2) - no 6x3 block, no 6x6 product.
    int slot = -1;
    double up[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) up[i] = 0.0;
    double s0 = 0.0, sx = 0.0, sy = 0.0, sz = 0.0, sxx = 0.0, sxy = 0.0, sxz = 0.0, syy = 0.0, syz = 0.0, szz = 0.0;
#pragma unroll
    for (int q = 0; q < VDO_TILE_EPT; ++q) {
      if (q < ecnt) {
        const int j = q * VDO_TILE_THREADS + tid;
        const int key = keyb[q];
        slot = key >> 16;
        const int64_t l = T.pt_begin + (key & 0xffff);
        const FInc f = make_f<true>(d, T, j, 0, key, web[q], slotW, pts);
        if (!DYN || sgl[q]) {
          const double sw = dsc[q] * f.we * f.we;
          const double wx = sw * f.cx, wy = sw * f.cy, wz = sw * f.cz;
          s0 += sw; sx += wx; sy += wy; sz += wz;
          sxx += wx * f.cx; sxy += wx * f.cy; sxz += wx * f.cz; syy += wy * f.cy; syz += wy * f.cz; szz += wz * f.cz;
        } else {
          double B[18], M[36];
          expand_block(0, f, slotW + 12 * slot, B);
          bgbt(B, d.Gdiag + 9 * l, B, M);
          int k = 0;
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) up[k++] += M[6 * r + c];
        }
      }
    }
    // upper triangle (row-major, 21 entries) of  s [[I, -2[c]x], [2[c]x, 4 (|c|^2 I - c c^T)]]  summed over the thread's single points
    up[0] += s0; up[4] += 2.0 * sz; up[5] += -2.0 * sy;
    up[6] += s0; up[8] += -2.0 * sz; up[10] += 2.0 * sx;
    up[11] += s0; up[12] += 2.0 * sy; up[13] += -2.0 * sx;
    up[15] += 4.0 * (syy + szz); up[16] += -4.0 * sxy; up[17] += -4.0 * sxz;
    up[18] += 4.0 * (sxx + szz); up[19] += -4.0 * syz;
    up[20] += 4.0 * (sxx + syy);
    if (nb > 0) { const SegCtl16 sc_ = seg_ctl16(slot); seg_apply16<21>(up, sc_, seg_flags(sc_), accm + 21 * (slot >= 0 ? slot : 0)); }
  }
  for (int base = 0; DYN && base < nt; base += VDO_TILE_THREADS) {
    const int j = base + tid;
    int slot = -1;
    double up[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) up[i] = 0.0;
    if (j < nt) {
      const int64_t i1 = T.inc_begin + nb + j, i2 = T.inc_begin + nb + nt + j;
      const int k1 = d.inc_key[i1], k2 = d.inc_key[i2];
      slot = k1 >> 16;
      const int64_t l1 = T.pt_begin + (k1 & 0xffff), l2 = T.pt_begin + (k2 & 0xffff);
      double B1[18], B2[18], M11[36], M12[36], M22[36];
      const FInc f = make_f<true>(d, T, nb + nt + j, 2, k2, d.Finc[(int64_t)d.Eb + T.et_begin + j], slotW, pts);
      expand_block(1, f, slotW + 12 * slot, B1);
      expand_block(2, f, slotW + 12 * slot, B2);
      bgbt(B1, d.Gdiag + 9 * l1, B1, M11);
      bgbt(B1, d.Goff + 9 * l2, B2, M12);     // [Hll^-1]_{l1,l2}, l2 = l1 + 1 in chain order
      bgbt(B2, d.Gdiag + 9 * l2, B2, M22);
      int k = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) up[k++] = M11[6 * r + c] + M22[6 * r + c] + M12[6 * r + c] + M12[6 * c + r];
    }
    { const SegCtl16 sc_ = seg_ctl16(slot); seg_apply16<21>(up, sc_, seg_flags(sc_), accm + 21 * (slot >= 0 ? slot : 0)); }
  }
  __syncthreads();
  // one row of 16 + one of 8 doubles per (tile, slot), pose-major (k_precond_finalize streams a pose's rows)
  for (int i = tid; i < 24 * nslot; i += VDO_TILE_THREADS) {
    const int sidx = i / 24, k = i - 24 * sidx;
    if (k >= 21) continue;
    const int64_t row = sdst[sidx];
    if (k < 16) d.part_m[16 * row + k] = accm[21 * sidx + k];
    else d.part_m8[8 * row + (k - 16)] = accm[21 * sidx + k];
  }
}

// M_i = (Hpp_ii + lambda I - sum partials)^-1
// SRC 0: gather this GPU's partials and invert.  Shards: SRC 1 gathers into msum (all-reduced by the
// hook), SRC 2 inverts from the reduced msum.
template <int SRC>
__global__ __launch_bounds__(256) void k_precond_finalize(BADev d, double lambda) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per pose
  if (p >= d.P) return;
  double up[21];
  if (SRC != 2) {
    const double a16 = wave_gather_rows<16>(d.part_m, d.ps_off[p], d.ps_off[p + 1]), a8 = wave_gather_rows<8>(d.part_m8, d.ps_off[p], d.ps_off[p + 1]);
#pragma unroll
    for (int i = 0; i < 16; ++i) up[i] = __shfl(a16, i, 64);
#pragma unroll
    for (int i = 0; i < 5; ++i) up[16 + i] = __shfl(a8, i, 64);
  }
  if ((threadIdx.x & 63) != 0) return;
  if (SRC == 1) {
#pragma unroll
    for (int i = 0; i < 21; ++i) d.msum[21 * (int64_t)p + i] = up[i];
    if (p == 0) d.msum[21 * (int64_t)d.P] = (double)d.flags[0];     // a failed chain factor on ANY rank must stop every rank
    return;
  }
  if (SRC == 2) {
#pragma unroll
    for (int i = 0; i < 21; ++i) up[i] = d.msum[21 * (int64_t)p + i];
    if (p == 0 && d.msum[21 * (int64_t)d.P] > 0) atomicOr(d.flags, 1);
  }
  double A[36];
  for (int i = 0; i < 36; ++i) A[i] = d.Hpp[36 * (int64_t)p + i];
  for (int i = 0; i < 6; ++i) A[7 * i] += lambda;
  int k = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) { A[6 * r + c] -= up[k]; if (c != r) A[6 * c + r] -= up[k]; ++k; }
  for (int i = 0; i < 36; ++i) d.Adg[36 * (int64_t)p + i] = A[i];      // inverted / factored per pose chain by k_pchain_factor
}

// Block LDL^T of the block-tridiagonal preconditioner  M = blockdiag(S_pp) + (EdgeSE3 off-diagonal blocks)
// along every pose chain:  Delta_0 = A_0,  L_k = E_{k-1,k}^T Delta_{k-1}^-1,  Delta_k = A_k - L_k E_{k-1,k}.
// M is SPD: blockdiag(S) minus the EdgeSE3 diagonal terms is the (PSD) Schur complement of the landmark
// system, the EdgeSE3 terms themselves are J^T W J.  A handful of chains, once per Levenberg trial; Minv / Lc are indexed by chain position.
// One wave per chain (two for a twisted chain, see k_pchain_factor), the loop body straight-line code:
//  * the LDS hand-overs use a wave-scope fence, not __syncthreads - whose s_waitcnt vmcnt(0) makes every step wait for the loads it has
//    just requested for the steps ahead and for its own stores of Lc / Minv;
//  * no lane is masked off (lanes 36..63 repeat the work of lanes 0..27 and store the same values to the same places) and the inputs of the
//    steps past the end are loads from clamped, valid addresses: with branches around the loads the compiler loses count of the loads in
//    flight and waits for ALL of them in every step (measured: 1.04 us per step = one trip to HBM; the recurrence itself is a tenth of it).
__device__ __forceinline__ void chain_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Inverse of the SPD 6x6 block held one entry per lane (lane ln = 6 r + q; lanes 36 .. 63 mirror 0 .. 27).  INV = 1 (default): the in-place Gauss-Jordan of
// rounds 2-4 in registers - six DEPENDENT pivots (a v_readlane, two ds_bpermute, a reciprocal + two Newton steps, selects).  INV = 0 (VDO_BA_PCHAIN_CLOSED=1;
// built in round 5 on the estimate that it would take two thirds of the time - it takes 5 % MORE, see the launch site - and kept for the record):
// the closed form over 3x3 blocks, A = [P Q; Q^T S]:  P^-1 by cofactors, W = P^-1 Q, T = S - Q^T W, T^-1 by cofactors, and
//   A^-1 = [P^-1 + W T^-1 W^T, -W T^-1; -(W T^-1)^T, T^-1]
// computed REDUNDANTLY by every lane from one LDS broadcast of the block (21 doubles, conflict-free: all lanes read the same address): ~170 independent-ish
// multiply-adds and two reciprocals instead of six pivots in a row; the 21 results go back through LDS (every lane writes the same values to the same
// places), where the next step of the recurrence reads them anyway.  Positive definiteness = the six leading minors (the pivots of the other form).
// sm: 36 doubles of LDS of this wave (the block as a full symmetric matrix on return); returns this lane's entry.
template <int INV>
__device__ __forceinline__ double chain_inv6(double a, int ln, int r, int q, double* sm, bool& bad) {
  if (INV == 1) {
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) {
      const double pv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), kk * 7), __builtin_amdgcn_readlane(__double2loint(a), kk * 7));
      const double aik = __shfl(a, r * 6 + kk, 64), akj = __shfl(a, kk * 6 + q, 64);
      if (!(pv > 0)) bad = true;
      double rp = __builtin_amdgcn_rcp(pv);
      rp = __builtin_fma(__builtin_fma(-pv, rp, 1.0), rp, rp);
      rp = __builtin_fma(__builtin_fma(-pv, rp, 1.0), rp, rp);      // (the second step is free: the chain waits for the permutes - measured)
      const double rowv = akj * rp, colv = -aik * rp, gen = a - aik * rowv;      // (selects, not branches: all four are a few cycles)
      const double on_row = q == kk ? rp : rowv, off_row = q == kk ? colv : gen;
      a = r == kk ? on_row : off_row;
    }
    sm[ln] = a;
    return a;
  }
  sm[ln] = a;
  chain_wave_sync();
  const double p00 = sm[0], p01 = sm[1], p02 = sm[2], p11 = sm[7], p12 = sm[8], p22 = sm[14];
  const double q00 = sm[3], q01 = sm[4], q02 = sm[5], q10 = sm[9], q11 = sm[10], q12 = sm[11], q20 = sm[15], q21 = sm[16], q22 = sm[17];
  const double s00 = sm[21], s01 = sm[22], s02 = sm[23], s11 = sm[28], s12 = sm[29], s22 = sm[35];
  chain_wave_sync();                               // (everybody has read the block: sm may be overwritten below)
  auto recip = [](double x) { double y = __builtin_amdgcn_rcp(x); y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y); return __builtin_fma(__builtin_fma(-x, y, 1.0), y, y); };
  // P^-1
  double c00 = p11 * p22 - p12 * p12, c01 = p02 * p12 - p01 * p22, c02 = p01 * p12 - p02 * p11;
  double c11 = p00 * p22 - p02 * p02, c12 = p01 * p02 - p00 * p12, c22 = p00 * p11 - p01 * p01;
  const double detp = __builtin_fma(p02, c02, __builtin_fma(p01, c01, p00 * c00));
  if (!(p00 > 0) || !(c22 > 0) || !(detp > 0)) bad = true;
  const double ip = recip(detp);
  c00 *= ip; c01 *= ip; c02 *= ip; c11 *= ip; c12 *= ip; c22 *= ip;
  // W = P^-1 Q
  const double w00 = __builtin_fma(c02, q20, __builtin_fma(c01, q10, c00 * q00)), w01 = __builtin_fma(c02, q21, __builtin_fma(c01, q11, c00 * q01)), w02 = __builtin_fma(c02, q22, __builtin_fma(c01, q12, c00 * q02));
  const double w10 = __builtin_fma(c12, q20, __builtin_fma(c11, q10, c01 * q00)), w11 = __builtin_fma(c12, q21, __builtin_fma(c11, q11, c01 * q01)), w12 = __builtin_fma(c12, q22, __builtin_fma(c11, q12, c01 * q02));
  const double w20 = __builtin_fma(c22, q20, __builtin_fma(c12, q10, c02 * q00)), w21 = __builtin_fma(c22, q21, __builtin_fma(c12, q11, c02 * q01)), w22 = __builtin_fma(c22, q22, __builtin_fma(c12, q12, c02 * q02));
  // T = S - Q^T W  (symmetric)
  const double t00 = s00 - __builtin_fma(q20, w20, __builtin_fma(q10, w10, q00 * w00)), t01 = s01 - __builtin_fma(q20, w21, __builtin_fma(q10, w11, q00 * w01));
  const double t02 = s02 - __builtin_fma(q20, w22, __builtin_fma(q10, w12, q00 * w02)), t11 = s11 - __builtin_fma(q21, w21, __builtin_fma(q11, w11, q01 * w01));
  const double t12 = s12 - __builtin_fma(q21, w22, __builtin_fma(q11, w12, q01 * w02)), t22 = s22 - __builtin_fma(q22, w22, __builtin_fma(q12, w12, q02 * w02));
  // T^-1
  double d00 = t11 * t22 - t12 * t12, d01 = t02 * t12 - t01 * t22, d02 = t01 * t12 - t02 * t11;
  double d11 = t00 * t22 - t02 * t02, d12 = t01 * t02 - t00 * t12, d22 = t00 * t11 - t01 * t01;
  const double dett = __builtin_fma(t02, d02, __builtin_fma(t01, d01, t00 * d00));
  if (!(t00 > 0) || !(d22 > 0) || !(dett > 0)) bad = true;
  const double it = recip(dett);
  d00 *= it; d01 *= it; d02 *= it; d11 *= it; d12 *= it; d22 *= it;
  // X = -W T^-1   (the upper right block)
  const double x00 = -__builtin_fma(w02, d02, __builtin_fma(w01, d01, w00 * d00)), x01 = -__builtin_fma(w02, d12, __builtin_fma(w01, d11, w00 * d01)), x02 = -__builtin_fma(w02, d22, __builtin_fma(w01, d12, w00 * d02));
  const double x10 = -__builtin_fma(w12, d02, __builtin_fma(w11, d01, w10 * d00)), x11 = -__builtin_fma(w12, d12, __builtin_fma(w11, d11, w10 * d01)), x12 = -__builtin_fma(w12, d22, __builtin_fma(w11, d12, w10 * d02));
  const double x20 = -__builtin_fma(w22, d02, __builtin_fma(w21, d01, w20 * d00)), x21 = -__builtin_fma(w22, d12, __builtin_fma(w21, d11, w20 * d01)), x22 = -__builtin_fma(w22, d22, __builtin_fma(w21, d12, w20 * d02));
  // upper left: P^-1 - X W^T  (symmetric)
  const double u00 = c00 - __builtin_fma(x02, w02, __builtin_fma(x01, w01, x00 * w00)), u01 = c01 - __builtin_fma(x02, w12, __builtin_fma(x01, w11, x00 * w10));
  const double u02 = c02 - __builtin_fma(x02, w22, __builtin_fma(x01, w21, x00 * w20)), u11 = c11 - __builtin_fma(x12, w12, __builtin_fma(x11, w11, x10 * w10));
  const double u12 = c12 - __builtin_fma(x12, w22, __builtin_fma(x11, w21, x10 * w20)), u22 = c22 - __builtin_fma(x22, w22, __builtin_fma(x21, w21, x20 * w20));
  // back through LDS as the full symmetric matrix (same values from every lane)
  sm[0] = u00; sm[1] = u01; sm[2] = u02; sm[3] = x00; sm[4] = x01; sm[5] = x02;
  sm[6] = u01; sm[7] = u11; sm[8] = u12; sm[9] = x10; sm[10] = x11; sm[11] = x12;
  sm[12] = u02; sm[13] = u12; sm[14] = u22; sm[15] = x20; sm[16] = x21; sm[17] = x22;
  sm[18] = x00; sm[19] = x10; sm[20] = x20; sm[21] = d00; sm[22] = d01; sm[23] = d02;
  sm[24] = x01; sm[25] = x11; sm[26] = x21; sm[27] = d01; sm[28] = d11; sm[29] = d12;
  sm[30] = x02; sm[31] = x12; sm[32] = x22; sm[33] = d02; sm[34] = d12; sm[35] = d22;
  chain_wave_sync();
  return sm[ln];
}

// Two waves per chain (round 5).  An untwisted chain: wave 0 walks it, wave 1 has nothing to do.  A TWISTED chain (capi_ba.hip): wave 0 factorises the first
// half [b, far], wave 1 the second half - stored backwards, its first position has no link - [far + 1, e - 2]; then wave 1 does the joint (position e - 1), which
// takes Delta^-1 of BOTH halves' last positions:  Delta_joint = A - L E - L_far E_far.  Half the depth of the one recurrence that cannot be partitioned.
template <int INV>
__global__ __launch_bounds__(128) void k_pchain_factor(BADev d) {
  // lane = 6*row + col of a 6x6 block, blocks exchanged through LDS (one set per wave)
  __shared__ __attribute__((aligned(16))) double sEw[2][36], sDw[2][36], sLw[2][36];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ln = lane < 36 ? lane : lane - 36;
  const int r = ln / 6, q = ln % 6;
  const int cb = d.pc_off[c], ce = d.pc_off[c + 1];
  const int far = d.pc_far_pos[c];                                      // -1: not twisted
  const int b = far < 0 ? cb : (wave == 0 ? cb : far + 1);              // this wave's range [b, e)
  const int e = far < 0 ? (wave == 0 ? ce : cb) : (wave == 0 ? far + 1 : ce - 1);
  double* sE = sEw[wave]; double* sD = sDw[wave]; double* sL = sLw[wave];
  bool bad = false;
  // the inputs of a step (A_k, E_{k-1,k}) do not depend on the recursion and sit behind a two-level pointer chase through HBM
  // (pc_pose / pc_edge -> Adg / Hpp_ep): they are requested a CHUNK of 8 steps ahead, their indices two chunks ahead, into register sets
  // that rotate at the end of a chunk (a rotation inside every step would wait for the loads of that very step)
  constexpr int CH = 8;
  const double* __restrict__ Hbase = d.Ep > 0 ? d.Hpp_ep : d.Adg;      // (a graph without EdgeSE3: every entry is -1, the address only has to be valid)
  const int last = ce - 1;
  auto idx_p = [&](int k) -> int { return d.pc_pose[k < e ? k : last]; };
  auto idx_e = [&](int k) -> int { const int t = d.pc_edge[k < e ? k : last]; return k < e ? t : -1; };      // (-1: a chain head, or the head of a twisted chain's second half)
  auto fetch_a = [&](int p) -> double { return d.Adg[36 * (int64_t)p + ln]; };
  auto fetch_e = [&](int ent) -> double {
    const int en = ent < 0 ? 0 : ent;
    return Hbase[36 * (int64_t)(en >> 1) + ((en & 1) ? q * 6 + r : ln)];      // E = block (previous pose, this pose); ent < 0: unused
  };
  // the joint's inputs (wave 1 of a twisted chain), requested up front
  double Aj = 0.0, Ej = 0.0, Ef = 0.0;
  if (far >= 0) { Aj = fetch_a(d.pc_pose[last]); Ej = fetch_e(d.pc_edge[last]); Ef = fetch_e(d.pc_far_edge[c]); }
  int pi[CH], ti[CH];
  double A[CH], E[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { pi[j] = idx_p(b + j); ti[j] = idx_e(b + j); }
#pragma unroll
  for (int j = 0; j < CH; ++j) { A[j] = fetch_a(pi[j]); E[j] = ti[j] < 0 ? 0.0 : fetch_e(ti[j]); }
#pragma unroll
  for (int j = 0; j < CH; ++j) { pi[j] = idx_p(b + CH + j); ti[j] = idx_e(b + CH + j); }
  sD[ln] = 0.0;
  for (int k0 = b; k0 < e; k0 += CH) {
    double An[CH], En[CH];
    int pn[CH], tn[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { An[j] = fetch_a(pi[j]); En[j] = fetch_e(ti[j]); }
#pragma unroll
    for (int j = 0; j < CH; ++j) { pn[j] = idx_p(k0 + 2 * CH + j); tn[j] = idx_e(k0 + 2 * CH + j); }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int k = k0 + j;
      if (k < e) {
        double a = A[j];
        sE[ln] = E[j];                       // (0 at the head of the chain: L = 0, Delta = A)
        chain_wave_sync();
        // (dot products as two fused chains of three: the step is a latency chain, 4 dependent operations instead of 12)
        double t = __builtin_fma(sE[12 + r], sD[12 + q], __builtin_fma(sE[6 + r], sD[6 + q], sE[r] * sD[q])) +
                   __builtin_fma(sE[30 + r], sD[30 + q], __builtin_fma(sE[24 + r], sD[24 + q], sE[18 + r] * sD[18 + q]));   // L = E^T Delta_prev^-1
        sL[ln] = t; d.Lc[36 * (int64_t)k + ln] = t;
        chain_wave_sync();
        t = __builtin_fma(sL[r * 6 + 2], sE[12 + q], __builtin_fma(sL[r * 6 + 1], sE[6 + q], sL[r * 6] * sE[q])) +
            __builtin_fma(sL[r * 6 + 5], sE[30 + q], __builtin_fma(sL[r * 6 + 4], sE[24 + q], sL[r * 6 + 3] * sE[18 + q]));      // Delta = A - L E
        a -= t;
        a = chain_inv6<INV>(a, ln, r, q, sD, bad);      // (leaves the inverse in sD: the next step's Delta_prev^-1)
        d.Minv[36 * (int64_t)k + ln] = a;
        chain_wave_sync();
      }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) { A[j] = An[j]; E[j] = ti[j] < 0 ? 0.0 : En[j]; pi[j] = pn[j]; ti[j] = tn[j]; }
  }
  if (far >= 0) {                                    // (uniform over the workgroup)
    __syncthreads();                                 // both halves done: sDw[0] = Delta_far^-1, sDw[1] = Delta^-1 of position e - 2 ... of the chain
    if (wave == 1) {
      const double* sD0 = sDw[0];
      double a = Aj;
      sE[ln] = Ej;
      chain_wave_sync();
      double t = __builtin_fma(sE[12 + r], sD[12 + q], __builtin_fma(sE[6 + r], sD[6 + q], sE[r] * sD[q])) +
                 __builtin_fma(sE[30 + r], sD[30 + q], __builtin_fma(sE[24 + r], sD[24 + q], sE[18 + r] * sD[18 + q]));       // L = E^T Delta_prev^-1  (ordinary link)
      sL[ln] = t; d.Lc[36 * (int64_t)last + ln] = t;
      chain_wave_sync();
      a -= __builtin_fma(sL[r * 6 + 2], sE[12 + q], __builtin_fma(sL[r * 6 + 1], sE[6 + q], sL[r * 6] * sE[q])) +
           __builtin_fma(sL[r * 6 + 5], sE[30 + q], __builtin_fma(sL[r * 6 + 4], sE[24 + q], sL[r * 6 + 3] * sE[18 + q]));
      chain_wave_sync();
      sE[ln] = Ef;
      chain_wave_sync();
      t = __builtin_fma(sE[12 + r], sD0[12 + q], __builtin_fma(sE[6 + r], sD0[6 + q], sE[r] * sD0[q])) +
          __builtin_fma(sE[30 + r], sD0[30 + q], __builtin_fma(sE[24 + r], sD0[24 + q], sE[18 + r] * sD0[18 + q]));           // L_far = E_far^T Delta_far^-1
      sL[ln] = t; d.Lfar[36 * (int64_t)c + ln] = t;
      chain_wave_sync();
      a -= __builtin_fma(sL[r * 6 + 2], sE[12 + q], __builtin_fma(sL[r * 6 + 1], sE[6 + q], sL[r * 6] * sE[q])) +
           __builtin_fma(sL[r * 6 + 5], sE[30 + q], __builtin_fma(sL[r * 6 + 4], sE[24 + q], sL[r * 6 + 3] * sE[18 + q]));
      a = chain_inv6<INV>(a, ln, r, q, sD, bad);
      d.Minv[36 * (int64_t)last + ln] = a;
    }
  }
  if (bad && lane == 0) atomicOr(d.flags, 1);
}

// z = M^-1 r along ONE pose chain (forward / diagonal / backward block substitution) with r and z in global memory - the
// path for chains whose strip does not fit the LDS (pc_lds = 0).  Called by the single wave of the chain's workgroup;
// lane (row, col) of a 6x6 block = (lane >> 3, lane & 7); each step is one block mat-vec: multiply, 3 xor-shuffles over the
// columns, one broadcast of the new 6-vector.  The next step's block elements are fetched before the current
// step's arithmetic (they depend only on k), so the dependent chain per step is ALU + shuffles.
__device__ __forceinline__ double row_sum6(double t) {     // sum over the 8-lane group (columns 6,7 carry zeros)
  t += __shfl_xor(t, 1, 64);
  t += __shfl_xor(t, 2, 64);
  t += __shfl_xor(t, 4, 64);
  return t;
}
__device__ void pchain_apply(const BADev& d, int c, const double* __restrict__ r, double* __restrict__ z) {
  const int lane = threadIdx.x & 63;
  const int row = lane >> 3, col = lane & 7;
  const bool act = row < 6 && col < 6;
  const int el = act ? row * 6 + col : 0, elT = act ? col * 6 + row : 0;
  {
    const int b = d.pc_off[c], e = d.pc_off[c + 1];
    // ---- forward: y_b = r_b ; y_k = r_k - L_k y_{k-1}      (y kept in z)
    int64_t p = d.pc_pose[b];
    double ycol = act ? r[6 * p + col] : 0.0;
    if (act && col == 0) z[6 * p + row] = r[6 * p + row];
    double lnext = (b + 1 < e && act) ? d.Lc[36 * (int64_t)(b + 1) + el] : 0.0;
    for (int k = b + 1; k < e; ++k) {
      const double lk = lnext;
      if (k + 1 < e && act) lnext = d.Lc[36 * (int64_t)(k + 1) + el];
      p = d.pc_pose[k];
      const double rk = act ? r[6 * p + row] : 0.0;
      const double ynew = rk - row_sum6(act ? lk * ycol : 0.0);
      if (act && col == 0) z[6 * p + row] = ynew;
      ycol = __shfl(ynew, (col < 6 ? col : 0) * 8, 64);
    }
    __threadfence_block();      // the y parked in z by other lanes of this wave must be visible to the loads below
    // (twisted chain: the joint - the last position - also hangs on position far:  y_last -= L_far y_far, and on the way back z_far -= L_far^T z_last)
    const int far = d.pc_far_pos[c];
    const double lf = (far >= 0 && act) ? d.Lfar[36 * (int64_t)c + el] : 0.0, lfT = (far >= 0 && act) ? d.Lfar[36 * (int64_t)c + elT] : 0.0;
    if (far >= 0) {
      const int64_t pf = d.pc_pose[far], pl = d.pc_pose[e - 1];
      const double yf = act ? z[6 * pf + col] : 0.0;
      const double ynew = (act ? z[6 * pl + row] : 0.0) - row_sum6(act ? lf * yf : 0.0);
      __builtin_amdgcn_wave_barrier();
      if (act && col == 0) z[6 * pl + row] = ynew;
      ycol = __shfl(ynew, (col < 6 ? col : 0) * 8, 64);
      __threadfence_block();
    }
    // ---- backward: z_last = Dinv y ; z_k = Dinv_k y_k - L_{k+1}^T z_{k+1}
    //      after the forward loop ycol holds y_{e-1}[col]
    double zcol = 0.0, zlast = 0.0;
    double dnext = act ? d.Minv[36 * (int64_t)(e - 1) + el] : 0.0;
    double ltnext = 0.0;
    for (int k = e - 1; k >= b; --k) {
      const double dk = dnext, ltk = ltnext;       // ltk = L_{k+1}^T element (0 at the chain end)
      if (k - 1 >= b && act) { dnext = d.Minv[36 * (int64_t)(k - 1) + el]; ltnext = d.Lc[36 * (int64_t)k + elT]; }
      p = d.pc_pose[k];
      if (k < e - 1) ycol = act ? z[6 * p + col] : 0.0;       // y_k was parked in z by the forward pass
      const double farterm = k == far ? lfT * zlast : 0.0;      // z_far = Dinv_far y_far - L_{far+1}^T z_{far+1} - L_far^T z_last   (L_{far+1} = 0: the second half starts there)
      const double zk = row_sum6(act ? (dk * ycol - ltk * zcol - farterm) : 0.0);
      __builtin_amdgcn_wave_barrier();
      if (act && col == 0) z[6 * p + row] = zk;
      zcol = __shfl(zk, (col < 6 ? col : 0) * 8, 64);
      if (k == e - 1) zlast = zcol;
    }
  }
}

// The same operator with the chain in LDS, partitioned over the waves of the chain's workgroup.  k_pcg_chain forms r of the chain
// straight into the LDS strip yb [len][6]; the substitutions overwrite it in place (r -> y -> Dinv y -> z).
//
// A wave alone would walk the 2 (len - 1) dependent block mat-vecs of the two substitutions one after the other (~120 ns each: 49 us on the
// 200-pose chains of the roofline graph).  The recurrences are linear, so the chain is cut into S <= 16 segments of G positions, one wave each:
//   forward   y_k = r_k - L_k y_{k-1}:  every segment runs its recurrence from a ZERO input (yhat); the true value differs by P_k y_in, where y_in is
//             the true y at the end of the segment before and P_k = (-L_k)(-L_{k-1})...(-L_{k0}) depends on the factorisation only (k_pchain_prefix
//             stores it next to Lc, once per Levenberg trial).  Wave 0 carries y_in across the S - 1 boundaries, then every position is corrected
//             independently (no recurrence: one lane per (position, row));
//   backward  z_k = w_k - L_{k+1}^T z_{k+1}: the same with Q_k = (-L_{k+1}^T)...(-L_{e+1}^T) and the first z of the segment behind.
// Dependent steps per substitution: G + S instead of len (200 -> 29).
// Inside a segment: lane = (a, b) = (lane >> 3, lane & 7) holds one entry of the 6x6 block; a block mat-vec is one multiply and one 8-lane
// all-reduce - over the octet (sum over b) or over the lanes of equal b (sum over a).  The two kinds ALTERNATE: a vector that comes out
// indexed by a (replicated along the octet) is consumed by a step that holds its block transposed and reduces over a, whose result is indexed
// by b (replicated across the octets) - no lane permutation between steps.  The blocks do not depend on the recurrence: fetched eight steps ahead.
// -DPCG_PROF (tools/build_variant.sh pcgprof "-DPCG_PROF" ba_solve; debug builds only): shader-clock time stamps of thread 0 of workgroup 0 of k_pcg_chain<0>
// at the phase boundaries (tools/pcg_chain_phase_probe.py)
#ifdef PCG_PROF
__device__ unsigned long long g_pcg_prof[20];
__device__ __noinline__ long long* pcg_tk() { __shared__ long long tk[20]; return tk; }
#define PCG_TICK(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) pcg_tk()[slot] = clock64(); } while (0)
#else
#define PCG_TICK(slot) do { } while (0)
#endif
// LDS block of a chain's workgroup behind its strip, boundary vectors and reduction scratch: 14 boundary matrices P | 14 boundary matrices Q | L of the far link
#define PC_BMAT_P 0
#define PC_BMAT_Q (14 * 36)
#define PC_BMAT_F (28 * 36)
#define PC_BMAT_DOUBLES (29 * 36)
__host__ __device__ inline int pc_seg_len(int len, int nwave) { const int g = (len + nwave - 1) / nwave; return g < 8 ? 8 : g; }

// What a wave's part of the partitioned solve reads FIRST in each of its phases - the first chunk of L blocks of the two recurrences, the first two rounds of
// Delta^-1 rows - depends on the factorisation only: k_pcg_chain requests it at its head (pc_prefetch), under alpha and the vector updates, instead of paying a
// memory latency at the head of every phase (round 5, tools/pcg_chain_phase_probe.py: 2 x 9.7 k + 3.5 k cycles of the kernel's 57 k on 240-pose chains).
struct PcPre { double fL[8], bL[8], D[2][6]; };
__device__ __forceinline__ void pc_prefetch(const BADev& d, int bgn, int len, PcPre& q) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int G = pc_seg_len(len, nwave), S = (len + G - 1) / G;
  const int k0 = wave * G, n = wave < S ? (len - k0 < G ? len - k0 : G) : 0;
  const int64_t c0 = (int64_t)bgn + (n > 0 ? k0 : 0), last = (int64_t)bgn + len - 1;
  const int a = lane >> 3, b = lane & 7;
  const bool act = a < 6 && b < 6;
  const int el = act ? a * 6 + b : 0, elT = act ? b * 6 + a : 0;
  const int nst = n - 1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {                               // (unconditional loads from clamped positions; what lies outside the segment is zeroed where it is used)
    const int64_t pf = c0 + 1 + j, pb = c0 + n - 1 - j;
    q.fL[j] = d.Lc[36 * (pf <= last ? pf : last) + ((j & 1) ? elT : el)];
    q.bL[j] = d.Lc[36 * (pb >= bgn ? (pb <= last ? pb : last) : bgn) + ((j & 1) ? el : elT)];
    if (!(j < nst && act)) { q.fL[j] = 0.0; q.bL[j] = 0.0; }
  }
  const int pr_k = lane / 6, pr_a = lane - 6 * pr_k;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t pd = c0 + 10 * h + pr_k;
    const double* D = d.Minv + 36 * (pd <= last ? pd : last) + 6 * (pr_a < 6 ? pr_a : 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) q.D[h][i] = D[i];
  }
}

// y_0 = yb_0 ; y_j = yb_j - Lc[lc0 + j] y_{j-1}   (j < n), in place; one wave.  L0: the first chunk of blocks (pc_prefetch)
__device__ void pseg_forward(const BADev& d, int64_t lc0, int n, double* yb, const double (&L0)[8]) {
  const int lane = threadIdx.x & 63;
  const int a = lane >> 3, b = lane & 7;
  const bool act = a < 6 && b < 6;
  const int el = act ? a * 6 + b : 0, elT = act ? b * 6 + a : 0;
  const int ia = a < 6 ? a : 0, ib = b < 6 ? b : 0;
  // Step s = j - 1: even s holds L natural (vector indexed by b in, by a out), odd s holds it transposed (a in, b out)
  double v = b < 6 ? yb[ib] : 0.0;                       // y_0[b]
  const int nst = n - 1;
  double Lp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) Lp[j] = L0[j];
  for (int s0 = 0; s0 < nst; s0 += 8) {
    double Ln[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) Ln[j] = (s0 + 8 + j < nst && act) ? d.Lc[36 * (lc0 + 1 + s0 + 8 + j) + ((j & 1) ? elT : el)] : 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = s0 + j + 1;
      if (k < n) {
        if ((j & 1) == 0) {
          const double sum = octet_allsum(Lp[j] * v);
          v = yb[6 * k + ia] - sum;
          if (b == 0 && a < 6) yb[6 * k + a] = v;
        } else {
          const double sum = stride8_allsum(Lp[j] * v);
          v = yb[6 * k + ib] - sum;
          if (a == 0 && b < 6) yb[6 * k + b] = v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) Lp[j] = Ln[j];
  }
}

// z_{n-1} = yb_{n-1} ; z_j = yb_j - Lc[lc0 + j + 1]^T z_{j+1}, in place; one wave.  L0: the first chunk of blocks (pc_prefetch)
__device__ void pseg_backward(const BADev& d, int64_t lc0, int n, double* yb, const double (&L0)[8]) {
  const int lane = threadIdx.x & 63;
  const int a = lane >> 3, b = lane & 7;
  const bool act = a < 6 && b < 6;
  const int el = act ? a * 6 + b : 0, elT = act ? b * 6 + a : 0;
  const int ia = a < 6 ? a : 0, ib = b < 6 ? b : 0;
  // Step t = n - 2 - j: even t holds L transposed (b in, a out), odd t natural (a in, b out)
  const int nst = n - 1;
  double v = b < 6 ? yb[6 * (n - 1) + ib] : 0.0;          // z_last[b]
  double Lp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) Lp[j] = L0[j];
  for (int t0 = 0; t0 < nst; t0 += 8) {
    double Ln[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) Ln[j] = (t0 + 8 + j < nst && act) ? d.Lc[36 * (lc0 + n - 1 - (t0 + 8 + j)) + ((j & 1) ? el : elT)] : 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = n - 2 - (t0 + j);
      if (k >= 0) {
        if ((j & 1) == 0) {
          const double sum = octet_allsum(Lp[j] * v);
          v = yb[6 * k + ia] - sum;
          if (b == 0 && a < 6) yb[6 * k + a] = v;
        } else {
          const double sum = stride8_allsum(Lp[j] * v);
          v = yb[6 * k + ib] - sum;
          if (a == 0 && b < 6) yb[6 * k + b] = v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) Lp[j] = Ln[j];
  }
}

// lane i's double to every lane (i a compile-time constant): two v_readlane_b32 - no trip through the LDS crossbar, unlike __shfl
__device__ __forceinline__ double bcast_lane(double x, int i) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), i), __builtin_amdgcn_readlane(__double2loint(x), i));
}
__device__ __forceinline__ void wave_lds_sync() {      // LDS writes of this wave visible to its other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// yb [len][6] (r of the chain, staged by ALL threads of the workgroup before the call) -> z.  Called by every thread of the workgroup
// (blockDim = 64 * pc_nwave); bnd: [16][6] doubles of LDS for the boundary vectors.  Ends with a barrier.
__device__ void pchain_solve_partitioned(const BADev& d, int c, int bgn, int len, double* yb, double* bnd, const double* bmat, const PcPre& pre) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int G = pc_seg_len(len, nwave), S = (len + G - 1) / G;
  const int k0 = wave * G, n = wave < S ? (len - k0 < G ? len - k0 : G) : 0;
  const int64_t c0 = (int64_t)bgn + k0;                          // chain position of the segment's first pose
  const int pr_k = lane / 6, pr_a = lane - 6 * pr_k;             // lane -> (position inside a round of 10, row) for the recurrence-free passes
  __syncthreads();
  PCG_TICK(3);
  if (n > 0) pseg_forward(d, c0, n, yb + 6 * k0, pre.fL);
  __syncthreads();
  PCG_TICK(4);
  if (wave == 0 && S > 1) {                                      // Y_s = true y at the last position of segment s, s = 0 .. S-2 -> bnd[6 s]
    const int a = lane < 6 ? lane : 0;
    double Y = yb[6 * (G - 1) + a];
    if (lane < 6) bnd[a] = Y;
    // (the S - 2 boundary matrices P of this carry and Q of the one on the way back sit in LDS - bmat, staged by k_pcg_chain at its head: read from HBM one step
    //  ahead, as until round 5, every step of the carry waited a whole memory latency: 2 x 14 k of the kernel's 70 k cycles on 240-pose chains, tools/pcg_chain_phase_probe.py)
    const double* bP = bmat + PC_BMAT_P;
    double row[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) row[i] = S > 2 ? bP[6 * a + i] : 0.0;
    for (int s = 1; s <= S - 2; ++s) {
      const int e = (s + 1) * G - 1;
      double nx[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) nx[i] = s + 1 <= S - 2 ? bP[36 * s + 6 * a + i] : 0.0;
      double acc = yb[6 * e + a];
#pragma unroll
      for (int i = 0; i < 6; ++i) acc += row[i] * bcast_lane(Y, i);
      Y = acc;
      if (lane < 6) bnd[6 * s + a] = Y;
#pragma unroll
      for (int i = 0; i < 6; ++i) row[i] = nx[i];
    }
  }
  __syncthreads();
  PCG_TICK(5);
  // A TWISTED chain (capi_ba.hip; uniform over the workgroup): its last position - the joint - also hangs on position far_l, the end of the first half.  The
  // second half starts with L = 0, so the recurrences above and below run over the whole strip unchanged; what the far link adds is one block product on the
  // way down (y_last -= L_far y_far, once y_far is final) and one on the way up (w_far -= L_far^T z_last, z_last = w_last, before the backward recurrences start).
  const int far_l = d.pc_far_pos[c] < 0 ? -1 : d.pc_far_pos[c] - bgn;
  const double* Lf = bmat + PC_BMAT_F;                          // (L of the far link, staged with the boundary matrices)
  if (n > 0 && wave > 0) {                                       // y_k = yhat_k + P_k y_in
    double yin[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) yin[i] = bnd[6 * (wave - 1) + i];
    for (int base = 0; base < n; base += 10) {
      const int k = base + pr_k;
      if (lane < 60 && k < n) {
        const double* P = d.Pf + 36 * (c0 + k) + 6 * pr_a;
        double acc = yb[6 * (k0 + k) + pr_a];
#pragma unroll
        for (int i = 0; i < 6; ++i) acc += P[i] * yin[i];
        yb[6 * (k0 + k) + pr_a] = acc;
      }
    }
    wave_lds_sync();
  }
  if (far_l >= 0) {
    __syncthreads();
    if (threadIdx.x < 6) {
      const int a = threadIdx.x;
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) acc += Lf[6 * a + i] * yb[6 * far_l + i];
      yb[6 * (len - 1) + a] -= acc;
    }
    __syncthreads();
  }
  PCG_TICK(6);
  if (n > 0) {
    for (int base = 0; base < n; base += 10) {                   // w_k = Dinv_k y_k  (a round reads its 10 positions, then writes them)
      const int k = base + pr_k;
      const bool on = lane < 60 && k < n;
      double w = 0.0;
      if (on) {
        const double* y = yb + 6 * (k0 + k);
        if (base < 20) {                                         // (the first two rounds' rows came with pc_prefetch; uniform branch)
#pragma unroll
          for (int i = 0; i < 6; ++i) w += (base == 0 ? pre.D[0][i] : pre.D[1][i]) * y[i];
        } else {
          const double* D = d.Minv + 36 * (c0 + k) + 6 * pr_a;
#pragma unroll
          for (int i = 0; i < 6; ++i) w += D[i] * y[i];
        }
      }
      wave_lds_sync();
      if (on) yb[6 * (k0 + k) + pr_a] = w;
    }
    wave_lds_sync();
  }
  if (far_l >= 0) {
    __syncthreads();
    if (threadIdx.x < 6) {
      const int a = threadIdx.x;
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) acc += Lf[6 * i + a] * yb[6 * (len - 1) + i];
      yb[6 * far_l + a] -= acc;
    }
    __syncthreads();
  }
  PCG_TICK(7);
  if (n > 0) pseg_backward(d, c0, n, yb + 6 * k0, pre.bL);
  __syncthreads();
  PCG_TICK(8);
  if (wave == 0 && S > 1) {                                      // Z_s = true z at the first position of segment s, s = S-1 .. 1 -> bnd[6 s]
    const int a = lane < 6 ? lane : 0;
    double Z = yb[6 * ((S - 1) * G) + a];
    if (lane < 6) bnd[6 * (S - 1) + a] = Z;
    const double* bQ = bmat + PC_BMAT_Q;                         // bQ[36 (s - 1)] = Q of position s G, s = 1 .. S-2
    double row[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) row[i] = S > 2 ? bQ[36 * (S - 3) + 6 * a + i] : 0.0;
    for (int s = S - 2; s >= 1; --s) {
      const int f = s * G;
      double nx[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) nx[i] = s - 1 >= 1 ? bQ[36 * (s - 2) + 6 * a + i] : 0.0;
      double acc = yb[6 * f + a];
#pragma unroll
      for (int i = 0; i < 6; ++i) acc += row[i] * bcast_lane(Z, i);
      Z = acc;
      if (lane < 6) bnd[6 * s + a] = Z;
#pragma unroll
      for (int i = 0; i < 6; ++i) row[i] = nx[i];
    }
  }
  __syncthreads();
  PCG_TICK(9);
  if (n > 0 && wave < S - 1) {                                   // z_k = zhat_k + Q_k z_in
    double zin[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) zin[i] = bnd[6 * (wave + 1) + i];
    for (int base = 0; base < n; base += 10) {
      const int k = base + pr_k;
      if (lane < 60 && k < n) {
        const double* Q = d.Qb + 36 * (c0 + k) + 6 * pr_a;
        double acc = yb[6 * (k0 + k) + pr_a];
#pragma unroll
        for (int i = 0; i < 6; ++i) acc += Q[i] * zin[i];
        yb[6 * (k0 + k) + pr_a] = acc;
      }
    }
  }
  __syncthreads();
}

// P_k and Q_k of the partitioned substitutions (see above) for every chain position, from Lc: one workgroup per chain, one wave per segment,
// G dependent 6x6 products each way.  Once per Levenberg trial, behind k_pchain_factor.
__global__ __launch_bounds__(1024) void k_pchain_prefix(BADev d) {
  __shared__ double scr[16][36];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int bgn = d.pc_off[c], len = d.pc_off[c + 1] - bgn;
  const int G = pc_seg_len(len, nwave), S = (len + G - 1) / G;
  if (wave >= S || S < 2) return;
  const int k0 = wave * G, n = len - k0 < G ? len - k0 : G;
  const int a = lane >> 3, b = lane & 7;
  const bool act = a < 6 && b < 6;
  const int ia = act ? a : 0, ib = act ? b : 0;
  double* P = scr[wave];
  if (wave > 0) {                       // P_{k0} = -L_{k0} ; P_k = -L_k P_{k-1}
    double v = act ? -d.Lc[36 * ((int64_t)bgn + k0) + 6 * ia + ib] : 0.0;
    if (act) d.Pf[36 * ((int64_t)bgn + k0) + 6 * a + b] = v;
    double Lr[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) Lr[m] = n > 1 ? d.Lc[36 * ((int64_t)bgn + k0 + 1) + 6 * ia + m] : 0.0;
    for (int j = 1; j < n; ++j) {
      double Ln[6];
#pragma unroll
      for (int m = 0; m < 6; ++m) Ln[m] = j + 1 < n ? d.Lc[36 * ((int64_t)bgn + k0 + j + 1) + 6 * ia + m] : 0.0;
      if (act) P[6 * a + b] = v;
      wave_lds_sync();
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) t += Lr[m] * P[6 * m + ib];
      wave_lds_sync();
      v = -t;
      if (act) d.Pf[36 * ((int64_t)bgn + k0 + j) + 6 * a + b] = v;
#pragma unroll
      for (int m = 0; m < 6; ++m) Lr[m] = Ln[m];
    }
  }
  if (wave < S - 1) {                   // Q_e = -L_{e+1}^T ; Q_k = -L_{k+1}^T Q_{k+1}     (e = last position of the segment)
    const int e = k0 + n - 1;
    double v = act ? -d.Lc[36 * ((int64_t)bgn + e + 1) + 6 * ib + ia] : 0.0;
    if (act) d.Qb[36 * ((int64_t)bgn + e) + 6 * a + b] = v;
    double Lr[6];                       // column a of L_{k+1}
#pragma unroll
    for (int m = 0; m < 6; ++m) Lr[m] = n > 1 ? d.Lc[36 * ((int64_t)bgn + e) + 6 * m + ia] : 0.0;
    for (int k = e - 1; k >= k0; --k) {
      double Ln[6];
#pragma unroll
      for (int m = 0; m < 6; ++m) Ln[m] = k - 1 >= k0 ? d.Lc[36 * ((int64_t)bgn + k) + 6 * m + ia] : 0.0;
      if (act) P[6 * a + b] = v;
      wave_lds_sync();
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) t += Lr[m] * P[6 * m + ib];
      wave_lds_sync();
      v = -t;
      if (act) d.Qb[36 * ((int64_t)bgn + k) + 6 * a + b] = v;
#pragma unroll
      for (int m = 0; m < 6; ++m) Lr[m] = Ln[m];
    }
  }
}

// Tile operator.  MODE 0: part_q = B Hll^-1 B^T v     (Schur mat-vec)
//                 MODE 1: part_q = B Hll^-1 bl        (reduced rhs)
//                 MODE 2: xl     = Hll^-1 (bl - B^T v) (back-substitution)
// The landmark factors are NOT staged in LDS: a point that is a chain of its own (every static landmark) needs one scalar, read where it is
// used; the multi-point chains (dynamic tracks) stream their 3x3 factors from HBM one step ahead of the recursion.  12 KB + 192 B per pose
// slot of LDS instead of 54 KB + 168 B.  Measured on the roofline graph (2 190 poses, 690 k points of which 87 % static), mat-vec pass:
// staged for every point 57.6 us; this form 40.3 us; factors requested TWO steps ahead 48 us (170 VGPRs); separate launches for static tiles
// (no chain code) and dynamic tiles (factors staged) 23 + 22 us.
template <int MODE>
__global__ __launch_bounds__(VDO_TILE_THREADS, 4) void k_schur_tile(BADev d, const double* __restrict__ v, const double* __restrict__ v2) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // The head of a tile is a chain of dependent loads (ba_sweep.hip, same order here): every request is UNCONDITIONAL (clamped index: a load
  // under a branch is waited for at the end of the branch) and made as soon as its address is known -
  //   descriptor -> this thread's <= VDO_TILE_EPT EdgeSE3PointXYZ incidences (key, we), slot pose ids, points -> poses, v of the slots.
  const int tid = threadIdx.x;
  if (MODE == 0 && d.flags[1]) return;     // PCG already converged: the launches queued behind it are no-ops
  const Tile T = d.tiles[blockIdx.x];                      // (launch order: tiles with the longest landmark chains first - their serial solves would be the tail of the launch)
  const int npts = T.pt_end - T.pt_begin, nslot = T.slot_end - T.slot_begin;
  const int nb = T.eb_end - T.eb_begin, nt = T.et_end - T.et_begin, ninc = nb + 2 * nt;
  constexpr int PL = VDO_TILE_PLANE;       // u and pts: three planes x | y | z (round 5; [l][3] until then: bank conflicts on every read and LDS atomic, as in the sweep of round 4)
  double* u = smem;                        // [3][PL]
  double* vs = u + 3 * PL;                 // [6*S]
  double* qs = vs + 6 * d.max_slots;       // [6*S]
  double* slotW = qs + 6 * d.max_slots;    // [12*S] inverse poses of the slots (R^T | -R^T t)
  double* pts = slotW + 12 * d.max_slots;  // [3][PL]  the tile's points (linearisation point)
  int* sdst = reinterpret_cast<int*>(pts + 3 * PL);      // [S] rows of the slots' partials (requested at the head, not where they are stored to)
  const int my_slot = min(tid, max(nslot - 1, 0));         // (tile_pose / slot_dst carry one entry of padding)
  int my_pose = d.tile_pose[T.slot_begin + my_slot];
  int my_dst = 0;
  if (MODE != 2) my_dst = d.slot_dst[T.slot_begin + my_slot];
  // the landmark chain this thread solves between the two passes (a tile has <= 256 points, hence <= 256 chains: one per thread) - its range now,
  // its scalar factor (and right-hand side) behind the staging: none of them waits for a round trip in the middle of the kernel
  const int my_chain = min(T.chain_begin + tid, T.chain_end - 1);
  const int64_t cp0 = d.chain_off[my_chain], cp1 = d.chain_off[my_chain + 1];
  double pvl[3];
  {
    const double* __restrict__ point = d.point[0] + 3 * (int64_t)T.pt_begin;
#pragma unroll
    for (int k = 0; k < 3; ++k) pvl[k] = point[min(tid + k * VDO_TILE_THREADS, 3 * npts - 1)];
  }
  // EdgeSE3PointXYZ incidences: <= Tile::ept consecutive ones (pose-sorted order) of ONE pose slot per thread = the rows of its column of the
  // tile's edge block, one contiguous row per load (the key of such an incidence is the eb_key of its edge); c is formed behind the staging
  // barrier; the slot's inverse pose and its part of v are read once, the thread's B w add up in registers and go through ONE segmented
  // scan.  The incidences of the ternary edges (dynamic tiles only) follow in strided loops, one scan per round.
  const int ebase = (T.ept ? T.eb_begin : 0) + tid, jmax = max(T.ept - 1, 0);
  int keyb[VDO_TILE_EPT];
  double web[VDO_TILE_EPT];
#pragma unroll
  for (int q = 0; q < VDO_TILE_EPT; ++q) { const int e = ebase + min(q, jmax) * VDO_TILE_THREADS; keyb[q] = __builtin_nontemporal_load(d.eb_key + e); web[q] = __builtin_nontemporal_load(d.Finc + e);      /* streamed once per launch: past the resident lines of L2, not through them (ba_sweep.hip VDO_NT_LOAD) */ }      // (eb_key has >= 1 entry)
  for (int i = tid; i < 3 * PL; i += VDO_TILE_THREADS) u[i] = 0.0;
  if (MODE != 2)
    for (int i = tid; i < 6 * nslot; i += VDO_TILE_THREADS) qs[i] = 0.0;
  auto stage_slot = [&](int sidx, int pid) {
    const IsoD W = iso_inv(iso_load(d.pose[0] + 12 * (int64_t)pid));
    double* o = slotW + 12 * sidx;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = W.r[i];
    o[9] = W.t.x; o[10] = W.t.y; o[11] = W.t.z;
    if (MODE == 0) {                       // the search direction of this PCG iteration: p = z + beta p_old (v = z, v2 = p_old; k_pcg_q stores it)
      const double beta = d.scal[S_BETA];
#pragma unroll
      for (int i = 0; i < 6; ++i) vs[6 * sidx + i] = v[6 * (int64_t)pid + i] + beta * v2[6 * (int64_t)pid + i];
    }
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 6; ++i) vs[6 * sidx + i] = v[6 * (int64_t)pid + i];
    }
  };
  asm volatile("" : "+v"(my_pose));         // (keeps the request where it was made: the compiler would sink it into the branch, behind a wait for every other request)
  if (tid < nslot) stage_slot(tid, my_pose);
  for (int sidx = tid + VDO_TILE_THREADS; sidx < nslot; sidx += VDO_TILE_THREADS) { stage_slot(sidx, d.tile_pose[T.slot_begin + sidx]); if (MODE != 2) sdst[sidx] = d.slot_dst[T.slot_begin + sidx]; }
  if (MODE != 2) sdst[my_slot] = my_dst;
  const double cg = d.dscal[cp0];                          // (meaningful for a chain of one point)
  D3 cbl{0.0, 0.0, 0.0};
  if (MODE != 0) cbl = D3{d.bl[3 * cp0], d.bl[3 * cp0 + 1], d.bl[3 * cp0 + 2]};
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int i = tid + k * VDO_TILE_THREADS, q3 = (i * 0xAAAB) >> 17; if (i < 3 * npts) pts[(i - 3 * q3) * PL + q3] = pvl[k]; }      // (q3 = i / 3 for i < 768)
  __syncthreads();
  int ecnt = 0;
#pragma unroll
  for (int q = 0; q < VDO_TILE_EPT; ++q) ecnt += (q < T.ept && keyb[q] >= 0) ? 1 : 0;
  const int slotb = ecnt ? (keyb[0] >> 16) : -1;
  // The slot's inverse pose and c = W p + t_W of the thread's incidences are formed where they are used - in pass A and again in pass C -
  // instead of living across the chain solves between the two: those hold the kernel's register peak (four 3x3 factors in flight), and
  // 24 + 6 per incidence registers on top of it cost a resident wave per SIMD (150 -> 178 registers at six incidences per thread).
  auto load_w = [&](double (&Wb)[12]) {
    const double* Ws = slotW + 12 * (slotb >= 0 ? slotb : 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) Wb[i] = Ws[i];
  };
  auto make_c = [&](const double (&Wb)[12], int q) {
    const int lp = q < ecnt ? (keyb[q] & 0xffff) : 0;
    return cam_point(Wb, D3{pts[lp], pts[PL + lp], pts[2 * PL + lp]});
  };
  // one incidence of a ternary edge (li >= nb): kind 1 = (H, p1), kind 2 = (H, p2)
  auto tern_load = [&](int li, int& key, int& kind, FInc& f) {
    key = d.inc_key[T.inc_begin + li];
    int64_t fidx;
    inc_locate(T, li, d.Eb, kind, fidx);
    f = make_f<true>(d, T, li, kind, key, d.Finc[fidx], slotW, pts);
  };
  if (MODE != 1) {   // pass A: u_l += B^T v_slot = sgn*we * (I or R) (vt - s c x vr)
    double pv[6], Wb[12];
    load_w(Wb);
#pragma unroll
    for (int i = 0; i < 6; ++i) pv[i] = vs[6 * (slotb >= 0 ? slotb : 0) + i];
#pragma unroll
    for (int q = 0; q < VDO_TILE_EPT; ++q) {
      if (q < ecnt) {
        const D3 c = make_c(Wb, q);
        const D3 t{pv[0] - 2.0 * (c.y * pv[5] - c.z * pv[4]), pv[1] - 2.0 * (c.z * pv[3] - c.x * pv[5]), pv[2] - 2.0 * (c.x * pv[4] - c.y * pv[3])};
        const D3 o = (-web[q]) * rotT(Wb, t);              // R t  (W starts with R^T)
        double* ul = u + (keyb[q] & 0xffff);
        atomicAdd(ul, o.x); atomicAdd(ul + PL, o.y); atomicAdd(ul + 2 * PL, o.z);
      }
    }
    for (int li = nb + tid; li < ninc; li += VDO_TILE_THREADS) {
      int key, kind; FInc f;
      tern_load(li, key, kind, f);
      const int sl = key >> 16;
      const double* pw = vs + 6 * sl;
      const D3 t{pw[0] - (f.cy * pw[5] - f.cz * pw[4]), pw[1] - (f.cz * pw[3] - f.cx * pw[5]), pw[2] - (f.cx * pw[4] - f.cy * pw[3])};
      D3 o;
      if (kind == 1) o = f.we * t;
      else o = (-f.we) * rotT(slotW + 12 * sl, t);
      double* ul = u + (key & 0xffff);
      atomicAdd(ul, o.x); atomicAdd(ul + PL, o.y); atomicAdd(ul + 2 * PL, o.z);
    }
    __syncthreads();
  }
  // chain solves: w = Hll^-1 y,  y = u (MODE 0) | bl (MODE 1) | bl - u (MODE 2); w overwrites u
  for (int c = T.chain_begin + tid; c < T.chain_end; c += VDO_TILE_THREADS) {      // (one trip: <= 256 chains per tile)
    const bool pre = c == my_chain;
    const int64_t p0 = pre ? cp0 : d.chain_off[c], p1 = pre ? cp1 : d.chain_off[c + 1];
    if (p1 - p0 == 1) {                                    // w = y / (Hll + lambda)
      double* ul = u + (p0 - T.pt_begin);
      D3 y{ul[0], ul[PL], ul[2 * PL]};
      const D3 blv = (MODE == 0 || pre) ? cbl : D3{d.bl[3 * p0], d.bl[3 * p0 + 1], d.bl[3 * p0 + 2]};
      if (MODE == 1) y = blv;
      if (MODE == 2) y = blv - y;
      const double g = pre ? cg : d.dscal[p0];
      ul[0] = g * y.x; ul[PL] = g * y.y; ul[2 * PL] = g * y.z;
      continue;
    }
    D3 yprev{0, 0, 0};
    double Dn[9], Gn[9];                                   // factors of the next step, requested before this step's arithmetic
#pragma unroll
    for (int i = 0; i < 9; ++i) { Dn[i] = d.Dinv[9 * p0 + i]; Gn[i] = 0.0; }
    for (int64_t l = p0; l < p1; ++l) {
      double Dc[9], Gc[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) { Dc[i] = Dn[i]; Gc[i] = Gn[i]; }
      if (l + 1 < p1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) { Dn[i] = d.Dinv[9 * (l + 1) + i]; Gn[i] = d.Gl[9 * (l + 1) + i]; }
      }
      double* ul = u + (l - T.pt_begin);
      D3 y{ul[0], ul[PL], ul[2 * PL]};
      if (MODE == 1) y = D3{d.bl[3 * l], d.bl[3 * l + 1], d.bl[3 * l + 2]};
      if (MODE == 2) y = D3{d.bl[3 * l], d.bl[3 * l + 1], d.bl[3 * l + 2]} - y;
      if (l > p0) y = y - rotT(Gc, yprev);                 // y_k = u_k - G_k^T y_{k-1}
      yprev = y;
      const D3 z = rot(Dc, y);
      ul[0] = z.x; ul[PL] = z.y; ul[2 * PL] = z.z;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) Gn[i] = d.Gl[9 * (p1 - 1) + i];
    for (int64_t l = p1 - 2; l >= p0; --l) {                // w_k = z_k - G_{k+1} w_{k+1}
      double Gc[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Gc[i] = Gn[i];
      if (l - 1 >= p0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) Gn[i] = d.Gl[9 * l + i];
      }
      const double* un = u + (l + 1 - T.pt_begin);
      const D3 wnext{un[0], un[PL], un[2 * PL]};
      double* ul = u + (l - T.pt_begin);
      const D3 z = D3{ul[0], ul[PL], ul[2 * PL]} - rot(Gc, wnext);
      ul[0] = z.x; ul[PL] = z.y; ul[2 * PL] = z.z;
    }
  }
  __syncthreads();
  if (MODE == 2) {
    double* xo = d.xl + 3 * (int64_t)T.pt_begin;
    for (int i = tid; i < 3 * npts; i += VDO_TILE_THREADS) { const int q3 = (i * 0xAAAB) >> 17; xo[i] = u[(i - 3 * q3) * PL + q3]; }
    return;
  }
  // pass C: q_slot += B w_l  (segmented wave reduction; incidences are slot-sorted per part)
  {
    double q[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, Wb[12];
    load_w(Wb);
#pragma unroll
    for (int k = 0; k < VDO_TILE_EPT; ++k) {
      if (k < ecnt) {
        const double* wl = u + (keyb[k] & 0xffff);
        const D3 y = rot(Wb, D3{wl[0], wl[PL], wl[2 * PL]});       // R^T w
        const D3 c = make_c(Wb, k);
        const double sg = -web[k];
        q[0] += sg * y.x; q[1] += sg * y.y; q[2] += sg * y.z;
        q[3] += sg * 2.0 * (c.y * y.z - c.z * y.y);
        q[4] += sg * 2.0 * (c.z * y.x - c.x * y.z);
        q[5] += sg * 2.0 * (c.x * y.y - c.y * y.x);
      }
    }
    if (nb > 0) { const SegCtl16 sc_ = seg_ctl16(slotb); seg_apply16<6>(q, sc_, seg_flags(sc_), qs + 6 * (slotb >= 0 ? slotb : 0)); }
  }
  for (int base = nb; base < ninc; base += VDO_TILE_THREADS) {      // (uniform trip count: every lane takes part in the scans)
    const int li = base + tid;
    double q[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int skey = -1;
    if (li < ninc) {
      int key, kind; FInc f;
      tern_load(li, key, kind, f);
      const int sl = key >> 16;
      const double* wl = u + (key & 0xffff);
      D3 y{wl[0], wl[PL], wl[2 * PL]};
      if (kind != 1) y = rot(slotW + 12 * sl, y);            // R^T w
      const double sg = kind == 1 ? f.we : -f.we;
      q[0] = sg * y.x; q[1] = sg * y.y; q[2] = sg * y.z;
      q[3] = sg * (f.cy * y.z - f.cz * y.y);
      q[4] = sg * (f.cz * y.x - f.cx * y.z);
      q[5] = sg * (f.cx * y.y - f.cy * y.x);
      skey = sl;
    }
    { const SegCtl16 sc_ = seg_ctl16(skey); seg_apply16<6>(q, sc_, seg_flags(sc_), qs + 6 * (skey >= 0 ? skey : 0)); }
  }
  __syncthreads();
  // one row of 8 doubles (6 used) per (tile, slot), pose-major (k_pcg_q / k_gather_q stream a pose's rows)
  for (int i = tid; i < 8 * nslot; i += VDO_TILE_THREADS) {
    const int sidx = i >> 3, k = i & 7;
    if (k < 6) d.part_q[8 * (int64_t)sdst[sidx] + k] = qs[6 * sidx + k];
  }
}

// qs[pose] = sum over the pose's (tile,slot) partials, fixed order
__global__ __launch_bounds__(256) void k_gather_q(BADev d, double* out, int pcg) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per pose
  if (p >= d.P || (pcg && d.flags[1])) return;
  const double s = wave_gather_rows<8>(d.part_q, d.ps_off[p], d.ps_off[p + 1]);
  const int lane = threadIdx.x & 63;
  if (lane < 6) out[6 * (int64_t)p + lane] = s;
}

// (Hpp v)_p including lambda and the EdgeSE3 off-diagonal blocks (CSR per pose, no atomics)
__device__ __forceinline__ void hpp_mv(const BADev& d, int p, const double* v, double lambda, double* out) {
  const double* Hm = d.Hpp + 36 * (int64_t)p;
  const double* pv = v + 6 * (int64_t)p;
  for (int i = 0; i < 6; ++i) {
    double s = lambda * pv[i];
    for (int j = 0; j < 6; ++j) s += Hm[i * 6 + j] * pv[j];
    out[i] = s;
  }
  for (int k = d.pe_off[p]; k < d.pe_off[p + 1]; ++k) {
    const int ent = d.pe_idx[k];
    const int e = ent >> 1, side = ent & 1;
    const double* He = d.Hpp_ep + 36 * (int64_t)e;       // block (i,j)
    if (side == 0) {
      const double* vj = v + 6 * (int64_t)d.ep_j[e];
      for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) out[a] += He[a * 6 + b] * vj[b];
    } else {
      const double* vi = v + 6 * (int64_t)d.ep_i[e];
      for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) out[a] += He[b * 6 + a] * vi[b];
    }
  }
}

// ---- The vector phases of a PCG iteration, spread over the chip.
// Round 2 ran them in ONE workgroup (k_pcg_vec: (Hpp + lambda) p - qs, two dot products, three axpys and the chain preconditioner between
// barriers): 127 us per iteration on the 2 190-pose graph, most of it a single CU chasing pe_off -> pe_idx -> ep_j -> p through HBM with
// 1 024 threads.  Now an iteration is three launches, none of them serial in the number of poses:
//   k_schur_tile<0>   part_q = B Hll^-1 B^T p         one workgroup per tile   (p = z + beta p_old formed while it is staged)
//   k_pcg_q           q = (Hpp + lambda) p - sum part_q, p stored, p.q per workgroup     one WAVE per pose
//   k_pcg_chain<0>    alpha = rz / p.q ; x += alpha p ; r -= alpha q ; z = M^-1 r ; r.z per chain     one workgroup (one wave) per pose chain;
//                     the workgroup that arrives last adds the chains' r.z in chain order: beta, convergence flags.
// p lives in two buffers (pp, pp2) that alternate: k_pcg_q reads its neighbours' p_old while other workgroups store p.  All sums have a
// fixed order (per-workgroup partials added by index): run-independent bits, and identical bits on every rank of a sharded solve.

// q = (Hpp + lambda I) p + off-diagonal EdgeSE3 blocks - qs, with p = z + beta p_old ; partial p.q of the workgroup's 4 poses
__global__ __launch_bounds__(256) void k_pcg_q(BADev d, double lambda, int par, int gather) {
  __shared__ double lds[4];
  if (d.flags[1]) return;                       // converged earlier: no-op
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + wv;
  const double* __restrict__ pold = par ? d.pp2 : d.pp;
  double* __restrict__ pnew = par ? d.pp : d.pp2;
  const double beta = d.scal[S_BETA];
  double dot = 0.0;
  if (p < d.P) {
    const int l = lane < 6 ? lane : 0;
    // the Schur part of q: this pose's partial rows (sharded: gathered and all-reduced into qs by the launches before)
    const double ql = gather ? wave_gather_rows<8>(d.part_q, d.ps_off[p], d.ps_off[p + 1]) : d.qs[6 * (int64_t)p + l];
    const double pn = d.zp[6 * (int64_t)p + l] + beta * pold[6 * (int64_t)p + l];
    if (lane < 6) pnew[6 * (int64_t)p + lane] = pn;
    const double* Hm = d.Hpp + 36 * (int64_t)p + 6 * l;
    double q = lambda * pn;
#pragma unroll
    for (int j = 0; j < 6; ++j) q += Hm[j] * __shfl(pn, j, 64);
    for (int k = d.pe_off[p]; k < d.pe_off[p + 1]; ++k) {
      const int ent = d.pe_idx[k];
      const int e = ent >> 1, side = ent & 1;
      const double* He = d.Hpp_ep + 36 * (int64_t)e;       // block (i,j)
      const int64_t o = side == 0 ? d.ep_j[e] : d.ep_i[e];
      const double on = d.zp[6 * o + l] + beta * pold[6 * o + l];
#pragma unroll
      for (int b = 0; b < 6; ++b) q += (side == 0 ? He[l * 6 + b] : He[b * 6 + l]) * __shfl(on, b, 64);
    }
    q -= ql;
    if (lane < 6) { d.qp[6 * (int64_t)p + lane] = q; dot = pn * q; }
  }
  dot = wave_sum(dot);
  if (lane == 0) lds[wv] = dot;
  __syncthreads();
  if (threadIdx.x == 0) d.part_pq[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// INIT:  bs = bp - qs ; x = 0 ; r = bs ; z = M^-1 r ; p = z ; rz = rz0 = r.z ; beta = 0
// else:  one CG update given q (k_pcg_q): alpha = rz / p.q ; x += alpha p ; r -= alpha q ; z = M^-1 r ; rz_new ; beta = rz_new / rz
// One workgroup per pose chain (64 * pc_nwave threads: the chain's segments, see pchain_solve_partitioned); the chain's r is formed straight
// into its LDS strip.  LDS: strip [6 * pc_maxlen] | boundary vectors [96] | reduction scratch [24]
template <int INIT>
__global__ __launch_bounds__(1024) void k_pcg_chain(BADev d, double tol2, int par, int npart) {
  extern __shared__ __attribute__((aligned(16))) double strip[];
  if (!INIT && d.flags[1]) return;
  const int c = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
  const bool in_lds = d.pc_lds != 0;
  double* bnd = strip + (in_lds ? 6 * (size_t)d.pc_maxlen : 0);
  double* red = bnd + 96;
  double* bmat = red + 24;
  const int bgn = d.pc_off[c], len = d.pc_off[c + 1] - bgn;
  // the matrices the serial parts of the solve need - boundary products of the partitioned substitutions, the far link of a twisted chain - are requested
  // NOW (they depend on the factorisation only) and parked in LDS below, behind alpha and the vector updates
  double bm_val[2] = {0.0, 0.0};
  int bm_dst[2] = {-1, -1};
  if (in_lds) {
    const int G = pc_seg_len(len, nth >> 6), S = (len + G - 1) / G, nbm = S > 2 ? S - 2 : 0;
    const int total = 72 * nbm + (d.pc_far_pos[c] >= 0 ? 36 : 0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                            // (<= 1 044 items for <= 1 024 threads; the loads are unconditional - a load under a branch is waited for at its end)
      const int it = tid + h * nth;
      const double* src = d.Pf + 36 * (int64_t)bgn;
      int dst = -1;
      if (it < 36 * nbm) { const int s = it / 36 + 1, el = it - 36 * (s - 1); dst = PC_BMAT_P + it; src = d.Pf + 36 * ((int64_t)bgn + (s + 1) * G - 1) + el; }
      else if (it < 72 * nbm) { const int t = it - 36 * nbm, s = t / 36 + 1, el = t - 36 * (s - 1); dst = PC_BMAT_Q + t; src = d.Qb + 36 * ((int64_t)bgn + s * G) + el; }
      else if (it < total) { const int t = it - 72 * nbm; dst = PC_BMAT_F + t; src = d.Lfar + 36 * (int64_t)c + t; }
      bm_dst[h] = dst;
      bm_val[h] = *src;
    }
  }
  PcPre pre;
  if (in_lds) pc_prefetch(d, bgn, len, pre);
  double alpha = 0.0, pq = 0.0, rz = 0.0;
  PCG_TICK(0);
  if (!INIT) {
    double a = 0.0;
    for (int i = tid; i < npart; i += nth) a += d.part_pq[i];
    pq = block_sum1(a, red);
    rz = d.scal[S_RZ];
    alpha = rz / pq;
  }
  PCG_TICK(1);
  const double* __restrict__ pn = par ? d.pp : d.pp2;      // p of this iteration (k_pcg_q stored it)
  for (int i = tid; i < 6 * len; i += nth) {
    const int k = i / 6;
    const int64_t g = 6 * (int64_t)d.pc_pose[bgn + k] + (i - 6 * k);
    double r;
    if (INIT) { r = d.bp[g] - d.qs[g]; d.bs[g] = r; d.xp[g] = 0.0; }
    else { d.xp[g] += alpha * pn[g]; r = d.rp[g] - alpha * d.qp[g]; }
    d.rp[g] = r;
    if (in_lds) strip[i] = r;
  }
  if (bm_dst[0] >= 0) bmat[bm_dst[0]] = bm_val[0];
  if (bm_dst[1] >= 0) bmat[bm_dst[1]] = bm_val[1];
  PCG_TICK(2);
  if (in_lds) pchain_solve_partitioned(d, c, bgn, len, strip, bnd, bmat, pre);
  else {                                             // (a chain too long for the LDS: one wave, vectors in global memory)
    __threadfence();
    __syncthreads();
    if (tid < 64) pchain_apply(d, c, d.rp, d.zp);
    __threadfence();
    __syncthreads();
  }
  PCG_TICK(10);
  double acc = 0.0;
  for (int i = tid; i < 6 * len; i += nth) {
    const int k = i / 6;
    const int64_t g = 6 * (int64_t)d.pc_pose[bgn + k] + (i - 6 * k);
    double z;
    if (in_lds) { z = strip[i]; d.zp[g] = z; } else z = d.zp[g];
    if (INIT) d.pp[g] = z;
    acc += d.rp[g] * z;
  }
  acc = block_sum1(acc, red);
  PCG_TICK(11);
#ifdef PCG_PROF
  if (!INIT && blockIdx.x == 0 && threadIdx.x == 0) {
    const long long* tk = pcg_tk();
    for (int i = 0; i < 11; ++i) atomicAdd(&g_pcg_prof[i], (unsigned long long)(tk[i + 1] - tk[i]));
    atomicAdd(&g_pcg_prof[19], 1ull);
  }
#endif
  // ---- the chain that arrives last closes the iteration (its sum runs over the chains in index order, whoever it is)
  __shared__ int s_last;
  if (tid == 0) {
    d.part_rz[c] = acc;
    __threadfence();
    s_last = atomicAdd(d.flags + 3, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  double a = 0.0;
  for (int i = tid; i < (int)gridDim.x; i += nth) a += __builtin_nontemporal_load(d.part_rz + i);
  const double rznew = block_sum1(a, red);
  if (tid == 0) {
    d.flags[3] = 0;
    if (INIT) {
      d.scal[S_RZ] = rznew; d.scal[S_RZ0] = rznew; d.scal[S_RZNEW] = rznew; d.scal[S_BETA] = 0.0;
      d.flags[1] = (rznew > 0) ? 0 : 1; d.flags[2] = 0;
    } else {
      d.scal[S_BETA] = rznew / rz;
      d.scal[S_RZ] = rznew; d.scal[S_RZNEW] = rznew; d.scal[S_PQ] = pq;
      int f = 0;
      if (!(pq > 0) || !(rznew == rznew)) f = 2;          // breakdown
      else if (rznew <= tol2 * d.scal[S_RZ0]) f = 1;      // converged
      d.flags[1] = f;
      d.flags[2] += 1;
    }
  }
}

// trial estimate = current (+) x ; scale = sum x (lambda x + b)   (computeScale, levenberg.cpp:182-189)
// Grid-wide (the points are the bulk): every block leaves one partial, k_reduce_part sums them in block order.
__global__ __launch_bounds__(1024) void k_update(BADev d, double lambda, int ortho) {
  __shared__ double lds[17];
  double acc = 0;
  const bool own_poses = !d.sharded || d.shard_rank == 0;     // replicated vertices count once in computeScale
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = gtid; p < d.P; p += gsz) {
    const double* x = d.xp + 6 * p;
    if (own_poses)
      for (int i = 0; i < 6; ++i) acc += x[i] * (lambda * x[i] + d.bp[6 * p + i]);
    const IsoD X = iso_load(d.pose[0] + 12 * p);
    iso_store(d.pose[1] + 12 * p, iso_oplus(X, x, ortho != 0));
  }
  for (int64_t i = gtid; i < 3 * (int64_t)d.L; i += gsz) {
    const double x = d.xl[i];
    acc += x * (lambda * x + d.bl[i]);
    d.point[1][i] = d.point[0][i] + x;
  }
  acc = block_sum1(acc, lds);
  if (threadIdx.x == 0) d.part_red[blockIdx.x] = acc;
}

// fixed-order sum (op 0) / max (op 1) of the per-block partials of k_update / k_max_diag
__global__ __launch_bounds__(256) void k_reduce_part(BADev d, int n, int op) {
  __shared__ double lds[24];
  double a = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a = op ? fmax(a, d.part_red[i]) : a + d.part_red[i];
  a = op ? block_max1(a, lds) : block_sum1(a, lds);
  if (threadIdx.x == 0) {
    if (op) d.scal[S_MAXDIAG] = a;
    else if (d.sharded) d.red_chi[2] = a;
    else d.scal[S_SCALE] = a;
  }
}

// Finc -> explicit 6x3 blocks Binc[18][Ninc] (download / debugging only; never on the solve path)
__global__ __launch_bounds__(VDO_TILE_THREADS) void k_expand_binc(BADev d) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Tile T = d.tiles[blockIdx.x];
  const int nb = T.eb_end - T.eb_begin, nt = T.et_end - T.et_begin, ninc = nb + 2 * nt;
  double* slotW = smem;                     // [12*S]
  double* pts = slotW + 12 * d.max_slots;   // [3*TP]
  stage_slot_w_pts(d, T, slotW, pts);
  __syncthreads();
  const int64_t N = d.Ninc;
  for (int li = threadIdx.x; li < ninc; li += VDO_TILE_THREADS) {
    int kind; int64_t fidx;
    inc_locate(T, li, d.Eb, kind, fidx);
    const int key = d.inc_key[T.inc_begin + li];
    if (key < 0) continue;                  // (an entry of the tile's edge block without an edge)
    const int sl = key >> 16;
    double B[18];
    expand_block(kind, make_f(d, T, li, kind, key, d.Finc[fidx], slotW, pts), slotW + 12 * sl, B);
#pragma unroll
    for (int i = 0; i < 18; ++i) d.Binc[i * N + T.inc_begin + li] = B[i];
  }
}

// -DDENSE_PROF (tools/build_variant.sh, debug builds only): shader-clock cycles per phase of k_schur_dense_tile, summed over its workgroups
#ifdef DENSE_PROF
__device__ unsigned long long g_asm_prof[16];
#define AP_TICK(slot) do { if (threadIdx.x == 0) { const long long t_ = clock64(); ap_t[slot] += t_ - ap_prev; ap_prev = t_; } } while (0)
#else
#define AP_TICK(slot) do { } while (0)
#endif
#ifndef VDO_BA_DENSE_CHUNK_DEFAULT
#define VDO_BA_DENSE_CHUNK_DEFAULT 4
#endif
// ---- explicit reduced-camera matrix for the dense solver (ba_dense.hip) ---------------------------------------------
// S (row-major, leading dimension ld >= 6P, padded rows/columns = identity) = blockdiag(Hpp + lambda I) + EdgeSE3 off-diagonal
// blocks - sum over tiles of B Hll^-1 B^T.
__global__ __launch_bounds__(256) void k_dense_init(BADev d, double* __restrict__ S, int64_t ld, double lambda) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t N = 6 * (int64_t)d.P;
  if (i < 36 * (int64_t)d.P) {
    const int64_t p = i / 36; const int a = (int)(i % 36) / 6, b = (int)(i % 6);
    atomicAdd(S + (6 * p + a) * ld + 6 * p + b, d.Hpp[i] + (a == b ? lambda : 0.0));
  } else if (i < 36 * (int64_t)(d.P + d.Ep)) {
    const int64_t e = (i - 36 * (int64_t)d.P) / 36; const int a = (int)(i % 36) / 6, b = (int)(i % 6);
    const double v = d.Hpp_ep[36 * e + 6 * a + b];                     // block (ep_i, ep_j)
    const int64_t pi = d.ep_i[e], pj = d.ep_j[e];
    atomicAdd(S + (6 * pi + a) * ld + 6 * pj + b, v);
    atomicAdd(S + (6 * pj + b) * ld + 6 * pi + a, v);
  } else {
    const int64_t k = i - 36 * (int64_t)(d.P + d.Ep);
    if (N + k < ld) S[(N + k) * ld + N + k] = 1.0;
  }
}

// One workgroup per tile; for every pose slot s of the tile the six unit vectors e_(s,b) go through B^T, the landmark chain
// solves and B at once (6 right-hand sides); the resulting 6x6 blocks against every slot r that shares a point (or a
// dynamic track) with s are subtracted from S with fp64 atomics.  Only the points reached from slot s are touched.
// Incidences a thread of the dense assembly keeps: the tile's PADDED count - its thread-transposed EdgeSE3PointXYZ block of 256 * ept <= 1536 entries,
// then two incidences per ternary edge (<= 255 of those in a tile of 256 points) - is at most 256 * (VDO_TILE_EPT + 2).  (Rounds 3-4 kept VDO_TILE_EPT
// and silently dropped what lay beyond 1536: the ternary incidences of a tile whose block is full - ADVICE r4.)
#define VDO_DENSE_EPT (VDO_TILE_EPT + 2)

__global__ __launch_bounds__(VDO_TILE_THREADS) void k_schur_dense_tile(BADev d, double* __restrict__ S, int64_t ld, int chunk) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Tile T = d.tiles[blockIdx.x];
  const int npts = T.pt_end - T.pt_begin, nslot = T.slot_end - T.slot_begin;
  const int s_begin = (int)blockIdx.y * chunk, s_end = min(nslot, s_begin + chunk);      // (the slots of a tile go in runs of `chunk` to the workgroups (tile, y))
  if (s_begin >= nslot) return;
#ifdef DENSE_PROF
  long long ap_t[10], ap_prev = clock64();
  for (int i = 0; i < 10; ++i) ap_t[i] = 0;
#endif
  const int nb = T.eb_end - T.eb_begin, nt = T.et_end - T.et_begin, ninc = nb + 2 * nt;
  double* u6 = smem;                          // [6][3*TP]
  double* dinv = u6 + 18 * VDO_TILE_PTS;      // [9*TP]
  double* gl = dinv + 9 * VDO_TILE_PTS;       // [9*TP]
  double* q36 = gl + 9 * VDO_TILE_PTS;        // [36*S]  block (r, s): [b][a]
  double* slotW = q36 + 36 * d.max_slots;     // [12*S]
  double* pts = slotW + 12 * d.max_slots;     // [3*TP]
  int* touched = (int*)(pts + 3 * VDO_TILE_PTS);    // [TP]
  int* choff = touched + VDO_TILE_PTS;              // [TP+1] the tile's chain offsets (read in every pass over the slots: not from L2 each time)
  int* spose = choff + VDO_TILE_PTS + 1;            // [S] pose id of every slot
  int* psingle = spose + d.max_slots;               // [TP] 1: the point is a chain of its own (Hll^-1 = dscal * I3, applied in pass A)
  int* mch = psingle + VDO_TILE_PTS;                // [TP/2 + 1] the tile's chains of two or more points (the only ones the chain solves visit); last entry: their number
  const int tid = threadIdx.x;
  // ---- head: what the tile needs from HBM / L2 in TWO rounds of requests, every request of a round made before the first use of any of them
  // (the round-3 form walked the 9 * npts factor entries in a loop whose every pass waited for a flag and then for a value: ~18 round trips of
  // ~1.5 us per workgroup, and with the slots dealt to many workgroups per tile every one of them pays for the head).
  // round 1: pose id of slot `tid`, point `tid`'s single-point flag, the points, this thread's incidences (key + information scalar)
  const int my_slot = min(tid, nslot - 1);
  const int my_pose = d.tile_pose[T.slot_begin + my_slot];
  const int my_pt = min(tid, max(npts - 1, 0));
  const int64_t my_l = T.pt_begin + my_pt;
  const int my_single = d.pt_single[my_l];
  double pv[3];
  {
    const double* __restrict__ point = d.point[0] + 3 * (int64_t)T.pt_begin;
#pragma unroll
    for (int k = 0; k < 3; ++k) pv[k] = point[min(tid + k * VDO_TILE_THREADS, max(3 * npts - 1, 0))];
  }
  int key[VDO_DENSE_EPT], kind[VDO_DENSE_EPT];       // (<= 256 * VDO_DENSE_EPT padded incidences per tile)
  FInc F[VDO_DENSE_EPT];
  double we[VDO_DENSE_EPT];
#pragma unroll
  for (int j = 0; j < VDO_DENSE_EPT; ++j) { key[j] = -1; kind[j] = 1; we[j] = 0.0; }
  if (ninc > 0) {                                  // (uniform)
#pragma unroll
    for (int j = 0; j < VDO_DENSE_EPT; ++j) {
      const int li = tid + VDO_TILE_THREADS * j, lic = min(li, ninc - 1);
      int64_t fidx;
      inc_locate(T, lic, d.Eb, kind[j], fidx);
      const int kv = d.inc_key[T.inc_begin + lic];
      const double wv = d.Finc[fidx];
      key[j] = li < ninc ? kv : -1;
      we[j] = li < ninc ? wv : 0.0;
    }
  }
  const int nch = T.chain_end - T.chain_begin;
  for (int i = tid; i <= nch; i += VDO_TILE_THREADS) choff[i] = d.chain_off[T.chain_begin + i];
  // round 2: the slot's pose, the point's factor
  const IsoD my_iso = iso_load(d.pose[0] + 12 * (int64_t)my_pose);
  double fd[9], fg[9];
  {
    const double ds = d.dscal[my_l];
    const double* gd = d.Dinv + 9 * my_l;
    const double* gg = d.Gl + 9 * my_l;
    if (my_single) {                               // (a point on its own: dscal * I3, ba_dev.hpp - its rows of Dinv / Gl are never written)
#pragma unroll
      for (int q = 0; q < 9; ++q) { fd[q] = q % 4 == 0 ? ds : 0.0; fg[q] = 0.0; }
    } else {
#pragma unroll
      for (int q = 0; q < 9; ++q) { fd[q] = gd[q]; fg[q] = gg[q]; }
    }
  }
  {
    auto stage_slot = [&](int sidx, int pid, const IsoD& iso) {
      const IsoD W = iso_inv(iso);
      double* o = slotW + 12 * sidx;
#pragma unroll
      for (int i = 0; i < 9; ++i) o[i] = W.r[i];
      o[9] = W.t.x; o[10] = W.t.y; o[11] = W.t.z;
      spose[sidx] = pid;
    };
    if (tid < nslot) stage_slot(tid, my_pose, my_iso);
    for (int sidx = tid + VDO_TILE_THREADS; sidx < nslot; sidx += VDO_TILE_THREADS) {       // (more than 256 slots in a tile: not in any graph of the bench)
      const int pid = d.tile_pose[T.slot_begin + sidx];
      stage_slot(sidx, pid, iso_load(d.pose[0] + 12 * (int64_t)pid));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { const int i = tid + k * VDO_TILE_THREADS; if (i < 3 * npts) pts[i] = pv[k]; }
    if (tid < npts) {
#pragma unroll
      for (int q = 0; q < 9; ++q) { dinv[9 * tid + q] = fd[q]; gl[9 * tid + q] = fg[q]; }
      psingle[tid] = my_single;
    }
    if (tid == 0) mch[VDO_TILE_PTS / 2] = 0;
  }
  __syncthreads();
  for (int c = tid; c < nch; c += VDO_TILE_THREADS)
    if (choff[c + 1] - choff[c] >= 2) mch[atomicAdd(&mch[VDO_TILE_PTS / 2], 1)] = c;       // (any order: the chains are independent of each other)
  AP_TICK(0);
#pragma unroll
  for (int j = 0; j < VDO_DENSE_EPT; ++j) F[j] = key[j] >= 0 ? make_f(d, T, tid + VDO_TILE_THREADS * j, kind[j], key[j], we[j], slotW, pts) : FInc{0, 0, 0, 0};
  AP_TICK(1);
  for (int s = s_begin; s < s_end; ++s) {
    __syncthreads();
    for (int i = tid; i < 18 * VDO_TILE_PTS; i += VDO_TILE_THREADS) u6[i] = 0.0;
    for (int i = 36 * s + tid; i < 36 * nslot; i += VDO_TILE_THREADS) q36[i] = 0.0;      // (blocks (r, s), r >= s)
    for (int i = tid; i < npts; i += VDO_TILE_THREADS) touched[i] = 0;
    __syncthreads();
    AP_TICK(2);
    // pass A: u_b[l] += row b of the explicit 6x3 block of every incidence (s, l)
#pragma unroll
    for (int j = 0; j < VDO_DENSE_EPT; ++j) {
      if (key[j] >= 0 && (key[j] >> 16) == s) {
        double B[18];
        expand_block(kind[j], F[j], slotW + 12 * s, B);
        const int lp = key[j] & 0xffff;
        const double sc = psingle[lp] ? dinv[9 * lp] : 1.0;      // a point on its own: w = dscal * u, applied term by term (no chain solve visits it)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          double* ul = u6 + b * 3 * VDO_TILE_PTS + 3 * lp;
          atomicAdd(ul, B[3 * b] * sc); atomicAdd(ul + 1, B[3 * b + 1] * sc); atomicAdd(ul + 2, B[3 * b + 2] * sc);
        }
        touched[lp] = 1;
      }
    }
    __syncthreads();
    AP_TICK(3);
    // chain solves w = Hll^-1 u for the multi-point chains slot s reaches, one (chain, right-hand side) per thread-iteration; every task looks
    // for itself whether a point of its chain was touched (no marking pass, no barrier for it), and the task of right-hand side 0 marks the
    // whole chain afterwards: its points all carry w (pass C, behind the barrier, sees them; the other five tasks of the chain may meet a
    // 2 where there was a 0 - they only ask whether ANY point is touched, and one was).
    const int n_mch = mch[VDO_TILE_PTS / 2];
    AP_TICK(4);
    // (Every step's operands - the next point's right-hand side and factor blocks - are requested before the current step's result is stored:
    // the recurrence y_l = u_l - G_l^T y_(l-1) then costs three dependent multiply-adds per point instead of an LDS round trip behind every store.)
    for (int task = tid; task < 6 * n_mch; task += VDO_TILE_THREADS) {
      const int c = mch[task / 6], b = task % 6;
      const int p0 = choff[c] - T.pt_begin, p1 = choff[c + 1] - T.pt_begin;       // local point range (>= 2 points)
      int any = 0;
      for (int l = p0; l < p1; ++l) any |= touched[l];
      if (!any) continue;
      double* u = u6 + b * 3 * VDO_TILE_PTS;
      auto ld3 = [](const double* p) { return D3{p[0], p[1], p[2]}; };
      struct M9 { double m[9]; };
      auto ld9 = [](const double* p) { M9 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; };
      // forward: y_l = u_l - G_l^T y_(l-1), z_l = Dinv_l y_l
      D3 uc = ld3(u + 3 * p0), yprev{0, 0, 0};
      M9 gc = ld9(gl + 9 * p0), dc = ld9(dinv + 9 * p0);
      for (int l = p0; l < p1; ++l) {
        const int ln = min(l + 1, p1 - 1);
        const D3 un = ld3(u + 3 * ln);
        const M9 gn = ld9(gl + 9 * ln), dn = ld9(dinv + 9 * ln);
        D3 y = uc;
        if (l > p0) y = y - rotT(gc.m, yprev);
        yprev = y;
        const D3 z = rot(dc.m, y);
        double* ul = u + 3 * l;
        ul[0] = z.x; ul[1] = z.y; ul[2] = z.z;
        uc = un; gc = gn; dc = dn;
      }
      // backward: w_l = z_l - G_(l+1) w_(l+1)   (w of the last point is its z)
      {
        D3 wnext = ld3(u + 3 * (p1 - 1));
        M9 gnx = ld9(gl + 9 * (p1 - 1));           // G of point l + 1
        D3 zc = ld3(u + 3 * (p1 - 2));
        for (int l = p1 - 2; l >= p0; --l) {
          const int lm = max(l - 1, p0);
          const D3 zm = ld3(u + 3 * lm);
          const M9 gm = ld9(gl + 9 * l);           // G of point l: the next pass's G of "l + 1"
          const D3 w = zc - rot(gnx.m, wnext);
          double* ul = u + 3 * l;
          ul[0] = w.x; ul[1] = w.y; ul[2] = w.z;
          wnext = w; gnx = gm; zc = zm;
        }
      }
      if (b == 0) for (int l = p0; l < p1; ++l) touched[l] = 2;
    }
    __syncthreads();
    AP_TICK(5);
    // pass C: block (r, s) += B_r w_b for every incidence (r, l) on a touched point - of the slots r >= s only: block (s, r) is the transpose
    // (threads and ternary incidences are in slot order, so about half of the waves have nothing to do; S comes out exactly symmetric).  The EdgeSE3PointXYZ entries of a thread (rows j < T.ept of
    // the tile's edge block) belong to ONE pose slot (capi_ba.hip): their 36 sums add up in registers and go through ONE segmented DPP
    // reduction per thread (threads are in slot order; many lanes share a slot: plain LDS atomics would serialise); the ternary incidences
    // (rows behind the block, slot-sorted) one reduction each.
    {
      const int rt = key[0] >= 0 && T.ept > 0 ? (key[0] >> 16) : -1;       // (a thread's entries fill its column from row 0)
      double g36[36];
#pragma unroll
      for (int i = 0; i < 36; ++i) g36[i] = 0.0;
      bool mine = false;
#pragma unroll
      for (int j = 0; j < VDO_DENSE_EPT; ++j) {
        if (j < T.ept && key[j] >= 0 && rt >= s && touched[key[j] & 0xffff]) {
          const int lp = key[j] & 0xffff;
          mine = true;
          double B[18];
          expand_block(0, F[j], slotW + 12 * rt, B);
#pragma unroll
          for (int b = 0; b < 6; ++b) {
            const double* w = u6 + b * 3 * VDO_TILE_PTS + 3 * lp;
            const double w0 = w[0], w1 = w[1], w2 = w[2];
#pragma unroll
            for (int a = 0; a < 6; ++a) g36[6 * b + a] += B[3 * a] * w0 + B[3 * a + 1] * w1 + B[3 * a + 2] * w2;
          }
        }
      }
      if (__any(mine)) {                                           // (wave-uniform)
        const SegCtl16 sc = seg_ctl16(rt);
        const SegFlags sf = seg_flags(sc);
        double* q = q36 + 36 * (rt >= 0 ? rt : 0);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          double g[6];
#pragma unroll
          for (int a = 0; a < 6; ++a) g[a] = g36[6 * b + a];
          seg_apply16<6>(g, sc, sf, q + 6 * b);                    // q layout [b][a]
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VDO_DENSE_EPT; ++j) {
      if (j < T.ept) continue;                                     // (uniform)
      const bool on = key[j] >= 0 && (key[j] >> 16) >= s && touched[key[j] & 0xffff];
      const int r = on ? (key[j] >> 16) : -1, lp = on ? (key[j] & 0xffff) : 0;
      if (!__any(on)) continue;                                  // (wave-uniform: nothing of this wave's incidences is reached from slot s)
      double B[18];
      expand_block(kind[j], F[j], slotW + 12 * (r >= 0 ? r : 0), B);
      const SegCtl16 sc = seg_ctl16(r);
      const SegFlags sf = seg_flags(sc);
      double* q = q36 + 36 * (r >= 0 ? r : 0);
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const double* w = u6 + b * 3 * VDO_TILE_PTS + 3 * lp;
        const double w0 = on ? w[0] : 0.0, w1 = on ? w[1] : 0.0, w2 = on ? w[2] : 0.0;
        double g[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) g[a] = B[3 * a] * w0 + B[3 * a + 1] * w1 + B[3 * a + 2] * w2;
        seg_apply16<6>(g, sc, sf, q + 6 * b);                    // q layout [b][a]
      }
    }
    __syncthreads();
    AP_TICK(6);
    const int64_t gs = spose[s];
    for (int i = 36 * s + tid; i < 36 * nslot; i += VDO_TILE_THREADS) {
      const double v = q36[i];
      if (v == 0.0) continue;
      const int r = i / 36, b = (i % 36) / 6, a = i % 6;         // q36[r][b][a]
      const int64_t gr = spose[r];
      atomicAdd(S + (6 * gr + a) * ld + 6 * gs + b, -v);
      if (r != s) atomicAdd(S + (6 * gs + b) * ld + 6 * gr + a, -v);
    }
      AP_TICK(7);
  }
#ifdef DENSE_PROF
  if (tid == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_asm_prof[i], (unsigned long long)ap_t[i]);
    atomicAdd(&g_asm_prof[8], (unsigned long long)(s_end - s_begin));
    atomicAdd(&g_asm_prof[9], 1ull);
  }
#endif
}

size_t dense_tile_lds(const BADev& d) { return (39 * VDO_TILE_PTS + 48 * (size_t)d.max_slots) * sizeof(double) + (3 * VDO_TILE_PTS + VDO_TILE_PTS / 2 + 4 + (size_t)d.max_slots) * sizeof(int); }

// S <- reduced-camera matrix at this lambda (launch_factor must have run: it leaves the landmark chain factors of Hll + lambda I)
void launch_dense_assemble(const BADev& d, double* S, int64_t ld, double lambda, hipStream_t s, const Reducer& R, bool init, bool clean) {
  if (!clean) hipMemsetAsync(S, 0, sizeof(double) * (size_t)ld * (size_t)ld, s);       // (clean: k_dense_small left S zeroed behind itself)
  // A tile's columns (one per pose slot: up to max_slots sequential passes with five barriers each, and a dynamic track's chain solves walk
  // its points on 6 threads per chain) are independent of each other: they go in runs of `chunk` to the workgroups (tile, 0 .. max_slots / chunk) -
  // most tiles of the bench graph carry a dozen slots, the ones with the long dynamic tracks 81: one workgroup per tile left the device waiting
  // for those (1.82 ms).  VDO_BA_DENSE_CHUNK overrides.
  // (round 6) a graph of a few tiles - a 20-frame window: ~30 tiles x 20 slots - leaves most of the device idle at 4 slots per workgroup: one slot per workgroup there
  // (k_schur_dense_tile 34 -> 13 us per Levenberg trial, profiles/r06_window_lm_trial_timeline.txt), two up to 8 k (tile, slot) pairs
  const int64_t pairs = (int64_t)d.n_tiles * d.max_slots;
  const int chunk_auto = pairs <= 2048 ? 1 : pairs <= 8192 ? 2 : VDO_BA_DENSE_CHUNK_DEFAULT;
  const int chunk = std::getenv("VDO_BA_DENSE_CHUNK") ? std::max(1, std::atoi(std::getenv("VDO_BA_DENSE_CHUNK"))) : chunk_auto;
  if (d.n_tiles) hipLaunchKernelGGL(k_schur_dense_tile, dim3(d.n_tiles, (d.max_slots + chunk - 1) / chunk), dim3(VDO_TILE_THREADS), raise_lds(k_schur_dense_tile, dense_tile_lds(d)), s, d, S, ld, chunk);
  if (d.sharded) R(S, ld * ld);                     // landmark-side contributions of every rank (all-reduce of S)
  if (!init) return;                                // (k_dense_small adds the pose side itself)
  const int64_t n = 36 * (int64_t)(d.P + d.Ep) + (ld - 6 * (int64_t)d.P);
  hipLaunchKernelGGL(k_dense_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, S, ld, lambda);
}

// rhs of the reduced system: bs = bp - B Hll^-1 bl (qs from launch_reduced_rhs), zero-padded to ld
__global__ void k_dense_rhs(BADev d, double* __restrict__ rhs, int64_t ld) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= ld) return;
  rhs[i] = i < 6 * (int64_t)d.P ? d.bp[i] - d.qs[i] : 0.0;
}
void launch_dense_rhs(const BADev& d, double* rhs, int64_t ld, hipStream_t s) {
  hipLaunchKernelGGL(k_dense_rhs, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, s, d, rhs, ld);
}

// ------------------------------------------------------------------------------ launchers
static size_t schur_lds(const BADev& d) { return (6 * VDO_TILE_PLANE + 24 * (size_t)d.max_slots + ((size_t)d.max_slots + 1) / 2) * sizeof(double); }      // (+ the slots' row ids, int32)

void launch_expand_binc(const BADev& d, hipStream_t s) {
  if (d.n_tiles) hipLaunchKernelGGL(k_expand_binc, dim3(d.n_tiles), dim3(VDO_TILE_THREADS), raise_lds(k_expand_binc, (12 * (size_t)d.max_slots + 3 * VDO_TILE_PTS) * sizeof(double)), s, d);
}

static int red_blocks(const BADev& d) { return (int)std::min<int64_t>(256, std::max<int64_t>(1, (3 * (int64_t)d.L + 6 * (int64_t)d.P + 4095) / 4096)); }

void launch_max_diag(const BADev& d, hipStream_t s, const Reducer& R) {
  const int nb = red_blocks(d);
  hipLaunchKernelGGL(k_max_diag, dim3(nb), dim3(1024), 0, s, d);
  hipLaunchKernelGGL(k_reduce_part, dim3(1), dim3(256), 0, s, d, nb, 1);
  if (d.sharded) R(d.scal + S_MAXDIAG, 1, 1);
}

// The factorisations of a Levenberg trial and the reduced right-hand side, which only needs the landmark factors.  The pose-chain
// factorisation is a recurrence over the chain (a few workgroups, 0.7 us per step): on one GPU the reduced right-hand side (k_schur_tile<1>,
// k_gather_q: every CU) runs beside it on the stream `side`, forked / joined with the two events.  Sharded solves keep one stream: the
// exchanges of the all-reduce hook are ordered on it.
void launch_factor_and_rhs(const BADev& d, double lambda, hipStream_t s, const Reducer& R, hipStream_t side, hipEvent_t fork, hipEvent_t join, bool precond, bool lin_pending) {
  hipMemsetAsync(d.flags, 0, 4 * sizeof(int32_t), s);
  if (d.n_chains) hipLaunchKernelGGL(k_factor_chains, dim3((d.n_chains + 127) / 128), dim3(128), 0, s, d, lambda);
  precond = precond || d.sharded;          // (the dense solver needs the landmark factors and the reduced right-hand side only; a sharded run keeps its exchanges as they are)
  if (precond) {
    const size_t lds = (33 * (size_t)d.max_slots + 3 * VDO_TILE_PLANE + ((size_t)d.max_slots + 1) / 2) * sizeof(double);
    const int nd = d.n_tiles < 1024 ? d.n_tiles : std::min(d.n_dyn_tiles, d.n_tiles);       // tiles with dynamic tracks come first in the launch order (a graph of few tiles: one launch - a second one costs more than the registers)
    if (nd > 0) hipLaunchKernelGGL(k_precond_tile<true>, dim3(nd), dim3(VDO_TILE_THREADS), raise_lds(k_precond_tile<true>, lds), s, d, 0);
    if (d.n_tiles > nd) hipLaunchKernelGGL(k_precond_tile<false>, dim3(d.n_tiles - nd), dim3(VDO_TILE_THREADS), raise_lds(k_precond_tile<false>, lds), s, d, nd);
    launch_hub_precond(d, s);
  }
  const dim3 g((d.P + 3) / 4), b(256);
  if (!precond) { }
  else if (!d.sharded) hipLaunchKernelGGL(k_precond_finalize<0>, g, b, 0, s, d, lambda);
  else {
    // Sharded (round 6): the reduced right-hand side needs the landmark factors only, so its tile pass runs BEFORE the exchange and the block-Jacobi sums (21 P + 1) and
    // qs (6 P) - contiguous in memory - cross the ranks in ONE all-reduce per trial instead of two dependent ones.
    hipLaunchKernelGGL(k_precond_finalize<1>, g, b, 0, s, d, lambda);
    if (d.n_tiles) hipLaunchKernelGGL(k_schur_tile<1>, dim3(d.n_tiles), dim3(VDO_TILE_THREADS), raise_lds(k_schur_tile<1>, schur_lds(d)), s, d, (const double*)nullptr, (const double*)nullptr);
    launch_hub_schur(d, 1, nullptr, nullptr, s);
    hipLaunchKernelGGL(k_gather_q, dim3((d.P + 3) / 4), dim3(256), 0, s, d, d.qs, 0);
    // lin_pending: the linearisation in front of this trial left its exchange to us - Hpp | bp | chi2 (42 P + 4) lie right in front of msum | qs: one all-reduce of
    // 69 P + 5 doubles for the first trial of an LM iteration instead of two dependent ones
    if (lin_pending) { R(d.Hpp, 69 * (int64_t)d.P + 5); launch_linearize_finish(d, s); }
    else R(d.msum, 27 * (int64_t)d.P + 1);
    hipLaunchKernelGGL(k_precond_finalize<2>, g, b, 0, s, d, lambda);
  }
  const bool two = side && fork && join && !d.sharded && precond;      // (without the pose-chain factorisation nothing runs beside the reduced right-hand side: no fork / join - two cross-stream waits of ~10 us on a window-sized graph)
  hipStream_t sr = two ? side : s;
  if (two) { hipEventRecord(fork, s); hipStreamWaitEvent(side, fork, 0); }
  if (precond) {
    // (A/B on one box, ms per LM iteration, 60-frame | 1 M-point graph: twisted + Gauss-Jordan 0.410 | 0.906, twisted + closed form 0.416 | 0.939,
    // untwisted + Gauss-Jordan - rounds 2-4 - 0.423 | 0.979, untwisted + closed form 0.435 | 1.030; profiles/r05_chain_ab.txt)
    if (d.pc_closed) hipLaunchKernelGGL(k_pchain_factor<0>, dim3(d.n_pchains), dim3(128), 0, s, d);
    else hipLaunchKernelGGL(k_pchain_factor<1>, dim3(d.n_pchains), dim3(128), 0, s, d);
    if (d.pc_lds && d.pc_nwave > 1) hipLaunchKernelGGL(k_pchain_prefix, dim3(d.n_pchains), dim3(64 * d.pc_nwave), 0, s, d);
  }
  if (!d.sharded) {
    if (d.n_tiles) hipLaunchKernelGGL(k_schur_tile<1>, dim3(d.n_tiles), dim3(VDO_TILE_THREADS), raise_lds(k_schur_tile<1>, schur_lds(d)), sr, d, (const double*)nullptr, (const double*)nullptr);
    launch_hub_schur(d, 1, nullptr, nullptr, sr);
    hipLaunchKernelGGL(k_gather_q, dim3((d.P + 3) / 4), dim3(256), 0, sr, d, d.qs, 0);
  }
  if (two) { hipEventRecord(join, side); hipStreamWaitEvent(s, join, 0); }
}

static size_t pc_strip_bytes(const BADev& d) {
  const size_t bytes = ((d.pc_lds ? 6 * (size_t)d.pc_maxlen : 0) + 96 + 24 + PC_BMAT_DOUBLES) * sizeof(double);
  // more than the default 64 KB of dynamic LDS: tell the runtime.  The attribute is per DEVICE (a process may hold BA contexts on
  // several GPUs): the size already granted is remembered per device id; the calls are idempotent.
  static std::atomic<size_t> raised[64];
  if (bytes > (size_t)(48 * 1024)) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (bytes > raised[dev].load(std::memory_order_relaxed)) {
      const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pcg_chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pcg_chain<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e1 == hipSuccess && e2 == hipSuccess) raised[dev].store(bytes, std::memory_order_relaxed);
    }
  }
  return bytes;
}
void launch_pcg_init(const BADev& d, hipStream_t s) { hipLaunchKernelGGL(k_pcg_chain<1>, dim3(d.n_pchains), dim3(64 * d.pc_nwave), pc_strip_bytes(d), s, d, 0.0, 0, 0); }

void launch_pcg_iter(const BADev& d, double lambda, double tol2, int parity, hipStream_t s, const Reducer& R) {
  if (d.n_tiles) hipLaunchKernelGGL(k_schur_tile<0>, dim3(d.n_tiles), dim3(VDO_TILE_THREADS), raise_lds(k_schur_tile<0>, schur_lds(d)), s, d, (const double*)d.zp, (const double*)(parity ? d.pp2 : d.pp));
  launch_hub_schur(d, 0, d.zp, parity ? d.pp2 : d.pp, s);
  const int nq = (d.P + 3) / 4;
  if (d.sharded) {
    hipLaunchKernelGGL(k_gather_q, dim3(nq), dim3(256), 0, s, d, d.qs, 1);
    R(d.qs, 6 * (int64_t)d.P);                     // the one exchange per CG iteration: 6P doubles
    hipLaunchKernelGGL(k_pcg_q, dim3(nq), dim3(256), 0, s, d, lambda, parity, 0);
  } else hipLaunchKernelGGL(k_pcg_q, dim3(nq), dim3(256), 0, s, d, lambda, parity, 1);
  hipLaunchKernelGGL(k_pcg_chain<0>, dim3(d.n_pchains), dim3(64 * d.pc_nwave), pc_strip_bytes(d), s, d, tol2, parity, nq);
}

// (profiling: the Schur mat-vec of one CG iteration alone, on whatever direction the last solve left in zp / pp; the caller clears flags[1])
void launch_schur_matvec_only(const BADev& d, hipStream_t s) {
  if (d.n_tiles) hipLaunchKernelGGL(k_schur_tile<0>, dim3(d.n_tiles), dim3(VDO_TILE_THREADS), raise_lds(k_schur_tile<0>, schur_lds(d)), s, d, (const double*)d.zp, (const double*)d.pp);
}

// the LM scalars and flags -> the handle's mapped pinned block [S_COUNT doubles][4 int32] (visible to the host once the stream has been waited for)
__global__ __launch_bounds__(64) void k_publish_scalars(const double* __restrict__ scal, const int32_t* __restrict__ flags, double* __restrict__ h_block, uint32_t ticket) {
  const int t = threadIdx.x;
  if (t < S_COUNT) h_block[t] = scal[t];
  if (t < 4) ((int32_t*)(h_block + S_COUNT))[t] = flags[t];
  if (ticket) {                                          // the host polls this word (ba_lm.hip fetch): the data above is visible system-wide before it
    __threadfence_system();
    if (t == 0) ((volatile uint32_t*)(h_block + S_COUNT))[4] = ticket;
  }
}
void launch_publish_scalars(const BADev& d, double* h_block_dev, hipStream_t s, uint32_t ticket) {
  static_assert(S_COUNT <= 64, "one wave publishes the scalars");
  hipLaunchKernelGGL(k_publish_scalars, dim3(1), dim3(64), 0, s, (const double*)d.scal, (const int32_t*)d.flags, h_block_dev, ticket);
}

void launch_backsub_update(const BADev& d, double lambda, bool ortho, hipStream_t s) {
  if (d.n_tiles) hipLaunchKernelGGL(k_schur_tile<2>, dim3(d.n_tiles), dim3(VDO_TILE_THREADS), raise_lds(k_schur_tile<2>, schur_lds(d)), s, d, (const double*)d.xp, (const double*)nullptr);
  launch_hub_schur(d, 2, d.xp, nullptr, s);
  const int nb = red_blocks(d);
  hipLaunchKernelGGL(k_update, dim3(nb), dim3(1024), 0, s, d, lambda, ortho ? 1 : 0);
  hipLaunchKernelGGL(k_reduce_part, dim3(1), dim3(256), 0, s, d, nb, 0);
}

}  // namespace vdo

#ifdef DENSE_PROF
extern "C" int vdo_debug_dense_prof(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(vdo::g_asm_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(vdo::g_asm_prof), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

#ifdef PCG_PROF
extern "C" int vdo_debug_pcg_prof(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(vdo::g_pcg_prof), sizeof(unsigned long long) * 20) != hipSuccess) return 1;
  if (reset) { unsigned long long z[20] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(vdo::g_pcg_prof), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif
