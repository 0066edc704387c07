// HUB landmarks of the batch graph (round 6): a static point observed from more pose vertices than a tile holds (256 slots / 256 per-pose pieces / 1 536 edges) -
// g2o has no such limit (g2o/core/block_solver.hpp:143-295 builds its Schur complement over any vertex degree).  Such a point belongs to no tile: ONE WORKGROUP PER HUB
// walks its EdgeSE3PointXYZ edges (g2o/types/edge_se3_pointxyz.cpp) in rounds of 256 and does, for them, what the tile kernels do for the edges of a tile:
//   k_hub_sweep<BUILD>   = k_sweep_tile     errors, Huber, the edge's `we`, the landmark's Hll / bl (fixed-order block sums), ONE pose-major partial row per edge
//   k_hub_precond        = k_precond_tile   the 21 sums of  B Hll^-1 B^T  of every edge -> its row of part_m / part_m8
//   k_hub_schur<MODE>    = k_schur_tile     0: part_q = B Hll^-1 B^T p     1: part_q = B Hll^-1 bl     2: xl = Hll^-1 (bl - B^T x_p)
// Every hub edge owns a row of the pose-major partial arrays (capi_ba.hip appends one slot per hub edge to tile_pose / slot_dst / ps_off), so k_finalize_pose, k_gather_q /
// k_pcg_q and k_precond_finalize pick the hub's contributions up with the rows of the (tile, slot) pairs - no change on the pose side.  The landmark side is the scalar
// Hll[l] (the block is Hll * I3: J_point^T J_point = R R^T) and bl[l], x_l of the hub's device point; k_factor_chains gives it dscal like any single point.
// Same arithmetic as the tile kernels (se3_dev.hpp: cam_point, chi2_w3, huber_dev; the 16 running sums of ba_sweep.hip acc_terms); sums over a hub's edges in a fixed
// order (strided per-thread partial sums, wave butterfly, waves in order): run-independent bits.  Graphs with hubs are rare and their hubs few: nothing here is tuned.
#include <hip/hip_runtime.h>

#include "ba_dev.hpp"
#include "ba_tile.hpp"
#include "se3_dev.hpp"

namespace vdo {

namespace {

struct HubEdge { double W[12]; D3 zc; };

// the inverse pose of the edge's vertex and the point in its frame, as the tile kernels form them (linearisation point: estimate[which])
__device__ __forceinline__ HubEdge hub_edge(const BADev& d, int which, int e, D3 p) {
  HubEdge h;
  const IsoD W = iso_inv(iso_load(d.pose[which] + 12 * (int64_t)d.hub_pose[e]));
#pragma unroll
  for (int i = 0; i < 9; ++i) h.W[i] = W.r[i];
  h.W[9] = W.t.x; h.W[10] = W.t.y; h.W[11] = W.t.z;
  h.zc = cam_point(h.W, p);
  return h;
}

template <int N>
__device__ __forceinline__ void block_sums(double (&v)[N], double* lds /* [16 * N + N] */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < N; ++i) lds[16 * i + wv] = v[i];
  __syncthreads();
  if (threadIdx.x < N) {
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += lds[16 * threadIdx.x + w];
    lds[16 * N + threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = lds[16 * N + i];
}

}  // namespace

template <bool BUILD>
__global__ __launch_bounds__(256) void k_hub_sweep(BADev d, int which) {
  __shared__ double lds[16 * 6 + 6];
  const int hub = blockIdx.x, tid = threadIdx.x;
  const int e0 = d.hub_off[hub], e1 = d.hub_off[hub + 1], Eh = d.n_hub_edges;
  const int64_t l = d.hub_point[hub];
  const D3 p{d.point[which][3 * l], d.point[which][3 * l + 1], d.point[which][3 * l + 2]};
  double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // chi2, robust chi2, Hll, bl
  for (int e = e0 + tid; e < e1; e += 256) {
    const HubEdge h = hub_edge(d, which, e, p);
    const D3 z{d.hub_z[e], d.hub_z[Eh + e], d.hub_z[2 * (int64_t)Eh + e]};
    const double w = d.hub_w[e];
    const D3 er = h.zc - z;
    const double c2 = chi2_w3(w, er);
    double rho0, rho1;
    huber_dev(c2, d.huber_eb, d.dsqr_eb, rho0, rho1);
    s[0] += c2; s[1] += rho0;
    if (BUILD) {
      const double we = w * rho1;
      d.hub_we[e] = we;
      const double* Wp = h.W;
      const D3 Re{__builtin_fma(Wp[6], er.z, __builtin_fma(Wp[3], er.y, Wp[0] * er.x)), __builtin_fma(Wp[7], er.z, __builtin_fma(Wp[4], er.y, Wp[1] * er.x)),
                  __builtin_fma(Wp[8], er.z, __builtin_fma(Wp[5], er.y, Wp[2] * er.x))};
      s[2] += we; s[3] += -we * Re.x; s[4] += -we * Re.y; s[5] += -we * Re.z;
      // the edge's 16 running sums (ba_sweep.hip acc_terms on zeroed accumulators) -> its own pose-major row
      const D3 c = h.zc;
      const double wx = we * c.x, wy = we * c.y, wz = we * c.z;
      double a[16];
      a[0] = we; a[1] = wx; a[2] = wy; a[3] = wz;
      a[4] = wx * c.x; a[5] = wx * c.y; a[6] = wx * c.z; a[7] = wy * c.y; a[8] = wy * c.z; a[9] = wz * c.z;
      a[10] = we * er.x; a[11] = we * er.y; a[12] = we * er.z;
      a[13] = we * __builtin_fma(c.y, er.z, -(c.z * er.y)); a[14] = we * __builtin_fma(c.z, er.x, -(c.x * er.z)); a[15] = we * __builtin_fma(c.x, er.y, -(c.y * er.x));
      double* row = d.part_sums + (int64_t)d.ps_stride * d.hub_row[e];
#pragma unroll
      for (int i = 0; i < 16; ++i) row[i] = a[i];
      if (d.ps_stride == 32)
#pragma unroll
        for (int i = 16; i < 32; ++i) row[i] = 0.0;      // (the ternary half of a wide row: a hub is a static point)
    }
  }
  block_sums<6>(s, lds);
  if (tid == 0) {
    d.hub_chi[hub] = s[0]; d.hub_chi[d.n_hubs + hub] = s[1];
    if (BUILD) { d.Hll[l] = s[2]; d.bl[3 * l] = s[3]; d.bl[3 * l + 1] = s[4]; d.bl[3 * l + 2] = s[5]; }
  }
}

// the 21 block-Jacobi sums of every hub edge (ba_solve.hip k_precond_tile, single points): upper triangle of  s [[I, -2[c]x], [2[c]x, 4 (|c|^2 I - c c^T)]],  s = we^2 / (Hll + lambda)
__global__ __launch_bounds__(256) void k_hub_precond(BADev d) {
  const int hub = blockIdx.x;
  const int e0 = d.hub_off[hub], e1 = d.hub_off[hub + 1];
  const int64_t l = d.hub_point[hub];
  const D3 p{d.point[0][3 * l], d.point[0][3 * l + 1], d.point[0][3 * l + 2]};
  const double dsc = d.dscal[l];
  for (int e = e0 + threadIdx.x; e < e1; e += 256) {
    const D3 c = hub_edge(d, 0, e, p).zc;
    const double we = d.hub_we[e];
    const double sw = dsc * we * we;
    const double wx = sw * c.x, wy = sw * c.y, wz = sw * c.z;
    const double sxx = wx * c.x, sxy = wx * c.y, sxz = wx * c.z, syy = wy * c.y, syz = wy * c.z, szz = wz * c.z;
    double up[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) up[i] = 0.0;
    up[0] = sw; up[4] = 2.0 * wz; up[5] = -2.0 * wy;
    up[6] = sw; up[8] = -2.0 * wz; up[10] = 2.0 * wx;
    up[11] = sw; up[12] = 2.0 * wy; up[13] = -2.0 * wx;
    up[15] = 4.0 * (syy + szz); up[16] = -4.0 * sxy; up[17] = -4.0 * sxz;
    up[18] = 4.0 * (sxx + szz); up[19] = -4.0 * syz;
    up[20] = 4.0 * (sxx + syy);
    const int64_t row = d.hub_row[e];
#pragma unroll
    for (int k = 0; k < 16; ++k) d.part_m[16 * row + k] = up[k];
#pragma unroll
    for (int k = 16; k < 21; ++k) d.part_m8[8 * row + (k - 16)] = up[k];
  }
}

// MODE 0: part_q = B Hll^-1 B^T (v + beta v2)   MODE 1: part_q = B Hll^-1 bl   MODE 2: xl = Hll^-1 (bl - B^T v)      (ba_solve.hip k_schur_tile, a chain of one point)
template <int MODE>
__global__ __launch_bounds__(256) void k_hub_schur(BADev d, const double* __restrict__ v, const double* __restrict__ v2) {
  __shared__ double lds[16 * 3 + 3];
  if (MODE == 0 && d.flags[1]) return;     // PCG already converged: the launches queued behind it are no-ops
  const int hub = blockIdx.x, tid = threadIdx.x;
  const int e0 = d.hub_off[hub], e1 = d.hub_off[hub + 1];
  const int64_t l = d.hub_point[hub];
  const D3 p{d.point[0][3 * l], d.point[0][3 * l + 1], d.point[0][3 * l + 2]};
  const double g = d.dscal[l];
  double u[3] = {0.0, 0.0, 0.0};
  if (MODE != 1) {                         // u = sum_e B_e^T v_pose(e) = -we R (v_t - 2 c x v_r)
    const double beta = MODE == 0 ? d.scal[S_BETA] : 0.0;
    for (int e = e0 + tid; e < e1; e += 256) {
      const HubEdge h = hub_edge(d, 0, e, p);
      const int64_t pid = d.hub_pose[e];
      double pv[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) pv[i] = MODE == 0 ? v[6 * pid + i] + beta * v2[6 * pid + i] : v[6 * pid + i];
      const D3 c = h.zc;
      const D3 t{pv[0] - 2.0 * (c.y * pv[5] - c.z * pv[4]), pv[1] - 2.0 * (c.z * pv[3] - c.x * pv[5]), pv[2] - 2.0 * (c.x * pv[4] - c.y * pv[3])};
      const D3 o = (-d.hub_we[e]) * rotT(h.W, t);
      u[0] += o.x; u[1] += o.y; u[2] += o.z;
    }
    block_sums<3>(u, lds);
  }
  D3 y{u[0], u[1], u[2]};
  const D3 blv{d.bl[3 * l], d.bl[3 * l + 1], d.bl[3 * l + 2]};
  if (MODE == 1) y = blv;
  if (MODE == 2) y = blv - y;
  const D3 w{g * y.x, g * y.y, g * y.z};
  if (MODE == 2) {
    if (tid == 0) { d.xl[3 * l] = w.x; d.xl[3 * l + 1] = w.y; d.xl[3 * l + 2] = w.z; }
    return;
  }
  for (int e = e0 + tid; e < e1; e += 256) {      // q_row = B_e w = -we [ R^T w ; 2 c x (R^T w) ]
    const HubEdge h = hub_edge(d, 0, e, p);
    const D3 yy = rot(h.W, w);
    const D3 c = h.zc;
    const double sg = -d.hub_we[e];
    double* row = d.part_q + 8 * (int64_t)d.hub_row[e];
    row[0] = sg * yy.x; row[1] = sg * yy.y; row[2] = sg * yy.z;
    row[3] = sg * 2.0 * (c.y * yy.z - c.z * yy.y);
    row[4] = sg * 2.0 * (c.z * yy.x - c.x * yy.z);
    row[5] = sg * 2.0 * (c.x * yy.y - c.y * yy.x);
  }
}

// explicit 6x3 pose-landmark blocks of the hub edges (vdo_ba_download_system): B = -we [ I ; 2 [c]x ] R^T   (ba_solve.hip expand_block, kind 0)
__global__ __launch_bounds__(256) void k_hub_expand_binc(BADev d, double* __restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= d.n_hub_edges) return;
  int hub = 0;
  while (hub + 1 < d.n_hubs && d.hub_off[hub + 1] <= e) ++hub;
  const int64_t l = d.hub_point[hub];
  const D3 p{d.point[0][3 * l], d.point[0][3 * l + 1], d.point[0][3 * l + 2]};
  const HubEdge h = hub_edge(d, 0, e, p);
  const double we = d.hub_we[e], s2 = -2.0 * we;
  const D3 c = h.zc;
  double* B = out + 18 * (int64_t)e;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double a = h.W[j], b = h.W[3 + j], cc = h.W[6 + j];   // column j of R^T
    B[0 * 3 + j] = -we * a; B[1 * 3 + j] = -we * b; B[2 * 3 + j] = -we * cc;
    B[3 * 3 + j] = s2 * (c.y * cc - c.z * b);
    B[4 * 3 + j] = s2 * (c.z * a - c.x * cc);
    B[5 * 3 + j] = s2 * (c.x * b - c.y * a);
  }
}

void launch_hub_sweep(const BADev& d, int which, bool build, hipStream_t s) {
  if (!d.n_hubs) return;
  if (build) hipLaunchKernelGGL(k_hub_sweep<true>, dim3(d.n_hubs), dim3(256), 0, s, d, which);
  else hipLaunchKernelGGL(k_hub_sweep<false>, dim3(d.n_hubs), dim3(256), 0, s, d, which);
}
void launch_hub_precond(const BADev& d, hipStream_t s) {
  if (d.n_hubs) hipLaunchKernelGGL(k_hub_precond, dim3(d.n_hubs), dim3(256), 0, s, d);
}
void launch_hub_schur(const BADev& d, int mode, const double* v, const double* v2, hipStream_t s) {
  if (!d.n_hubs) return;
  if (mode == 0) hipLaunchKernelGGL(k_hub_schur<0>, dim3(d.n_hubs), dim3(256), 0, s, d, v, v2);
  else if (mode == 1) hipLaunchKernelGGL(k_hub_schur<1>, dim3(d.n_hubs), dim3(256), 0, s, d, v, v2);
  else hipLaunchKernelGGL(k_hub_schur<2>, dim3(d.n_hubs), dim3(256), 0, s, d, v, v2);
}
void launch_hub_expand_binc(const BADev& d, double* binc18, hipStream_t s) {
  if (d.n_hub_edges) hipLaunchKernelGGL(k_hub_expand_binc, dim3((d.n_hub_edges + 255) / 256), dim3(256), 0, s, d, binc18);
}

}  // namespace vdo
