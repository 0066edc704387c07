// Batch-BA linearisation sweep for gfx950 (K18 in SURVEY.md §2.1): per-edge SE(3) residual +
// Jacobian, Huber weight, and block accumulation — the work of g2o's
// SparseOptimizer::computeActiveErrors (g2o/core/sparse_optimizer.cpp:61-114) and
// BlockSolver::buildSystem (g2o/core/block_solver.hpp:502-560) with the edge classes
//   EdgeSE3PointXYZ            g2o/types/edge_se3_pointxyz.cpp:99-140
//   LandmarkMotionTernaryEdge  g2o/types/types_dyn_slam3d.cpp:53-85   (F4 kept: no factor 2)
//   EdgeSE3 / EdgeSE3Prior     g2o/types/edge_se3.cpp:77-104, edge_se3_prior.cpp:89-102,
//                              isometry3d_gradients.h:191-325, dquat2mat.cpp:35-84
//
// HBM-bound design (MI355X): one workgroup per *chunk* (<=1024 consecutive edges that share a
// pose vertex).  The pose is wave-uniform (scalar loads), edge data is SoA and read fully
// coalesced (40 B/edge), the 6x3 pose-landmark block is written once, coalesced (144 B/edge).
// Because J_pose = [-I | 2[zc]x] (resp. [I | -[v]x]) the whole 6x6+6 pose contribution of an
// edge is a function of 16 running sums (Σw, Σw·zc, Σw·zc zcᵀ, Σw·e, Σw·zc×e); they are reduced
// with wave shuffles + one LDS stage per chunk and written as per-chunk partials (no atomics on
// the pose side, fixed summation order).  Landmark 3x3+3 go through fp64 L2 atomics.
#include "ba_dev.hpp"
#include "se3_dev.hpp"

namespace vdo {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Reduce N per-thread doubles over a 256-thread block; result valid in thread 0.
template <int N>
__device__ __forceinline__ void block_sum(double (&acc)[N], double* lds /* [4][N] */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double v = wave_sum(acc[i]);
    if (lane == 0) lds[wv * N + i] = v;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double s = lds[threadIdx.x];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += lds[w * N + threadIdx.x];
    lds[threadIdx.x] = s;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------- EdgeSE3PointXYZ
template <bool BUILD>
__global__ __launch_bounds__(VDO_SWEEP_THREADS) void k_sweep_eb(BADev d, int which) {
  __shared__ double lds[4 * 18];
  const Chunk c = d.chunks_b[blockIdx.x];
  const double* __restrict__ pose = d.pose[which];
  const double* __restrict__ point = d.point[which];
  // camera pose (uniform): w2l = X^-1
  const IsoD X = iso_load(pose + 12 * (int64_t)c.pose);
  const IsoD W = iso_inv(X);     // W.r = R^T = Jl (row-major)
  double M[6];                   // Jl^T Jl = R R^T, upper triangle
  if (BUILD) {
    M[0] = X.r[0] * X.r[0] + X.r[1] * X.r[1] + X.r[2] * X.r[2];
    M[1] = X.r[0] * X.r[3] + X.r[1] * X.r[4] + X.r[2] * X.r[5];
    M[2] = X.r[0] * X.r[6] + X.r[1] * X.r[7] + X.r[2] * X.r[8];
    M[3] = X.r[3] * X.r[3] + X.r[4] * X.r[4] + X.r[5] * X.r[5];
    M[4] = X.r[3] * X.r[6] + X.r[4] * X.r[7] + X.r[5] * X.r[8];
    M[5] = X.r[6] * X.r[6] + X.r[7] * X.r[7] + X.r[8] * X.r[8];
  }
  double acc[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] = 0.0;
  const int64_t Eb = d.Eb;
  for (int e = c.begin + (int)threadIdx.x; e < c.end; e += VDO_SWEEP_THREADS) {
    const int pt = d.eb_point[e];
    const double w = d.eb_w[e];
    const D3 z{d.eb_z[e], d.eb_z[Eb + e], d.eb_z[2 * Eb + e]};
    const D3 p{point[3 * (int64_t)pt], point[3 * (int64_t)pt + 1], point[3 * (int64_t)pt + 2]};
    const D3 zc = iso_apply(W, p);
    const D3 er = zc - z;
    const double chi = er.x * (w * er.x) + er.y * (w * er.y) + er.z * (w * er.z);
    double rho0, rho1;
    huber(chi, d.huber_eb, d.dsqr_eb, rho0, rho1);
    acc[16] += chi;
    acc[17] += rho0;
    if (BUILD) {
      const double we = w * rho1;
      // 6x3 block: rows 0..2 = -we*Jl ; rows 3..5 = -we * 2[zc]x * Jl   (column j of Jl = W.r[.][j])
      double* B = d.Binc + e;
      const int64_t N = d.Ninc;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double a = W.r[j], b = W.r[3 + j], cc = W.r[6 + j];   // column j of Jl
        B[(0 * 3 + j) * N] = -we * a;
        B[(1 * 3 + j) * N] = -we * b;
        B[(2 * 3 + j) * N] = -we * cc;
        const double s = -2.0 * we;
        B[(3 * 3 + j) * N] = s * (zc.y * cc - zc.z * b);
        B[(4 * 3 + j) * N] = s * (zc.z * a - zc.x * cc);
        B[(5 * 3 + j) * N] = s * (zc.x * b - zc.y * a);
      }
      // landmark side: Hll += we * R R^T, bl += -we * R e     (Jl^T = R)
      double* H = d.Hll + 9 * (int64_t)pt;
      atomicAdd(H + 0, we * M[0]); atomicAdd(H + 1, we * M[1]); atomicAdd(H + 2, we * M[2]);
      atomicAdd(H + 4, we * M[3]); atomicAdd(H + 5, we * M[4]); atomicAdd(H + 8, we * M[5]);
      const D3 Re = rot(X.r, er);
      double* bl = d.bl + 3 * (int64_t)pt;
      atomicAdd(bl + 0, -we * Re.x); atomicAdd(bl + 1, -we * Re.y); atomicAdd(bl + 2, -we * Re.z);
      // pose side running sums
      acc[0] += we;
      acc[1] += we * zc.x; acc[2] += we * zc.y; acc[3] += we * zc.z;
      acc[4] += we * zc.x * zc.x; acc[5] += we * zc.x * zc.y; acc[6] += we * zc.x * zc.z;
      acc[7] += we * zc.y * zc.y; acc[8] += we * zc.y * zc.z; acc[9] += we * zc.z * zc.z;
      acc[10] += we * er.x; acc[11] += we * er.y; acc[12] += we * er.z;
      acc[13] += we * (zc.y * er.z - zc.z * er.y);
      acc[14] += we * (zc.z * er.x - zc.x * er.z);
      acc[15] += we * (zc.x * er.y - zc.y * er.x);
    }
  }
  block_sum<18>(acc, lds);
  const int nchunks = d.n_chunks_b + d.n_chunks_t;
  if (BUILD) {
    if (threadIdx.x < 16) d.chunk_sums[threadIdx.x * nchunks + blockIdx.x] = lds[threadIdx.x];
  }
  if (threadIdx.x < 2) d.chunk_chi[threadIdx.x * (nchunks + 1) + blockIdx.x] = lds[16 + threadIdx.x];
}

// ------------------------------------------------------------------ LandmarkMotionTernaryEdge
template <bool BUILD>
__global__ __launch_bounds__(VDO_SWEEP_THREADS) void k_sweep_et(BADev d, int which) {
  __shared__ double lds[4 * 18];
  const Chunk c = d.chunks_t[blockIdx.x];
  const double* __restrict__ pose = d.pose[which];
  const double* __restrict__ point = d.point[which];
  const IsoD H = iso_load(pose + 12 * (int64_t)c.pose);
  const IsoD Hi = iso_inv(H);    // Hi.r = R_H^T ; J2 = -Hi.r
  double M[6];
  if (BUILD) {
    M[0] = H.r[0] * H.r[0] + H.r[1] * H.r[1] + H.r[2] * H.r[2];
    M[1] = H.r[0] * H.r[3] + H.r[1] * H.r[4] + H.r[2] * H.r[5];
    M[2] = H.r[0] * H.r[6] + H.r[1] * H.r[7] + H.r[2] * H.r[8];
    M[3] = H.r[3] * H.r[3] + H.r[4] * H.r[4] + H.r[5] * H.r[5];
    M[4] = H.r[3] * H.r[6] + H.r[4] * H.r[7] + H.r[5] * H.r[8];
    M[5] = H.r[6] * H.r[6] + H.r[7] * H.r[7] + H.r[8] * H.r[8];
  }
  double acc[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] = 0.0;
  const int64_t Et = d.Et, N = d.Ninc;
  for (int e = c.begin + (int)threadIdx.x; e < c.end; e += VDO_SWEEP_THREADS) {
    const int i1 = d.et_p1[e], i2 = d.et_p2[e];
    const double w = d.et_w[e];
    const D3 z{d.et_z[e], d.et_z[Et + e], d.et_z[2 * Et + e]};
    const D3 p1{point[3 * (int64_t)i1], point[3 * (int64_t)i1 + 1], point[3 * (int64_t)i1 + 2]};
    const D3 p2{point[3 * (int64_t)i2], point[3 * (int64_t)i2 + 1], point[3 * (int64_t)i2 + 2]};
    const D3 v = iso_apply(Hi, p2);
    const D3 er = p1 - v - z;
    const double chi = er.x * (w * er.x) + er.y * (w * er.y) + er.z * (w * er.z);
    double rho0, rho1;
    huber(chi, d.huber_et, d.dsqr_et, rho0, rho1);
    acc[16] += chi;
    acc[17] += rho0;
    if (BUILD) {
      const double we = w * rho1;
      // O = we * J1^T J2 = -we * Hi.r  (3x3, p1 x p2)
      double* O = d.Oll + e;
#pragma unroll
      for (int i = 0; i < 9; ++i) O[i * Et] = -we * Hi.r[i];
      // incidence (H,p1): 6x3 = we * Jh^T : top we*I, bottom we*[v]x
      double* B1 = d.Binc + d.Eb + e;
      B1[0 * N] = we; B1[1 * N] = 0;  B1[2 * N] = 0;
      B1[3 * N] = 0;  B1[4 * N] = we; B1[5 * N] = 0;
      B1[6 * N] = 0;  B1[7 * N] = 0;  B1[8 * N] = we;
      B1[9 * N] = 0;           B1[10 * N] = -we * v.z;  B1[11 * N] = we * v.y;
      B1[12 * N] = we * v.z;   B1[13 * N] = 0;          B1[14 * N] = -we * v.x;
      B1[15 * N] = -we * v.y;  B1[16 * N] = we * v.x;   B1[17 * N] = 0;
      // incidence (H,p2): 6x3 = Jh^T we J2 = -we * Jh^T Hi.r : top -we*Hi.r ; bottom -we*[v]x*Hi.r
      double* B2 = d.Binc + d.Eb + Et + e;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double a = Hi.r[j], b = Hi.r[3 + j], cc = Hi.r[6 + j];  // column j of Hi.r
        B2[(0 * 3 + j) * N] = -we * a;
        B2[(1 * 3 + j) * N] = -we * b;
        B2[(2 * 3 + j) * N] = -we * cc;
        B2[(3 * 3 + j) * N] = -we * (v.y * cc - v.z * b);
        B2[(4 * 3 + j) * N] = -we * (v.z * a - v.x * cc);
        B2[(5 * 3 + j) * N] = -we * (v.x * b - v.y * a);
      }
      // landmark diagonals / rhs: p1: +we*I, b1 += -we*e ; p2: +we*R_H R_H^T, b2 += we * R_H e
      double* H1 = d.Hll + 9 * (int64_t)i1;
      atomicAdd(H1 + 0, we); atomicAdd(H1 + 4, we); atomicAdd(H1 + 8, we);
      double* b1 = d.bl + 3 * (int64_t)i1;
      atomicAdd(b1 + 0, -we * er.x); atomicAdd(b1 + 1, -we * er.y); atomicAdd(b1 + 2, -we * er.z);
      double* H2 = d.Hll + 9 * (int64_t)i2;
      atomicAdd(H2 + 0, we * M[0]); atomicAdd(H2 + 1, we * M[1]); atomicAdd(H2 + 2, we * M[2]);
      atomicAdd(H2 + 4, we * M[3]); atomicAdd(H2 + 5, we * M[4]); atomicAdd(H2 + 8, we * M[5]);
      const D3 Re = rot(H.r, er);
      double* b2 = d.bl + 3 * (int64_t)i2;
      atomicAdd(b2 + 0, we * Re.x); atomicAdd(b2 + 1, we * Re.y); atomicAdd(b2 + 2, we * Re.z);
      acc[0] += we;
      acc[1] += we * v.x; acc[2] += we * v.y; acc[3] += we * v.z;
      acc[4] += we * v.x * v.x; acc[5] += we * v.x * v.y; acc[6] += we * v.x * v.z;
      acc[7] += we * v.y * v.y; acc[8] += we * v.y * v.z; acc[9] += we * v.z * v.z;
      acc[10] += we * er.x; acc[11] += we * er.y; acc[12] += we * er.z;
      acc[13] += we * (v.y * er.z - v.z * er.y);
      acc[14] += we * (v.z * er.x - v.x * er.z);
      acc[15] += we * (v.x * er.y - v.y * er.x);
    }
  }
  block_sum<18>(acc, lds);
  const int nchunks = d.n_chunks_b + d.n_chunks_t;
  const int slot = d.n_chunks_b + blockIdx.x;
  if (BUILD) {
    if (threadIdx.x < 16) d.chunk_sums[threadIdx.x * nchunks + slot] = lds[threadIdx.x];
  }
  if (threadIdx.x < 2) d.chunk_chi[threadIdx.x * (nchunks + 1) + slot] = lds[16 + threadIdx.x];
}

// Expand the 16 running sums of every chunk of a pose into its 6x6 diagonal block and rhs.
__global__ void k_finalize_pose(BADev d) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.P) return;
  const int nchunks = d.n_chunks_b + d.n_chunks_t;
  double Hm[36], b[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) Hm[i] = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) b[i] = 0;
  for (int k = d.pc_off[p]; k < d.pc_off[p + 1]; ++k) {
    const int c = d.pc_idx[k];
    double s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = d.chunk_sums[i * nchunks + c];
    const bool bin = c < d.n_chunks_b;
    // top-left: S0 * I
    Hm[0] += s[0]; Hm[7] += s[0]; Hm[14] += s[0];
    // top-right (rows 0..2, cols 3..5): binary -2[S1]x ; ternary -[S1]x ; bottom-left = transpose
    const double f = bin ? -2.0 : -1.0;
    const double sx = f * s[1], sy = f * s[2], sz = f * s[3];
    // [a]x = [0 -az ay; az 0 -ax; -ay ax 0]
    Hm[0 * 6 + 4] += -sz; Hm[0 * 6 + 5] += sy;
    Hm[1 * 6 + 3] += sz;  Hm[1 * 6 + 5] += -sx;
    Hm[2 * 6 + 3] += -sy; Hm[2 * 6 + 4] += sx;
    // bottom-right: g * (tr(S2) I - S2), g = 4 (binary) / 1 (ternary)
    const double g = bin ? 4.0 : 1.0;
    const double tr = s[4] + s[7] + s[9];
    Hm[3 * 6 + 3] += g * (tr - s[4]); Hm[3 * 6 + 4] += -g * s[5];       Hm[3 * 6 + 5] += -g * s[6];
    Hm[4 * 6 + 4] += g * (tr - s[7]); Hm[4 * 6 + 5] += -g * s[8];
    Hm[5 * 6 + 5] += g * (tr - s[9]);
    if (bin) {
      b[0] += s[10]; b[1] += s[11]; b[2] += s[12];
      b[3] += 2.0 * s[13]; b[4] += 2.0 * s[14]; b[5] += 2.0 * s[15];
    } else {
      b[0] -= s[10]; b[1] -= s[11]; b[2] -= s[12];
      b[3] -= s[13]; b[4] -= s[14]; b[5] -= s[15];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < i; ++j) Hm[i * 6 + j] = Hm[j * 6 + i];
  double* Ho = d.Hpp + 36 * (int64_t)p;
#pragma unroll
  for (int i = 0; i < 36; ++i) Ho[i] = Hm[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) d.bp[6 * (int64_t)p + i] = b[i];
}

__global__ void k_mirror_points(BADev d) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= d.L) return;
  double* H = d.Hll + 9 * (int64_t)l;
  H[3] = H[1]; H[6] = H[2]; H[7] = H[5];
}

// ---------------------------------------------------------------- EdgeSE3 / EdgeSE3Prior
// d(q_xyz)/dR, dq[3][9], column index = i + 3j (column-major R); branch choice as in
// _q2m (g2o/types/dquat2mat.cpp:35-64).
__device__ void dq_dR_dev(const double* R, double (*dq)[9]) {
  for (int a = 0; a < 3; ++a) for (int c = 0; c < 9; ++c) dq[a][c] = 0;
  const double r00 = R[0], r11 = R[4], r22 = R[8];
  const double tr = r00 + r11 + r22;
  double qw;
  if (tr > 0) {
    const double S = sqrt(tr + 1.0) * 2;
    qw = 0.25 * S;
    const double a = 0.25 / qw, dd = -0.03125 / (qw * qw * qw);
    const int hi[3][2] = {{2, 1}, {0, 2}, {1, 0}};
    for (int k = 0; k < 3; ++k) {
      const int i = hi[k][0], j = hi[k][1];
      const double num = R[3 * i + j] - R[3 * j + i];
      dq[k][0] = dq[k][4] = dq[k][8] = num * dd;
      dq[k][i + 3 * j] = a;
      dq[k][j + 3 * i] = -a;
    }
  } else {
    int dm;
    if ((r00 > r11) & (r00 > r22)) dm = 0; else if (r11 > r22) dm = 1; else dm = 2;
    double s[3] = {-1, -1, -1};
    s[dm] = 1;
    const double S = sqrt(1.0 + s[0] * r00 + s[1] * r11 + s[2] * r22) * 2;
    const int j = (dm + 1) % 3, k = (dm + 2) % 3;
    qw = (R[3 * k + j] - R[3 * j + k]) / S;
    const double qd = 0.25 * S;
    const double a = 0.25 / qd, g = 0.125 / qd, d3 = 0.03125 / (qd * qd * qd);
    for (int i = 0; i < 3; ++i) dq[dm][i + 3 * i] = s[i] * g;
    for (int o = 0; o < 3; ++o) {
      if (o == dm) continue;
      const double num = R[3 * dm + o] + R[3 * o + dm];
      for (int i = 0; i < 3; ++i) dq[o][i + 3 * i] = -s[i] * d3 * num;
      dq[o][dm + 3 * o] = a;
      dq[o][o + 3 * dm] = a;
    }
  }
  if (qw <= 0) for (int a = 0; a < 3; ++a) for (int c = 0; c < 9; ++c) dq[a][c] = -dq[a][c];
}

// skew(Sx,Sy,Sz,R) with sign (isometry3d_gradients.h:57-85); S[k] row-major 3x3
__device__ void skew3_dev(const double* R, double sgn, double (*S)[9]) {
  double r[9];
  for (int i = 0; i < 9; ++i) r[i] = sgn * 2 * R[i];
  const double Sx[9] = {0, 0, 0, -r[6], -r[7], -r[8], r[3], r[4], r[5]};
  const double Sy[9] = {r[6], r[7], r[8], 0, 0, 0, -r[0], -r[1], -r[2]};
  const double Sz[9] = {-r[3], -r[4], -r[5], r[0], r[1], r[2], 0, 0, 0};
  for (int i = 0; i < 9; ++i) { S[0][i] = Sx[i]; S[1][i] = Sy[i]; S[2][i] = Sz[i]; }
}
// J(3..5,3..5) = dq * [vec(A Sx) vec(A Sy) vec(A Sz)]   (column-major vec)
__device__ void rot_block_dev(const double (*dq)[9], const double* A, const double (*S)[9], double* J) {
  for (int c = 0; c < 3; ++c) {
    double Pm[9];
    mat3_mul(A, S[c], Pm);
    for (int a = 0; a < 3; ++a) {
      double s = 0;
      for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row) s += dq[a][row + 3 * col] * Pm[3 * row + col];
      J[(3 + a) * 6 + 3 + c] = s;
    }
  }
}

__device__ void edge_se3_dev(const IsoD& Z, const IsoD& Xi, const IsoD& Xj, double* e, double* Ji, double* Jj) {
  const IsoD A = iso_inv(Z);
  const IsoD B = iso_mul(iso_inv(Xi), Xj);
  const IsoD E = iso_mul(A, B);
  const D3 q = compact_quat(E.r);
  e[0] = E.t.x; e[1] = E.t.y; e[2] = E.t.z; e[3] = q.x; e[4] = q.y; e[5] = q.z;
  if (!Ji) return;
  for (int i = 0; i < 36; ++i) Ji[i] = Jj[i] = 0;
  double dq[3][9];
  dq_dR_dev(E.r, dq);
  // Ra * skewT(tb): skewT = 2[tb]x
  const double x = 2 * B.t.x, y = 2 * B.t.y, z = 2 * B.t.z;
  const double St[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  double RaS[9];
  mat3_mul(A.r, St, RaS);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      Ji[r * 6 + c] = -A.r[3 * r + c];
      Jj[r * 6 + c] = E.r[3 * r + c];
      Ji[r * 6 + 3 + c] = RaS[3 * r + c];
    }
  double S[3][9];
  skew3_dev(B.r, -1.0, S);
  rot_block_dev(dq, A.r, S, Ji);
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  skew3_dev(I3, 1.0, S);
  rot_block_dev(dq, E.r, S, Jj);
}

__device__ void edge_prior_dev(const IsoD& Z, const IsoD& X, double* e, double* J) {
  const IsoD A = iso_mul(iso_inv(Z), X);
  const D3 q = compact_quat(A.r);
  e[0] = A.t.x; e[1] = A.t.y; e[2] = A.t.z; e[3] = q.x; e[4] = q.y; e[5] = q.z;
  if (!J) return;
  for (int i = 0; i < 36; ++i) J[i] = 0;
  double dq[3][9];
  dq_dR_dev(A.r, dq);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J[r * 6 + c] = A.r[3 * r + c];
  double S[3][9];
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  skew3_dev(I3, 1.0, S);
  rot_block_dev(dq, A.r, S, J);
}

__device__ double chi2_6(const double* e, const double* info) {
  double s = 0;
  for (int i = 0; i < 6; ++i) {
    double t = 0;
    for (int j = 0; j < 6; ++j) t += info[i * 6 + j] * e[j];
    s += e[i] * t;
  }
  return s;
}

// out(6x6) = Ja^T (w * Omega) Jb
__device__ void jtwj6(const double* Ja, const double* Om, double w, const double* Jb, double* out) {
  double WJ[36];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += Om[i * 6 + k] * Jb[k * 6 + j];
      WJ[i * 6 + j] = w * s;
    }
  for (int a = 0; a < 6; ++a)
    for (int c = 0; c < 6; ++c) {
      double s = 0;
      for (int i = 0; i < 6; ++i) s += Ja[i * 6 + a] * WJ[i * 6 + c];
      out[a * 6 + c] = s;
    }
}

// one thread per EdgeSE3 (k < Ep) or prior (k >= Ep).  ep_chi: [2][Ep+Npr] (chi2, robust chi2)
template <bool BUILD>
__global__ void k_posepose(BADev d, int which, double* ep_chi) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = d.Ep + d.Npr;
  if (k >= n) return;
  const double* pose = d.pose[which];
  double e[6], Ji[36], Jj[36], Hm[36];
  if (k < d.Ep) {
    const int vi = d.ep_i[k], vj = d.ep_j[k];
    const double* info = d.ep_info + 36 * (int64_t)k;
    const IsoD Z = iso_load(d.ep_z + 12 * (int64_t)k);
    const IsoD Xi = iso_load(pose + 12 * (int64_t)vi), Xj = iso_load(pose + 12 * (int64_t)vj);
    edge_se3_dev(Z, Xi, Xj, e, BUILD ? Ji : nullptr, BUILD ? Jj : nullptr);
    const double chi = chi2_6(e, info);
    double rho0, rho1;
    huber(chi, d.huber_ep, d.dsqr_ep, rho0, rho1);
    ep_chi[k] = chi; ep_chi[n + k] = rho0;
    if (BUILD) {
      double r[6];
      for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += info[i * 6 + j] * e[j]; r[i] = -s * rho1; }
      jtwj6(Ji, info, rho1, Ji, Hm);
      for (int i = 0; i < 36; ++i) atomicAdd(d.Hpp + 36 * (int64_t)vi + i, Hm[i]);
      jtwj6(Jj, info, rho1, Jj, Hm);
      for (int i = 0; i < 36; ++i) atomicAdd(d.Hpp + 36 * (int64_t)vj + i, Hm[i]);
      jtwj6(Ji, info, rho1, Jj, Hm);
      for (int i = 0; i < 36; ++i) d.Hpp_ep[36 * (int64_t)k + i] = Hm[i];
      for (int a = 0; a < 6; ++a) {
        double si = 0, sj = 0;
        for (int i = 0; i < 6; ++i) { si += Ji[i * 6 + a] * r[i]; sj += Jj[i * 6 + a] * r[i]; }
        atomicAdd(d.bp + 6 * (int64_t)vi + a, si);
        atomicAdd(d.bp + 6 * (int64_t)vj + a, sj);
      }
    }
  } else {
    const int q = k - d.Ep;
    const int v = d.pr_pose[q];
    const double* info = d.pr_info + 36 * (int64_t)q;
    const IsoD Z = iso_load(d.pr_z + 12 * (int64_t)q);
    const IsoD X = iso_load(pose + 12 * (int64_t)v);
    edge_prior_dev(Z, X, e, BUILD ? Ji : nullptr);
    const double chi = chi2_6(e, info);
    ep_chi[k] = chi; ep_chi[n + k] = chi;    // no robust kernel on the prior
    if (BUILD) {
      double r[6];
      for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += info[i * 6 + j] * e[j]; r[i] = -s; }
      jtwj6(Ji, info, 1.0, Ji, Hm);
      for (int i = 0; i < 36; ++i) atomicAdd(d.Hpp + 36 * (int64_t)v + i, Hm[i]);
      for (int a = 0; a < 6; ++a) {
        double si = 0;
        for (int i = 0; i < 6; ++i) si += Ji[i * 6 + a] * r[i];
        atomicAdd(d.bp + 6 * (int64_t)v + a, si);
      }
    }
  }
}

// Fixed-order final reduction of the chi2 partials: chunk partials then pose-pose edges.
__global__ __launch_bounds__(256) void k_reduce_chi(BADev d, const double* ep_chi) {
  __shared__ double lds[4 * 2];
  const int nchunks = d.n_chunks_b + d.n_chunks_t, n2 = d.Ep + d.Npr;
  double acc[2] = {0, 0};
  for (int i = threadIdx.x; i < nchunks; i += blockDim.x) { acc[0] += d.chunk_chi[i]; acc[1] += d.chunk_chi[(nchunks + 1) + i]; }
  for (int i = threadIdx.x; i < n2; i += blockDim.x) { acc[0] += ep_chi[i]; acc[1] += ep_chi[n2 + i]; }
  block_sum<2>(acc, lds);
  if (threadIdx.x == 0) { d.scal[S_CHI2] = lds[0]; d.scal[S_RCHI2] = lds[1]; }
}

// ---------------------------------------------------------------------------- launchers
static double* ep_chi_buf(const BADev& d) { return d.chunk_chi + 2 * (int64_t)(d.n_chunks_b + d.n_chunks_t + 1); }

void launch_errors(const BADev& d, int which, hipStream_t s) {
  if (d.n_chunks_b) hipLaunchKernelGGL(k_sweep_eb<false>, dim3(d.n_chunks_b), dim3(VDO_SWEEP_THREADS), 0, s, d, which);
  if (d.n_chunks_t) hipLaunchKernelGGL(k_sweep_et<false>, dim3(d.n_chunks_t), dim3(VDO_SWEEP_THREADS), 0, s, d, which);
  const int n2 = d.Ep + d.Npr;
  if (n2) hipLaunchKernelGGL(k_posepose<false>, dim3((n2 + 63) / 64), dim3(64), 0, s, d, which, ep_chi_buf(d));
  hipLaunchKernelGGL(k_reduce_chi, dim3(1), dim3(256), 0, s, d, ep_chi_buf(d));
}

void launch_sweep_eb_only(const BADev& d, hipStream_t s) {
  if (d.n_chunks_b) hipLaunchKernelGGL(k_sweep_eb<true>, dim3(d.n_chunks_b), dim3(VDO_SWEEP_THREADS), 0, s, d, 0);
}

void launch_linearize(const BADev& d, hipStream_t s) {
  hipMemsetAsync(d.Hll, 0, sizeof(double) * 9 * (size_t)d.L, s);
  hipMemsetAsync(d.bl, 0, sizeof(double) * 3 * (size_t)d.L, s);
  if (d.n_chunks_b) hipLaunchKernelGGL(k_sweep_eb<true>, dim3(d.n_chunks_b), dim3(VDO_SWEEP_THREADS), 0, s, d, 0);
  if (d.n_chunks_t) hipLaunchKernelGGL(k_sweep_et<true>, dim3(d.n_chunks_t), dim3(VDO_SWEEP_THREADS), 0, s, d, 0);
  hipLaunchKernelGGL(k_finalize_pose, dim3((d.P + 127) / 128), dim3(128), 0, s, d);
  if (d.L) hipLaunchKernelGGL(k_mirror_points, dim3((d.L + 255) / 256), dim3(256), 0, s, d);
  const int n2 = d.Ep + d.Npr;
  if (n2) hipLaunchKernelGGL(k_posepose<true>, dim3((n2 + 63) / 64), dim3(64), 0, s, d, 0, ep_chi_buf(d));
  hipLaunchKernelGGL(k_reduce_chi, dim3(1), dim3(256), 0, s, d, ep_chi_buf(d));
}

}  // namespace vdo
