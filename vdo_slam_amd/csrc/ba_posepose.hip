// Pose-pose factors of the batch graph on the GPU: EdgeSE3 (odometry, motion smoothness) and
// EdgeSE3Prior.  One thread per edge (there are only O(#frames x #objects) of them).
//   EdgeSE3 / EdgeSE3Prior      g2o/types/edge_se3.cpp:77-104, edge_se3_prior.cpp:89-102
//   computeEdgeSE3Gradient      g2o/types/isometry3d_gradients.h:191-261
//   computeEdgeSE3PriorGradient g2o/types/isometry3d_gradients.h:264-325
//   compute_dq_dR               g2o/types/dquat2mat.cpp:35-84 (entries re-derived analytically)
//   quadratic form + Huber      g2o/core/base_binary_edge.hpp:55-120, base_unary_edge.hpp:43-72
#include "ba_dev.hpp"
#include "se3_dev.hpp"

namespace vdo {

// ---------------------------------------------------------------- EdgeSE3 / EdgeSE3Prior
// d(q_xyz)/dR, dq[3][9], column index = i + 3j (column-major R); branch choice as in
// _q2m (g2o/types/dquat2mat.cpp:35-64).
template <int DM>
__device__ __forceinline__ double dq_dR_neg_trace(const double* R, double (*dq)[9]) {   // constant indices: dq stays in registers
  constexpr int j = (DM + 1) % 3, k = (DM + 2) % 3;
  const double r00 = R[0], r11 = R[4], r22 = R[8];
  double s[3] = {-1, -1, -1};
  s[DM] = 1;
  const double S = sqrt(1.0 + s[0] * r00 + s[1] * r11 + s[2] * r22) * 2;
  const double qw = (R[3 * k + j] - R[3 * j + k]) / S;
  const double qd = 0.25 * S;
  const double a = 0.25 / qd, g = 0.125 / qd, d3 = 0.03125 / (qd * qd * qd);
#pragma unroll
  for (int i = 0; i < 3; ++i) dq[DM][i + 3 * i] = s[i] * g;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    if (o == DM) continue;
    const double num = R[3 * DM + o] + R[3 * o + DM];
#pragma unroll
    for (int i = 0; i < 3; ++i) dq[o][i + 3 * i] = -s[i] * d3 * num;
    dq[o][DM + 3 * o] = a;
    dq[o][o + 3 * DM] = a;
  }
  return qw;
}
__device__ __forceinline__ void dq_dR_dev(const double* R, double (*dq)[9]) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 9; ++c) dq[a][c] = 0;
  const double r00 = R[0], r11 = R[4], r22 = R[8];
  const double tr = r00 + r11 + r22;
  double qw;
  if (tr > 0) {
    const double S = sqrt(tr + 1.0) * 2;
    qw = 0.25 * S;
    const double a = 0.25 / qw, dd = -0.03125 / (qw * qw * qw);
    constexpr int hi[3][2] = {{2, 1}, {0, 2}, {1, 0}};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = hi[k][0], j = hi[k][1];
      const double num = R[3 * i + j] - R[3 * j + i];
      dq[k][0] = dq[k][4] = dq[k][8] = num * dd;
      dq[k][i + 3 * j] = a;
      dq[k][j + 3 * i] = -a;
    }
  } else {
    if ((r00 > r11) & (r00 > r22)) qw = dq_dR_neg_trace<0>(R, dq);
    else if (r11 > r22) qw = dq_dR_neg_trace<1>(R, dq);
    else qw = dq_dR_neg_trace<2>(R, dq);
  }
  if (qw <= 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 9; ++c) dq[a][c] = -dq[a][c];
  }
}

// skew(Sx,Sy,Sz,R) with sign (isometry3d_gradients.h:57-85); S[k] row-major 3x3
__device__ __forceinline__ void skew3_dev(const double* R, double sgn, double (*S)[9]) {
  double r[9];
  #pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = sgn * 2 * R[i];
  const double Sx[9] = {0, 0, 0, -r[6], -r[7], -r[8], r[3], r[4], r[5]};
  const double Sy[9] = {r[6], r[7], r[8], 0, 0, 0, -r[0], -r[1], -r[2]};
  const double Sz[9] = {-r[3], -r[4], -r[5], r[0], r[1], r[2], 0, 0, 0};
  #pragma unroll
  for (int i = 0; i < 9; ++i) { S[0][i] = Sx[i]; S[1][i] = Sy[i]; S[2][i] = Sz[i]; }
}
// J(3..5,3..5) = dq * [vec(A Sx) vec(A Sy) vec(A Sz)]   (column-major vec)
__device__ __forceinline__ void rot_block_dev(const double (*dq)[9], const double* A, const double (*S)[9], double* J) {
  #pragma unroll
  for (int c = 0; c < 3; ++c) {
    double Pm[9];
    mat3_mul(A, S[c], Pm);
    #pragma unroll
    for (int a = 0; a < 3; ++a) {
      double s = 0;
#pragma unroll
      for (int col = 0; col < 3; ++col)
  #pragma unroll
        for (int row = 0; row < 3; ++row) s += dq[a][row + 3 * col] * Pm[3 * row + col];
      J[(3 + a) * 6 + 3 + c] = s;
    }
  }
}

__device__ __forceinline__ void edge_se3_dev(const IsoD& Z, const IsoD& Xi, const IsoD& Xj, double* e, double* Ji, double* Jj) {
  const IsoD A = iso_inv(Z);
  const IsoD B = iso_mul(iso_inv(Xi), Xj);
  const IsoD E = iso_mul(A, B);
  const D3 q = compact_quat(E.r);
  e[0] = E.t.x; e[1] = E.t.y; e[2] = E.t.z; e[3] = q.x; e[4] = q.y; e[5] = q.z;
  if (!Ji) return;
  #pragma unroll
  for (int i = 0; i < 36; ++i) Ji[i] = Jj[i] = 0;
  double dq[3][9];
  dq_dR_dev(E.r, dq);
  // Ra * skewT(tb): skewT = 2[tb]x
  const double x = 2 * B.t.x, y = 2 * B.t.y, z = 2 * B.t.z;
  const double St[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  double RaS[9];
  mat3_mul(A.r, St, RaS);
  #pragma unroll
  for (int r = 0; r < 3; ++r)
    #pragma unroll
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

__device__ __forceinline__ void edge_prior_dev(const IsoD& Z, const IsoD& X, double* e, double* J) {
  const IsoD A = iso_mul(iso_inv(Z), X);
  const D3 q = compact_quat(A.r);
  e[0] = A.t.x; e[1] = A.t.y; e[2] = A.t.z; e[3] = q.x; e[4] = q.y; e[5] = q.z;
  if (!J) return;
  #pragma unroll
  for (int i = 0; i < 36; ++i) J[i] = 0;
  double dq[3][9];
  dq_dR_dev(A.r, dq);
  #pragma unroll
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J[r * 6 + c] = A.r[3 * r + c];
  double S[3][9];
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  skew3_dev(I3, 1.0, S);
  rot_block_dev(dq, A.r, S, J);
}

__device__ __forceinline__ double chi2_6(const double* e, const double* info) {
  double s = 0;
  #pragma unroll
  for (int i = 0; i < 6; ++i) {
    double t = 0;
    #pragma unroll
    for (int j = 0; j < 6; ++j) t += info[i * 6 + j] * e[j];
    s += e[i] * t;
  }
  return s;
}

// out(6x6) = Ja^T (w * Omega) Jb
__device__ __forceinline__ void jtwj6(const double* Ja, const double* Om, double w, const double* Jb, double* out) {
  double WJ[36];
  #pragma unroll
  for (int i = 0; i < 6; ++i)
    #pragma unroll
    for (int j = 0; j < 6; ++j) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += Om[i * 6 + k] * Jb[k * 6 + j];
      WJ[i * 6 + j] = w * s;
    }
  #pragma unroll
  for (int a = 0; a < 6; ++a)
    #pragma unroll
    for (int c = 0; c < 6; ++c) {
      double s = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) s += Ja[i * 6 + a] * WJ[i * 6 + c];
      out[a * 6 + c] = s;
    }
}

// one thread per EdgeSE3 (k < Ep) or prior (k >= Ep).  ep_chi: [2][Ep+Npr] (chi2, robust chi2)
template <bool BUILD>
__global__ __launch_bounds__(64) void k_posepose(BADev d, int which, double* ep_chi, int acc) {
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
#pragma unroll
      for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += info[i * 6 + j] * e[j]; r[i] = -s * rho1; }
      jtwj6(Ji, info, rho1, Ji, Hm);
      if (acc) {
#pragma unroll
        for (int i = 0; i < 36; ++i) atomicAdd(d.Hpp + 36 * (int64_t)vi + i, Hm[i]);
      }
      jtwj6(Jj, info, rho1, Jj, Hm);
      if (acc) {
#pragma unroll
        for (int i = 0; i < 36; ++i) atomicAdd(d.Hpp + 36 * (int64_t)vj + i, Hm[i]);
      }
      jtwj6(Ji, info, rho1, Jj, Hm);
#pragma unroll
      for (int i = 0; i < 36; ++i) d.Hpp_ep[36 * (int64_t)k + i] = Hm[i];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double si = 0, sj = 0;
  #pragma unroll
        for (int i = 0; i < 6; ++i) { si += Ji[i * 6 + a] * r[i]; sj += Jj[i * 6 + a] * r[i]; }
        if (acc) { atomicAdd(d.bp + 6 * (int64_t)vi + a, si); atomicAdd(d.bp + 6 * (int64_t)vj + a, sj); }
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
#pragma unroll
      for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += info[i * 6 + j] * e[j]; r[i] = -s; }
      jtwj6(Ji, info, 1.0, Ji, Hm);
      if (acc) {
#pragma unroll
        for (int i = 0; i < 36; ++i) atomicAdd(d.Hpp + 36 * (int64_t)v + i, Hm[i]);
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double si = 0;
  #pragma unroll
        for (int i = 0; i < 6; ++i) si += Ji[i * 6 + a] * r[i];
        if (acc) atomicAdd(d.bp + 6 * (int64_t)v + a, si);
      }
    }
  }
}


void launch_posepose(const BADev& d, int which, bool build, double* ep_chi, hipStream_t s) {
  const int n2 = d.Ep + d.Npr;
  if (!n2) return;
  // Shards: the fp64 atomics below commit in a run-dependent order, so only rank 0 accumulates the
  // (replicated) pose-pose terms and the all-reduce hands every rank the same bits.
  const int acc = (!d.sharded || d.shard_rank == 0) ? 1 : 0;
  if (build) hipLaunchKernelGGL(k_posepose<true>, dim3((n2 + 63) / 64), dim3(64), 0, s, d, which, ep_chi, acc);
  else hipLaunchKernelGGL(k_posepose<false>, dim3((n2 + 63) / 64), dim3(64), 0, s, d, which, ep_chi, acc);
}

}  // namespace vdo
