// Pose-pose factors of the batch graph on the GPU: EdgeSE3 (odometry, motion smoothness) and
// EdgeSE3Prior.  One wave per edge (there are only O(#frames x #objects) of them).
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
  // computeError multiplies left to right - (Z^-1 Xi^-1) Xj, edge_se3.cpp:77-82 -, the gradient forms E = A (Xi^-1 Xj) (isometry3d_gradients.h:203-206):
  // the two differ in the last bit, and near convergence of a pure pose graph that decides a Levenberg trial (found by compiling the reference's own file:
  // oracle/_ref, tests/test_ref_g2o.py)
  {
    const IsoD Ee = iso_mul(iso_mul(A, iso_inv(Xi)), Xj);
    const D3 q = compact_quat(Ee.r);
    e[0] = Ee.t.x; e[1] = Ee.t.y; e[2] = Ee.t.z; e[3] = q.x; e[4] = q.y; e[5] = q.z;
  }
  if (!Ji) return;
  const IsoD B = iso_mul(iso_inv(Xi), Xj);
  const IsoD E = iso_mul(A, B);
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

// One WAVE per EdgeSE3 (k < Ep) or prior (k >= Ep).  ep_chi: [2][Ep+Npr] (chi2, robust chi2).
// The residual and the two 6x6 Jacobians are a long scalar computation (dq/dR, the quaternion branches): lane 0 runs it and leaves
// Ji, Jj, the information matrix and the weighted residual in LDS; the products J^T (rho' Omega) J - 3 x 432 multiply-adds when one
// thread formed them - are then taken entry by entry: lane l < 36 owns entry (l / 6, l % 6) of all three blocks, lanes 36..41 the two
// right-hand sides, each with the summation order of the one-thread routine (same bits).  Nothing is accumulated here: the blocks of
// edge k go to ep_blk[k] = Hii (36) | Hjj (36) | bi (6) | bj (6) and Hpp_ep[k] = Hij; k_finalize_pose adds them to the pose blocks
// in the fixed order of the pose's edge list (no atomics, run-independent bits).
template <bool BUILD>
__global__ __launch_bounds__(64) void k_posepose(BADev d, int which, double* ep_chi) {
  __shared__ double sJi[36], sJj[36], sOm[36], sR[6], sRho;
  const int k = blockIdx.x, lane = threadIdx.x;
  const int n = d.Ep + d.Npr;
  if (k >= n) return;
  const double* pose = d.pose[which];
  const bool is_edge = k < d.Ep;
  const double* info = is_edge ? d.ep_info + 36 * (int64_t)k : d.pr_info + 36 * (int64_t)(k - d.Ep);
  double rho1 = 1.0;
  if (lane == 0) {
    double e[6], Ji[36], Jj[36];
    if (is_edge) {
      const IsoD Z = iso_load(d.ep_z + 12 * (int64_t)k);
      const IsoD Xi = iso_load(pose + 12 * (int64_t)d.ep_i[k]), Xj = iso_load(pose + 12 * (int64_t)d.ep_j[k]);
      edge_se3_dev(Z, Xi, Xj, e, BUILD ? Ji : nullptr, BUILD ? Jj : nullptr);
    } else {
      const int q = k - d.Ep;
      edge_prior_dev(iso_load(d.pr_z + 12 * (int64_t)q), iso_load(pose + 12 * (int64_t)d.pr_pose[q]), e, BUILD ? Ji : nullptr);
    }
    const double chi = chi2_6(e, info);
    double rho0 = chi;
    if (is_edge) huber(chi, d.huber_ep, d.dsqr_ep, rho0, rho1);     // (no robust kernel on the prior)
    ep_chi[k] = chi; ep_chi[n + k] = rho0;
    if (BUILD) {
#pragma unroll
      for (int i = 0; i < 36; ++i) { sJi[i] = Ji[i]; sJj[i] = is_edge ? Jj[i] : 0.0; }
#pragma unroll
      for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += info[i * 6 + j] * e[j]; sR[i] = -s * rho1; }
      sRho = rho1;                                         // (the other lanes did not run the robust kernel)
    }
  }
  if (!BUILD) return;
  if (lane < 36) sOm[lane] = info[lane];
  __syncthreads();
  rho1 = sRho;
  double* blk = d.ep_blk + 84 * (int64_t)k;
  if (lane < 36) {
    const int a = lane / 6, c = lane - 6 * a;
    double wi[6], wj[6];                                   // column c of rho1 * Omega * Ji  /  rho1 * Omega * Jj
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double si = 0, sj = 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) { si += sOm[i * 6 + q] * sJi[q * 6 + c]; sj += sOm[i * 6 + q] * sJj[q * 6 + c]; }
      wi[i] = rho1 * si; wj[i] = rho1 * sj;
    }
    double hii = 0, hjj = 0, hij = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { hii += sJi[i * 6 + a] * wi[i]; hjj += sJj[i * 6 + a] * wj[i]; hij += sJi[i * 6 + a] * wj[i]; }
    blk[lane] = hii;
    blk[36 + lane] = hjj;
    if (is_edge) d.Hpp_ep[36 * (int64_t)k + lane] = hij;
  } else if (lane < 42) {
    const int a = lane - 36;
    double si = 0, sj = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { si += sJi[i * 6 + a] * sR[i]; sj += sJj[i * 6 + a] * sR[i]; }
    blk[72 + a] = si; blk[78 + a] = sj;
  }
}


void launch_posepose(const BADev& d, int which, bool build, double* ep_chi, hipStream_t s) {
  const int n2 = d.Ep + d.Npr;
  if (!n2) return;
  if (build) hipLaunchKernelGGL(k_posepose<true>, dim3(n2), dim3(64), 0, s, d, which, ep_chi);
  else hipLaunchKernelGGL(k_posepose<false>, dim3(n2), dim3(64), 0, s, d, which, ep_chi);
}

}  // namespace vdo
