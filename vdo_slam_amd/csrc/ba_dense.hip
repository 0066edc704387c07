// Dense solve of the reduced-camera normal equations on the MFMA units (BASELINE north_star: "MFMA used only on the small
// dense reduced-camera normal equations").  What g2o does with LinearSolverDense / CSparse on the
// Schur complement (g2o/solvers/linear_solver_dense.h:65-113, linear_solver_csparse.h:108-144): factorise S = L L^T, fail when a
// pivot is not positive, solve.  Here: blocked right-looking Cholesky, 64x64 blocks, fp64 throughout -
//   k_potrf64      diagonal block: two-level blocked Cholesky (16x16 diagonal factors + inverses in one wave's registers) and W = L_kk^-1
//   k_panel_syrk   one launch per panel step: the workgroups first form the panel L_ik = A_ik W^T (a 64x64x64 product), then - after
//                  the next launch boundary - k_syrk applies A_ij -= L_ik L_jk^T to the lower triangle of the trailing matrix.
//                  Both products are the same 64x64 = (64x64) (64x64)^T kernel body on v_mfma_f64_16x16x4_f64: four waves,
//                  each a 16-row strip of the output, operands staged through LDS.
//   k_trsv_step    forward and backward substitution with the stored W blocks, one launch per 64-row block (right-looking).
// Used when the pose graph is not a set of paths (the chain preconditioner of the PCG then degrades to block-Jacobi), when the
// PCG needs many iterations, or on request (vdo_lm_options.solver = 3).
#include <hip/hip_runtime.h>

#include "ba_dev.hpp"

namespace vdo {

constexpr int NB = 64;
typedef double d4 __attribute__((ext_vector_type(4)));

// D(64x64) = A(64x64) * B(64x64)^T for one workgroup of 256 threads (4 waves).  As, Bs: LDS copies, row-major with stride LDS_LD
// (padded: rows land on different banks).  Wave w computes rows [16w, 16w+16); acc[t] = the 16x16 tile of columns [16t, 16t+16):
// lane l holds D[16w + (l>>4) + 4*reg][16t + (l&15)]  (f64 MFMA C/D layout).
constexpr int LDS_LD = NB + 2;
__device__ __forceinline__ void gemm64_abt(const double* As, const double* Bs, d4 (&acc)[4]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int k0 = 0; k0 < NB; k0 += 4) {
    const double a = As[(16 * wv + i) * LDS_LD + k0 + kq];            // A[row i][k]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double b = Bs[(16 * t + i) * LDS_LD + k0 + kq];           // B^T[k][col] = B[col][k]
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void stage64(const double* __restrict__ G, int64_t ld, double* Ls) {
  for (int i = threadIdx.x; i < NB * NB; i += blockDim.x) { const int r = i >> 6, c = i & 63; Ls[r * LDS_LD + c] = G[r * ld + c]; }
}

// broadcast of a double from a compile-time-known lane (two scalar v_readlane_b32: no LDS round trip)
__device__ __forceinline__ double bcast(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// 16x16 Cholesky + inverse of the factor, rows in the registers of lanes 0..15 of one wave (fully unrolled: every broadcast comes
// from a lane known at compile time).  a: row r of the SPD block in, row r of L out; w: row r of L^-1 out.
__device__ __forceinline__ bool potrf16_regs(double (&a)[16], double (&w)[16], int r) {
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 16; ++c) w[c] = 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double p = bcast(a[j], j);
    if (!(p > 0.0)) { bad = true; p = 1.0; }
    const double dj = sqrt(p);
    const double l = (r == j) ? dj : a[j] / dj;
    a[j] = l;
    if (r == j) {
#pragma unroll
      for (int c = 0; c <= j; ++c) w[c] = ((c == j ? 1.0 : 0.0) + w[c]) / dj;      // row j of L^-1 is final
    }
#pragma unroll
    for (int c = j + 1; c < 16; ++c) a[c] -= l * bcast(l, c);                        // trailing update (meaningful for r >= c)
#pragma unroll
    for (int c = 0; c <= j; ++c) { const double wj = bcast(w[c], j); if (r > j) w[c] -= l * wj; }
  }
  return bad;
}

// diagonal block k: A_kk = L L^T in place (lower), W[k] = L^-1; flags[0] |= 1 when a pivot is not positive (g2o: "Cholesky failure").
// Two-level blocking inside one workgroup: 16-wide sub-panels - the 16x16 diagonal factor and its inverse in one wave's registers,
// the sub-panel solve and the trailing update by all 256 threads on the LDS copy - 12 barriers instead of the 192 of a
// column-by-column version; then L^-1 of the whole block from the four 16x16 inverses by block forward substitution.
__global__ __launch_bounds__(256) void k_potrf64(double* __restrict__ S, int64_t ld, int k, double* __restrict__ Winv, int32_t* __restrict__ flags) {
  __shared__ double A[NB * LDS_LD];
  __shared__ double W[NB * LDS_LD];
  __shared__ double Tm[NB * LDS_LD];
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double* G = S + (int64_t)k * NB * ld + (int64_t)k * NB;
  stage64(G, ld, A);
  for (int i = tid; i < NB * LDS_LD; i += 256) W[i] = 0.0;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  for (int kb = 0; kb < 4; ++kb) {
    const int o = 16 * kb;
    if (wv == 0) {
      double a[16], w[16];
      const int r = lane & 15;
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = A[(o + r) * LDS_LD + o + c];
      const bool bad = potrf16_regs(a, w, r);
      if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { A[(o + r) * LDS_LD + o + c] = c <= r ? a[c] : 0.0; W[(o + r) * LDS_LD + o + c] = c <= r ? w[c] : 0.0; }
      }
      if (__ballot(bad && lane < 16) && lane == 0) s_bad = 1;
    }
    __syncthreads();
    const int m = NB - o - 16;                  // rows below the diagonal sub-block
    // sub-panel: X = A_sub * Wd^T   (X[r][c] = sum_t A_sub[r][t] Wd[c][t], t <= c)
    for (int e = tid; e < m * 16; e += 256) {
      const int rr = o + 16 + (e >> 4), c = e & 15;
      double acc = 0.0;
      for (int t = 0; t <= c; ++t) acc += A[rr * LDS_LD + o + t] * W[(o + c) * LDS_LD + o + t];
      Tm[rr * LDS_LD + c] = acc;
    }
    __syncthreads();
    for (int e = tid; e < m * 16; e += 256) { const int rr = o + 16 + (e >> 4), c = e & 15; A[rr * LDS_LD + o + c] = Tm[rr * LDS_LD + c]; }
    __syncthreads();
    // trailing update of the lower triangle: A[rr][cc] -= sum_t X[rr][t] X[cc][t]
    for (int e = tid; e < m * m; e += 256) {
      const int i2 = e / m, j2 = e - i2 * m;     // (m is 48, 32, 16: cheap constant-ish division, 9/4/1 iterations per thread)
      if (j2 > i2) continue;
      const int rr = o + 16 + i2, cc = o + 16 + j2;
      double acc = 0.0;
#pragma unroll
      for (int t = 0; t < 16; ++t) acc += A[rr * LDS_LD + o + t] * A[cc * LDS_LD + o + t];
      A[rr * LDS_LD + cc] -= acc;
    }
    __syncthreads();
  }
  // W = L^-1 of the 64x64 block: diagonal 16x16 blocks are in W; W_ij = -W_ii * sum_{t=j}^{i-1} L_it W_tj for i > j, by block distance
  for (int dist = 1; dist < 4; ++dist) {
    const int nbk = 4 - dist;                   // blocks (i, j) = (j + dist, j), j = 0 .. nbk-1
    for (int e = tid; e < nbk * 256; e += 256) {
      const int j = e >> 8, i = j + dist, rr = (e >> 4) & 15, cc = e & 15;
      double acc = 0.0;
      for (int t = j; t < i; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += A[(16 * i + rr) * LDS_LD + 16 * t + q] * W[(16 * t + q) * LDS_LD + 16 * j + cc];
      Tm[(16 * i + rr) * LDS_LD + 16 * j + cc] = acc;
    }
    __syncthreads();
    for (int e = tid; e < nbk * 256; e += 256) {
      const int j = e >> 8, i = j + dist, rr = (e >> 4) & 15, cc = e & 15;
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += W[(16 * i + rr) * LDS_LD + 16 * i + q] * Tm[(16 * i + q) * LDS_LD + 16 * j + cc];
      W[(16 * i + rr) * LDS_LD + 16 * j + cc] = -acc;
    }
    __syncthreads();
  }
  double* Wk = Winv + (int64_t)k * NB * NB;
  for (int i = tid; i < NB * NB; i += 256) {
    const int rr = i >> 6, c = i & 63;
    G[(int64_t)rr * ld + c] = c <= rr ? A[rr * LDS_LD + c] : 0.0;
    Wk[i] = c <= rr ? W[rr * LDS_LD + c] : 0.0;
  }
  if (tid == 0 && s_bad) atomicOr(flags, 1);
}

// panel: L_ik = A_ik W_k^T for every block row i > k (blockIdx.x = i - k - 1), in place
__global__ __launch_bounds__(256) void k_panel(double* __restrict__ S, int64_t ld, int k, const double* __restrict__ Winv) {
  __shared__ double As[NB * LDS_LD];
  __shared__ double Bs[NB * LDS_LD];
  const int bi = k + 1 + blockIdx.x;
  double* G = S + (int64_t)bi * NB * ld + (int64_t)k * NB;
  stage64(G, ld, As);
  stage64(Winv + (int64_t)k * NB * NB, NB, Bs);
  __syncthreads();
  d4 acc[4];
  gemm64_abt(As, Bs, acc);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) G[(int64_t)(16 * wv + (lane >> 4) + 4 * r) * ld + 16 * t + (lane & 15)] = acc[t][r];
}

// trailing update: A_ij -= L_ik L_jk^T for k < j <= i; blockIdx.x enumerates the lower-triangular block pairs
__global__ __launch_bounds__(256) void k_syrk(double* __restrict__ S, int64_t ld, int k, int nblk) {
  __shared__ double As[NB * LDS_LD];
  __shared__ double Bs[NB * LDS_LD];
  // pair index -> (i, j), m = nblk - k - 1 trailing block rows
  int t = blockIdx.x, ri = 0;
  while (t > ri) { t -= ri + 1; ++ri; }
  const int bi = k + 1 + ri, bj = k + 1 + t;
  stage64(S + (int64_t)bi * NB * ld + (int64_t)k * NB, ld, As);
  stage64(S + (int64_t)bj * NB * ld + (int64_t)k * NB, ld, Bs);
  __syncthreads();
  d4 acc[4];
  gemm64_abt(As, Bs, acc);
  double* G = S + (int64_t)bi * NB * ld + (int64_t)bj * NB;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* g = G + (int64_t)(16 * wv + (lane >> 4) + 4 * r) * ld + 16 * tt + (lane & 15);
      *g -= acc[tt][r];
    }
}

// Substitutions, one launch per 64-row block, right-looking: workgroup 0 turns the (already updated) right-hand side block into the
// solution block x_kb = W_kb b_kb (forward) resp. W_kb^T b_kb (backward); every other workgroup recomputes x_kb for itself (4096
// multiply-adds) and subtracts its 64x64 block's product from its own part of the right-hand side.  x is updated in place.
template <bool FWD>
__global__ __launch_bounds__(256) void k_trsv_step(const double* __restrict__ S, int64_t ld, int kb, const double* __restrict__ Winv, double* __restrict__ x) {
  __shared__ double bk[NB], xk[NB];
  __shared__ double part[4][NB];
  const int col = threadIdx.x & 63, chunk = threadIdx.x >> 6;
  if (threadIdx.x < NB) bk[threadIdx.x] = x[kb * NB + threadIdx.x];
  __syncthreads();
  {
    const double* W = Winv + (int64_t)kb * NB * NB;
    double acc = 0.0;
    for (int t = chunk; t < NB; t += 4) acc += (FWD ? W[col * NB + t] : W[t * NB + col]) * bk[t];      // x[col] = sum_t W[col][t] b[t]  |  W[t][col] b[t]
    part[chunk][col] = acc;
  }
  __syncthreads();
  if (threadIdx.x < NB) xk[threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
  __syncthreads();
  if (blockIdx.x == 0) {
    if (threadIdx.x < NB) x[kb * NB + threadIdx.x] = xk[threadIdx.x];
    return;
  }
  // forward: rows of block bi = kb + blockIdx.x: b_bi -= L[bi][kb] x_kb ; backward: rows of block bi = kb - blockIdx.x: b_bi -= L[kb][bi]^T x_kb
  const int bi = FWD ? kb + (int)blockIdx.x : kb - (int)blockIdx.x;
  double acc = 0.0;
  if (FWD) {
    const double* Lb = S + (int64_t)bi * NB * ld + (int64_t)kb * NB;             // row `col`, columns chunk-strided
    for (int t = chunk; t < NB; t += 4) acc += Lb[(int64_t)col * ld + t] * xk[t];
  } else {
    const double* Lb = S + (int64_t)kb * NB * ld + (int64_t)bi * NB;             // L[kb rows][bi cols]: column `col`, rows chunk-strided (coalesced over col)
    for (int t = chunk; t < NB; t += 4) acc += Lb[(int64_t)t * ld + col] * xk[t];
  }
  __syncthreads();
  part[chunk][col] = acc;
  __syncthreads();
  if (threadIdx.x < NB) x[bi * NB + threadIdx.x] -= part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

__global__ void k_copy_xp(BADev d, const double* __restrict__ x) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < 6 * (int64_t)d.P) d.xp[i] = x[i];
}

// factor S (ld x ld, ld a multiple of 64) and solve S x = rhs -> d.xp.  d.flags[0] reports a failed factorisation.
void launch_dense_solve(const BADev& d, double* S, int64_t ld, double* Winv, double* rhs, hipStream_t s) {
  const int nblk = (int)(ld / NB);
  for (int k = 0; k < nblk; ++k) {
    hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(256), 0, s, S, ld, k, Winv, d.flags);
    const int m = nblk - k - 1;
    if (m > 0) {
      hipLaunchKernelGGL(k_panel, dim3(m), dim3(256), 0, s, S, ld, k, (const double*)Winv);
      hipLaunchKernelGGL(k_syrk, dim3(m * (m + 1) / 2), dim3(256), 0, s, S, ld, k, nblk);
    }
  }
  for (int kb = 0; kb < nblk; ++kb) hipLaunchKernelGGL(k_trsv_step<true>, dim3(nblk - kb), dim3(256), 0, s, (const double*)S, ld, kb, (const double*)Winv, rhs);
  for (int kb = nblk - 1; kb >= 0; --kb) hipLaunchKernelGGL(k_trsv_step<false>, dim3(kb + 1), dim3(256), 0, s, (const double*)S, ld, kb, (const double*)Winv, rhs);
  hipLaunchKernelGGL(k_copy_xp, dim3((unsigned)((6 * (int64_t)d.P + 255) / 256)), dim3(256), 0, s, d, (const double*)rhs);
}

}  // namespace vdo
