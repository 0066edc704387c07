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

#include <cstdlib>

#include "ba_dev.hpp"

#ifndef VDO_BA_DENSE_DEFAULT
#define VDO_BA_DENSE_DEFAULT 4
#endif

namespace vdo {

constexpr int NB = 64;
typedef double d4 __attribute__((ext_vector_type(4)));

// -DDENSE_PROF (tools/dense_check.hip, debug build only): shader-clock cycles per phase of the workgroup of k_chol_step that factorises the next
// diagonal block - the critical path of a step
#ifdef DENSE_PROF
struct DProf { long long t[16]; long long prev; };
__device__ long long g_dense_prof[64][16];
#define DP_ARG , DProf& pr
#define DP_PASS , pr
#define DP_TICK(slot) do { if (threadIdx.x == 0) { const long long t_ = clock64(); pr.t[slot] += t_ - pr.prev; pr.prev = t_; } } while (0)
#else
#define DP_ARG
#define DP_PASS
#define DP_TICK(slot) do { } while (0)
#endif

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

// The same in two halves - request (16 values per thread into registers), commit (-> LDS): the blocks a kernel needs are requested together and
// arrive in ONE round trip; stage64's loop waits for every value before it asks for the next (16 trips of ~700 cycles each from L2: the phase
// probe of k_chol_step showed 21 k cycles of its 101 k there).
__device__ __forceinline__ void stage64_request(const double* __restrict__ G, int64_t ld, double (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) { const int i = threadIdx.x + 256 * q; v[q] = G[(int64_t)(i >> 6) * ld + (i & 63)]; }
}
__device__ __forceinline__ void stage64_commit(const double (&v)[16], double* Ls) {
#pragma unroll
  for (int q = 0; q < 16; ++q) { const int i = threadIdx.x + 256 * q; Ls[(i >> 6) * LDS_LD + (i & 63)] = v[q]; }
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
// (sol: where workgroup 0 puts the solved block - NOT x itself, which the other workgroups of the launch are still reading block kb of)
template <bool FWD>
__global__ __launch_bounds__(256) void k_trsv_step(const double* __restrict__ S, int64_t ld, int kb, const double* __restrict__ Winv, double* __restrict__ x,
                                                   double* __restrict__ sol) {
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
    if (threadIdx.x < NB) sol[kb * NB + threadIdx.x] = xk[threadIdx.x];
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


// ---------------------------------------------------------------------------------------------------------------- version 2
// One launch per 64-column step (instead of potrf + panel + syrk), the forward substitution folded into it:
//   k_chol_step(k)  workgroup (i, j), k < j <= i: stages the raw A_ik, A_jk and W_k = L_kk^-1, forms L_ik = A_ik W_k^T and L_jk itself
//                   (two 64^3 products, redundant across the row - they cost less than a launch boundary and a trip through HBM), then
//                   A_ij -= L_ik L_jk^T.  The workgroups of the first trailing column (j = k + 1) store the panel block - into the UPPER
//                   triangle, block (k, i), which nothing of the factorisation reads: the raw A_ik stays in place for the others of the
//                   launch - and advance the right-hand side: y_k = W_k b_k, b_i -= L_ik y_k.  Workgroup (k+1, k+1) keeps its updated
//                   block in LDS and factorises it on the spot (potrf64_lds): L, W of step k + 1 are ready when the launch ends.
//   potrf64_lds     the sub-panel, the trailing update and the assembly of L^-1 from the four 16x16 inverses on v_mfma_f64_16x16x4_f64
//                   (16x16 tiles, one per wave); the 16x16 diagonal factor multiplies by 1 / sqrt(pivot) (v_rsq_f64 + two Newton steps)
//                   instead of dividing 136 times per block.
//   backward        x = L^-T y right-looking, one launch per block as before, reading L from the upper triangle.
template <typename F>
__device__ __forceinline__ void for_acc(F f) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int r = 0; r < 4; ++r) f((lane >> 4) + 4 * r, lane & 15, r);      // (row, column, register) of a 16x16 f64 MFMA accumulator
}

// 16x16 tile  D[i][j] = sum_{k < K} a(i, k) b(j, k),  a(i, k) = Ap[i * sai + k * sak],  b(j, k) = Bp[j * sbj + k * sbk]   (one wave; K a multiple of 4)
__device__ __forceinline__ d4 mma16(const double* Ap, int sai, int sak, const double* Bp, int sbj, int sbk, int K) {
  const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  d4 acc = d4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const double a = Ap[i * sai + (k0 + kq) * sak];
    const double b = Bp[i * sbj + (k0 + kq) * sbk];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// 1 / sqrt(p), p > 0: hardware estimate + two Newton steps  y <- y + y (1/2 - (p y) (y / 2))   (quadratic: ~2^-23 -> 2^-45 -> rounding)
__device__ __forceinline__ double rsqrt_nr(double p) {
  double y = __builtin_amdgcn_rsq(p);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double h = 0.5 * y;
    const double e = __builtin_fma(-(p * y), h, 0.5);
    y = __builtin_fma(y, e, y);
  }
  return y;
}

// potrf16_regs with one reciprocal square root per column and multiplications everywhere else
__device__ __forceinline__ bool potrf16_regs_v2(double (&a)[16], double (&w)[16], int r) {
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 16; ++c) w[c] = 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double p = bcast(a[j], j);
    if (!(p > 0.0)) { bad = true; p = 1.0; }
    const double inv = rsqrt_nr(p);
    const double l = (r == j) ? p * inv : a[j] * inv;
    a[j] = l;
    if (r == j) {
#pragma unroll
      for (int c = 0; c <= j; ++c) w[c] = ((c == j ? 1.0 : 0.0) + w[c]) * inv;       // row j of L^-1 is final
    }
#pragma unroll
    for (int c = j + 1; c < 16; ++c) a[c] -= l * bcast(l, c);
#pragma unroll
    for (int c = 0; c <= j; ++c) { const double wj = bcast(w[c], j); if (r > j) w[c] -= l * wj; }
  }
  return bad;
}

// A (LDS, 64 x LDS_LD, SPD in its lower triangle) -> L in its lower triangle, W = L^-1 (LDS, lower; upper zero); Tm: scratch of the same size.
// Called by all 256 threads; *s_bad (LDS, initialised by the caller before its last barrier) is set when a pivot is not positive.
__device__ void potrf64_lds(double* A, double* W, double* Tm, int* s_bad) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < NB * LDS_LD; i += 256) W[i] = 0.0;
  __syncthreads();
  for (int kb = 0; kb < 4; ++kb) {
    const int o = 16 * kb;
    if (wv == 0) {
      double a[16], w[16];
      const int r = lane & 15;
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = A[(o + r) * LDS_LD + o + c];
      const bool bad = potrf16_regs_v2(a, w, r);
      if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { A[(o + r) * LDS_LD + o + c] = c <= r ? a[c] : 0.0; W[(o + r) * LDS_LD + o + c] = c <= r ? w[c] : 0.0; }
      }
      if (__ballot(bad && lane < 16) && lane == 0) *s_bad = 1;
    }
    __syncthreads();
    const int nrt = 3 - kb;                       // 16-row tiles below the diagonal sub-block
    if (wv < nrt) {                               // sub-panel X = A_sub Wd^T, in place (a wave reads and writes only its own tile)
      const int R0 = o + 16 + 16 * wv;
      const d4 x = mma16(A + R0 * LDS_LD + o, LDS_LD, 1, W + o * LDS_LD + o, LDS_LD, 1, 16);
      for_acc([&](int rr, int cc, int q) { A[(R0 + rr) * LDS_LD + o + cc] = x[q]; });
    }
    __syncthreads();
    const int ntile = nrt * (nrt + 1) / 2;        // trailing update of the lower triangle, tile (ti, tj), tj <= ti:  A -= X_ti X_tj^T
    for (int q = wv; q < ntile; q += 4) {
      int tj = q, ti = 0;
      while (tj > ti) { tj -= ti + 1; ++ti; }
      const int R0 = o + 16 + 16 * ti, C0 = o + 16 + 16 * tj;
      const d4 u = mma16(A + R0 * LDS_LD + o, LDS_LD, 1, A + C0 * LDS_LD + o, LDS_LD, 1, 16);
      for_acc([&](int rr, int cc, int qq) { A[(R0 + rr) * LDS_LD + C0 + cc] -= u[qq]; });
    }
    __syncthreads();
  }
  // L^-1: W_ij = -W_ii * sum_{t = j}^{i - 1} L_it W_tj  for the 16x16 blocks i > j, by block distance
  for (int dist = 1; dist < 4; ++dist) {
    const int nbk = 4 - dist;
    const int j = wv, i = wv + dist;
    if (wv < nbk) {
      const d4 t = mma16(A + 16 * i * LDS_LD + 16 * j, LDS_LD, 1, W + 16 * j * LDS_LD + 16 * j, 1, LDS_LD, 16 * dist);
      for_acc([&](int rr, int cc, int q) { Tm[(16 * i + rr) * LDS_LD + 16 * j + cc] = t[q]; });
    }
    __syncthreads();
    if (wv < nbk) {
      const d4 v = mma16(W + 16 * i * LDS_LD + 16 * i, LDS_LD, 1, Tm + 16 * i * LDS_LD + 16 * j, 1, LDS_LD, 16);
      for_acc([&](int rr, int cc, int q) { W[(16 * i + rr) * LDS_LD + 16 * j + cc] = -v[q]; });
    }
    __syncthreads();
  }
}

// The same, with the 16-column panels factorised whole in the registers of wave 0 - lane r holds row o + r of the panel, so the sub-panel
// X = A_sub Ld^-T falls out of the column loop that produces Ld (its broadcasts are wave-wide anyway) and no inverse is needed inside
// the loop; the four 16x16 inverses are formed afterwards, one per wave, by forward substitution on the unit vectors (lane c: column c;
// the entries of L are LDS broadcasts).  dinv: LDS, 64 doubles (reciprocal pivots).
__device__ void potrf64_lds_v3(double* A, double* W, double* Tm, double* dinv, int* s_bad DP_ARG) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < NB * LDS_LD; i += 256) W[i] = 0.0;
  __syncthreads();
  DP_TICK(5);
  for (int kb = 0; kb < 4; ++kb) {
    const int o = 16 * kb;
    if (wv == 0) {
      const bool active = o + lane < NB;
      const int row = active ? o + lane : NB - 1;
      double a[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = A[row * LDS_LD + o + c];
      bool bad = false;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        double p = bcast(a[j], j);                  // (lane j holds row o + j)
        if (!(p > 0.0)) { bad = true; p = 1.0; }
        const double inv = rsqrt_nr(p);
        const double l = (lane == j) ? p * inv : a[j] * inv;
        a[j] = l;
        if (lane == j) dinv[o + j] = inv;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] = __builtin_fma(-l, bcast(l, c), a[c]);
      }
      if (active) {
#pragma unroll
        for (int c = 0; c < 16; ++c) A[row * LDS_LD + o + c] = (lane < 16 && c > lane) ? 0.0 : a[c];
      }
      if (bad && lane == 0) *s_bad = 1;             // (p is wave-uniform)
    }
    __syncthreads();
    DP_TICK(6);
    const int nrt = 3 - kb;
    const int ntile = nrt * (nrt + 1) / 2;          // trailing update of the lower triangle, tile (ti, tj), tj <= ti:  A -= X_ti X_tj^T
    for (int q = wv; q < ntile; q += 4) {
      int tj = q, ti = 0;
      while (tj > ti) { tj -= ti + 1; ++ti; }
      const int R0 = o + 16 + 16 * ti, C0 = o + 16 + 16 * tj;
      const d4 u = mma16(A + R0 * LDS_LD + o, LDS_LD, 1, A + C0 * LDS_LD + o, LDS_LD, 1, 16);
      for_acc([&](int rr, int cc, int qq) { A[(R0 + rr) * LDS_LD + C0 + cc] -= u[qq]; });
    }
    __syncthreads();
    DP_TICK(7);
  }
  {   // Wd of block wv: column c of Ld^-1 on lane c (lanes >= 16 repeat column c & 15 and do not store)
    const int o = 16 * wv, c = lane & 15;
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double sacc = (r == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) sacc -= A[(o + r) * LDS_LD + o + k] * x[k];
      x[r] = sacc * dinv[o + r];
    }
    if (lane < 16) {
#pragma unroll
      for (int r = 0; r < 16; ++r) W[(o + r) * LDS_LD + o + c] = x[r];
    }
  }
  __syncthreads();
  DP_TICK(8);
  for (int dist = 1; dist < 4; ++dist) {
    const int nbk = 4 - dist;
    const int j = wv, i = wv + dist;
    if (wv < nbk) {
      const d4 t = mma16(A + 16 * i * LDS_LD + 16 * j, LDS_LD, 1, W + 16 * j * LDS_LD + 16 * j, 1, LDS_LD, 16 * dist);
      for_acc([&](int rr, int cc, int q) { Tm[(16 * i + rr) * LDS_LD + 16 * j + cc] = t[q]; });
    }
    __syncthreads();
    if (wv < nbk) {
      const d4 v = mma16(W + 16 * i * LDS_LD + 16 * i, LDS_LD, 1, Tm + 16 * i * LDS_LD + 16 * j, 1, LDS_LD, 16);
      for_acc([&](int rr, int cc, int q) { W[(16 * i + rr) * LDS_LD + 16 * j + cc] = -v[q]; });
    }
    __syncthreads();
  }
  DP_TICK(9);
}

__device__ __forceinline__ void store_factor(const double* A, const double* W, double* __restrict__ G, int64_t ld, double* __restrict__ Wk) {
  for (int i = threadIdx.x; i < NB * NB; i += 256) {
    const int rr = i >> 6, c = i & 63;
    G[(int64_t)rr * ld + c] = c <= rr ? A[rr * LDS_LD + c] : 0.0;
    Wk[i] = c <= rr ? W[rr * LDS_LD + c] : 0.0;
  }
}

// diagonal block 0 (the others are factorised inside k_chol_step)
__global__ __launch_bounds__(256) void k_potrf64_v2(double* __restrict__ S, int64_t ld, int k, double* __restrict__ Winv, int32_t* __restrict__ flags, int variant) {
  __shared__ double A[NB * LDS_LD];
  __shared__ double W[NB * LDS_LD];
  __shared__ double Tm[NB * LDS_LD];
  __shared__ double dinv[NB];
  __shared__ int s_bad;
  double* G = S + (int64_t)k * NB * ld + (int64_t)k * NB;
  {
    double v[16];
    stage64_request(G, ld, v);
    stage64_commit(v, A);
  }
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
#ifdef DENSE_PROF
  DProf pr; pr.prev = clock64();
  for (int i = 0; i < 16; ++i) pr.t[i] = 0;
#endif
  if (variant == 3) potrf64_lds_v3(A, W, Tm, dinv, &s_bad DP_PASS);
  else potrf64_lds(A, W, Tm, &s_bad);
  store_factor(A, W, G, ld, Winv + (int64_t)k * NB * NB);
  if (threadIdx.x == 0 && s_bad) atomicOr(flags, 1);
}

__global__ __launch_bounds__(256) void k_chol_step(double* __restrict__ S, int64_t ld, int k, double* __restrict__ Winv, double* __restrict__ rhs,
                                                   double* __restrict__ yv, int32_t* __restrict__ flags, int variant, int npairs) {
  __shared__ double As[NB * LDS_LD];
  __shared__ double Bs[NB * LDS_LD];
  __shared__ double Ws[NB * LDS_LD];
  __shared__ double bk[NB], yk[NB];
  __shared__ double part[4][NB];
  __shared__ int s_bad;
  const int tid = threadIdx.x, wv = tid >> 6;
  // Workgroup npairs (one past the pairs) is the AUXILIARY of block row k + 1: it forms L_{k+1,k} as workgroup (k+1, k+1) does, stores it and
  // advances the right-hand side of that row - so that the workgroup every other launch waits for goes from its products straight into the
  // factorisation (the phase probe: 4.8 k of its 73 k cycles were this substitution).
  const bool aux = (int)blockIdx.x == npairs;
  int t = aux ? 0 : blockIdx.x, ri = 0;             // pair index -> (bi, bj), k < bj <= bi
  while (t > ri) { t -= ri + 1; ++ri; }
  const int bi = k + 1 + ri, bj = k + 1 + t;
  const bool diag = bi == bj, next = diag && bj == k + 1 && !aux;     // next: block (k+1, k+1), factorised below
  const bool first = bj == k + 1 && !next;          // stores its panel block and substitutes its block row
#ifdef DENSE_PROF
  DProf pr; pr.prev = clock64();
  for (int i = 0; i < 16; ++i) pr.t[i] = 0;
  const long long wall0 = wall_clock64(), clk0 = pr.prev;
#endif
  double* G = S + (int64_t)bi * NB * ld + (int64_t)bj * NB;
  d4 gv[4];                                         // this thread's 16 entries of the block it updates: requested now, used after the three products
  {
    double va[16], vb[16], vw[16];
    stage64_request(S + (int64_t)bi * NB * ld + (int64_t)k * NB, ld, va);
    stage64_request(S + (int64_t)bj * NB * ld + (int64_t)k * NB, ld, vb);           // (the same block again in a diagonal workgroup: not committed)
    stage64_request(Winv + (int64_t)k * NB * NB, NB, vw);
    const double bkv = (first && tid < NB) ? rhs[k * NB + tid] : 0.0;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) for_acc([&](int rr, int cc, int q) { gv[tt][q] = G[(int64_t)(16 * wv + rr) * ld + 16 * tt + cc]; });
    stage64_commit(va, As);
    if (!diag) stage64_commit(vb, Bs);
    stage64_commit(vw, Ws);
    if (first && tid < NB) bk[tid] = bkv;
  }
  if (tid == 0) s_bad = 0;
  __syncthreads();
  DP_TICK(0);
  d4 li[4], lj[4];
  gemm64_abt(As, Ws, li);                           // L_ik = A_ik W_k^T
  if (!diag) gemm64_abt(Bs, Ws, lj);
  __syncthreads();                                  // (every wave has read the raw blocks)
  DP_TICK(1);
#pragma unroll
  for (int tt = 0; tt < 4; ++tt)
    for_acc([&](int rr, int cc, int q) {
      As[(16 * wv + rr) * LDS_LD + 16 * tt + cc] = li[tt][q];
      if (!diag) Bs[(16 * wv + rr) * LDS_LD + 16 * tt + cc] = lj[tt][q];
    });
  if (first) {                                      // the panel block, kept in the upper triangle: block (k, bi) = L_{bi,k}
    double* P = S + (int64_t)k * NB * ld + (int64_t)bi * NB;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) for_acc([&](int rr, int cc, int q) { P[(int64_t)(16 * wv + rr) * ld + 16 * tt + cc] = li[tt][q]; });
  }
  __syncthreads();
  DP_TICK(2);
  if (!aux) {
    d4 acc[4];
    gemm64_abt(As, diag ? As : Bs, acc);            // L_ik L_jk^T
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
      for_acc([&](int rr, int cc, int q) {
        const double u = gv[tt][q] - acc[tt][q];
        if (next) Bs[(16 * wv + rr) * LDS_LD + 16 * tt + cc] = u;       // (Bs is free in a diagonal workgroup)
        else G[(int64_t)(16 * wv + rr) * ld + 16 * tt + cc] = u;
      });
  }
  DP_TICK(3);
  if (first) {                                      // forward substitution: y_k = W_k b_k, b_i -= L_ik y_k
    const int col = tid & 63, chunk = tid >> 6;
    {
      double a = 0.0;
      for (int c = chunk; c < NB; c += 4) a += Ws[col * LDS_LD + c] * bk[c];
      part[chunk][col] = a;
    }
    __syncthreads();
    if (tid < NB) {
      const double y = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
      yk[tid] = y;
      if (aux) yv[k * NB + tid] = y;
    }
    __syncthreads();
    {
      double a = 0.0;
      for (int c = chunk; c < NB; c += 4) a += As[col * LDS_LD + c] * yk[c];
      part[chunk][col] = a;
    }
    __syncthreads();
    if (tid < NB) rhs[bi * NB + tid] -= part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
  }
  if (next) {
    __syncthreads();
    DP_TICK(4);
    if (variant == 3) potrf64_lds_v3(Bs, Ws, As, yk, &s_bad DP_PASS);
    else potrf64_lds(Bs, Ws, As, &s_bad);
    store_factor(Bs, Ws, G, ld, Winv + (int64_t)(k + 1) * NB * NB);
    if (tid == 0 && s_bad) atomicOr(flags, 1);
    DP_TICK(10);
#ifdef DENSE_PROF
    if (tid == 0 && k < 64) {
      for (int i = 0; i < 11; ++i) g_dense_prof[k][i] = pr.t[i];
      g_dense_prof[k][11] = wall_clock64() - wall0;
      g_dense_prof[k][12] = clock64() - clk0;
    }
#endif
  }
#ifdef DENSE_PROF
  if (!next && tid == 0 && k < 64 && blockIdx.x + 2 == gridDim.x) { g_dense_prof[k][13] = clock64() - clk0; g_dense_prof[k][14] = wall_clock64() - wall0; }
#endif
}

// y of the last block (no step launch follows its factorisation)
__global__ __launch_bounds__(256) void k_fwd_last(int kb, const double* __restrict__ Winv, const double* __restrict__ rhs, double* __restrict__ yv) {
  __shared__ double bk[NB];
  __shared__ double part[4][NB];
  const int col = threadIdx.x & 63, chunk = threadIdx.x >> 6;
  if (threadIdx.x < NB) bk[threadIdx.x] = rhs[kb * NB + threadIdx.x];
  __syncthreads();
  const double* W = Winv + (int64_t)kb * NB * NB;
  double a = 0.0;
  for (int c = chunk; c < NB; c += 4) a += W[col * NB + c] * bk[c];
  part[chunk][col] = a;
  __syncthreads();
  if (threadIdx.x < NB) yv[kb * NB + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// backward step kb on the version-2 layout: L_{kb,bi} (bi < kb) lives in block (bi, kb) of S.  y: the forward-substituted vector, updated in
// place; xo: the solution (a separate vector: workgroup 0 must not overwrite what the others are still reading).
__global__ __launch_bounds__(256) void k_trsv_back_v2(const double* __restrict__ S, int64_t ld, int kb, const double* __restrict__ Winv, double* __restrict__ y,
                                                      double* __restrict__ xo) {
  __shared__ double bk[NB], xk[NB];
  __shared__ double part[4][NB];
  const int col = threadIdx.x & 63, chunk = threadIdx.x >> 6;
  if (threadIdx.x < NB) bk[threadIdx.x] = y[kb * NB + threadIdx.x];
  __syncthreads();
  {
    const double* W = Winv + (int64_t)kb * NB * NB;
    double acc = 0.0;
    for (int t = chunk; t < NB; t += 4) acc += W[t * NB + col] * bk[t];         // x_kb = W_kb^T y_kb
    part[chunk][col] = acc;
  }
  __syncthreads();
  if (threadIdx.x < NB) xk[threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
  __syncthreads();
  if (blockIdx.x == 0) {
    if (threadIdx.x < NB) xo[kb * NB + threadIdx.x] = xk[threadIdx.x];
    return;
  }
  const int bi = kb - (int)blockIdx.x;
  const double* Lb = S + (int64_t)bi * NB * ld + (int64_t)kb * NB;              // [t][col] = L_{kb,bi}[t][col]
  double acc = 0.0;
  for (int t = chunk; t < NB; t += 4) acc += Lb[(int64_t)t * ld + col] * xk[t];
  __syncthreads();
  part[chunk][col] = acc;
  __syncthreads();
  if (threadIdx.x < NB) y[bi * NB + threadIdx.x] -= part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// G block rows per launch (kb, kb-1, .., kb-G+1, all >= 0): every workgroup solves the G x G block triangle itself - all its operands
// (the G inverses, the G (G-1) / 2 coupling blocks, and its own G blocks of the update) are requested at the head of the kernel, so the G
// dependent rounds cost LDS round trips, not trips to L2 - then workgroup b >= 1 updates block row kb - G + 1 - b.
template <int G>
__global__ __launch_bounds__(256) void k_trsv_back_multi(const double* __restrict__ S, int64_t ld, int kb, const double* __restrict__ Winv, double* __restrict__ y,
                                                         double* __restrict__ xo) {
  __shared__ double xs[G][NB];
  __shared__ double vk[NB];
  __shared__ double part[4][NB];
  const int tid = threadIdx.x, col = tid & 63, chunk = tid >> 6;
  const bool upd = blockIdx.x > 0;
  const int bi = kb - G + 1 - (int)blockIdx.x;
  double wq[G][16], lq[G * (G - 1) / 2 + 1][16], lu[G][16], yq[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const double* W = Winv + (int64_t)(kb - g) * NB * NB;
#pragma unroll
    for (int i = 0; i < 16; ++i) wq[g][i] = W[(chunk + 4 * i) * NB + col];
    yq[g] = tid < NB ? y[(kb - g) * NB + tid] : 0.0;
  }
  {
    int idx = 0;
#pragma unroll
    for (int g = 1; g < G; ++g)
#pragma unroll
      for (int gp = 0; gp < g; ++gp, ++idx) {
        const double* Lb = S + (int64_t)(kb - g) * NB * ld + (int64_t)(kb - gp) * NB;        // block (kb-g, kb-gp) holds L_{kb-gp, kb-g}
#pragma unroll
        for (int i = 0; i < 16; ++i) lq[idx][i] = Lb[(int64_t)(chunk + 4 * i) * ld + col];
      }
  }
  if (upd) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const double* Lb = S + (int64_t)bi * NB * ld + (int64_t)(kb - g) * NB;
#pragma unroll
      for (int i = 0; i < 16; ++i) lu[g][i] = Lb[(int64_t)(chunk + 4 * i) * ld + col];
    }
  }
  {
    int idx = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      double acc = 0.0;
#pragma unroll
      for (int gp = 0; gp < g; ++gp, ++idx)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += lq[idx][i] * xs[gp][chunk + 4 * i];
      if (g > 0) {
        part[chunk][col] = acc;
        __syncthreads();
      }
      if (tid < NB) vk[tid] = g > 0 ? yq[g] - (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]) : yq[g];
      __syncthreads();
      acc = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += wq[g][i] * vk[chunk + 4 * i];          // x_q = W_q^T v
      part[chunk][col] = acc;
      __syncthreads();
      if (tid < NB) {
        const double v = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        xs[g][tid] = v;
        if (!upd) xo[(kb - g) * NB + tid] = v;
      }
      __syncthreads();
    }
  }
  if (upd) {
    double acc = 0.0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc += lu[g][i] * xs[g][chunk + 4 * i];
    part[chunk][col] = acc;
    __syncthreads();
    if (tid < NB) y[bi * NB + tid] -= part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
  }
}

__global__ void k_copy_xp(BADev d, const double* __restrict__ x) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < 6 * (int64_t)d.P) d.xp[i] = x[i];
}

// ---------------------------------------------------------------------------------------------------------------- small systems
// The reduced system of a 20-frame window of PartialBatchOptimization has 120 unknowns: eight launches (k_dense_init, k_dense_rhs, k_potrf64_v2, k_chol_step,
// k_fwd_last, two k_trsv_back_v2, k_copy_xp) of 4 .. 34 us each - 95 us of a 235-us Levenberg trial - for a matrix that fits the LDS of ONE workgroup.
// k_dense_small does all of it for 6P <= 128: stages the lower triangle of S (what k_schur_dense_tile left: - sum B Hll^-1 B^T), adds blockdiag(Hpp + lambda I) and the
// EdgeSE3 blocks there, appends the right-hand side bp - qs as ROW 128 of the matrix (the factorisation then leaves y = L^-1 b in that row: no forward substitution),
// factorises by 16-column panels - EVERY wave repeats the 16x16 diagonal factor in lanes 0 .. 15 (so the pivots and the columns of L are wave-wide broadcasts, no barrier
// inside a panel) and carries 48 rows of the sub-panel in lanes 16 .. 63 -, trailing update on v_mfma_f64_16x16x4_f64 tiles, then x = L^-T y by 16-row blocks -> d.xp.
constexpr int SN = 128;                 // padded order (identity on the padding)
constexpr int SLD = SN + 2;             // leading dimension in LDS: consecutive rows two doubles apart in bank space - the MFMA operand reads (16 rows x 2 consecutive k per half-wave) are conflict-free (as LDS_LD above; with SN + 1 they were 4-way: 1 050 cycles per 16x16x16 tile)
size_t dense_small_lds() { return ((size_t)(SN + 1) * SLD + SN) * sizeof(double); }

__global__ __launch_bounds__(256) void k_dense_small(BADev d, double* __restrict__ S, int64_t ld, double lambda) {
  extern __shared__ __attribute__((aligned(16))) double sm_small[];
  double* A = sm_small;                 // [SN + 1][SLD]
  double* xs = A + (SN + 1) * SLD;      // [SN]  the solution
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = 6 * d.P, np = (n + 15) / 16, npad = 16 * np;
  if (tid == 0) s_bad = 0;
#ifdef DENSE_PROF
  DProf pr; pr.prev = clock64();
  for (int i = 0; i < 16; ++i) pr.t[i] = 0;
#endif
  // the lower triangle, rows r and SN - 1 - r folded into one line of SN + 1 entries (r + 1 of the one, SN - r of the other): 64 x 129 entries, every load
  // unconditional (clamped: the padding is never read from memory) and 8 in flight per thread; S is zeroed behind the read for the next assembly
  constexpr int FOLD = (SN / 2) * (SN + 1), NLD = (FOLD + 255) / 256;      // 33 entries per thread: requested together, ONE round trip (five rounds of 8 cost 13 us of the kernel's 55)
  {
    double v[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int i = min(tid + 256 * q, FOLD - 1), pr = i / (SN + 1), k = i - pr * (SN + 1);
      const int r = k <= pr ? pr : SN - 1 - pr, c = k <= pr ? k : k - pr - 1;
      v[q] = S[(int64_t)min(r, n - 1) * ld + min(c, n - 1)];
    }
    // the pose side and the right-hand side travel with them (clamped: every thread asks)
    double hp[3], rb = 0.0, rq = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) hp[q] = d.Hpp[min(tid + 256 * q, 36 * d.P - 1)];
    if (tid < SN) { rb = d.bp[min(tid, n - 1)]; rq = d.qs[min(tid, n - 1)]; }
    double he[3];
    int er[3], ec[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {                        // EdgeSE3 blocks (ep_i, ep_j): three entries per thread cover 21 edges - a window's 19
      const int i = min(tid + 256 * q, max(36 * d.Ep - 1, 0)), e = i / 36;
      he[q] = d.Ep ? d.Hpp_ep[i] : 0.0;
      er[q] = d.Ep ? d.ep_i[e] : 0; ec[q] = d.Ep ? d.ep_j[e] : 0;
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int i = tid + 256 * q, pr = i / (SN + 1), k = i - pr * (SN + 1);
      const int r = k <= pr ? pr : SN - 1 - pr, c = k <= pr ? k : k - pr - 1;
      if (i < FOLD) A[r * SLD + c] = r < n ? v[q] : (r == c ? 1.0 : 0.0);
    }
    if (tid < SN) A[SN * SLD + tid] = tid < n ? rb - rq : 0.0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = tid + 256 * q;
      if (i < 36 * d.P) {
        const int p = i / 36, a = (i % 36) / 6, b = i % 6;
        if (b <= a) A[(6 * p + a) * SLD + 6 * p + b] += hp[q] + (a == b ? lambda : 0.0);
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {                        // the block or its transpose: whichever lies below the diagonal
      const int i = tid + 256 * q;
      if (i < 36 * d.Ep) {
        const int r = 6 * er[q] + (i % 36) / 6, c = 6 * ec[q] + i % 6;
        atomicAdd(A + (r > c ? r * SLD + c : c * SLD + r), he[q]);
      }
    }
  }
  for (int i = tid + 768; i < 36 * d.P; i += 256) {       // (more than 21 poses' worth: never at <= 128 unknowns, kept for the bound's sake)
    const int p = i / 36, a = (i % 36) / 6, b = i % 6;
    if (b <= a) A[(6 * p + a) * SLD + 6 * p + b] += d.Hpp[i] + (a == b ? lambda : 0.0);
  }
  for (int i = tid + 768; i < 36 * d.Ep; i += 256) {    // (more than 21 edges)
    const int e = i / 36, a = (i % 36) / 6, b = i % 6;
    const double v = d.Hpp_ep[i];
    const int r = 6 * d.ep_i[e] + a, c = 6 * d.ep_j[e] + b;
    atomicAdd(A + (r > c ? r * SLD + c : c * SLD + r), v);
  }
  __syncthreads();
  DP_TICK(0);
  for (int o = 0; o < npad; o += 16) {
    // rows of this lane: lanes 0..15 the diagonal block (every wave its own copy), lanes 16..63 row sub = 48 wv + lane - 16 of the sub-panel; the last one is the right-hand side
    const int nsub = npad - o - 16;                       // matrix rows below the diagonal block
    if (wv == 0 || 48 * wv <= nsub) {                     // (wave-uniform: this wave carries rows of the panel)
      const int sub = 48 * wv + lane - 16;
      const bool diag = lane < 16, act = diag ? wv == 0 : sub <= nsub;
      const int row = diag ? o + lane : (sub < nsub ? o + 16 + sub : SN);
      double a[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = A[row * SLD + o + c];
      // No test of the pivot inside the recurrence: a pivot that is not positive turns its column into NaN (v_rsq_f64 of a negative number, 0 * inf) and the NaN spreads
      // over everything behind it - the diagonal of L is looked at once, by the back-substitution.  The diagonal lane's own entry is the pivot: l = a_j / sqrt(p) for all.
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double inv = rsqrt_nr(bcast(a[j], j));
        const double l = a[j] * inv;
        a[j] = l;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] = __builtin_fma(-l, bcast(l, c), a[c]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (act) {
#pragma unroll
        for (int c = 0; c < 16; ++c) A[row * SLD + o + c] = a[c];      // (the diagonal block's upper part receives garbage: nothing reads it)
      }
    } else if (wv == 3) {
      // wave 3 never carries rows of a panel (<= 113 of them, 48 per wave): it zeroes S for the next assembly meanwhile, 16 rows per panel - 16-byte stores, a row per
      // step (n = 6P is even, rows are 1 KB apart)
      for (int r = o; r < min(o + 16, n); ++r)
        for (int c2 = 2 * lane; c2 < n; c2 += 128) *reinterpret_cast<double2*>(S + (int64_t)r * ld + c2) = double2{0.0, 0.0};
    }
    DP_TICK(2);
    __syncthreads();
    DP_TICK(3);
    if (tid < nsub) {                                    // the right-hand side's row: b_c -= sum_k X_b[k] X_c[k]
      const int c = o + 16 + tid;
      double acc = A[SN * SLD + c];
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = __builtin_fma(-A[SN * SLD + o + k], A[c * SLD + o + k], acc);
      A[SN * SLD + c] = acc;
    }
    DP_TICK(7);
    // trailing update, 16x16 tiles (ti, tj), tj <= ti, round-robin over the waves: C -= X_ti X_tj^T as ONE accumulation chain D = (-X_ti) X_tj^T + C.  The eight operand
    // values and the four C values of a lane are requested together (mma16's loop paid an LDS round trip per MFMA: 1 050 cycles per tile), two tiles per trip.
    const int nrt = nsub / 16, ntile = nrt * (nrt + 1) / 2;
    {
      const int i16 = lane & 15, kq = lane >> 4;
      for (int q0 = wv; q0 < ntile; q0 += 8) {
        int R0[2], C0[2];
        bool on[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = q0 + 4 * h;
          on[h] = q < ntile;
          int tj = on[h] ? q : 0, ti = 0;
          while (tj > ti) { tj -= ti + 1; ++ti; }
          R0[h] = o + 16 + 16 * ti; C0[h] = o + 16 + 16 * tj;
        }
        double av[2][4], bv[2][4];
        d4 acc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { av[h][k] = A[(R0[h] + i16) * SLD + o + 4 * k + kq]; bv[h][k] = A[(C0[h] + i16) * SLD + o + 4 * k + kq]; }
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[h][r] = A[(R0[h] + kq + 4 * r) * SLD + C0[h] + i16];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[h][k], bv[h][k], acc[h], 0, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (on[h]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) A[(R0[h] + kq + 4 * r) * SLD + C0[h] + i16] = acc[h][r];
          }
        }
      }
    }
    DP_TICK(4);
    __syncthreads();
    DP_TICK(5);
  }
  // x = L^-T y, y = row SN: 16-row blocks from the end; wave 0 solves the block in registers (column j of the block's L^T is row ob + j of L: requested before the
  // recurrence starts; a lane's y is final when its own step comes: x = y / L_jj afterwards), everybody takes the block out of the rows above.  The factorisation failed
  // (g2o: "Cholesky failure") when an entry of L's diagonal is not a positive finite number.
  for (int ob = npad - 16; ob >= 0; ob -= 16) {
    if (wv == 0) {
      const int l15 = lane & 15;
      double yv = A[SN * SLD + ob + l15];
      const double ljj = A[(ob + l15) * SLD + ob + l15];
      double Lc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) Lc[j] = A[(ob + j) * SLD + ob + l15];
      const double dv = 1.0 / ljj;
      if (__ballot(!(ljj > 0.0 && ljj < __builtin_inf())) != 0ull && lane == 0) s_bad = 1;
#pragma unroll
      for (int j = 15; j >= 1; --j) {
        const double xj = bcast(yv * dv, j);
        yv = lane < j ? __builtin_fma(-Lc[j], xj, yv) : yv;
      }
      if (lane < 16) xs[ob + lane] = yv * dv;
    }
    __syncthreads();
    if (tid < ob) {
      double acc = A[SN * SLD + tid];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc = __builtin_fma(-A[(ob + c) * SLD + tid], xs[ob + c], acc);
      A[SN * SLD + tid] = acc;
    }
    __syncthreads();
  }
  DP_TICK(6);
  if (tid < n) d.xp[tid] = xs[tid];
  if (tid == 0 && s_bad) atomicOr(d.flags, 1);
#ifdef DENSE_PROF
  if (tid == 0) for (int i = 0; i < 8; ++i) g_dense_prof[63][i] = pr.t[i];
#endif
}

void launch_dense_small(const BADev& d, double* S, int64_t ld, double lambda, hipStream_t s) {
  hipLaunchKernelGGL(k_dense_small, dim3(1), dim3(256), raise_lds(k_dense_small, dense_small_lds()), s, d, S, ld, lambda);
}

// VDO_BA_DENSE selects: 1 = the round-2 launch sequence (potrf + panel + syrk per step, two substitution sweeps); 2 = one launch per step
// (above), potrf64_lds on the diagonal blocks; 3 = the same with potrf64_lds_v3; 4 = 3 + three block rows per launch of the backward
// substitution; 5 = 2 + the same.
static int dense_version() {
  const char* e = std::getenv("VDO_BA_DENSE");
  return e && e[0] >= '1' && e[0] <= '5' ? e[0] - '0' : VDO_BA_DENSE_DEFAULT;
}

// factor S (ld x ld, ld a multiple of 64) and solve S x = rhs -> d.xp.  d.flags[0] reports a failed factorisation.  rhs: [2][ld] (the second
// half receives the forward-substituted vector of version 2).
void launch_dense_solve(const BADev& d, double* S, int64_t ld, double* Winv, double* rhs, hipStream_t s) {
  const int nblk = (int)(ld / NB);
  const int ver = dense_version();
  if (ver >= 2) {
    const int potrf = (ver == 2 || ver == 5) ? 2 : 3;            // routine of the diagonal blocks
    const bool multi = ver >= 4;                                  // three block rows per launch of the backward substitution
    double* yv = rhs + ld;
    hipLaunchKernelGGL(k_potrf64_v2, dim3(1), dim3(256), 0, s, S, ld, 0, Winv, d.flags, potrf);
    for (int k = 0; k + 1 < nblk; ++k) {
      const int m = nblk - k - 1;
      hipLaunchKernelGGL(k_chol_step, dim3(m * (m + 1) / 2 + 1), dim3(256), 0, s, S, ld, k, Winv, rhs, yv, d.flags, potrf, m * (m + 1) / 2);
    }
    hipLaunchKernelGGL(k_fwd_last, dim3(1), dim3(256), 0, s, nblk - 1, (const double*)Winv, (const double*)rhs, yv);
    int kb = nblk - 1;                                            // (b is dead by now: the solution goes where it was)
    if (multi) for (; kb >= 2; kb -= 3) hipLaunchKernelGGL(k_trsv_back_multi<3>, dim3(kb - 1), dim3(256), 0, s, (const double*)S, ld, kb, (const double*)Winv, yv, rhs);
    for (; kb >= 0; --kb) hipLaunchKernelGGL(k_trsv_back_v2, dim3(kb + 1), dim3(256), 0, s, (const double*)S, ld, kb, (const double*)Winv, yv, rhs);
    hipLaunchKernelGGL(k_copy_xp, dim3((unsigned)((6 * (int64_t)d.P + 255) / 256)), dim3(256), 0, s, d, (const double*)rhs);
    return;
  }
  for (int k = 0; k < nblk; ++k) {
    hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(256), 0, s, S, ld, k, Winv, d.flags);
    const int m = nblk - k - 1;
    if (m > 0) {
      hipLaunchKernelGGL(k_panel, dim3(m), dim3(256), 0, s, S, ld, k, (const double*)Winv);
      hipLaunchKernelGGL(k_syrk, dim3(m * (m + 1) / 2), dim3(256), 0, s, S, ld, k, nblk);
    }
  }
  double* yv = rhs + ld;      // forward: b (updated in place) -> y; backward: y (updated in place) -> x where b was
  for (int kb = 0; kb < nblk; ++kb) hipLaunchKernelGGL(k_trsv_step<true>, dim3(nblk - kb), dim3(256), 0, s, (const double*)S, ld, kb, (const double*)Winv, rhs, yv);
  for (int kb = nblk - 1; kb >= 0; --kb) hipLaunchKernelGGL(k_trsv_step<false>, dim3(kb + 1), dim3(256), 0, s, (const double*)S, ld, kb, (const double*)Winv, yv, rhs);
  hipLaunchKernelGGL(k_copy_xp, dim3((unsigned)((6 * (int64_t)d.P + 255) / 256)), dim3(256), 0, s, d, (const double*)rhs);
}

}  // namespace vdo
