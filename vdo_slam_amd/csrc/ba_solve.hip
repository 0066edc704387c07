// Linear solve of one Levenberg trial, (H + lambda I) x = b, for the batch graph on gfx950.
// The reference hands the whole (poses+motions+points) system to CSparse Cholesky
// (g2o/solvers/linear_solver_csparse.h:108-144; no Schur).  Here the landmark
// part is eliminated first — block-diagonal for static points, block-tridiagonal along each
// dynamic track (the ternary edge couples consecutive observations, src/Optimizer.cc:1704-1741)
// — and the reduced pose/motion system S = Hpp - Hpl Hll^-1 Hlp is solved matrix-free with
// block-Jacobi preconditioned conjugate gradients.  x is identical to the direct solve up to
// the PCG tolerance.
#include "ba_dev.hpp"
#include "se3_dev.hpp"

namespace vdo {

__device__ __forceinline__ double wave_sum2(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// block-wide sum for up to 1024 threads; result broadcast to all threads
__device__ double block_sum1(double v, double* lds /*[17]*/) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum2(v);
  __syncthreads();
  if (lane == 0) lds[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int w = 0; w < nw; ++w) s += lds[w];
    lds[16] = s;
  }
  __syncthreads();
  return lds[16];
}
__device__ double block_max1(double v, double* lds) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  __syncthreads();
  if (lane == 0) lds[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int w = 0; w < nw; ++w) s = fmax(s, lds[w]);
    lds[16] = s;
  }
  __syncthreads();
  return lds[16];
}

// computeLambdaInit (g2o/core/optimization_algorithm_levenberg.cpp:166-180): max |H(j,j)|
__global__ __launch_bounds__(1024) void k_max_diag(BADev d) {
  __shared__ double lds[17];
  double m = 0;
  for (int64_t i = threadIdx.x; i < 6 * (int64_t)d.P; i += blockDim.x) {
    const int64_t p = i / 6, j = i % 6;
    m = fmax(m, fabs(d.Hpp[36 * p + 7 * j]));
  }
  for (int64_t i = threadIdx.x; i < 3 * (int64_t)d.L; i += blockDim.x) {
    const int64_t l = i / 3, j = i % 3;
    m = fmax(m, fabs(d.Hll[9 * l + 4 * j]));
  }
  m = block_max1(m, lds);
  if (threadIdx.x == 0) d.scal[S_MAXDIAG] = m;
}

// 3x3 SPD inverse with positive-definiteness test (leading minors).
__device__ __forceinline__ bool spd3_inv(const double* a, double* o) {
  const double det = sym3_inv(a, o);
  const double m2 = a[0] * a[4] - a[1] * a[3];
  return (a[0] > 0) && (m2 > 0) && (det > 0);
}

// Block LDL^T along every landmark chain: Delta_k = D_k + lambda I - O^T Delta_{k-1}^-1 O.
__global__ void k_factor_chains(BADev d, double lambda) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.n_chains) return;
  const int off = d.chain_off[c], m = d.chain_off[c + 1] - off;
  double prev[9];
  bool ok = true;
  const int64_t Et = d.Et;
  for (int k = 0; k < m; ++k) {
    const int64_t l = d.chain_pt[off + k];
    double D[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = d.Hll[9 * l + i];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    if (k > 0) {
      const int64_t e = d.chain_edge[off + k - 1];
      double O[9], G[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) O[i] = d.Oll[i * Et + e];
      mat3_mul(prev, O, G);
#pragma unroll
      for (int i = 0; i < 9; ++i) d.Gl[9 * l + i] = G[i];
      // D -= O^T G
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) D[3 * i + j] -= O[i] * G[j] + O[3 + i] * G[3 + j] + O[6 + i] * G[6 + j];
    }
    ok &= spd3_inv(D, prev);
#pragma unroll
    for (int i = 0; i < 9; ++i) d.Dinv[9 * l + i] = prev[i];
  }
  if (!ok) atomicOr(d.flags, 1);
}

// w = Hll(lambda)^-1 u along every chain.  sign/addb: u_eff = addb ? (bl - u) : u ; u is zeroed after use.
__global__ void k_chain_solve(BADev d, const double* bl_or_null, double* u, double* w) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.n_chains) return;
  const int off = d.chain_off[c], m = d.chain_off[c + 1] - off;
  D3 yprev{0, 0, 0};
  for (int k = 0; k < m; ++k) {
    const int64_t l = d.chain_pt[off + k];
    D3 y{u[3 * l], u[3 * l + 1], u[3 * l + 2]};
    u[3 * l] = 0; u[3 * l + 1] = 0; u[3 * l + 2] = 0;
    if (bl_or_null) y = D3{bl_or_null[3 * l], bl_or_null[3 * l + 1], bl_or_null[3 * l + 2]} - y;
    if (k > 0) y = y - rotT(d.Gl + 9 * l, yprev);   // y_k = u_k - G_k^T y_{k-1}
    yprev = y;
    const D3 z = rot(d.Dinv + 9 * l, y);
    w[3 * l] = z.x; w[3 * l + 1] = z.y; w[3 * l + 2] = z.z;
  }
  D3 wnext{0, 0, 0};
  int64_t lnext = -1;
  for (int k = m - 1; k >= 0; --k) {
    const int64_t l = d.chain_pt[off + k];
    D3 z{w[3 * l], w[3 * l + 1], w[3 * l + 2]};
    if (lnext >= 0) {                               // w_k = z_k - G_{k+1} w_{k+1}
      z = z - rot(d.Gl + 9 * lnext, wnext);
      w[3 * l] = z.x; w[3 * l + 1] = z.y; w[3 * l + 2] = z.z;
    }
    wnext = z; lnext = l;
  }
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
  // invert L (lower) in place -> Li, then Ainv = Li^T Li
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

// Block-Jacobi preconditioner: M_i = (Hpp_ii + lambda I)^-1
__global__ void k_precond(BADev d, double lambda) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.P) return;
  double A[36], Ai[36];
  for (int i = 0; i < 36; ++i) A[i] = d.Hpp[36 * (int64_t)p + i];
  for (int i = 0; i < 6; ++i) A[7 * i] += lambda;
  if (!spd6_inv(A, Ai)) atomicOr(d.flags, 1);
  for (int i = 0; i < 36; ++i) d.Minv[36 * (int64_t)p + i] = Ai[i];
}

// pass A: u_l += B_inc^T v_pose  (one workgroup per incidence chunk; pose vector is uniform)
__global__ __launch_bounds__(VDO_SWEEP_THREADS) void k_pass_a(BADev d, const double* v, double* u) {
  const Chunk c = d.chunks_inc[blockIdx.x];
  double pv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) pv[i] = v[6 * (int64_t)c.pose + i];
  const int64_t N = d.Ninc;
  for (int e = c.begin + (int)threadIdx.x; e < c.end; e += VDO_SWEEP_THREADS) {
    const int64_t pt = d.inc_point[e];
    const double* B = d.Binc + e;
    double t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      t0 += B[(3 * r + 0) * N] * pv[r];
      t1 += B[(3 * r + 1) * N] * pv[r];
      t2 += B[(3 * r + 2) * N] * pv[r];
    }
    atomicAdd(u + 3 * pt, t0); atomicAdd(u + 3 * pt + 1, t1); atomicAdd(u + 3 * pt + 2, t2);
  }
}

// pass C: chunk_q[.,chunk] = sum_inc B_inc w_point
__global__ __launch_bounds__(VDO_SWEEP_THREADS) void k_pass_c(BADev d, const double* w) {
  __shared__ double lds[4 * 6];
  const Chunk c = d.chunks_inc[blockIdx.x];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const int64_t N = d.Ninc;
  for (int e = c.begin + (int)threadIdx.x; e < c.end; e += VDO_SWEEP_THREADS) {
    const int64_t pt = d.inc_point[e];
    const double w0 = w[3 * pt], w1 = w[3 * pt + 1], w2 = w[3 * pt + 2];
    const double* B = d.Binc + e;
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[r] += B[(3 * r) * N] * w0 + B[(3 * r + 1) * N] * w1 + B[(3 * r + 2) * N] * w2;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double s = wave_sum2(acc[i]);
    if (lane == 0) lds[wv * 6 + i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 6)
    d.chunk_q[threadIdx.x * (int64_t)d.n_chunks_inc + blockIdx.x] =
        lds[threadIdx.x] + lds[6 + threadIdx.x] + lds[12 + threadIdx.x] + lds[18 + threadIdx.x];
}

// qs[pose] = sum over the pose's chunks of chunk_q  (fixed order)
__global__ void k_gather_q(BADev d, double* qs) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.P) return;
  double s[6] = {0, 0, 0, 0, 0, 0};
  const int64_t nc = d.n_chunks_inc;
  for (int k = d.pc_off[p]; k < d.pc_off[p + 1]; ++k) {
    const int c = d.pc_idx[k];
    if (c < d.n_chunks_b) {
#pragma unroll
      for (int i = 0; i < 6; ++i) s[i] += d.chunk_q[i * nc + c];
    } else {
      const int c1 = c, c2 = c + d.n_chunks_t;
#pragma unroll
      for (int i = 0; i < 6; ++i) s[i] += d.chunk_q[i * nc + c1] + d.chunk_q[i * nc + c2];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) qs[6 * (int64_t)p + i] = s[i];
}

// L2-coherent load: the value may have been updated by atomics (performed at L2) after this
// CU cached the line in its vector L1.
__device__ __forceinline__ double ld_l2(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out += Hpp_offdiag * v for the EdgeSE3 blocks (single workgroup)
__device__ void offdiag_mv(const BADev& d, const double* v, double* out) {
  for (int k = threadIdx.x; k < d.Ep; k += blockDim.x) {
    const int64_t vi = d.ep_i[k], vj = d.ep_j[k];
    const double* Hm = d.Hpp_ep + 36 * (int64_t)k;
    for (int a = 0; a < 6; ++a) {
      double si = 0, sj = 0;
      for (int b = 0; b < 6; ++b) { si += Hm[a * 6 + b] * v[6 * vj + b]; sj += Hm[b * 6 + a] * v[6 * vi + b]; }
      atomicAdd(out + 6 * vi + a, si);
      atomicAdd(out + 6 * vj + a, sj);
    }
  }
}

// bs = bp - qs ; x = 0 ; r = bs ; z = Minv r ; p = z ; rz = rz0 = r.z       (single workgroup)
__global__ __launch_bounds__(1024) void k_pcg_init(BADev d, const double* qs) {
  __shared__ double lds[17];
  double acc = 0;
  for (int p = threadIdx.x; p < d.P; p += blockDim.x) {
    double r[6];
    for (int i = 0; i < 6; ++i) {
      r[i] = d.bp[6 * (int64_t)p + i] - qs[6 * (int64_t)p + i];
      d.bs[6 * (int64_t)p + i] = r[i];
      d.rp[6 * (int64_t)p + i] = r[i];
      d.xp[6 * (int64_t)p + i] = 0;
    }
    const double* Mi = d.Minv + 36 * (int64_t)p;
    for (int i = 0; i < 6; ++i) {
      double z = 0;
      for (int j = 0; j < 6; ++j) z += Mi[i * 6 + j] * r[j];
      d.zp[6 * (int64_t)p + i] = z;
      d.pp[6 * (int64_t)p + i] = z;
      acc += r[i] * z;
    }
  }
  acc = block_sum1(acc, lds);
  if (threadIdx.x == 0) { d.scal[S_RZ] = acc; d.scal[S_RZ0] = acc; d.scal[S_RZNEW] = acc; d.flags[1] = (acc <= 0) ? 1 : 0; }
}

// One CG update given qs = Hpl Hll^-1 Hlp p (Schur part).                     (single workgroup)
__global__ __launch_bounds__(1024) void k_pcg_vec(BADev d, const double* qs, double lambda, double tol2) {
  __shared__ double lds[17];
  if (d.flags[1]) return;                       // converged earlier: no-op
  // q = (Hpp_diag + lambda I) p - qs
  for (int p = threadIdx.x; p < d.P; p += blockDim.x) {
    const double* Hm = d.Hpp + 36 * (int64_t)p;
    const double* pv = d.pp + 6 * (int64_t)p;
    for (int i = 0; i < 6; ++i) {
      double s = lambda * pv[i] - qs[6 * (int64_t)p + i];
      for (int j = 0; j < 6; ++j) s += Hm[i * 6 + j] * pv[j];
      d.qp[6 * (int64_t)p + i] = s;
    }
  }
  __syncthreads();
  offdiag_mv(d, d.pp, d.qp);
  __syncthreads();
  double acc = 0;
  for (int64_t i = threadIdx.x; i < 6 * (int64_t)d.P; i += blockDim.x) acc += d.pp[i] * ld_l2(d.qp + i);
  const double pq = block_sum1(acc, lds);
  const double rz = d.scal[S_RZ];
  const double alpha = rz / pq;
  acc = 0;
  for (int p = threadIdx.x; p < d.P; p += blockDim.x) {
    double r[6];
    for (int i = 0; i < 6; ++i) {
      const int64_t idx = 6 * (int64_t)p + i;
      d.xp[idx] += alpha * d.pp[idx];
      r[i] = d.rp[idx] - alpha * ld_l2(d.qp + idx);
      d.rp[idx] = r[i];
    }
    const double* Mi = d.Minv + 36 * (int64_t)p;
    for (int i = 0; i < 6; ++i) {
      double z = 0;
      for (int j = 0; j < 6; ++j) z += Mi[i * 6 + j] * r[j];
      d.zp[6 * (int64_t)p + i] = z;
      acc += r[i] * z;
    }
  }
  const double rznew = block_sum1(acc, lds);
  const double beta = rznew / rz;
  for (int64_t i = threadIdx.x; i < 6 * (int64_t)d.P; i += blockDim.x) d.pp[i] = d.zp[i] + beta * d.pp[i];
  if (threadIdx.x == 0) {
    d.scal[S_RZ] = rznew;
    d.scal[S_RZNEW] = rznew;
    d.scal[S_PQ] = pq;
    int f = 0;
    if (!(pq > 0) || !(rznew == rznew)) f = 2;          // breakdown
    else if (rznew <= tol2 * d.scal[S_RZ0]) f = 1;      // converged
    d.flags[1] = f;
    d.flags[2] += 1;
  }
}

// trial estimate = current (+) x ; scale = sum x (lambda x + b)   (computeScale, levenberg.cpp:182-189)
__global__ __launch_bounds__(1024) void k_update(BADev d, double lambda, int ortho) {
  __shared__ double lds[17];
  double acc = 0;
  for (int p = threadIdx.x; p < d.P; p += blockDim.x) {
    const double* x = d.xp + 6 * (int64_t)p;
    for (int i = 0; i < 6; ++i) acc += x[i] * (lambda * x[i] + d.bp[6 * (int64_t)p + i]);
    const IsoD X = iso_load(d.pose[0] + 12 * (int64_t)p);
    iso_store(d.pose[1] + 12 * (int64_t)p, iso_oplus(X, x, ortho != 0));
  }
  for (int64_t i = threadIdx.x; i < 3 * (int64_t)d.L; i += blockDim.x) {
    const double x = d.xl[i];
    acc += x * (lambda * x + d.bl[i]);
    d.point[1][i] = d.point[0][i] + x;
  }
  acc = block_sum1(acc, lds);
  if (threadIdx.x == 0) d.scal[S_SCALE] = acc;
}

// ------------------------------------------------------------------------------ launchers
void launch_max_diag(const BADev& d, hipStream_t s) { hipLaunchKernelGGL(k_max_diag, dim3(1), dim3(1024), 0, s, d); }

void launch_factor(const BADev& d, double lambda, hipStream_t s) {
  hipMemsetAsync(d.flags, 0, 4 * sizeof(int32_t), s);
  if (d.n_chains) hipLaunchKernelGGL(k_factor_chains, dim3((d.n_chains + 127) / 128), dim3(128), 0, s, d, lambda);
  hipLaunchKernelGGL(k_precond, dim3((d.P + 63) / 64), dim3(64), 0, s, d, lambda);
}

// qs (in d.zp as scratch? no: dedicated) — we reuse d.qp as the "qs" buffer before PCG starts.
void launch_reduced_rhs(const BADev& d, hipStream_t s) {
  // w = Hll^-1 bl ; qs = Hpl w
  hipMemsetAsync(d.ul, 0, sizeof(double) * 3 * (size_t)d.L, s);
  if (d.n_chains) hipLaunchKernelGGL(k_chain_solve, dim3((d.n_chains + 127) / 128), dim3(128), 0, s, d, (const double*)d.bl, d.ul, d.wl);
  if (d.n_chunks_inc) hipLaunchKernelGGL(k_pass_c, dim3(d.n_chunks_inc), dim3(VDO_SWEEP_THREADS), 0, s, d, (const double*)d.wl);
  hipLaunchKernelGGL(k_gather_q, dim3((d.P + 127) / 128), dim3(128), 0, s, d, d.bs);
}

void launch_pcg_init(const BADev& d, hipStream_t s) {
  hipLaunchKernelGGL(k_pcg_init, dim3(1), dim3(1024), 0, s, d, (const double*)d.bs);
}

void launch_pcg_iter_tol(const BADev& d, double lambda, double tol2, double* qs, hipStream_t s) {
  if (d.n_chunks_inc) hipLaunchKernelGGL(k_pass_a, dim3(d.n_chunks_inc), dim3(VDO_SWEEP_THREADS), 0, s, d, (const double*)d.pp, d.ul);
  if (d.n_chains) hipLaunchKernelGGL(k_chain_solve, dim3((d.n_chains + 127) / 128), dim3(128), 0, s, d, (const double*)nullptr, d.ul, d.wl);
  if (d.n_chunks_inc) hipLaunchKernelGGL(k_pass_c, dim3(d.n_chunks_inc), dim3(VDO_SWEEP_THREADS), 0, s, d, (const double*)d.wl);
  hipLaunchKernelGGL(k_gather_q, dim3((d.P + 127) / 128), dim3(128), 0, s, d, qs);
  hipLaunchKernelGGL(k_pcg_vec, dim3(1), dim3(1024), 0, s, d, (const double*)qs, lambda, tol2);
}

void launch_backsub_update(const BADev& d, double lambda, bool ortho, hipStream_t s) {
  // x_l = Hll^-1 (bl - Hlp x_p)
  if (d.n_chunks_inc) hipLaunchKernelGGL(k_pass_a, dim3(d.n_chunks_inc), dim3(VDO_SWEEP_THREADS), 0, s, d, (const double*)d.xp, d.ul);
  if (d.n_chains) hipLaunchKernelGGL(k_chain_solve, dim3((d.n_chains + 127) / 128), dim3(128), 0, s, d, (const double*)d.bl, d.ul, d.xl);
  hipLaunchKernelGGL(k_update, dim3(1), dim3(1024), 0, s, d, lambda, ortho ? 1 : 0);
}

}  // namespace vdo
