// Device helpers shared by the per-frame Levenberg-Marquardt kernels (flow2.hip, pose_only.hip):
// SE3Quat / Eigen quaternion arithmetic in the reference's operation order
// (g2o/types/se3quat.h:41-301), the pivoted 6x6 LDLT of Eigen as LinearSolverDense uses it
// (g2o/solvers/linear_solver_dense.h:65-113), RobustKernelHuber (robust_kernel_impl.cpp:78-91)
// and the fixed-order block reduction.
#pragma once
#include <hip/hip_runtime.h>

#ifndef F2_THREADS
#define F2_THREADS 256
#endif
#define F2_WAVES (F2_THREADS / 64)

namespace vdo {

struct Q4 { double x, y, z, w; };
struct SE3d { Q4 r; double t[3]; };

__device__ __forceinline__ void q_rotate(const Q4& q, const double* v, double* o) {
  // Eigen _transformVector: v + w*uv + qv x uv, uv = 2 qv x v
  double ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
  ux += ux; uy += uy; uz += uz;
  o[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
  o[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
  o[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
template <int I>
__device__ __forceinline__ Q4 q_from_R_neg_trace(const double* m) {   // constant indices: R stays in registers
  constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
  double t = sqrt(m[4 * I] - m[4 * J] - m[4 * K] + 1.0);
  double c[3];
  c[I] = 0.5 * t; t = 0.5 / t;
  Q4 q;
  q.w = (m[3 * K + J] - m[3 * J + K]) * t;
  c[J] = (m[3 * J + I] + m[3 * I + J]) * t;
  c[K] = (m[3 * K + I] + m[3 * I + K]) * t;
  q.x = c[0]; q.y = c[1]; q.z = c[2];
  return q;
}
__device__ __forceinline__ Q4 q_from_R(const double* m) {   // Eigen Quaterniond(Matrix3d)
  Q4 q;
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t;
    q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > (i == 1 ? m[4] : m[0])) i = 2;
    q = i == 0 ? q_from_R_neg_trace<0>(m) : (i == 1 ? q_from_R_neg_trace<1>(m) : q_from_R_neg_trace<2>(m));
  }
  return q;
}
__device__ inline void q_normalize_pos(Q4& q) {   // SE3Quat::normalizeRotation
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
__device__ inline void m3mul(const double* a, const double* b, double* o) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
// x^3 with ONE rounding (x^2 and x^2*x carried as exact two-term sums): what a correctly rounded pow(x, 3) returns - the
// library pow costs several hundred cycles of a latency-bound section for the same bits.
__device__ __forceinline__ double cube_rn(double x) {
  const double p = x * x, pe = __builtin_fma(x, x, -p);
  const double h = p * x, he = __builtin_fma(p, x, -h);
  if (!(fabs(h) < 1.7976931348623157e308)) return h;      // overflow / NaN: as the plain product
  return h + (he + pe * x);
}
// sin and cos of a small angle (|x| < 0.3: every LM update): the kernel polynomials of fdlibm (k_sin.c / k_cos.c coefficients),
// no range reduction, the two Horner chains interleave; < 1 ulp like the library calls they replace.
__device__ __forceinline__ void sincos_small(double x, double& sn, double& cs) {
  const double z = x * x;
  const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  sn = x + (z * x) * (-1.66666666666666324348e-01 + z * rs);
  cs = 1.0 - (0.5 * z - z * rc);
}

// SE3Quat::exp(update) * T      (se3quat.h:229-262, :106-112)
__device__ inline SE3d se3_exp_compose(const double* u, const SE3d& T) {
  const double ox = u[0], oy = u[1], oz = u[2];
  const double theta = sqrt(ox * ox + oy * oy + oz * oz);
  const double Om[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
  double Om2[9], R[9], V[9];
  m3mul(Om, Om, Om2);
  if (theta < 0.00001) {
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i];
    for (int i = 0; i < 9; ++i) V[i] = R[i];
  } else {
    double sn, cs;
    if (theta < 0.3) sincos_small(theta, sn, cs); else { sn = sin(theta); cs = cos(theta); }
    const double a = sn / theta, b = (1 - cs) / (theta * theta), c = (theta - sn) / cube_rn(theta);
    for (int i = 0; i < 9; ++i) {
      const double id = (i % 4 == 0) ? 1.0 : 0.0;
      R[i] = (id + a * Om[i]) + b * Om2[i];
      V[i] = (id + b * Om[i]) + c * Om2[i];
    }
  }
  SE3d E;
  E.r = q_from_R(R);
  for (int i = 0; i < 3; ++i) E.t[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
  q_normalize_pos(E.r);
  // E * T
  SE3d o;
  double rt[3];
  q_rotate(E.r, T.t, rt);
  for (int i = 0; i < 3; ++i) o.t[i] = E.t[i] + rt[i];
  const Q4 &a = E.r, &b = T.r;
  o.r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  o.r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  o.r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  o.r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  q_normalize_pos(o.r);
  return o;
}

// Eigen::LDLT<MatrixXd,Lower> (unblocked, diagonal pivoting) for n=6; solve in place.  Returns isPositive().
__device__ inline bool ldlt6_solve(double* m /*36, destroyed*/, const double* b, double* x) {
  int tr[6];
  int sign = 0;
  const int n = 6;
  for (int k = 0; k < n; ++k) {
    int big = k;
    double bv = fabs(m[k * 6 + k]);
    for (int i = k + 1; i < n; ++i) if (fabs(m[i * 6 + i]) > bv) { bv = fabs(m[i * 6 + i]); big = i; }
    tr[k] = big;
    if (k != big) {
      const int s = n - big - 1;
      for (int j = 0; j < k; ++j) { const double t = m[k * 6 + j]; m[k * 6 + j] = m[big * 6 + j]; m[big * 6 + j] = t; }
      for (int i = 0; i < s; ++i) { const double t = m[(big + 1 + i) * 6 + k]; m[(big + 1 + i) * 6 + k] = m[(big + 1 + i) * 6 + big]; m[(big + 1 + i) * 6 + big] = t; }
      { const double t = m[k * 6 + k]; m[k * 6 + k] = m[big * 6 + big]; m[big * 6 + big] = t; }
      for (int i = k + 1; i < big; ++i) { const double t = m[i * 6 + k]; m[i * 6 + k] = m[big * 6 + i]; m[big * 6 + i] = t; }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      double temp[6];
      for (int j = 0; j < k; ++j) temp[j] = m[j * 6 + j] * m[k * 6 + j];
      double s = 0;
      for (int j = 0; j < k; ++j) s += m[k * 6 + j] * temp[j];
      m[k * 6 + k] -= s;
      for (int i = 0; i < rs; ++i) {
        double t = 0;
        for (int j = 0; j < k; ++j) t += m[(k + 1 + i) * 6 + j] * temp[j];
        m[(k + 1 + i) * 6 + k] -= t;
      }
    }
    const double akk = m[k * 6 + k];
    const bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) { sign = 0; for (int j = 0; j < n; ++j) tr[j] = j; break; }
    if (rs > 0 && valid) for (int i = 0; i < rs; ++i) m[(k + 1 + i) * 6 + k] /= akk;
    if (sign == 1) { if (akk < 0) sign = 2; }
    else if (sign == -1) { if (akk > 0) sign = 2; }
    else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = -1; }
  }
  if (!(sign == 1 || sign == 0)) return false;
  for (int i = 0; i < n; ++i) x[i] = b[i];
  for (int k = 0; k < n; ++k) { const double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) x[i] -= m[i * 6 + j] * x[j];
  for (int i = 0; i < n; ++i) { if (fabs(m[i * 6 + i]) > 2.2250738585072014e-308) x[i] /= m[i * 6 + i]; else x[i] = 0; }
  for (int i = n - 1; i >= 0; --i) for (int j = i + 1; j < n; ++j) x[i] -= m[j * 6 + i] * x[j];
  for (int k = n - 1; k >= 0; --k) { const double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
  return true;
}

// The same factorisation without its branches.  Eigen's unblocked LDLT is left-looking: when step k looks for its pivot, the trailing
// diagonal entries have not been touched yet - the pivot ORDER is a function of the original diagonal alone (selection with
// Eigen's swap semantics: first maximum wins, the displaced entry takes the pivot's old place).  So: find the order first,
// gather the symmetrically permuted matrix (dynamic indices: from LDS), and run an LDLT WITHOUT pivoting on it with constant
// indices in registers - the same products and sums in the same order as the pivoted version, minus its branches and swaps.
// A: full symmetric 6x6 in LDS (lower part is read), b: LDS, x: LDS (written in the original ordering).  Returns isPositive().
__device__ __forceinline__ bool ldlt6_solve_perm(const double* A, const double* b, double* x) {
  int ord[6] = {0, 1, 2, 3, 4, 5};
  {
    double d[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = fabs(A[7 * i]);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      int big = k;
      double bv = d[k];
#pragma unroll
      for (int i = k + 1; i < 6; ++i) if (d[i] > bv) { bv = d[i]; big = i; }
      // swap positions k and big (values and indices travel together)
#pragma unroll
      for (int bb = k + 1; bb < 6; ++bb)
        if (big == bb) { const double t = d[k]; d[k] = d[bb]; d[bb] = t; const int u = ord[k]; ord[k] = ord[bb]; ord[bb] = u; }
    }
  }
  double m[21];                      // lower triangle of P A P^T, row-major packed: (i, j) at i*(i+1)/2 + j
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      const int r = ord[i], c = ord[j];
      m[i * (i + 1) / 2 + j] = r >= c ? A[r * 6 + c] : A[c * 6 + r];
    }
  }
#define VDO_M(i, j) m[(i) * ((i) + 1) / 2 + (j)]
  int sign = 0;
  bool zero = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (zero) continue;
    if (k > 0) {
      double temp[6];
#pragma unroll
      for (int j = 0; j < k; ++j) temp[j] = VDO_M(j, j) * VDO_M(k, j);
      double s = 0;
#pragma unroll
      for (int j = 0; j < k; ++j) s += VDO_M(k, j) * temp[j];
      VDO_M(k, k) -= s;
#pragma unroll
      for (int i = k + 1; i < 6; ++i) {
        double t = 0;
#pragma unroll
        for (int j = 0; j < k; ++j) t += VDO_M(i, j) * temp[j];
        VDO_M(i, k) -= t;
      }
    }
    const double akk = VDO_M(k, k);
    const bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) { sign = 0; zero = true; continue; }      // the largest diagonal entry is 0: Eigen stops, the solve below yields x = 0
    if (valid) {
#pragma unroll
      for (int i = k + 1; i < 6; ++i) VDO_M(i, k) /= akk;
    }
    if (sign == 1) { if (akk < 0) sign = 2; }
    else if (sign == -1) { if (akk > 0) sign = 2; }
    else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = -1; }
  }
  if (!(sign == 1 || sign == 0)) return false;
  double y[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) y[k] = b[ord[k]];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) y[i] -= VDO_M(i, j) * y[j];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) { if (fabs(VDO_M(i, i)) > 2.2250738585072014e-308) y[i] /= VDO_M(i, i); else y[i] = 0; }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
#pragma unroll
    for (int j = i + 1; j < 6; ++j) y[i] -= VDO_M(j, i) * y[j];
  }
#undef VDO_M
#pragma unroll
  for (int k = 0; k < 6; ++k) x[ord[k]] = y[k];
  return true;
}

// The same solve spread over the lanes of ONE FULL WAVE (every lane of the wave must call it): lane i < 6 owns row i of
// P A P^T, the uniform quantities (pivot order, row k, the pivots) travel through v_readlane.  Every entry goes through the
// same products, sums and quotients in the same order as in ldlt6_solve_perm - identical bits (tools/lm_dev_check.hip compares
// them) - but a step costs one dot product, one division and one update instead of 5 - k of each: ~40 % fewer instructions
// on the serial path of an LM trial.  A, b, x: LDS as above.  Returns isPositive() (uniform).
__device__ __forceinline__ double readlane_f64(double v, const int lane) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ bool ldlt6_solve_lanes(const double* A, const double* b, double* x) {
  const int lane = threadIdx.x & 63;
  const int li = lane < 6 ? lane : 5;
  // ---- pivot order.  Without ties among the |diagonal| entries Eigen's selection (first maximum wins, the displaced entry takes
  // the pivot's old place) is the descending order: lane i counts the entries above its own, a ballot per rank turns the
  // ranks into the order.  Ties (or NaNs: ranks that are no permutation) take the step-by-step selection of ldlt6_solve_perm.
  int ord[6];
  {
    double dd[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) dd[i] = fabs(A[7 * i]);
    const double dm = fabs(A[7 * li]);
    int rank = 0, ties = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) { rank += dd[j] > dm ? 1 : 0; ties += (dd[j] == dm && j != li) ? 1 : 0; }
    bool simple = __ballot(lane < 6 && ties != 0) == 0ull;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const unsigned long long mk = __ballot(lane < 6 && rank == q);
      simple = simple && mk != 0ull;
      ord[q] = (int)__builtin_ctzll(mk | (1ull << 63));
    }
    if (!simple) {
#pragma unroll
      for (int i = 0; i < 6; ++i) ord[i] = i;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        int big = k;
        double bv = dd[k];
#pragma unroll
        for (int i = k + 1; i < 6; ++i) if (dd[i] > bv) { bv = dd[i]; big = i; }
#pragma unroll
        for (int bb = k + 1; bb < 6; ++bb)
          if (big == bb) { const double t = dd[k]; dd[k] = dd[bb]; dd[bb] = t; const int u = ord[k]; ord[k] = ord[bb]; ord[bb] = u; }
      }
    }
  }
  int r = ord[0];
#pragma unroll
  for (int i = 1; i < 6; ++i) r = li == i ? ord[i] : r;
  double m[6];                       // row li of P A P^T (columns <= li are meaningful)
#pragma unroll
  for (int j = 0; j < 6; ++j) { const int c = ord[j]; m[j] = r >= c ? A[r * 6 + c] : A[c * 6 + r]; }
  double y = b[r];
  // ---- the largest |diagonal| entry is 0 (or NaN): Eigen stops at once and solves with the untouched matrix - the rare path
  // goes to the one-lane routine
  const double a00 = readlane_f64(m[0], 0);
  if (!(fabs(a00) > 0.0)) {
    bool ok = false;
    if (lane == 0) ok = ldlt6_solve_perm(A, b, x);
    return __builtin_amdgcn_readfirstlane((int)ok) != 0;
  }
  double D[6];                       // pivots (uniform)
  double dself = 0.0;                // this lane's pivot
  bool neg = false;                  // isPositive(): no negative pivot (the sign bookkeeping of Eigen reduces to this)
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (k > 0) {
      double temp[6];
#pragma unroll
      for (int j = 0; j < k; ++j) temp[j] = D[j] * readlane_f64(m[j], k);
      double t = 0;
#pragma unroll
      for (int j = 0; j < k; ++j) t += m[j] * temp[j];
      m[k] -= t;                     // row k: the pivot; rows below: column k
    }
    const double akk = readlane_f64(m[k], k);
    D[k] = akk;
    dself = lane == k ? akk : dself;
    const bool valid = fabs(akk) > 0.0;
    const double q = m[k] / akk;
    m[k] = (valid && lane > k) ? q : m[k];
    neg = neg || akk < 0;
  }
  if (neg) return false;
#pragma unroll
  for (int j = 0; j < 5; ++j) { const double yj = readlane_f64(y, j); const double u = y - m[j] * yj; y = lane > j ? u : y; }
  y = fabs(dself) > 2.2250738585072014e-308 ? y / dself : 0.0;
  double yf[6];
  yf[5] = readlane_f64(y, 5);
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    double yi = readlane_f64(y, i);
#pragma unroll
    for (int j = i + 1; j < 6; ++j) yi -= readlane_f64(m[i], j) * yf[j];
    yf[i] = yi;
  }
  double mine = yf[0];
#pragma unroll
  for (int i = 1; i < 6; ++i) mine = li == i ? yf[i] : mine;
  if (lane < 6) x[r] = mine;
  return true;
}

__device__ __forceinline__ void inv3_dev(const double* a, double* o) {   // Eigen fixed 3x3 inverse
  const double C00 = a[4] * a[8] - a[5] * a[7];
  const double C10 = a[2] * a[7] - a[1] * a[8];
  const double C20 = a[1] * a[5] - a[2] * a[4];
  const double det = (C00 * a[0] + C10 * a[3]) + C20 * a[6];
  const double id = 1.0 / det;
  o[0] = C00 * id; o[1] = C10 * id; o[2] = C20 * id;
  o[3] = (a[5] * a[6] - a[3] * a[8]) * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  o[6] = (a[3] * a[7] - a[4] * a[6]) * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

__device__ __forceinline__ void huber_f2(double e, double delta, double dsqr, double& r0, double& r1) {
  if (e <= dsqr) { r0 = e; r1 = 1.0; }
  else { const double s = sqrt(e); r0 = 2 * s * delta - dsqr; r1 = delta / s; }
}

// block reduction of K values per thread -> out[K] in LDS (valid after return for all threads)
// Sum of K per-thread values over the workgroup (F2_THREADS threads): shuffle tree inside each wave,
// one LDS stage across the waves, fixed order.  scratch: [F2_WAVES*K], out: [K].
template <int K>
__device__ __forceinline__ void block_reduce(double (&v)[K], double* scratch, double* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    double t = v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if (lane == 0) scratch[wv * K + i] = t;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double a = 0.0;
#pragma unroll
    for (int w = 0; w < F2_WAVES; w += 4)
      a += (scratch[w * K + threadIdx.x] + scratch[(w + 1) * K + threadIdx.x]) + (scratch[(w + 2) * K + threadIdx.x] + scratch[(w + 3) * K + threadIdx.x]);
    out[threadIdx.x] = a;
  }
  __syncthreads();
}

// Wide variant for the 27/28 running sums of a linearisation: K shuffle trees cost ~20k cycles (each
// 64-bit __shfl_down is two ds_bpermute with ~100-cycle dependent latency, 6 levels deep, 28 chains);
// going through LDS is an order of magnitude cheaper: every thread stores its K values (conflict-free,
// column per quantity), 8 threads per quantity each add 32 of the 256 entries (stride-8 interleave, also
// conflict-free), and an 8-lane shuffle finishes.  wide: [K][F2_THREADS + 1] doubles of LDS.
template <int K>
__device__ __forceinline__ void block_reduce_wide(const double (&v)[K], double* wide, double* out) {
  static_assert(K * 8 <= F2_THREADS, "one 8-lane group per quantity");
  constexpr int LD = F2_THREADS + 1;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < K; ++i) wide[i * LD + tid] = v[i];
  __syncthreads();
  const int q = tid >> 3, part = tid & 7;
  double a = 0.0;
  if (q < K) {
    const double* col = wide + q * LD + part;
#pragma unroll 8
    for (int j = 0; j < F2_THREADS / 8; ++j) a += col[8 * j];
  }
  a += __shfl_down(a, 4, 8);
  a += __shfl_down(a, 2, 8);
  a += __shfl_down(a, 1, 8);
  if (q < K && part == 0) out[q] = a;
  __syncthreads();
}

// Butterfly variant (gfx950): the K (<= 32) running sums never leave the registers inside a wave.  A halving step pairs lanes
// and registers: each lane of a pair keeps half of the quantities and receives the partner's partial sums of those - after
// the six steps every lane holds the wave total of ONE quantity.  The two widest steps are single instructions on gfx950
// (v_permlane32_swap / v_permlane16_swap exchange 32- / 16-lane rows between two registers: swap + add, no selects); the
// steps inside a 16-lane row use DPP (row_ror:8, row_half_mirror, quad_perm).  ~130 VALU instructions for 32 quantities
// against 32 LDS stores + 32 dependent LDS loads per thread in block_reduce_wide; then one small LDS stage across the waves
// (added in wave order by everybody: fixed order, same bits in every workgroup of a cluster).  part: [F2_WAVES][32], out: [K].
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const long long u = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)u >> 32), CTRL, 0xF, 0xF, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
template <bool ROW32>
__device__ __forceinline__ void rows_swap_f64(double& a, double& b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  const auto lo = ROW32 ? __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false) : __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
  const auto hi = ROW32 ? __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false)
                        : __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  a = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
  b = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
// quantity whose wave total lane `lane` holds after wave_reduce32
__device__ __forceinline__ int bfly_quantity(int lane) { return ((lane >> 5) << 4) | (((lane >> 4) & 1) << 3) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) | ((lane >> 1) & 1); }
template <int K>
__device__ __forceinline__ double wave_reduce32(const double (&acc)[K]) {
  static_assert(K <= 32, "at most 32 quantities");
  const int lane = threadIdx.x & 63;
  double v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = i < K ? acc[i] : 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) { rows_swap_f64<true>(v[j], v[j + 16]); v[j] = v[j] + v[j + 16]; }    // lanes >= 32 now carry quantities 16..31
#pragma unroll
  for (int j = 0; j < 8; ++j) { rows_swap_f64<false>(v[j], v[j + 8]); v[j] = v[j] + v[j + 8]; }       // odd rows: +8
  {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const double keep = up ? v[j + 4] : v[j], send = up ? v[j] : v[j + 4]; v[j] = keep + dpp_f64<0x128>(send); }      // row_ror:8
  }
  {
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) { const double keep = up ? v[j + 2] : v[j], send = up ? v[j] : v[j + 2]; v[j] = keep + dpp_f64<0x141>(send); }      // row_half_mirror: i <-> 7 - i
  }
  {
    const bool up = (lane & 2) != 0;
    const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
    v[0] = keep + dpp_f64<0x4E>(send);                                                                                                       // quad_perm [2,3,0,1]
  }
  return v[0] + dpp_f64<0xB1>(v[0]);                                                                                                          // quad_perm [1,0,3,2]
}
template <int K>
__device__ __forceinline__ void block_reduce_bfly(const double (&acc)[K], double* part, double* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double t = wave_reduce32<K>(acc);
  if ((lane & 1) == 0) part[wv * 32 + bfly_quantity(lane)] = t;
  __syncthreads();
  if (threadIdx.x < K) {
    double a = part[threadIdx.x];
#pragma unroll
    for (int w = 1; w < F2_WAVES; ++w) a += part[w * 32 + threadIdx.x];
    out[threadIdx.x] = a;
  }
  __syncthreads();
}

// SE3Quat::to_homogeneous_matrix, row-major 4x4
__device__ inline void se3_to_matrix(const SE3d& S, double* T) {
  const Q4 q = S.r;
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  T[0] = 1 - (tyy + tzz); T[1] = txy - twz; T[2] = txz + twy; T[3] = S.t[0];
  T[4] = txy + twz; T[5] = 1 - (txx + tzz); T[6] = tyz - twx; T[7] = S.t[1];
  T[8] = txz - twy; T[9] = tyz + twx; T[10] = 1 - (txx + tyy); T[11] = S.t[2];
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

}  // namespace vdo
