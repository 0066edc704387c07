// Stand-alone checks of lm_dev.hpp: block_reduce_bfly against exact integer sums, ldlt6_solve_lanes against ldlt6_solve_perm (bit for bit).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/lm_dev_check.hip -o tools/lm_dev_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include "../vdo_slam_amd/csrc/lm_dev.hpp"
using namespace vdo;
template <int K>
__global__ void k(const double* in, double* out) {
  __shared__ double part[F2_WAVES * 32], red[32];
  double acc[K];
  for (int i = 0; i < K; ++i) acc[i] = in[(size_t)threadIdx.x * 32 + i];
  block_reduce_bfly<K>(acc, part, red);
  if (threadIdx.x < K) out[threadIdx.x] = red[threadIdx.x];
}
template <int K>
int run() {
  std::vector<double> h(F2_THREADS * 32), ref(K, 0.0), got(K);
  for (int t = 0; t < F2_THREADS; ++t) for (int i = 0; i < 32; ++i) { h[t * 32 + i] = (double)((t * 131 + i * 7919) % 1009) + 1000.0 * i; if (i < K) ref[i] += h[t * 32 + i]; }
  double *d_in, *d_out;
  hipMalloc(&d_in, h.size() * 8); hipMalloc(&d_out, K * 8);
  hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k<K>, dim3(1), dim3(F2_THREADS), 0, 0, d_in, d_out);
  hipMemcpy(got.data(), d_out, K * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < K; ++i) if (got[i] != ref[i]) { if (bad < 5) printf("K=%d q=%d got %.1f want %.1f\n", K, i, got[i], ref[i]); ++bad; }
  printf("K=%d: %s\n", K, bad ? "MISMATCH" : "ok");
  return bad;
}

__global__ void k_ldlt(const double* in, double* out, int n) {
  __shared__ double A[36], b[6], x[6], x2[6];
  __shared__ int okv[2];
  for (int c = 0; c < n; ++c) {
    if (threadIdx.x < 36) A[threadIdx.x] = in[(size_t)c * 42 + threadIdx.x];
    if (threadIdx.x < 6) { b[threadIdx.x] = in[(size_t)c * 42 + 36 + threadIdx.x]; x[threadIdx.x] = -7.0; x2[threadIdx.x] = -7.0; }
    __syncthreads();
    if (threadIdx.x == 0) okv[0] = ldlt6_solve_perm(A, b, x) ? 1 : 0;
    if (threadIdx.x >= 64 && threadIdx.x < 128) { const bool ok = ldlt6_solve_lanes(A, b, x2); if (threadIdx.x == 64) okv[1] = ok ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x < 6) { out[(size_t)c * 14 + threadIdx.x] = x[threadIdx.x]; out[(size_t)c * 14 + 6 + threadIdx.x] = x2[threadIdx.x]; }
    if (threadIdx.x < 2) out[(size_t)c * 14 + 12 + threadIdx.x] = okv[threadIdx.x];
    __syncthreads();
  }
}
int run_ldlt() {
  const int n = 20000;
  std::vector<double> h((size_t)n * 42), got((size_t)n * 14);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
  for (int c = 0; c < n; ++c) {
    double J[8][6], *A = &h[(size_t)c * 42];
    const int kind = c % 10;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 6; ++j) J[i][j] = rnd() * (kind == 3 ? (j + 1) * 100.0 : 1.0);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { double a = 0; for (int q = 0; q < 8; ++q) a += J[q][i] * J[q][j]; A[i * 6 + j] = a; }
    if (kind == 1) for (int i = 0; i < 36; ++i) A[i] = 0.0;                       // zero matrix
    if (kind == 2) for (int i = 0; i < 36; ++i) A[i] = -A[i];                      // negative definite
    if (kind == 4) { A[7] = A[0]; A[14] = A[0]; }                                  // ties on the diagonal
    if (kind == 5) for (int i = 0; i < 6; ++i) { A[2 * 6 + i] = 0; A[i * 6 + 2] = 0; }   // a zero row / column
    if (kind == 6) A[21] = -A[21];                                                 // indefinite
    if (kind == 7) for (int i = 0; i < 36; ++i) A[i] *= 1e-300;                    // tiny pivots
    for (int j = 0; j < 6; ++j) A[36 + j] = rnd();
  }
  double *d_in, *d_out;
  hipMalloc(&d_in, h.size() * 8); hipMalloc(&d_out, got.size() * 8);
  hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_ldlt, dim3(1), dim3(F2_THREADS), 0, 0, d_in, d_out, n);
  hipMemcpy(got.data(), d_out, got.size() * 8, hipMemcpyDeviceToHost);
  int bad = 0, npos = 0;
  for (int c = 0; c < n; ++c) {
    const double* g = &got[(size_t)c * 14];
    bool same = g[12] == g[13];
    if (g[12] != 0) { ++npos; for (int i = 0; i < 6; ++i) if (std::memcmp(&g[i], &g[6 + i], 8) != 0) same = false; }
    if (!same) { if (bad < 5) printf("ldlt case %d (kind %d): ok %g/%g x %.17g/%.17g\n", c, c % 10, g[12], g[13], g[0], g[6]); ++bad; }
  }
  printf("ldlt lanes vs serial: %d cases, %d positive, %s\n", n, npos, bad ? "MISMATCH" : "ok");
  return bad;
}
int main() { return (run<29>() + run<27>() + run<32>() + run<1>() + run_ldlt()) ? 1 : 0; }
