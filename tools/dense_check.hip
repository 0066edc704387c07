// Stand-alone check of the dense reduced-camera solver (vdo_slam_amd/csrc/ba_dense.hip): random SPD systems of 1 .. 34 blocks of 64 through
// the launch sequences (VDO_BA_DENSE=1: potrf + panel + syrk per step; =2: one launch per step; =3: the same, other diagonal-block routine; =4, 5: 3, 2 with three block rows per launch of the backward substitution) against a host Cholesky in long double,
// an indefinite matrix (the failure flag must rise), and microseconds per solve of either.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/dense_check.hip -o tools/dense_check
// Output: one line per case "n=<n> v<version>: relerr <e> resid <r> us <t> flag <f> ok|BAD"; exit code = number of bad cases.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../vdo_slam_amd/csrc/ba_dense.hip"

using namespace vdo;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand() {   // xorshift64*, (-1, 1)
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}

// SPD, banded-ish like a reduced-camera matrix plus a dense low-rank part: A = D + sum_k v_k v_k^T, n0 real unknowns, padded to n with the identity
static void make_system(int n, int n0, std::vector<double>& A, std::vector<double>& b, bool indefinite) {
  A.assign((size_t)n * n, 0.0);
  b.assign(n, 0.0);
  const int rank = 24;
  std::vector<double> V((size_t)rank * n0);
  for (auto& v : V) v = urand();
  for (int i = 0; i < n0; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
      for (int k = 0; k < rank; ++k) s += V[(size_t)k * n0 + i] * V[(size_t)k * n0 + j];
      if (std::abs(i - j) < 40) s += 0.3 * std::cos(0.37 * (i + j));
      if (i == j) s += 30.0 + 10.0 * (i % 7);
      A[(size_t)i * n + j] = A[(size_t)j * n + i] = s;
    }
  if (indefinite) { const int q = n0 / 2; A[(size_t)q * n + q] = -5.0; }
  for (int i = n0; i < n; ++i) A[(size_t)i * n + i] = 1.0;
  for (int i = 0; i < n0; ++i) b[i] = urand() * 10.0;
}

template <typename T>
static bool host_solve_t(int n, const std::vector<double>& A, const std::vector<double>& b, std::vector<double>& x) {
  typedef T ld_t;
  std::vector<ld_t> L((size_t)n * n, (ld_t)0);
  for (int j = 0; j < n; ++j) {
    ld_t d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(d > 0)) return false;
    const ld_t ljj = std::sqrt(d);
    L[(size_t)j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      ld_t s = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = s / ljj;
    }
  }
  std::vector<ld_t> y(n);
  for (int i = 0; i < n; ++i) { ld_t s = b[i]; for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * y[k]; y[i] = s / L[(size_t)i * n + i]; }
  x.resize(n);
  for (int i = n - 1; i >= 0; --i) { ld_t s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * (ld_t)x[k]; x[i] = (double)(s / L[(size_t)i * n + i]); }
  return true;
}
// long double up to 704 unknowns; plain double above (its own error, ~1e-13 on these matrices, is far below the 1e-10 asked of the device)
static bool host_solve(int n, const std::vector<double>& A, const std::vector<double>& b, std::vector<double>& x) {
  return n <= 1024 ? host_solve_t<long double>(n, A, b, x) : host_solve_t<double>(n, A, b, x);
}

int main(int argc, char** argv) {
  int nbad = 0;
  const int sizes[][2] = {{64, 64}, {64, 30}, {128, 128}, {192, 170}, {320, 318}, {704, 700}, {2176, 2130}};
  const int reps_big = argc > 1 ? atoi(argv[1]) : 5;
  const int only_ver = argc > 2 ? atoi(argv[2]) : 0;      // (for a kernel trace of one launch sequence)
  for (const auto& sz : sizes) {
    const int n = sz[0], n0 = sz[1];
    for (int indef = 0; indef < 2; ++indef) {
      if (indef && n != 320) continue;
      std::vector<double> A, b, xref;
      make_system(n, n0, A, b, indef != 0);
      const bool spd = host_solve(n, A, b, xref);
      double *dS, *dW, *dr, *dx;
      int32_t* dflags;
      hipMalloc(&dS, sizeof(double) * (size_t)n * n); hipMalloc(&dW, sizeof(double) * (size_t)n * 64); hipMalloc(&dr, sizeof(double) * 2 * n);
      hipMalloc(&dx, sizeof(double) * n); hipMalloc(&dflags, 16);
      BADev d;
      d.P = n0 / 6; d.xp = dx; d.flags = dflags;      // k_copy_xp moves 6 P entries
      for (int ver = 1; ver <= 5; ++ver) {
        if (only_ver && ver != only_ver) continue;
        const char vs[2] = {(char)('0' + ver), 0};
        setenv("VDO_BA_DENSE", vs, 1);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0.f;
        const int reps = n >= 2000 ? reps_big : 2;
        for (int rep = 0; rep < reps; ++rep) {
          hipMemcpy(dS, A.data(), sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice);
          hipMemcpy(dr, b.data(), sizeof(double) * n, hipMemcpyHostToDevice);
          hipMemset(dflags, 0, 16); hipMemset(dx, 0, sizeof(double) * n);
          hipDeviceSynchronize();
          hipEventRecord(e0, 0);
          launch_dense_solve(d, dS, n, dW, dr, 0);
          hipEventRecord(e1, 0);
          if (hipDeviceSynchronize() != hipSuccess) { printf("n=%d v%d: device error %s\n", n, ver, hipGetErrorString(hipGetLastError())); return 100; }
          hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<double> x(n, 0.0);
        int32_t flag = 0;
        hipMemcpy(x.data(), dx, sizeof(double) * (6 * (n0 / 6)), hipMemcpyDeviceToHost);
        hipMemcpy(&flag, dflags, 4, hipMemcpyDeviceToHost);
        bool ok;
        double relerr = 0.0, resid = 0.0;
        if (!spd) ok = flag != 0;
        else {
          const int m = 6 * (n0 / 6);
          double num = 0.0, den = 0.0;
          for (int i = 0; i < m; ++i) { num = std::fmax(num, std::fabs(x[i] - xref[i])); den = std::fmax(den, std::fabs(xref[i])); }
          relerr = num / den;
          if (m == n0) {     // residual |A x - b| / (|A| |x| + |b|) when the whole solution came back
            for (int i = 0; i < n0; ++i) { double s = -b[i], a = std::fabs(b[i]); for (int j = 0; j < n0; ++j) { s += A[(size_t)i * n + j] * x[j]; a += std::fabs(A[(size_t)i * n + j] * x[j]); } resid = std::fmax(resid, std::fabs(s) / a); }
          }
          ok = flag == 0 && relerr < 1e-10 && resid < 1e-12 && std::isfinite(relerr);
        }
        printf("n=%d n0=%d %s v%d: relerr %.3e resid %.3e us %.1f flag %d %s\n", n, n0, spd ? "spd" : "indefinite", ver, relerr, resid, ms * 1e3, flag, ok ? "ok" : "BAD");
        if (!ok) ++nbad;
        hipEventDestroy(e0); hipEventDestroy(e1);
      }
      hipFree(dS); hipFree(dW); hipFree(dr); hipFree(dx); hipFree(dflags);
    }
  }
#ifdef DENSE_PROF
  {   // phases of the critical workgroup of k_chol_step, mean over the steps of the LAST system solved (the 34-block one), shader cycles
    static long long h[64][16];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dense_prof), sizeof(h));
    const char* names[15] = {"stage", "panel products", "LDS write + panel store", "update product + G", "rhs", "potrf: zero W", "potrf: panels", "potrf: trailing",
                             "potrf: Wd", "potrf: assembly", "store", "wall ticks (100 MHz)", "cycles total", "last workgroup: cycles", "last workgroup: wall ticks"};
    for (int i = 0; i < 15; ++i) {
      double sum = 0; int cnt = 0;
      for (int k = 0; k < 33; ++k) if (i < 13 || k < 32) { sum += (double)h[k][i]; ++cnt; }
      printf("prof %-28s %10.0f\n", names[i], sum / cnt);
    }
    for (int k = 0; k < 33; k += 8) printf("prof step %d: total cycles %lld wall %lld | last wg cycles %lld wall %lld\n", k, h[k][12], h[k][11], h[k][13], h[k][14]);
  }
#endif
  printf("dense_check: %d bad\n", nbad);
  return nbad;
}
