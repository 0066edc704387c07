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

// (raise_lds reports a refused LDS size through the library's error slot: the stand-alone build has none)
namespace vdo { int set_error(int code, const char* fmt, ...) { fprintf(stderr, "set_error(%d): %s\n", code, fmt); return code; } }

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
    if (argc > 3) break;                                  // (a third argument: the k_dense_small section only)
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
  // ---- k_dense_small (<= 128 unknowns in one workgroup): S + blockdiag(Hpp + lambda I), right-hand side bp - qs, against the long-double Cholesky
  for (int n0 : {18, 36, 90, 120, 126}) {
    for (int indef = 0; indef < 2; ++indef) {
      if (indef && n0 != 120) continue;
      const int n = 128, P = n0 / 6;
      std::vector<double> A, b, xref;
      make_system(n, n0, A, b, indef != 0);
      // split: S = A - blockdiag(H), Hpp = H - lambda I with H the 6x6 diagonal blocks of 0.5 A; bp = 2 b, qs = b
      const double lambda = 0.37;
      std::vector<double> S(A), Hpp(36 * (size_t)P), bp(n0), qs(n0);
      for (int p = 0; p < P; ++p)
        for (int a = 0; a < 6; ++a)
          for (int c = 0; c < 6; ++c) {
            const double h = 0.5 * A[(size_t)(6 * p + a) * n + 6 * p + c];
            S[(size_t)(6 * p + a) * n + 6 * p + c] -= h;
            Hpp[36 * (size_t)p + 6 * a + c] = h - (a == c ? lambda : 0.0);
          }
      for (int i = 0; i < n0; ++i) { bp[i] = 2.0 * b[i]; qs[i] = b[i]; }
      const bool spd = host_solve(n, A, b, xref);
      double *dS, *dH, *dbp, *dqs, *dx;
      int32_t* dflags;
      hipMalloc(&dS, sizeof(double) * (size_t)n * n); hipMalloc(&dH, sizeof(double) * 36 * P); hipMalloc(&dbp, sizeof(double) * n0); hipMalloc(&dqs, sizeof(double) * n0);
      hipMalloc(&dx, sizeof(double) * n); hipMalloc(&dflags, 16);
      hipMemcpy(dH, Hpp.data(), sizeof(double) * 36 * P, hipMemcpyHostToDevice);
      hipMemcpy(dbp, bp.data(), sizeof(double) * n0, hipMemcpyHostToDevice); hipMemcpy(dqs, qs.data(), sizeof(double) * n0, hipMemcpyHostToDevice);
      BADev d;
      memset(&d, 0, sizeof d);
      d.P = P; d.Ep = 0; d.Hpp = dH; d.bp = dbp; d.qs = dqs; d.xp = dx; d.flags = dflags;
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0.f, best = 1e9f;
      for (int rep = 0; rep < 20; ++rep) {
        hipMemcpy(dS, S.data(), sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice);
        hipMemset(dflags, 0, 16); hipMemset(dx, 0, sizeof(double) * n);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        launch_dense_small(d, dS, n, lambda, 0);
        hipEventRecord(e1, 0);
        if (hipDeviceSynchronize() != hipSuccess) { printf("k_dense_small n0=%d: device error %s\n", n0, hipGetErrorString(hipGetLastError())); return 100; }
        hipEventElapsedTime(&ms, e0, e1);
        best = std::fmin(best, ms);
      }
      std::vector<double> x(n, 0.0), Sback((size_t)n * n);
      int32_t flag = 0;
      hipMemcpy(x.data(), dx, sizeof(double) * n0, hipMemcpyDeviceToHost);
      hipMemcpy(&flag, dflags, 4, hipMemcpyDeviceToHost);
      hipMemcpy(Sback.data(), dS, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost);
      bool zeroed = true;
      for (int i = 0; i < n0; ++i) for (int j = 0; j < n0; ++j) zeroed = zeroed && Sback[(size_t)i * n + j] == 0.0;
      bool ok;
      double relerr = 0.0, resid = 0.0;
      if (!spd) ok = flag != 0;
      else {
        double num = 0.0, den = 0.0;
        for (int i = 0; i < n0; ++i) { num = std::fmax(num, std::fabs(x[i] - xref[i])); den = std::fmax(den, std::fabs(xref[i])); }
        relerr = num / den;
        for (int i = 0; i < n0; ++i) { double s2 = -b[i], a = std::fabs(b[i]); for (int j = 0; j < n0; ++j) { s2 += A[(size_t)i * n + j] * x[j]; a += std::fabs(A[(size_t)i * n + j] * x[j]); } resid = std::fmax(resid, std::fabs(s2) / a); }
        ok = flag == 0 && relerr < 1e-10 && resid < 1e-12 && std::isfinite(relerr);
      }
      ok = ok && zeroed;
      printf("k_dense_small n0=%d %s: relerr %.3e resid %.3e us %.1f (best of 20) flag %d S zeroed %d %s\n", n0, spd ? "spd" : "indefinite", relerr, resid, best * 1e3, flag, (int)zeroed, ok ? "ok" : "BAD");
      if (!ok) ++nbad;
#ifdef DENSE_PROF
      {
        static long long h[64][16];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dense_prof), sizeof(h));
        const char* nm[7] = {"stage + pose side", "panel: load rows", "panel: 16 pivots + store", "barrier", "rhs row + MFMA tiles", "barrier", "back-substitution"};
        long long tot = 0;
        for (int i = 0; i < 8; ++i) tot += h[63][i];
        printf("   k_dense_small phases (shader cycles, thread 0):");
        for (int i = 0; i < 7; ++i) printf(" %s %lld,", nm[i], h[63][i]);
        printf(" (rhs row alone %lld)", h[63][7]);
        printf(" total %lld\n", tot);
      }
#endif
      hipEventDestroy(e0); hipEventDestroy(e1);
      hipFree(dS); hipFree(dH); hipFree(dbp); hipFree(dqs); hipFree(dx); hipFree(dflags);
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
