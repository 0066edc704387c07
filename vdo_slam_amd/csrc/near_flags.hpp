// "Is there a reference point within 1 px" for every query point — the O(n*m) test of
// Tracking::RenewFrameInfo (reference src/Tracking.cc:2727-2745 static, :2893-2907 objects).
// Float arithmetic as in the reference: sqrt((rx-qx)^2 + (ry-qy)^2) < 1 - evaluated as (rx-qx)^2 + (ry-qy)^2 < 1 on the same float sum:
// the correctly rounded square root of a float below 1 is below 1 (the largest, 1 - 2^-24, has the root 1 - 2^-25 - ..., which rounds down) and
// sqrt(1) = 1, so the two predicates agree for every input, and the quarter-rate v_sqrt_f32 leaves the 256-step inner loop.
// 2-D grid: block (bx, by) tests 256 queries against 256 references staged in LDS and ORs its verdict
// into used[] (callers zero it first): a few hundred workgroups instead of ~20 that each walk the whole
// reference set (83 -> ~6 us for 5.5 k x 3.9 k points).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdo {

static __global__ __launch_bounds__(256) void k_near_flags(int nq, const float* __restrict__ qx, const float* __restrict__ qy,
                                                           int nr, const float* __restrict__ rx, const float* __restrict__ ry, int32_t* __restrict__ used) {
  __shared__ float sx[256], sy[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int base = blockIdx.y * 256;
  const int j = base + threadIdx.x;
  sx[threadIdx.x] = j < nr ? rx[j] : 1e30f;
  sy[threadIdx.x] = j < nr ? ry[j] : 1e30f;
  __syncthreads();
  if (i >= nq) return;
  const float x = qx[i], y = qy[i];
  const int m = min(256, nr - base);
  int u = 0;
  for (int k = 0; k < m; ++k) {
    const float dx = sx[k] - x, dy = sy[k] - y;
    if (dx * dx + dy * dy < 1.0f) u = 1;
  }
  if (u) atomicOr(&used[i], 1);
}

// The same test against a SELECTION of the reference list (rsel[j] != 0), so that the selection itself can stay on the device
// (one-pass RenewFrameInfo); as_int: the reference position is the truncated one (the object keys are (int)x, (int)y).
static __global__ __launch_bounds__(256) void k_near_flags_sel(int nq, const float* __restrict__ qx, const float* __restrict__ qy,
                                                               int nr, const float* __restrict__ rx, const float* __restrict__ ry,
                                                               const int32_t* __restrict__ rsel, int as_int, int32_t* __restrict__ used) {
  __shared__ float sx[256], sy[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int base = blockIdx.y * 256;
  const int j = base + threadIdx.x;
  float vx = 1e30f, vy = 1e30f;
  if (j < nr && rsel[j]) { vx = rx[j]; vy = ry[j]; if (as_int) { vx = (float)(int)vx; vy = (float)(int)vy; } }
  sx[threadIdx.x] = vx; sy[threadIdx.x] = vy;
  __syncthreads();
  if (i >= nq) return;
  const float x = qx[i], y = qy[i];
  const int m = min(256, nr - base);
  int u = 0;
  for (int k = 0; k < m; ++k) {
    const float dx = sx[k] - x, dy = sy[k] - y;
    if (dx * dx + dy * dy < 1.0f) u = 1;
  }
  if (u) atomicOr(&used[i], 1);
}
// (zero = false: the caller's previous kernel cleared used[])
static inline void launch_near_flags_sel(hipStream_t s, int nq, const float* qx, const float* qy, int nr, const float* rx, const float* ry, const int32_t* rsel, int as_int, int32_t* used,
                                         bool zero = true) {
  if (zero) hipMemsetAsync(used, 0, sizeof(int32_t) * (size_t)nq, s);
  if (nq > 0 && nr > 0) hipLaunchKernelGGL(k_near_flags_sel, dim3((nq + 255) / 256, (nr + 255) / 256), dim3(256), 0, s, nq, qx, qy, nr, rx, ry, rsel, as_int, used);
}

// used[0..nq) = 0, then the 2-D launch
static inline void launch_near_flags(hipStream_t s, int nq, const float* qx, const float* qy, int nr, const float* rx, const float* ry, int32_t* used) {
  hipMemsetAsync(used, 0, sizeof(int32_t) * (size_t)nq, s);
  if (nq > 0 && nr > 0) hipLaunchKernelGGL(k_near_flags, dim3((nq + 255) / 256, (nr + 255) / 256), dim3(256), 0, s, nq, qx, qy, nr, rx, ry, used);
}

}  // namespace vdo
