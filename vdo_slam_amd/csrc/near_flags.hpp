// "Is there a reference point within 1 px" for every query point — the O(n*m) test of
// Tracking::RenewFrameInfo (reference src/Tracking.cc:2727-2745 static, :2893-2907 objects), tiled
// through LDS.  Float arithmetic as in the reference: sqrt((rx-qx)^2 + (ry-qy)^2) < 1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdo {

// used[i] = exists j: sqrt((rx[j]-qx[i])^2 + (ry[j]-qy[i])^2) < 1
static __global__ __launch_bounds__(256) void k_near_flags(int nq, const float* __restrict__ qx, const float* __restrict__ qy,
                                                           int nr, const float* __restrict__ rx, const float* __restrict__ ry, int32_t* __restrict__ used) {
  __shared__ float sx[256], sy[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float x = i < nq ? qx[i] : 0.f, y = i < nq ? qy[i] : 0.f;
  int u = 0;
  for (int base = 0; base < nr; base += 256) {
    const int j = base + threadIdx.x;
    sx[threadIdx.x] = j < nr ? rx[j] : 1e30f;
    sy[threadIdx.x] = j < nr ? ry[j] : 1e30f;
    __syncthreads();
    const int m = min(256, nr - base);
    for (int k = 0; k < m; ++k) {
      const float dx = sx[k] - x, dy = sy[k] - y;
      if (sqrtf(dx * dx + dy * dy) < 1.0f) u = 1;
    }
    __syncthreads();
  }
  if (i < nq) used[i] = u;
}

}  // namespace vdo
