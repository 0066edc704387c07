// Wave/workgroup primitives shared by the tile kernels (gfx950: 64-lane waves).
#pragma once
#include <hip/hip_runtime.h>

namespace vdo {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Segmented (by `key`, keys sorted within the wave) inclusive scan of N doubles, then the
// last lane of every segment adds its totals into lds_acc[key * stride + i] with LDS atomics.
// Lanes with key < 0 contribute nothing.  All 64 lanes must call this.
template <int N>
__device__ __forceinline__ void seg_reduce_to_lds(double (&v)[N], int key, double* lds_acc, int stride) {
  const int lane = threadIdx.x & 63;
  const int kprev1 = __shfl_up(key, 1, 64);
  const int head = (lane == 0 || kprev1 != key) ? 1 : 0;
  int f = head;                      // segment-head flag, propagated (Hillis–Steele segmented scan)
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int ft = __shfl_up(f, off, 64);
    const bool take = (lane >= off) && !f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const double t = __shfl_up(v[i], off, 64);
      if (take) v[i] += t;
    }
    if (take) f |= ft;
  }
  const int hnext = __shfl_down(head, 1, 64);
  const bool tail = (key >= 0) && (lane == 63 || hnext != 0);
  if (tail) {
    double* a = lds_acc + (size_t)key * stride;
#pragma unroll
    for (int i = 0; i < N; ++i) atomicAdd(a + i, v[i]);
  }
}

// One wave sums, for pose p, the K-vectors of all its (tile,slot) partials (part: [K][NPS] SoA).
// Lanes stride over the slot list, then a fixed shuffle tree -> deterministic.  Result in all lanes.
template <int K>
__device__ __forceinline__ void wave_gather(const double* __restrict__ part, int64_t NPS, const int32_t* __restrict__ ps_off,
                                            const int32_t* __restrict__ ps_idx, int p, double (&out)[K]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < K; ++i) out[i] = 0.0;
  for (int k = ps_off[p] + lane; k < ps_off[p + 1]; k += 64) {
    const int64_t c = ps_idx[k];
#pragma unroll
    for (int i = 0; i < K; ++i) out[i] += part[i * NPS + c];
  }
#pragma unroll
  for (int i = 0; i < K; ++i) out[i] = __shfl(wave_sum(out[i]), 0, 64);
}

// workgroup sum of one double (<= 1024 threads); result broadcast.  lds: >= 17 doubles
__device__ __forceinline__ double block_sum1(double v, double* lds) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
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

__device__ __forceinline__ double block_max1(double v, double* lds) {
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

}  // namespace vdo
