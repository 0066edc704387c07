// Wave/workgroup primitives shared by the tile kernels (gfx950: 64-lane waves).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
namespace vdo {

// Sum over the wave, valid in LANE 0: the tree  v[l] += v[l + 32], += v[l + 16], ... += v[l + 1]  (the additions and their order are those of the
// __shfl_down form this replaces - same bits), by gfx950 row swaps and DPP moves instead of ds_bpermute round trips (and without the
// lane-address registers those need).
__device__ __forceinline__ double wave_sum(double v) {
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);        // v[l] + v[l ^ 32]
  }
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);        // v[l] + v[l ^ 16]
  }
  auto mv = [](double x, auto ctrl) {
    constexpr int C = decltype(ctrl)::value;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), C, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), C, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };
  v += mv(v, std::integral_constant<int, 0x128>{});    // row_ror:8  lane l < 8 reads l + 8
  v += mv(v, std::integral_constant<int, 0x104>{});    // row_shl:4  lane l reads l + 4
  v += mv(v, std::integral_constant<int, 0x102>{});    // row_shl:2
  v += mv(v, std::integral_constant<int, 0x101>{});    // row_shl:1
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

// Same reduction split into a control word (computed once per wave-iteration from the keys)
// and small value groups, so that callers can produce the values just-in-time and keep the
// register footprint low (16 running sums scanned 4 at a time instead of 16 at once).
struct SegCtl { unsigned take; int key; bool tail; };
__device__ __forceinline__ SegCtl seg_ctl(int key) {
  const int lane = threadIdx.x & 63;
  const int kprev1 = __shfl_up(key, 1, 64);
  const int head = (lane == 0 || kprev1 != key) ? 1 : 0;
  int f = head;
  unsigned take = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int off = 1 << k;
    const int ft = __shfl_up(f, off, 64);
    if ((lane >= off) && !f) { take |= 1u << k; f |= ft; }
  }
  const int hnext = __shfl_down(head, 1, 64);
  return SegCtl{take, key, (key >= 0) && (lane == 63 || hnext != 0)};
}
template <int N>
__device__ __forceinline__ void seg_apply(double (&v)[N], const SegCtl& c, double* lds_acc_slot /* &acc[key*stride + first] */) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int off = 1 << k;
    const bool take = (c.take >> k) & 1u;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const double t = __shfl_up(v[i], off, 64);
      if (take) v[i] += t;
    }
  }
  if (c.tail) {
#pragma unroll
    for (int i = 0; i < N; ++i) atomicAdd(lds_acc_slot + i, v[i]);
  }
  asm volatile("" ::: "memory");   // keep the groups sequential: bounds live registers
}

// ---- DPP variant: segments are additionally cut at 16-lane row boundaries so every step is a
// VALU `row_shr` DPP move (no ds_bpermute round trip through the LDS crossbar).  A segment that
// spans several rows simply issues one LDS atomic per row (<=4 lanes on the same address).
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = dpp_i32<CTRL>(__double2loint(v), 0), hi = dpp_i32<CTRL>(__double2hiint(v), 0);
  return __hiloint2double(hi, lo);
}
struct SegCtl16 { unsigned take; bool tail; };
__device__ __forceinline__ SegCtl16 seg_ctl16(int key) {
  const int lr = threadIdx.x & 15;
  const int kprev = dpp_i32<0x111>(key, -2);          // row_shr:1
  const int head = (lr == 0 || kprev != key) ? 1 : 0;
  int f = head;
  unsigned take = 0;
  { const int ft = dpp_i32<0x111>(f, 1); if (lr >= 1 && !f) { take |= 1u; f |= ft; } }
  { const int ft = dpp_i32<0x112>(f, 1); if (lr >= 2 && !f) { take |= 2u; f |= ft; } }
  { const int ft = dpp_i32<0x114>(f, 1); if (lr >= 4 && !f) { take |= 4u; f |= ft; } }
  { const int ft = dpp_i32<0x118>(f, 1); if (lr >= 8 && !f) { take |= 8u; f |= ft; } }
  const int hnext = dpp_i32<0x101>(head, 1);          // row_shl:1 (lane 15 of a row keeps `old` = 1)
  return SegCtl16{take, (key >= 0) && (lr == 15 || hnext != 0)};
}
// zero-filling DPP move (bound_ctrl: lanes whose source is outside the row read 0; every lane is written, so there is no
// `old` operand to initialise)
template <int CTRL>
__device__ __forceinline__ double dpp_f64z(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// plain DPP move of a double (every lane has a source for the controls used with it)
template <int CTRL>
__device__ __forceinline__ double dpp_f64p(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Sum over the 8 lanes of a lane octet (lane & ~7), result in all of them: xor 1, xor 2, mirror inside the octet.
__device__ __forceinline__ double octet_allsum(double v) {
  v += dpp_f64p<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f64p<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f64p<0x141>(v);     // row_half_mirror: the two quads of an octet are uniform by now
  return v;
}
// Sum over the 8 lanes with the same (lane & 7), result in all of them: xor 8 (row_ror:8), xor 16 and xor 32 with the gfx950
// row-swap instructions (v_permlane16_swap / v_permlane32_swap of a register with its copy: both halves of a pair end up side by side).
__device__ __forceinline__ double stride8_allsum(double v) {
  v += dpp_f64p<0x128>(v);
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
  return v;
}
// The conditional add of a scan step is ONE fused multiply-add with a 0.0 / 1.0 flag: fma(t, 1, v) = round(t + v), the very
// result of the addition, and fma(t, 0, v) = v exactly (t is finite) - instead of two v_cndmask + v_add per double.
struct SegFlags { double f1, f2, f4, f8; };
__device__ __forceinline__ SegFlags seg_flags(const SegCtl16& c) {
  return SegFlags{(c.take & 1u) ? 1.0 : 0.0, (c.take & 2u) ? 1.0 : 0.0, (c.take & 4u) ? 1.0 : 0.0, (c.take & 8u) ? 1.0 : 0.0};
}
template <int N>
__device__ __forceinline__ void seg_apply16(double (&v)[N], const SegCtl16& c, const SegFlags& f, double* lds_acc_slot) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __builtin_fma(dpp_f64z<0x111>(v[i]), f.f1, v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __builtin_fma(dpp_f64z<0x112>(v[i]), f.f2, v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __builtin_fma(dpp_f64z<0x114>(v[i]), f.f4, v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __builtin_fma(dpp_f64z<0x118>(v[i]), f.f8, v[i]);
  if (c.tail) {
#pragma unroll
    for (int i = 0; i < N; ++i) atomicAdd(lds_acc_slot + i, v[i]);
  }
}

// Sum over the pose-major rows [r0, r1) of `part` (rows of STRIDE doubles, STRIDE = 8 or 16: every (tile, pose slot) pair owns one row, the
// rows of a pose are contiguous - slot_dst, ba_dev.hpp): the wave streams them, 64 / STRIDE rows (512 contiguous bytes) per step; lane l always
// meets component l % STRIDE, whose total is returned in every lane with that remainder.  Fixed order.
template <int STRIDE>
__device__ __forceinline__ double wave_gather_rows(const double* __restrict__ part, int r0, int r1) {
  static_assert(STRIDE == 8 || STRIDE == 16, "row of 8 or 16 doubles");
  const int lane = threadIdx.x & 63;
  const double* __restrict__ base = part + (int64_t)r0 * STRIDE;
  const int64_t n = (int64_t)(r1 - r0) * STRIDE;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int64_t i = lane;
  for (; i + 192 < n; i += 256) { a0 += base[i]; a1 += base[i + 64]; a2 += base[i + 128]; a3 += base[i + 192]; }
  for (; i < n; i += 64) a0 += base[i];
  double a = (a0 + a1) + (a2 + a3);
  a += __shfl_xor(a, 32, 64);
  a += __shfl_xor(a, 16, 64);
  if (STRIDE == 8) a += __shfl_xor(a, 8, 64);
  return a;
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
