// Frame construction kernels on gfx950 — the data-parallel part of Frame::Frame
// (reference src/Frame.cc:61-260):
//   K9  static keypoint filter + flow correspondence + depth gather   :100-128, :178-194
//   K10 semi-dense object sampling, stride 4, raster order            :201-228
// Both are ordered stream compactions: the output ORDER is part of the contract (feature
// indices are stored in the Map and in the tracklets), so flags are scanned with wave ballots /
// workgroup scans and written in input order — never with atomics.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"
#include "frame_images.hpp"

namespace vdo {

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int* total) {
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
  *total = __shfl(incl, 63, 64);
  return incl - v;
}

// workgroup exclusive scan (<=1024 threads); returns exclusive prefix, *total = sum. lds: int[17]
__device__ __forceinline__ int block_excl_scan(int v, int* lds, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  int wt;
  const int ex = wave_excl_scan(v, lane, &wt);
  __syncthreads();
  if (lane == 0) lds[wv] = wt;
  __syncthreads();
  if (threadIdx.x == 0) { int a = 0; for (int q = 0; q < nw; ++q) { const int t = lds[q]; lds[q] = a; a += t; } lds[16] = a; }
  __syncthreads();
  *total = lds[16];
  return ex + lds[wv];
}

// K9.  Single workgroup (n <= a few thousand keypoints), chunks of blockDim.x in input order.
__global__ __launch_bounds__(1024) void k_static_filter(int n, const float* __restrict__ kx, const float* __restrict__ ky,
                                                        const int32_t* __restrict__ mask, const float* __restrict__ depth,
                                                        const float* __restrict__ flow, int w, int h, float th_depth,
                                                        int32_t* __restrict__ keep_idx, float* __restrict__ corr_x, float* __restrict__ corr_y,
                                                        float* __restrict__ flow_x, float* __restrict__ flow_y, float* __restrict__ depth_out,
                                                        int* __restrict__ n_out, int sampled) {
  __shared__ int lds[17];
  int base_out = 0;
  for (int base = 0; base < n; base += blockDim.x) {
    const int i = base + threadIdx.x;
    int keep = 0;
    float fxe = 0, fye = 0, px = 0, py = 0, dd = 0;
    if (i < n) {
      px = kx[i]; py = ky[i];
      const int x = (int)px, y = (int)py;
      const size_t o = (size_t)y * w + x;
      if (mask[o] == 0) {
        const float d = depth[o];
        if (!(d > th_depth || d <= 0)) {
          fxe = flow[2 * o]; fye = flow[2 * o + 1];
          // ORB branch (Frame.cc:116-124): destination right/bottom bounds only; sampled branch (:156-160): all four sides
          const bool inb = sampled ? (px + fxe < w && py + fye < h && px + fxe > 0 && py + fye > 0) : (px + fxe < w && py + fye < h && px < w && py < h);
          if (fxe != 0 && fye != 0 && inb) { keep = 1; dd = d > 0 ? d : -1.f; }
        }
      }
    }
    int total;
    const int pos = base_out + block_excl_scan(keep, lds, &total);
    if (keep) {
      keep_idx[pos] = i; corr_x[pos] = px + fxe; corr_y[pos] = py + fye; flow_x[pos] = fxe; flow_y[pos] = fye; depth_out[pos] = dd;
    }
    base_out += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = base_out;
}

// K10, pass 1: flags + per-workgroup counts.  Probe index p -> (i = (p / ncol) * step, j = (p % ncol) * step)
__global__ __launch_bounds__(256) void k_obj_count(const int32_t* __restrict__ mask, const float* __restrict__ depth, const float* __restrict__ flow,
                                                   int w, int h, float th_obj, int step, int ncol, int nprobe, int* __restrict__ blk_cnt) {
  __shared__ int lds[17];
  const int p = blockIdx.x * 256 + threadIdx.x;
  int keep = 0;
  if (p < nprobe) {
    const int i = (p / ncol) * step, j = (p % ncol) * step;
    const size_t o = (size_t)i * w + j;
    const float d = depth[o];
    if (mask[o] != 0 && d < th_obj && d > 0) {
      const float fx = flow[2 * o], fy = flow[2 * o + 1];
      if (j + fx < w && j + fx > 0 && i + fy < h && i + fy > 0) keep = 1;
    }
  }
  int total;
  block_excl_scan(keep, lds, &total);
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = total;
}
// pass 2: exclusive scan of the workgroup counts (single workgroup)
__global__ __launch_bounds__(1024) void k_scan_blocks(int* __restrict__ blk_cnt, int nblk, int* __restrict__ n_out) {
  __shared__ int lds[17];
  int carry = 0;
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblk ? blk_cnt[i] : 0;
    int total;
    const int ex = block_excl_scan(v, lds, &total);
    if (i < nblk) blk_cnt[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = carry;
}
// pass 3: ordered scatter
__global__ __launch_bounds__(256) void k_obj_scatter(const int32_t* __restrict__ mask, const float* __restrict__ depth, const float* __restrict__ flow,
                                                     int w, int h, float th_obj, int step, int ncol, int nprobe, const int* __restrict__ blk_off, int cap,
                                                     float* __restrict__ key_x, float* __restrict__ key_y, float* __restrict__ corr_x, float* __restrict__ corr_y,
                                                     float* __restrict__ flow_x, float* __restrict__ flow_y, float* __restrict__ depth_out, int32_t* __restrict__ label) {
  __shared__ int lds[17];
  const int p = blockIdx.x * 256 + threadIdx.x;
  int keep = 0, i = 0, j = 0, lab = 0;
  float fx = 0, fy = 0, d = 0;
  if (p < nprobe) {
    i = (p / ncol) * step; j = (p % ncol) * step;
    const size_t o = (size_t)i * w + j;
    d = depth[o]; lab = mask[o];
    if (lab != 0 && d < th_obj && d > 0) {
      fx = flow[2 * o]; fy = flow[2 * o + 1];
      if (j + fx < w && j + fx > 0 && i + fy < h && i + fy > 0) keep = 1;
    }
  }
  int total;
  const int pos = blk_off[blockIdx.x] + block_excl_scan(keep, lds, &total);
  if (keep && pos < cap) {
    flow_x[pos] = fx; flow_y[pos] = fy; corr_x[pos] = j + fx; corr_y[pos] = i + fy;
    key_x[pos] = (float)j; key_y[pos] = (float)i; depth_out[pos] = d; label[pos] = lab;
  }
}

}  // namespace vdo

using namespace vdo;

extern "C" int vdo_frame_images_destroy(vdo_frame_images* f) {
  if (!f) return VDO_OK;
  if (f->ctx) ctx_bind(f->ctx);
  for (void* p : f->allocs) hipFree(p);
  if (f->h_pin) hipHostFree(f->h_pin);
  if (f->h_pin2) hipHostFree(f->h_pin2);
  delete f;
  return VDO_OK;
}

extern "C" int vdo_frame_images_set_ctx(vdo_frame_images* f, vdo_ctx* ctx) {
  if (!f || !ctx) return set_error(VDO_ERR_INVALID, "vdo_frame_images_set_ctx: null argument");
  f->ctx = ctx;
  return VDO_OK;
}

extern "C" int vdo_frame_images_create(vdo_ctx* ctx, int w, int h, vdo_frame_images** out) {
  if (!ctx || !out || w <= 0 || h <= 0) return set_error(VDO_ERR_INVALID, "bad argument");
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  vdo_frame_images* f = new vdo_frame_images();
  f->ctx = ctx; f->w = w; f->h = h;
  f->cap = ((w + 3) / 4) * ((h + 3) / 4) + 4096;
  auto dev = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) return nullptr; f->allocs.push_back(p); return p; };
  const size_t np = (size_t)w * h;
  f->d_mask = (int32_t*)dev(4 * np); f->d_depth = (float*)dev(4 * np); f->d_flow = (float*)dev(8 * np);
  f->d_rows = (float*)dev(4 * (size_t)f->cap * 10);
  if (f->d_rows) {
    for (int k = 0; k < 7; ++k) f->d_f[k] = f->d_rows + (size_t)k * f->cap;
    f->d_i[0] = (int32_t*)(f->d_rows + (size_t)7 * f->cap);
    f->d_f[7] = f->d_rows + (size_t)8 * f->cap;
    f->d_i[1] = (int32_t*)(f->d_rows + (size_t)9 * f->cap);
  }
  if (hipHostMalloc((void**)&f->h_pin, 4 * ((size_t)f->cap * 8 + 16)) != hipSuccess) f->h_pin = nullptr;
  if (hipHostMalloc((void**)&f->h_pin2, 4 * ((size_t)f->cap * 8 + 16)) != hipSuccess) f->h_pin2 = nullptr;
  f->d_rows2 = (float*)dev(4 * (size_t)f->cap * 8); f->d_cnt2 = (int*)dev(16);
  f->d_cnt = (int*)dev(16); f->d_blk = (int*)dev(4 * ((size_t)f->cap / 256 + 2));
  f->d_cand = (unsigned long long*)dev(8 * np);
  if (f->d_cand) hipMemsetAsync(f->d_cand, 0, 8 * np, ctx->stream);
  f->d_ticket = (int*)dev(64);
  if (f->d_ticket) hipMemsetAsync(f->d_ticket, 0, 64, ctx->stream);
  for (void* p : f->allocs) if (!p) { vdo_frame_images_destroy(f); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  if (!f->d_blk || !f->h_pin || !f->h_pin2) { vdo_frame_images_destroy(f); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  *out = f;
  return VDO_OK;
}

// Host images -> HBM.  The caller's buffers are pageable (cv::Mat data); plain stream copies move them at ~35 GB/s on this stack
// (measured: tools/host_input_probe.py; a pinned staging block filled by pool threads was tried and was no faster).
extern "C" int vdo_frame_images_upload(vdo_frame_images* f, const float* depth, const float* flow, const int32_t* mask) {
  if (!f) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = f->ctx->stream;
  const size_t np = (size_t)f->w * f->h;
  if (depth) hipMemcpyAsync(f->d_depth, depth, 4 * np, hipMemcpyHostToDevice, s);
  if (flow) hipMemcpyAsync(f->d_flow, flow, 8 * np, hipMemcpyHostToDevice, s);
  if (mask) hipMemcpyAsync(f->d_mask, mask, 4 * np, hipMemcpyHostToDevice, s);
  if (hipStreamSynchronize(s) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "frame upload failed");
  return VDO_OK;
}

extern "C" int vdo_frame_images_upload_on(vdo_ctx* ctx, vdo_frame_images* f, const float* depth, const float* flow, const int32_t* mask) {
  if (!ctx || !f) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ctx->stream;
  const size_t np = (size_t)f->w * f->h;
  if (depth) hipMemcpyAsync(f->d_depth, depth, 4 * np, hipMemcpyHostToDevice, s);
  if (flow) hipMemcpyAsync(f->d_flow, flow, 8 * np, hipMemcpyHostToDevice, s);
  if (mask) hipMemcpyAsync(f->d_mask, mask, 4 * np, hipMemcpyHostToDevice, s);
  if (hipStreamSynchronize(s) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "frame upload failed");
  return VDO_OK;
}

// Same as upload, but the sources are DEVICE pointers (e.g. torch tensors): device-to-device, stream-ordered, no sync.
extern "C" int vdo_frame_images_upload_device(vdo_frame_images* f, const float* depth, const float* flow, const int32_t* mask) {
  if (!f) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = f->ctx->stream;
  const size_t np = (size_t)f->w * f->h;
  if (depth) hipMemcpyAsync(f->d_depth, depth, 4 * np, hipMemcpyDeviceToDevice, s);
  if (flow) hipMemcpyAsync(f->d_flow, flow, 8 * np, hipMemcpyDeviceToDevice, s);
  if (mask) hipMemcpyAsync(f->d_mask, mask, 4 * np, hipMemcpyDeviceToDevice, s);
  return VDO_OK;
}

// GrabImageRGBD's ingest of device-resident inputs as ONE launch: the caller's flow and mask copied into the resident image set and the
// raw depth map converted on the way (K1: d < 0 -> 0, else bf / (d / factor) - the two correctly rounded divisions of k_depth_preprocess,
// src/Tracking.cc:180-204), instead of three device-to-device copies and a kernel (four dependent operations at the head of every frame's
// critical chain).  16 bytes per thread: one float4 of depth, two of flow, one int4 of mask.
__global__ __launch_bounds__(256) void k_ingest(const float4* __restrict__ depth, const float4* __restrict__ flow, const int4* __restrict__ mask, int64_t n4, int64_t n,
                                                float bf, float factor, int convert, float4* __restrict__ d_out, float4* __restrict__ f_out, int4* __restrict__ m_out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n4) {
    float4 d = depth[i];
    if (convert) {
      d.x = d.x < 0 ? 0.f : bf / (d.x / factor); d.y = d.y < 0 ? 0.f : bf / (d.y / factor);
      d.z = d.z < 0 ? 0.f : bf / (d.z / factor); d.w = d.w < 0 ? 0.f : bf / (d.w / factor);
    }
    d_out[i] = d;
    f_out[2 * i] = flow[2 * i]; f_out[2 * i + 1] = flow[2 * i + 1];
    m_out[i] = mask[i];
  }
  if (i == 0) {                                     // (a pixel count that is not a multiple of 4: the tail, element by element)
    const float* ds = (const float*)depth; const float* fs = (const float*)flow; const int32_t* ms = (const int32_t*)mask;
    float* dd = (float*)d_out; float* fd = (float*)f_out; int32_t* md = (int32_t*)m_out;
    for (int64_t k = 4 * n4; k < n; ++k) {
      const float v = ds[k];
      dd[k] = convert ? (v < 0 ? 0.f : bf / (v / factor)) : v;
      fd[2 * k] = fs[2 * k]; fd[2 * k + 1] = fs[2 * k + 1];
      md[k] = ms[k];
    }
  }
}

extern "C" int vdo_frame_images_ingest_device(vdo_frame_images* f, const float* depth, const float* flow, const int32_t* mask, float bf, float depth_map_factor, int convert_depth) {
  if (!f || !depth || !flow || !mask) return set_error(VDO_ERR_INVALID, "vdo_frame_images_ingest_device: null argument");
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  const int64_t n = (int64_t)f->w * f->h;
  if ((((uintptr_t)depth | (uintptr_t)flow | (uintptr_t)mask) & 15) != 0) {      // unaligned caller buffers: the plain copies + K1
    rc = vdo_frame_images_upload_device(f, depth, flow, mask);
    return rc != VDO_OK || !convert_depth ? rc : vdo_frame_images_depth_preprocess(f, bf, depth_map_factor);
  }
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(k_ingest, dim3((unsigned)((std::max<int64_t>(n4, 1) + 255) / 256)), dim3(256), 0, f->ctx->stream, (const float4*)depth, (const float4*)flow, (const int4*)mask, n4, n,
                     bf, depth_map_factor, convert_depth, (float4*)f->d_depth, (float4*)f->d_flow, (int4*)f->d_mask);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "vdo_frame_images_ingest_device: %s", hipGetErrorString(e));
  return VDO_OK;
}

// K1 on the resident depth image (in place)
extern "C" int vdo_frame_images_depth_preprocess(vdo_frame_images* f, float bf, float depth_map_factor) {
  if (!f) return set_error(VDO_ERR_INVALID, "null handle");
  return vdo_depth_preprocess(f->ctx, f->d_depth, (int64_t)f->w * f->h, bf, depth_map_factor, 1);
}

static int static_filter_impl(vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth, int sampled,
                              int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out, int* n_out);

extern "C" int vdo_frame_static_filter(vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth,
                                       int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out, int* n_out) {
  return static_filter_impl(f, n, kx, ky, th_depth, 0, keep_idx, corr_x, corr_y, flow_x, flow_y, depth_out, n_out);
}
extern "C" int vdo_frame_static_filter_sampled(vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth,
                                               int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out, int* n_out) {
  return static_filter_impl(f, n, kx, ky, th_depth, 1, keep_idx, corr_x, corr_y, flow_x, flow_y, depth_out, n_out);
}

// Frame::SampleKeyPoints (src/Frame.cc:672-737): N = 3000 random integer positions over a 20 x 20 grid, cv::RNG(seed) (the reference
// seeds with time(NULL): any seed is "the reference"), output in grid-cell order.  Host only.
extern "C" int vdo_sample_keypoints(int rows, int cols, uint64_t seed, int capacity, float* x_out, float* y_out, int* n_out) {
  if (rows <= 0 || cols <= 0 || !x_out || !y_out || !n_out) return set_error(VDO_ERR_INVALID, "vdo_sample_keypoints: bad argument");
  const int N = 3000, n_div = 20;
  if (capacity < N) return set_error(VDO_ERR_INVALID, "vdo_sample_keypoints: capacity %d < %d", capacity, N);
  if (cols < n_div || rows < n_div) return set_error(VDO_ERR_INVALID, "vdo_sample_keypoints: image smaller than the grid");
  uint64_t state = seed ? seed : 0xffffffffULL;                                    // cv::RNG (MWC, operations.hpp)
  auto next = [&]() -> unsigned { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; };
  auto uniform = [&](int a, int b) -> int { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); };
  std::vector<std::vector<float> > gx(n_div * n_div), gy(n_div * n_div);
  const int x_step = cols / n_div, y_step = rows / n_div;
  int key_num = 0;
  while (key_num < N) {
    bool done = false;
    for (int i = 0; i < n_div && !done; ++i)
      for (int j = 0; j < n_div; ++j) {
        const float x = (float)uniform(i * x_step, (i + 1) * x_step);
        const float y = (float)uniform(j * y_step, (j + 1) * y_step);
        if (x >= cols || y >= rows || x <= 0 || y <= 0) continue;
        gx[i * n_div + j].push_back(x); gy[i * n_div + j].push_back(y);
        if (++key_num >= N) { done = true; break; }
      }
  }
  int n = 0;
  for (int c = 0; c < n_div * n_div; ++c)
    for (size_t k = 0; k < gx[c].size(); ++k) { x_out[n] = gx[c][k]; y_out[n] = gy[c][k]; ++n; }
  *n_out = n;
  return VDO_OK;
}

static int static_filter_enqueue(vdo_frame_images* f, hipStream_t s, int n, const float* kx, const float* ky, float th_depth, int sampled) {
  // inputs: kx|ky -> pinned -> rows 8,9 in one strided H2D; outputs: rows 0..7 (5 float rows, 2 unused, idx) x n in one
  // strided D2H next to the count (m <= n is only known after the kernel): 1 sync, no pageable copies.
  float* pin = f->h_pin;
  std::memcpy(pin, kx, 4 * (size_t)n); std::memcpy(pin + n, ky, 4 * (size_t)n);
  hipMemcpy2DAsync(f->d_f[7], 4 * (size_t)f->cap, pin, 4 * (size_t)n, 4 * (size_t)n, 2, hipMemcpyHostToDevice, s);
  hipLaunchKernelGGL(k_static_filter, dim3(1), dim3(1024), 0, s, n, (const float*)f->d_f[7], (const float*)f->d_i[1], (const int32_t*)f->d_mask,
                     (const float*)f->d_depth, (const float*)f->d_flow, f->w, f->h, th_depth, f->d_i[0], f->d_f[0], f->d_f[1], f->d_f[2], f->d_f[3], f->d_f[4], f->d_cnt, sampled);
  int* pcnt = (int*)(pin + (size_t)8 * f->cap);
  float* stage = pin + 2 * (size_t)n;          // behind the inputs (the H2D above is stream-ordered before the D2H)
  hipMemcpyAsync(pcnt, f->d_cnt, 4, hipMemcpyDeviceToHost, s);
  hipMemcpy2DAsync(stage, 4 * (size_t)n, f->d_rows, 4 * (size_t)f->cap, 4 * (size_t)n, 8, hipMemcpyDeviceToHost, s);
  return VDO_OK;
}
static void static_filter_collect(vdo_frame_images* f, int n, int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out, int* n_out) {
  float* pin = f->h_pin;
  const int m = *(int*)(pin + (size_t)8 * f->cap);
  float* stage = pin + 2 * (size_t)n;
  *n_out = m;
  if (m) {
    float* dst[5] = {corr_x, corr_y, flow_x, flow_y, depth_out};
    for (int k = 0; k < 5; ++k) std::memcpy(dst[k], stage + (size_t)k * n, 4 * (size_t)m);
    std::memcpy(keep_idx, stage + (size_t)7 * n, 4 * (size_t)m);
  }
}
static int static_filter_impl(vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth, int sampled,
                              int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out, int* n_out) {
  if (!f || !n_out || n < 0 || 10 * (size_t)n > 8 * (size_t)f->cap) return set_error(VDO_ERR_INVALID, "bad argument / too many keypoints for the staging buffer");
  *n_out = 0;
  if (n == 0) return VDO_OK;
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  static_filter_enqueue(f, f->ctx->stream, n, kx, ky, th_depth, sampled);
  if (hipStreamSynchronize(f->ctx->stream) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "static filter failed: %s", hipGetErrorString(hipGetLastError()));
  static_filter_collect(f, n, keep_idx, corr_x, corr_y, flow_x, flow_y, depth_out, n_out);
  return VDO_OK;
}

// device-only variant used by the per-frame pipeline / bench: results stay in HBM, count is returned
constexpr int kObjSpec = 8192;     // columns of the 8 result rows that come back with the count (a second copy only if more were kept)
static int object_sample_enqueue(vdo_frame_images* f, hipStream_t s, float th_depth_obj, int step, bool want_host, float* rows, int* cnt, float* pin) {
  const int ncol = (f->w + step - 1) / step, nrow = (f->h + step - 1) / step, nprobe = ncol * nrow;
  const int nblk = (nprobe + 255) / 256;
  if (nprobe > f->cap) return set_error(VDO_ERR_INVALID, "sampling step too small for the scratch capacity");
  const size_t C = (size_t)f->cap;
  hipLaunchKernelGGL(k_obj_count, dim3(nblk), dim3(256), 0, s, (const int32_t*)f->d_mask, (const float*)f->d_depth, (const float*)f->d_flow, f->w, f->h, th_depth_obj, step, ncol, nprobe, f->d_blk);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, f->d_blk, nblk, cnt);
  hipLaunchKernelGGL(k_obj_scatter, dim3(nblk), dim3(256), 0, s, (const int32_t*)f->d_mask, (const float*)f->d_depth, (const float*)f->d_flow, f->w, f->h, th_depth_obj, step, ncol, nprobe,
                     (const int*)f->d_blk, f->cap, rows, rows + C, rows + 2 * C, rows + 3 * C, rows + 4 * C, rows + 5 * C, rows + 6 * C, (int32_t*)(rows + 7 * C));
  // count + the first kObjSpec columns of the 8 result rows in one go (pinned); a second strided copy only if m > kObjSpec
  int* pcnt = (int*)(pin + (size_t)8 * f->cap);
  const int spec = want_host ? std::min(kObjSpec, f->cap) : 0;
  hipMemcpyAsync(pcnt, cnt, 4, hipMemcpyDeviceToHost, s);
  if (spec) hipMemcpy2DAsync(pin, 4 * (size_t)spec, rows, 4 * C, 4 * (size_t)spec, 8, hipMemcpyDeviceToHost, s);
  return VDO_OK;
}
static int object_sample_collect(vdo_frame_images* f, hipStream_t s, int cap, float* rows, float* pin, float* key_x, float* key_y, float* corr_x, float* corr_y,
                                 float* flow_x, float* flow_y, float* depth_out, int32_t* label, int* n_out) {
  const int m = *(int*)(pin + (size_t)8 * f->cap);
  *n_out = m;
  if (key_x) {   // host outputs requested
    const int spec = std::min(kObjSpec, f->cap);
    if (m > cap) return set_error(VDO_ERR_INVALID, "object sampling: %d points exceed the output capacity %d", m, cap);
    float* dst[7] = {key_x, key_y, corr_x, corr_y, flow_x, flow_y, depth_out};
    const int m0 = std::min(m, spec);
    for (int k = 0; k < 7; ++k) if (dst[k] && m0) std::memcpy(dst[k], pin + (size_t)k * spec, 4 * (size_t)m0);
    if (label && m0) std::memcpy(label, pin + (size_t)7 * spec, 4 * (size_t)m0);
    if (m > spec) {
      const int r = m - spec;
      hipMemcpy2DAsync(pin, 4 * (size_t)r, rows + spec, 4 * (size_t)f->cap, 4 * (size_t)r, 8, hipMemcpyDeviceToHost, s);
      if (hipStreamSynchronize(s) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "object sampling D2H failed");
      for (int k = 0; k < 7; ++k) if (dst[k]) std::memcpy(dst[k] + spec, pin + (size_t)k * r, 4 * (size_t)r);
      if (label) std::memcpy(label + spec, pin + (size_t)7 * r, 4 * (size_t)r);
    }
  }
  return VDO_OK;
}
extern "C" int vdo_frame_object_sample(vdo_frame_images* f, float th_depth_obj, int step, int cap,
                                       float* key_x, float* key_y, float* corr_x, float* corr_y,
                                       float* flow_x, float* flow_y, float* depth_out, int32_t* label, int* n_out) {
  if (!f || !n_out || step <= 0) return set_error(VDO_ERR_INVALID, "bad argument");
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  rc = object_sample_enqueue(f, f->ctx->stream, th_depth_obj, step, key_x != nullptr, f->d_rows, f->d_cnt, f->h_pin);
  if (rc != VDO_OK) return rc;
  if (hipStreamSynchronize(f->ctx->stream) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "object sampling failed: %s", hipGetErrorString(hipGetLastError()));
  return object_sample_collect(f, f->ctx->stream, cap, f->d_rows, f->h_pin, key_x, key_y, corr_x, corr_y, flow_x, flow_y, depth_out, label, n_out);
}

// K10 on the stream of `on_ctx` and the second scratch set: for a caller that samples the objects on another host thread while the
// owning thread runs K9 / RenewFrameInfo on the same image set.
extern "C" int vdo_frame_object_sample_on(vdo_ctx* on_ctx, vdo_frame_images* f, float th_depth_obj, int step, int cap,
                                          float* key_x, float* key_y, float* corr_x, float* corr_y,
                                          float* flow_x, float* flow_y, float* depth_out, int32_t* label, int* n_out) {
  if (!f || !n_out || step <= 0 || !key_x) return set_error(VDO_ERR_INVALID, "vdo_frame_object_sample_on: bad argument");
  vdo_ctx* c = on_ctx ? on_ctx : f->ctx;
  int rc = ctx_bind(c);
  if (rc != VDO_OK) return rc;
  rc = object_sample_enqueue(f, c->stream, th_depth_obj, step, true, f->d_rows2, f->d_cnt2, f->h_pin2);
  if (rc != VDO_OK) return rc;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "object sampling failed: %s", hipGetErrorString(hipGetLastError()));
  return object_sample_collect(f, c->stream, cap, f->d_rows2, f->h_pin2, key_x, key_y, corr_x, corr_y, flow_x, flow_y, depth_out, label, n_out);
}

// K9 + K10 of one image with ONE synchronisation: the two results use separate scratch sets, so both pipelines are queued
// back to back (Frame::Frame runs them back to back too: src/Frame.cc:104-131, 168-199).
// (on_ctx: the stream / device binding to use instead of the image set's own context - a caller that runs this on a second host
// thread while the owning thread keeps using the image set; the two scratch sets used here are not touched by anything else)
extern "C" int vdo_frame_filters_on(vdo_ctx* on_ctx, vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth, int sampled,
                                    int32_t* keep_idx, float* s_corr_x, float* s_corr_y, float* s_flow_x, float* s_flow_y, float* s_depth, int* n_static,
                                    float th_depth_obj, int step, int cap,
                                    float* key_x, float* key_y, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out, int32_t* label, int* n_obj) {
  if (!f || !n_static || !n_obj || n < 0 || step <= 0 || !key_x || 10 * (size_t)n > 8 * (size_t)f->cap) return set_error(VDO_ERR_INVALID, "vdo_frame_filters: bad argument");
  vdo_ctx* c = on_ctx ? on_ctx : f->ctx;
  int rc = ctx_bind(c);
  if (rc != VDO_OK) return rc;
  hipStream_t s = c->stream;
  *n_static = 0;
  if (n) static_filter_enqueue(f, s, n, kx, ky, th_depth, sampled);
  rc = object_sample_enqueue(f, s, th_depth_obj, step, true, f->d_rows2, f->d_cnt2, f->h_pin2);
  if (rc != VDO_OK) return rc;
  if (hipStreamSynchronize(s) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "frame filters failed: %s", hipGetErrorString(hipGetLastError()));
  if (n) static_filter_collect(f, n, keep_idx, s_corr_x, s_corr_y, s_flow_x, s_flow_y, s_depth, n_static);
  return object_sample_collect(f, s, cap, f->d_rows2, f->h_pin2, key_x, key_y, corr_x, corr_y, flow_x, flow_y, depth_out, label, n_obj);
}
extern "C" int vdo_frame_filters(vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth, int sampled,
                                 int32_t* keep_idx, float* s_corr_x, float* s_corr_y, float* s_flow_x, float* s_flow_y, float* s_depth, int* n_static,
                                 float th_depth_obj, int step, int cap,
                                 float* key_x, float* key_y, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out, int32_t* label, int* n_obj) {
  return vdo_frame_filters_on(nullptr, f, n, kx, ky, th_depth, sampled, keep_idx, s_corr_x, s_corr_y, s_flow_x, s_flow_y, s_depth, n_static,
                              th_depth_obj, step, cap, key_x, key_y, corr_x, corr_y, flow_x, flow_y, depth_out, label, n_obj);
}
