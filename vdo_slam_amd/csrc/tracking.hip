// Tracking-side kernels on gfx950 — the data-parallel loops of Tracking::GrabImageRGBD / Track
// (reference src/Tracking.cc) over the HBM-resident depth / flow / mask images:
//   K11 propagate last-frame correspondences: depth + label gather            :259-305
//   K12 back-projection  Frame::UnprojectStereoObject, Optimizer::Get3DinWorld  src/Frame.cc:517-555, src/Optimizer.cc:2974-2995
//   K13 scene flow per object point (GetSceneFlowObj)                          :1278-1364
//   K14 RenewFrameInfo, static part: inlier carry-over, O(n*m) "already used"
//       distance test, validity predicate; the ORDER-dependent selection
//       (first-come truncation, stride-20 interleave) is applied on the host
//       from the flags, reproducing the sequential reference exactly             :2666-2778
//   K15 UpdateMask: label gather at flowed positions + mask warp                 :3015-3065
// All gathers are 4-byte random reads of L2-resident images: latency-bound, a few µs each.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"
#include "frame_images.hpp"
#include "near_flags.hpp"
#include "arena.hpp"
#include "tracking_shared.hpp"

namespace vdo {

// mode 0: K11 static (depth>0 else -1; bounds u<w-1,u>0,v<h-1,v>0)
// mode 1: K11 object (depth<th && depth>0 -> depth,label else 0.1,0)
// mode 2: K15 mask at (u<w,u>0,v<h,v>0) else -1
__global__ void k_gather(int mode, int n, const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ depth,
                         const int32_t* __restrict__ mask, int w, int h, float th, float* __restrict__ dout, int32_t* __restrict__ lout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int u = (int)kx[i], v = (int)ky[i];
  if (mode == 0) {
    float o = -1.f;
    if (u < (w - 1) && u > 0 && v < (h - 1) && v > 0) { const float d = depth[(size_t)v * w + u]; if (d > 0) o = d; }
    dout[i] = o;
  } else if (mode == 1) {
    float o = 0.1f; int l = 0;
    if (u < (w - 1) && u > 0 && v < (h - 1) && v > 0) {
      const float d = depth[(size_t)v * w + u];
      if (d < th && d > 0) { o = d; l = mask[(size_t)v * w + u]; }
    }
    dout[i] = o; lout[i] = l;
  } else {
    lout[i] = (u < w && u > 0 && v < h && v > 0) ? mask[(size_t)v * w + u] : -1;
  }
}


__global__ void k_get3d_world(int n, const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ d, Cam c, float* __restrict__ xyz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o[3];
  backproject(c, kx[i], ky[i], d[i], o);
  xyz[3 * i] = o[0]; xyz[3 * i + 1] = o[1]; xyz[3 * i + 2] = o[2];
}

__global__ void k_scene_flow(int n, const float* __restrict__ cx_, const float* __restrict__ cy_, const float* __restrict__ cd, const int32_t* __restrict__ cl, Cam cc,
                             const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ ld, const int32_t* __restrict__ ll, Cam lc,
                             float* __restrict__ flow3d, int32_t* __restrict__ objlab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (cl[i] <= 0 || ll[i] <= 0) { objlab[i] = -1; flow3d[3 * i] = 0; flow3d[3 * i + 1] = 0; flow3d[3 * i + 2] = 0; return; }
  float p[3], c[3];
  backproject(lc, lx[i], ly[i], ld[i], p);
  backproject(cc, cx_[i], cy_[i], cd[i], c);
  flow3d[3 * i] = c[0] - p[0]; flow3d[3 * i + 1] = c[1] - p[1]; flow3d[3 * i + 2] = c[2] - p[2];
}

// K14 validity predicate of RenewFrameInfo (static): writes flag, flow and depth for every candidate
__global__ void k_renew_pred(int n, const float* __restrict__ px, const float* __restrict__ py, const int32_t* __restrict__ mask,
                             const float* __restrict__ depth, const float* __restrict__ flow, int w, int h,
                             int32_t* __restrict__ ok, float* __restrict__ fx, float* __restrict__ fy, float* __restrict__ dout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x_ = px[i], y_ = py[i];
  const int x = (int)x_, y = (int)y_;
  int good = 0;
  float fxe = 0, fye = 0, d = -1.f;
  if (!(x >= w || y >= h || x <= 0 || y <= 0)) {
    const size_t o = (size_t)y * w + x;
    const float dd = depth[o];
    if (mask[o] == 0 && !(dd > 40 || dd <= 0)) {
      fxe = flow[2 * o]; fye = flow[2 * o + 1];
      if (fxe != 0 && fye != 0 && x_ + fxe < w && y_ + fye < h && x_ + fxe > 0 && y_ + fye > 0) { good = 1; d = dd > 0 ? dd : -1.f; }
    }
  }
  ok[i] = good; fx[i] = fxe; fy[i] = fye; dout[i] = d;
}

// carried[i] = ok[i] && (number of ok before i) <= max_keep: the first-come truncation of the carry-over loop ("stop once the
// size exceeds the limit", Tracking.cc:2703-2709) as a selection mask.  Single workgroup, chunks in input order.
__global__ __launch_bounds__(1024) void k_carry_select(int n, const int32_t* __restrict__ ok, int max_keep, int32_t* __restrict__ sel) {
  __shared__ int lds[17];
  int carry = 0;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < n && ok[i]) ? 1 : 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    __syncthreads();
    if (lane == 63) lds[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { int a = 0; for (int q = 0; q < 16; ++q) { const int t = lds[q]; lds[q] = a; a += t; } lds[16] = a; }
    __syncthreads();
    const int rank = carry + lds[wv] + incl - v;
    if (i < n) sel[i] = (v && rank <= max_keep) ? 1 : 0;
    carry += lds[16];
    __syncthreads();
  }
}

// K15b: every pixel of the previous mask with label `lab` writes `lab` at its flowed position
// (all writers store the same value, so the raster-order "last writer wins" of the reference is immaterial)
__global__ void k_mask_warp(const int32_t* __restrict__ mask_last, const float* __restrict__ flow_last, int w, int h, int32_t lab, int32_t* __restrict__ mask_cur) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (k >= w) return;
  const size_t o = (size_t)j * w + k;
  if (mask_last[o] != lab) return;
  const int fx = (int)flow_last[2 * o], fy = (int)flow_last[2 * o + 1];
  if (k + fx < w && k + fx > 0 && j + fy < h && j + fy > 0) mask_cur[(size_t)(j + fy) * w + (k + fx)] = lab;
}

}  // namespace vdo

using namespace vdo;

extern "C" int vdo_propagate_static(vdo_frame_images* f, int n, const float* kx, const float* ky, float* depth_out) {
  if (!f || n < 0) return set_error(VDO_ERR_INVALID, "bad argument");
  if (n == 0) return VDO_OK;
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  Arena S(f->ctx);
  if (!S.reserve(Arena::bytes_for(3 * (size_t)n))) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  float *dx = S.up(kx, n), *dy = S.up(ky, n), *dd = S.up<float>(nullptr, n);
  if (!dd) return set_error(VDO_ERR_OOM, "hipMalloc failed");
  hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), dim3(256), 0, S.stream(), 0, n, (const float*)dx, (const float*)dy, (const float*)f->d_depth, (const int32_t*)f->d_mask, f->w, f->h, 0.f, dd, (int32_t*)nullptr);
  S.down(depth_out, dd, n);
  return S.finish("vdo_propagate_static");
}

extern "C" int vdo_propagate_object(vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth_obj, float* depth_out, int32_t* label_out) {
  if (!f || n < 0) return set_error(VDO_ERR_INVALID, "bad argument");
  if (n == 0) return VDO_OK;
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  Arena S(f->ctx);
  if (!S.reserve(Arena::bytes_for(4 * (size_t)n))) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  float *dx = S.up(kx, n), *dy = S.up(ky, n), *dd = S.up<float>(nullptr, n);
  int32_t* dl = S.up<int32_t>(nullptr, n);
  if (!dl) return set_error(VDO_ERR_OOM, "hipMalloc failed");
  hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), dim3(256), 0, S.stream(), 1, n, (const float*)dx, (const float*)dy, (const float*)f->d_depth, (const int32_t*)f->d_mask, f->w, f->h, th_depth_obj, dd, dl);
  S.down(depth_out, dd, n); S.down(label_out, dl, n);
  return S.finish("vdo_propagate_object");
}

extern "C" int vdo_mask_at(vdo_frame_images* f, int n, const float* cx, const float* cy, int32_t* label_out) {
  if (!f || n < 0) return set_error(VDO_ERR_INVALID, "bad argument");
  if (n == 0) return VDO_OK;
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  Arena S(f->ctx);
  if (!S.reserve(Arena::bytes_for(3 * (size_t)n))) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  float *dx = S.up(cx, n), *dy = S.up(cy, n);
  int32_t* dl = S.up<int32_t>(nullptr, n);
  if (!dl) return set_error(VDO_ERR_OOM, "hipMalloc failed");
  hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), dim3(256), 0, S.stream(), 2, n, (const float*)dx, (const float*)dy, (const float*)f->d_depth, (const int32_t*)f->d_mask, f->w, f->h, 0.f, (float*)nullptr, dl);
  S.down(label_out, dl, n);
  return S.finish("vdo_mask_at");
}

extern "C" int vdo_mask_warp(vdo_frame_images* cur, vdo_frame_images* last, int32_t label) {
  if (!cur || !last || cur->w != last->w || cur->h != last->h) return set_error(VDO_ERR_INVALID, "bad argument");
  int rc = ctx_bind(cur->ctx);
  if (rc != VDO_OK) return rc;
  hipLaunchKernelGGL(k_mask_warp, dim3((cur->w + 255) / 256, cur->h), dim3(256), 0, cur->ctx->stream, (const int32_t*)last->d_mask, (const float*)last->d_flow, cur->w, cur->h, label, cur->d_mask);
  return VDO_OK;
}

extern "C" int vdo_frame_images_download_mask(vdo_frame_images* f, int32_t* mask_out) {
  if (!f || !mask_out) return set_error(VDO_ERR_INVALID, "bad argument");
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  hipMemcpyAsync(mask_out, f->d_mask, 4 * (size_t)f->w * f->h, hipMemcpyDeviceToHost, f->ctx->stream);
  hipError_t e = hipStreamSynchronize(f->ctx->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "vdo_frame_images_download_mask: %s", hipGetErrorString(e));
  return VDO_OK;
}

// the resident depth image (metres once K1 ran): what Tracking::GrabImageRGBD leaves in the caller's imD (src/Tracking.cc:180-204)
extern "C" int vdo_frame_images_download_depth(vdo_frame_images* f, float* depth_out) {
  if (!f || !depth_out) return set_error(VDO_ERR_INVALID, "bad argument");
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  hipMemcpyAsync(depth_out, f->d_depth, 4 * (size_t)f->w * f->h, hipMemcpyDeviceToHost, f->ctx->stream);
  hipError_t e = hipStreamSynchronize(f->ctx->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "vdo_frame_images_download_depth: %s", hipGetErrorString(e));
  return VDO_OK;
}

extern "C" int vdo_get3d_world(vdo_ctx* ctx, int n, const float* kx, const float* ky, const float* depth, const float K4[4], const float Twc[16], float* xyz_out) {
  if (!ctx || n < 0) return set_error(VDO_ERR_INVALID, "bad argument");
  if (n == 0) return VDO_OK;
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  Arena S(ctx);
  if (!S.reserve(Arena::bytes_for(6 * (size_t)n))) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  float *dx = S.up(kx, n), *dy = S.up(ky, n), *dd = S.up(depth, n), *dxyz = S.up<float>(nullptr, 3 * (size_t)n);
  if (!dxyz) return set_error(VDO_ERR_OOM, "hipMalloc failed");
  hipLaunchKernelGGL(k_get3d_world, dim3((n + 255) / 256), dim3(256), 0, S.stream(), n, (const float*)dx, (const float*)dy, (const float*)dd, make_cam_Twc(K4, Twc), dxyz);
  S.down(xyz_out, dxyz, 3 * (size_t)n);
  return S.finish("vdo_get3d_world");
}

extern "C" int vdo_scene_flow(vdo_ctx* ctx, int n, const float* cur_x, const float* cur_y, const float* cur_d, const int32_t* cur_label, const float Tcw_cur[16],
                              const float* last_x, const float* last_y, const float* last_d, const int32_t* last_label, const float Tcw_last[16],
                              const float K4[4], float* flow3d_out, int32_t* obj_label_inout) {
  if (!ctx || n < 0) return set_error(VDO_ERR_INVALID, "bad argument");
  if (n == 0) return VDO_OK;
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  Arena S(ctx);
  if (!S.reserve(Arena::bytes_for(12 * (size_t)n))) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  float *a = S.up(cur_x, n), *b = S.up(cur_y, n), *c = S.up(cur_d, n), *d = S.up(last_x, n), *e = S.up(last_y, n), *g = S.up(last_d, n);
  int32_t *cl = S.up(cur_label, n), *ll = S.up(last_label, n), *ol = S.up(obj_label_inout, n);
  float* fl = S.up<float>(nullptr, 3 * (size_t)n);
  if (!fl) return set_error(VDO_ERR_OOM, "hipMalloc failed");
  hipLaunchKernelGGL(k_scene_flow, dim3((n + 255) / 256), dim3(256), 0, S.stream(), n, (const float*)a, (const float*)b, (const float*)c, (const int32_t*)cl, make_cam_Tcw(K4, Tcw_cur),
                     (const float*)d, (const float*)e, (const float*)g, (const int32_t*)ll, make_cam_Tcw(K4, Tcw_last), fl, ol);
  S.down(flow3d_out, fl, 3 * (size_t)n); S.down(obj_label_inout, ol, n);
  return S.finish("vdo_scene_flow");
}

// K14, static part.  Same contract as Tracking::RenewFrameInfo :2666-2790 (see the oracle for the
// sequential statement).  GPU: predicates + O(n*m) distance flags; host: order-dependent selection.
// RenewFrameInfo (static) + Get3DinWorld of the new set in ONE pass over the device: both candidate lists (carried inliers,
// top-up keypoints) are judged, the carry-over truncation is a device-side selection, the "within 1 px of a carried key"
// test runs against that selection, every candidate is back-projected - one copy back, one synchronisation; the
// ORDER-dependent part (first-come, stride-20 interleave) stays on the host.  xyz_out may be NULL (K4 / Twc unused then).
extern "C" int vdo_renew_static_world(vdo_frame_images* f, int n_tm, const int32_t* tm_sta, const float* stat_x, const float* stat_y,
                                      int n_orb, const float* orb_x, const float* orb_y, int max_num_sta, const float K4[4], const float Twc[16],
                                      float* key_x, float* key_y, float* corr_x, float* corr_y, float* flow_x, float* flow_y,
                                      int32_t* inlier_id, float* depth_out, float* xyz_out, int* n_out) {
  if (!f || !n_out || n_tm < 0 || n_orb < 0 || (xyz_out && (!K4 || !Twc))) return set_error(VDO_ERR_INVALID, "bad argument");
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  Arena S(f->ctx);
  if (!S.reserve(Arena::bytes_for(24 * ((size_t)n_tm + (size_t)n_orb) + 4 * (size_t)max_num_sta))) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  // phase 1 candidates: the inlier static keys, in TM_sta order
  std::vector<float> cx1, cy1; std::vector<int32_t> id1;
  for (int i = 0; i < n_tm; ++i) if (tm_sta[i] != -1) { cx1.push_back(stat_x[tm_sta[i]]); cy1.push_back(stat_y[tm_sta[i]]); id1.push_back(tm_sta[i]); }
  const int n1 = (int)cx1.size();
  std::vector<int32_t> ok1(n1), ok2(n_orb), used2(n_orb, 0);
  std::vector<float> fx1(n1), fy1(n1), d1(n1), fx2(n_orb), fy2(n_orb), d2(n_orb), xyz1(xyz_out ? 3 * (size_t)n1 : 0), xyz2(xyz_out ? 3 * (size_t)n_orb : 0);
  float *dx1 = S.up(cx1.data(), n1), *dy1 = S.up(cy1.data(), n1);
  float *dx2 = S.up(orb_x, n_orb), *dy2 = S.up(orb_y, n_orb);
  int32_t *dok1 = S.up<int32_t>(nullptr, n1), *dsel1 = S.up<int32_t>(nullptr, n1);
  float *dfx1 = S.up<float>(nullptr, n1), *dfy1 = S.up<float>(nullptr, n1), *dd1 = S.up<float>(nullptr, n1), *dxyz1 = S.up<float>(nullptr, xyz_out ? 3 * (size_t)n1 : 0);
  int32_t *dok2 = S.up<int32_t>(nullptr, n_orb), *dused = S.up<int32_t>(nullptr, n_orb);
  float *dfx2 = S.up<float>(nullptr, n_orb), *dfy2 = S.up<float>(nullptr, n_orb), *dd2 = S.up<float>(nullptr, n_orb), *dxyz2 = S.up<float>(nullptr, xyz_out ? 3 * (size_t)n_orb : 0);
  if (!dxyz2 || !dxyz1 || !dd2 || !dd1) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  hipStream_t s = S.stream();
  const Cam cam = xyz_out ? make_cam_Twc(K4, Twc) : Cam{};
  if (n1) {
    hipLaunchKernelGGL(k_renew_pred, dim3((n1 + 255) / 256), dim3(256), 0, s, n1, (const float*)dx1, (const float*)dy1, (const int32_t*)f->d_mask, (const float*)f->d_depth, (const float*)f->d_flow, f->w, f->h, dok1, dfx1, dfy1, dd1);
    hipLaunchKernelGGL(k_carry_select, dim3(1), dim3(1024), 0, s, n1, (const int32_t*)dok1, max_num_sta, dsel1);
    if (xyz_out) hipLaunchKernelGGL(k_backproject_pts, dim3((n1 + 255) / 256), dim3(256), 0, s, n1, (const float*)dx1, (const float*)dy1, (const float*)dd1, cam, 0, dxyz1);
  }
  if (n_orb) {
    hipLaunchKernelGGL(k_renew_pred, dim3((n_orb + 255) / 256), dim3(256), 0, s, n_orb, (const float*)dx2, (const float*)dy2, (const int32_t*)f->d_mask, (const float*)f->d_depth, (const float*)f->d_flow, f->w, f->h, dok2, dfx2, dfy2, dd2);
    launch_near_flags_sel(s, n_orb, dx2, dy2, n1, dx1, dy1, dsel1, 0, dused);
    if (xyz_out) hipLaunchKernelGGL(k_backproject_pts, dim3((n_orb + 255) / 256), dim3(256), 0, s, n_orb, (const float*)dx2, (const float*)dy2, (const float*)dd2, cam, 0, dxyz2);
  }
  S.down(ok1.data(), dok1, n1); S.down(fx1.data(), dfx1, n1); S.down(fy1.data(), dfy1, n1); S.down(d1.data(), dd1, n1);
  S.down(used2.data(), dused, n_orb); S.down(ok2.data(), dok2, n_orb); S.down(fx2.data(), dfx2, n_orb); S.down(fy2.data(), dfy2, n_orb); S.down(d2.data(), dd2, n_orb);
  if (xyz_out) { S.down(xyz1.data(), dxyz1, 3 * (size_t)n1); S.down(xyz2.data(), dxyz2, 3 * (size_t)n_orb); }
  rc = S.finish("vdo_renew_static");
  if (rc != VDO_OK) return rc;
  // carry-over: first-come, stop once size > max (the reference checks after every element, :2703-2709)
  int m = 0;
  for (int i = 0; i < n1; ++i) {
    if (ok1[i]) {
      key_x[m] = cx1[i]; key_y[m] = cy1[i]; corr_x[m] = cx1[i] + fx1[i]; corr_y[m] = cy1[i] + fy1[i];
      flow_x[m] = fx1[i]; flow_y[m] = fy1[i]; inlier_id[m] = id1[i]; depth_out[m] = d1[i];
      if (xyz_out) { xyz_out[3 * m] = xyz1[3 * i]; xyz_out[3 * m + 1] = xyz1[3 * i + 1]; xyz_out[3 * m + 2] = xyz1[3 * i + 2]; }
      ++m;
    }
    if (m > max_num_sta) break;
  }
  if (m < max_num_sta && n_orb) {
    int tot = m, start_id = 0;
    const int step = 20;
    while (tot < max_num_sta) {                       // stride-20 interleave (:2722-2778)
      if (start_id == step) break;
      for (int i = start_id; i < n_orb; i += step) {
        if (used2[i]) continue;
        if (ok2[i]) {
          key_x[m] = orb_x[i]; key_y[m] = orb_y[i]; corr_x[m] = orb_x[i] + fx2[i]; corr_y[m] = orb_y[i] + fy2[i];
          flow_x[m] = fx2[i]; flow_y[m] = fy2[i]; inlier_id[m] = -1; depth_out[m] = d2[i];
          if (xyz_out) { xyz_out[3 * m] = xyz2[3 * i]; xyz_out[3 * m + 1] = xyz2[3 * i + 1]; xyz_out[3 * m + 2] = xyz2[3 * i + 2]; }
          ++m; ++tot;
        }
        if (tot >= max_num_sta) break;
      }
      ++start_id;
    }
  }
  *n_out = m;
  return VDO_OK;
}

extern "C" int vdo_renew_static(vdo_frame_images* f, int n_tm, const int32_t* tm_sta, const float* stat_x, const float* stat_y,
                                int n_orb, const float* orb_x, const float* orb_y, int max_num_sta,
                                float* key_x, float* key_y, float* corr_x, float* corr_y, float* flow_x, float* flow_y,
                                int32_t* inlier_id, float* depth_out, int* n_out) {
  return vdo_renew_static_world(f, n_tm, tm_sta, stat_x, stat_y, n_orb, orb_x, orb_y, max_num_sta, nullptr, nullptr,
                                key_x, key_y, corr_x, corr_y, flow_x, flow_y, inlier_id, depth_out, nullptr, n_out);
}
