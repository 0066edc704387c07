// TEST INFRASTRUCTURE — CPU oracle (see vdo_oracle.h).  Tracking-side stages of the hot path,
// restated sequentially exactly as the reference executes them:
//   K11 propagate last-frame correspondences (depth / label gather)   src/Tracking.cc:259-305
//   K12 back-projection Frame::UnprojectStereoObject / Optimizer::Get3DinWorld   src/Frame.cc:517-555, src/Optimizer.cc:2974-2995
//   K13 scene flow per object point                                    src/Tracking.cc:1278-1364
//   K14 RenewFrameInfo, static part (carry inliers, top-up, depth)     src/Tracking.cc:2660-2790
//   K15 UpdateMask: label gather + mask warp by the previous flow       src/Tracking.cc:3015-3065
// cv::Mat products of small fp32 matrices go through cv::gemm (OpenCV 3.4, modules/core/src/matmul.cpp), on one of two paths:
//   * TRANSPOSED operand (-Rlw.t()*tlw): the generic GEMMSingleMul<float,double> - float inputs accumulated in double, k ascending, times
//     alpha, one rounding to float;
//   * no transposition and 2..4 wide (Rwl*x3Dc + twl, mRwc*x3D + mtwc): the small-matrix fast path at the head of cv::gemm, in FLOAT:
//     t = a0*b0 + a1*b1 + a2*b2 left to right, d = (float)(t*alpha + c*beta) = t + c.
// Restated that way - parity unpinned (OpenCV not available).
#include <cmath>
#include <cstdint>
#include <vector>

#include "vdo_oracle.h"

namespace {
// generic path: A(3x3 float) * v accumulated in double, rounded to float once
inline void gemm3(const float* A, const float* v, float* o) {
  for (int i = 0; i < 3; ++i) o[i] = (float)((double)A[3 * i] * v[0] + (double)A[3 * i + 1] * v[1] + (double)A[3 * i + 2] * v[2]);
}
// fast path (flags == 0, len == 3): o = A * v + c in float, left to right
inline void gemm3_fast_add(const float* A, const float* v, const float* c, float* o) {
  for (int i = 0; i < 3; ++i) {
    const float t = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
    o[i] = (float)((double)t * 1.0 + (double)c[i] * 1.0);
  }
}
}  // namespace

extern "C" void vdo_oracle_propagate_static(int n, const float* kx, const float* ky, const float* depth, int w, int h, float* depth_out) {
  for (int i = 0; i < n; ++i) {
    depth_out[i] = -1;
    const int v = (int)ky[i], u = (int)kx[i];
    if (u < (w - 1) && u > 0 && v < (h - 1) && v > 0) {
      const float d = depth[(size_t)v * w + u];
      if (d > 0) depth_out[i] = d;
    }
  }
}

extern "C" void vdo_oracle_propagate_object(int n, const float* kx, const float* ky, const float* depth, const int32_t* mask, int w, int h,
                                            float th_obj, float* depth_out, int32_t* label_out) {
  for (int i = 0; i < n; ++i) {
    const int u = (int)kx[i], v = (int)ky[i];
    if (u < (w - 1) && u > 0 && v < (h - 1) && v > 0 && depth[(size_t)v * w + u] < th_obj && depth[(size_t)v * w + u] > 0) {
      depth_out[i] = depth[(size_t)v * w + u];
      label_out[i] = mask[(size_t)v * w + u];
    } else {
      depth_out[i] = 0.1f;
      label_out[i] = 0;
    }
  }
}

// Frame::UnprojectStereoObject (Tcw given): Rwl*x3Dc + twl with Rwl = Rlw^T, twl = -Rlw^T*tlw
static void unproject_tcw(float u, float v, float z, const float* K4, const float* Tcw, float* out) {
  const float invfx = 1.0f / K4[0], invfy = 1.0f / K4[1], cx = K4[2], cy = K4[3];
  const float xc[3] = {(u - cx) * z * invfx, (v - cy) * z * invfy, z};
  float Rwl[9], nRwl[9], tlw[3] = {Tcw[3], Tcw[7], Tcw[11]}, twl[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Rwl[3 * i + j] = Tcw[4 * j + i]; nRwl[3 * i + j] = -Tcw[4 * j + i]; }
  gemm3(nRwl, tlw, twl);
  gemm3_fast_add(Rwl, xc, twl, out);
}

// K13: GetSceneFlowObj.  obj_label_inout[i] is set to -1 where either semantic label is <= 0.
extern "C" void vdo_oracle_scene_flow(int n, const float* cur_x, const float* cur_y, const float* cur_d, const int32_t* cur_lab, const float* Tcw_cur,
                                      const float* last_x, const float* last_y, const float* last_d, const int32_t* last_lab, const float* Tcw_last,
                                      const float* K4, float* flow3d, int32_t* obj_label_inout) {
  for (int i = 0; i < n; ++i) {
    if (cur_lab[i] <= 0 || last_lab[i] <= 0) { obj_label_inout[i] = -1; flow3d[3 * i] = flow3d[3 * i + 1] = flow3d[3 * i + 2] = 0; continue; }
    float p[3], c[3];
    unproject_tcw(last_x[i], last_y[i], last_d[i], K4, Tcw_last, p);
    unproject_tcw(cur_x[i], cur_y[i], cur_d[i], K4, Tcw_cur, c);
    for (int k = 0; k < 3; ++k) flow3d[3 * i + k] = c[k] - p[k];
  }
}

// Optimizer::Get3DinWorld(kp, d, K, Twc): mRwc*x3D + mtwc
extern "C" void vdo_oracle_get3d_world(int n, const float* kx, const float* ky, const float* d, const float* K4, const float* Twc, float* xyz) {
  const float invfx = 1.0f / K4[0], invfy = 1.0f / K4[1], cx = K4[2], cy = K4[3];
  float R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = Twc[4 * i + j];
  for (int i = 0; i < n; ++i) {
    const float z = d[i];
    const float xc[3] = {(kx[i] - cx) * z * invfx, (ky[i] - cy) * z * invfy, z};
    const float t[3] = {Twc[3], Twc[7], Twc[11]};
    gemm3_fast_add(R, xc, t, xyz + 3 * i);
  }
}

// K14, static part of RenewFrameInfo (Tracking.cc:2666-2778).  Inputs: inlier list TM_sta over the
// current static keys, the ORB keypoints of the current frame (top-up source) and the images.
// Outputs (in order): key x,y ; corr x,y ; flow ; inlier id (-1 for topped-up) ; depth.  Returns the count.
extern "C" int vdo_oracle_renew_static(int n_tm, const int32_t* tm_sta, const float* stat_x, const float* stat_y,
                                       int n_orb, const float* orb_x, const float* orb_y,
                                       const int32_t* mask, const float* depth, const float* flow, int w, int h, int max_num_sta,
                                       float* key_x, float* key_y, float* corr_x, float* corr_y, float* flow_x, float* flow_y,
                                       int32_t* inlier_id, float* depth_out) {
  int m = 0;
  auto accept = [&](float px, float py, bool strict_src) -> bool {
    const int x = (int)px, y = (int)py;
    if (x >= w || y >= h || x <= 0 || y <= 0) return false;
    const size_t o = (size_t)y * w + x;
    if (mask[o] != 0) return false;
    if (depth[o] > 40 || depth[o] <= 0) return false;
    const float fxe = flow[2 * o], fye = flow[2 * o + 1];
    if (fxe != 0 && fye != 0 && px + fxe < w && py + fye < h && px + fxe > 0 && py + fye > 0) {
      key_x[m] = px; key_y[m] = py; corr_x[m] = px + fxe; corr_y[m] = py + fye; flow_x[m] = fxe; flow_y[m] = fye;
      return true;
    }
    (void)strict_src;
    return false;
  };
  for (int i = 0; i < n_tm; ++i) {
    if (tm_sta[i] == -1) continue;
    if (accept(stat_x[tm_sta[i]], stat_y[tm_sta[i]], true)) { inlier_id[m] = tm_sta[i]; ++m; }
    if (m > max_num_sta) break;
  }
  const int n_check = m;
  int tot = m, start_id = 0;
  const int step = 20;
  while (tot < max_num_sta) {
    if (start_id == step) break;
    for (int i = start_id; i < n_orb; i += step) {
      float min_dist = 100;
      bool used = false;
      for (int j = 0; j < n_check; ++j) {
        const float dx = key_x[j] - orb_x[i], dy = key_y[j] - orb_y[i];
        const float cur = std::sqrt(dx * dx + dy * dy);
        if (cur < min_dist) min_dist = cur;
        if (min_dist < 1.0) { used = true; break; }
      }
      if (used) continue;
      if (accept(orb_x[i], orb_y[i], false)) { inlier_id[m] = -1; ++m; ++tot; }
      if (tot >= max_num_sta) break;
    }
    ++start_id;
  }
  for (int i = 0; i < m; ++i) {
    depth_out[i] = -1;
    const float d = depth[(size_t)((int)key_y[i]) * w + (int)key_x[i]];
    if (d > 0) depth_out[i] = d;
  }
  return m;
}

// K15a: labels of the current mask at the flowed positions of the last frame's object points (-1: outside)
extern "C" void vdo_oracle_mask_at(int n, const float* cx, const float* cy, const int32_t* mask, int w, int h, int32_t* out) {
  for (int i = 0; i < n; ++i) {
    const int u = (int)cx[i], v = (int)cy[i];
    out[i] = (u < w && u > 0 && v < h && v > 0) ? mask[(size_t)v * w + u] : -1;
  }
}

// K15b: warp label `lab` of the previous mask into the current mask with the previous flow (int-truncated)
extern "C" void vdo_oracle_mask_warp(const int32_t* mask_last, const float* flow_last, int w, int h, int32_t lab, int32_t* mask_cur) {
  for (int j = 0; j < h; ++j)
    for (int k = 0; k < w; ++k) {
      const size_t o = (size_t)j * w + k;
      if (mask_last[o] == lab) {
        const int fx = (int)flow_last[2 * o], fy = (int)flow_last[2 * o + 1];
        if (k + fx < w && k + fx > 0 && j + fy < h && j + fy > 0) mask_cur[(size_t)(j + fy) * w + (k + fx)] = lab;
      }
    }
}
