// Pieces of tracking.hip that tracking_logic.hip's fused object-chain entry point launches too.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdo {

struct Cam { float invfx, invfy, cx, cy; float R[9]; float t[3]; };   // R|t applied to the camera-frame point

inline Cam make_cam_Twc(const float* K4, const float* Twc) {   // Get3DinWorld: R = Twc[:3,:3], t = Twc[:3,3]
  Cam c;
  c.invfx = 1.0f / K4[0]; c.invfy = 1.0f / K4[1]; c.cx = K4[2]; c.cy = K4[3];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) c.R[3 * i + j] = Twc[4 * i + j]; c.t[i] = Twc[4 * i + 3]; }
  return c;
}
inline Cam make_cam_Tcw(const float* K4, const float* Tcw) {   // UnprojectStereo*: Rwl = Rlw^T, twl = -Rlw^T tlw (cv::gemm rounding)
  Cam c;
  c.invfx = 1.0f / K4[0]; c.invfy = 1.0f / K4[1]; c.cx = K4[2]; c.cy = K4[3];
  float nR[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { c.R[3 * i + j] = Tcw[4 * j + i]; nR[3 * i + j] = -Tcw[4 * j + i]; }
  for (int i = 0; i < 3; ++i) c.t[i] = (float)((double)nR[3 * i] * Tcw[3] + (double)nR[3 * i + 1] * Tcw[7] + (double)nR[3 * i + 2] * Tcw[11]);
  return c;
}

// cv::gemm semantics for small float matrices: accumulate in double, round once
__device__ __forceinline__ void gemm3_dev(const float* A, const float* v, float* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = (float)((double)A[3 * i] * v[0] + (double)A[3 * i + 1] * v[1] + (double)A[3 * i + 2] * v[2]);
}
__device__ __forceinline__ void backproject(const Cam& c, float u, float v, float z, float* out) {
  const float xc[3] = {(u - c.cx) * z * c.invfx, (v - c.cy) * z * c.invfy, z};
  float r[3];
  gemm3_dev(c.R, xc, r);
  out[0] = r[0] + c.t[0]; out[1] = r[1] + c.t[1]; out[2] = r[2] + c.t[2];
}
// K12 over a candidate list (one-pass RenewFrameInfo: the 3-D point of every candidate, the host keeps the selected ones);
// as_int: the key is the truncated position (objects, Tracking.cc:2846-2851)
static __global__ void k_backproject_pts(int n, const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ d, Cam c, int as_int,
                                         float* __restrict__ xyz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = kx[i], y = ky[i];
  if (as_int) { x = (float)(int)x; y = (float)(int)y; }
  float o[3];
  backproject(c, x, y, d[i], o);
  xyz[3 * i] = o[0]; xyz[3 * i + 1] = o[1]; xyz[3 * i + 2] = o[2];
}

// defined in tracking.hip
__global__ void k_gather(int mode, int n, const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ depth,
                         const int32_t* __restrict__ mask, int w, int h, float th, float* __restrict__ dout, int32_t* __restrict__ lout);
__global__ void k_scene_flow(int n, const float* __restrict__ cx_, const float* __restrict__ cy_, const float* __restrict__ cd, const int32_t* __restrict__ cl, Cam cc,
                             const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ ld, const int32_t* __restrict__ ll, Cam lc,
                             float* __restrict__ flow, int32_t* __restrict__ olab);

}  // namespace vdo
