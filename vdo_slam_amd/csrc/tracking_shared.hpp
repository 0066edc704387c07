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

// `Rwl * x3Dc + twl` / `mRwc * x3D + mtwc` (src/Frame.cc:511,548, src/Optimizer.cc:2995): a cv::gemm of UNTRANSPOSED CV_32F operands, 3 wide. OpenCV 3.4
// runs such products (flags == 0, 2 <= len <= 4) through the small-matrix fast path at the head of cv::gemm (modules/core/src/matmul.cpp), which
// works in FLOAT: t = a0*b0 + a1*b1 + a2*b2 left to right, d = (float)(t * alpha + c * beta) with alpha = beta = 1.0 - a float addition.  (Round 1-3
// accumulated these in double like the generic GEMMSingleMul<float, double> path; that path is what TRANSPOSED products take - twl = -Rlw.t() * tlw in
// make_cam_Tcw above, Converter::toInvMatrix.  Parity unpinned either way: OpenCV is not in the image; tools/pin_reference settles it.)
__device__ __forceinline__ void backproject(const Cam& c, float u, float v, float z, float* out) {
  const float xc[3] = {(u - c.cx) * z * c.invfx, (v - c.cy) * z * c.invfy, z};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float t = c.R[3 * i] * xc[0] + c.R[3 * i + 1] * xc[1] + c.R[3 * i + 2] * xc[2];      // (-ffp-contract=off: no fused multiply-add)
    out[i] = t + c.t[i];
  }
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
