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


// defined in tracking.hip
__global__ void k_gather(int mode, int n, const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ depth,
                         const int32_t* __restrict__ mask, int w, int h, float th, float* __restrict__ dout, int32_t* __restrict__ lout);
__global__ void k_scene_flow(int n, const float* __restrict__ cx_, const float* __restrict__ cy_, const float* __restrict__ cd, const int32_t* __restrict__ cl, Cam cc,
                             const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ ld, const int32_t* __restrict__ ll, Cam lc,
                             float* __restrict__ flow, int32_t* __restrict__ olab);

}  // namespace vdo
