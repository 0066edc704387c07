// HBM-resident images of one frame + scratch (shared by frame.hip and tracking.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "ctx.hpp"

struct vdo_frame_images {
  vdo_ctx* ctx = nullptr;
  int w = 0, h = 0;
  int32_t* d_mask = nullptr; float *d_depth = nullptr, *d_flow = nullptr;
  // scratch
  float* d_f[8] = {nullptr}; int32_t* d_i[2] = {nullptr}; int* d_cnt = nullptr; int* d_blk = nullptr;
  int cap = 0;
  std::vector<void*> allocs;
};
