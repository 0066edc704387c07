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
  // scratch: ONE allocation of 10 rows x cap (4-byte elements) so that results leave the device in a
  // single strided copy: rows 0-6 = d_f[0..6], row 7 = d_i[0], row 8 = d_f[7], row 9 = d_i[1]
  float* d_rows = nullptr;
  float* d_f[8] = {nullptr}; int32_t* d_i[2] = {nullptr}; int* d_cnt = nullptr; int* d_blk = nullptr;
  int cap = 0;
  unsigned long long* d_cand = nullptr;   // [h][w] UpdateMask: labels whose warp lands on a pixel (bit = label slot); all zero between calls
  int* d_ticket = nullptr;                // UpdateMask: arrival counter of k_votes_par, zero between launches
  float* h_pin = nullptr;          // pinned staging, 8 rows x cap + 16 (count lives at h_pin[8*cap])
  // second scratch set (rows 0-7 + count + pinned staging): K10 next to K9 in one synchronisation (vdo_frame_filters)
  float* d_rows2 = nullptr; int* d_cnt2 = nullptr; float* h_pin2 = nullptr;
  std::vector<void*> allocs;
};
