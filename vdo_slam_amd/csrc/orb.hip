// ORB front-end on gfx950 — replaces ORBextractor::operator() (reference src/ORBextractor.cc:1035-1110):
//   K3 ComputePyramid          :1112-1137  cv::resize(INTER_LINEAR, 8u) from the previous level + REFLECT_101 border (19 px)
//   K4 ComputeKeyPointsOctTree :754-818    cv::FAST(cell, thr=iniTh, NMS) with fallback minTh, per 30-px cell (+6 px overlap)
//   K5 DistributeOctTree       :470-752    quadtree, best response per node      (host C++, sequential by nature)
//   K6 IC_Angle                :66-93      31x31 circular patch moments + cv::fastAtan2
//   K7 GaussianBlur 7x7 s=2    :1083-1084  executed by the reference although its consumer (BRIEF) is commented out
//   K8 computeOrbDescriptor    :97-136     256 rotated pair tests on the blurred level (call site commented out :1083-1091;
//                                          here on request: vdo_orb_descriptors / vdo_orb_extract_desc)
// Image-sized work is HBM/L2-bound integer arithmetic: tiles are staged in LDS (cell ROI 37x37,
// blur tile + 3-px halo), one workgroup per FAST cell over ALL pyramid levels in a single launch,
// candidates are compacted in raster order with a workgroup scan so that the reference's feature
// order — and therefore every downstream feature index — is reproduced exactly.
// OpenCV 3.4 fixed-point semantics are restated (parity unpinned).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <cstring>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"
#include "host_pool.hpp"
#include "orb_pattern.hpp"

namespace vdo {

constexpr int kEdge = 19, kHalfPatch = 15, kPatch = 31;
constexpr int kSpecCand = 24576;       // candidates fetched speculatively together with the header (typ. 10-15 k per KITTI frame)
constexpr int kCellCap = 160;          // max keypoints kept per FAST cell (NMS => <= ~(37/2)^2/2)
constexpr int kMaxCellDim = 68;        // hCell = ceil(height/nRows) < 60, +6 overlap

struct LevelDesc { int w, h, bw, bh; int64_t off; int64_t off_inner; int64_t off_blur; };   // bordered image at img + off; blurred interior (w x h, contiguous) at blur + off_blur
struct CellDesc { int level, x0, y0, x1, y1, addx, addy, pad; };          // ROI in level interior coords; add = j*wCell, i*hCell

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * (len - 1) - p; }
  return p;
}

// ------------------------------------------------------------------------------------ K2
__global__ void k_rgb2gray(const uint8_t* __restrict__ src, int64_t n, int ch, int rgb_order, uint8_t* __restrict__ dst) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = src + i * ch;
  const int r = rgb_order ? p[0] : p[2], g = p[1], b = rgb_order ? p[2] : p[0];
  dst[i] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14);
}

// K1: d<0 -> 0 ; else bf / (d / factor)       (src/Tracking.cc:180-204)
__global__ void k_depth_preprocess(float* __restrict__ d, int64_t n, float bf, float factor) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = d[i];
  d[i] = v < 0 ? 0.f : bf / (v / factor);
}

// ------------------------------------------------------------------------------------ K3
// level 0: copy + border.
__global__ void k_border_copy(const uint8_t* __restrict__ src, int sw, int sh, int sstride, uint8_t* __restrict__ dst, int bw, int bh) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= bw || y >= bh) return;
  dst[(size_t)y * bw + x] = src[(size_t)reflect101(y - kEdge, sh) * sstride + reflect101(x - kEdge, sw)];
}
// level l>0: bilinear (11-bit fixed point) from the previous level's interior, border by reflection.
__global__ void k_resize_border(const uint8_t* __restrict__ src /*interior of level l-1*/, int sw, int sh, int sstride,
                                uint8_t* __restrict__ dst, int dw, int dh, double scale_x, double scale_y) {
  const int bx = blockIdx.x * blockDim.x + threadIdx.x, by = blockIdx.y * blockDim.y + threadIdx.y;
  const int bw = dw + 2 * kEdge, bh = dh + 2 * kEdge;
  if (bx >= bw || by >= bh) return;
  const int dx = reflect101(bx - kEdge, dw), dy = reflect101(by - kEdge, dh);
  float fx = (float)((dx + 0.5) * scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= sx;
  bool xedge = false;
  if (sx < 0) { fx = 0; sx = 0; }
  if (sx + 1 >= sw) { xedge = true; fx = 0; sx = sw - 1; }
  const int a0 = max(-32768, min(32767, (int)rintf((1.f - fx) * 2048))), a1 = max(-32768, min(32767, (int)rintf(fx * 2048)));
  float fy = (float)((dy + 0.5) * scale_y - 0.5);
  int sy = (int)floorf(fy);
  fy -= sy;
  const int b0 = max(-32768, min(32767, (int)rintf((1.f - fy) * 2048))), b1 = max(-32768, min(32767, (int)rintf(fy * 2048)));
  const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
  const uint8_t* S0 = src + (size_t)sy0 * sstride;
  const uint8_t* S1 = src + (size_t)sy1 * sstride;
  int r0, r1;
  if (!xedge) { r0 = S0[sx] * a0 + S0[sx + 1] * a1; r1 = S1[sx] * a0 + S1[sx + 1] * a1; }
  else { r0 = S0[sx] * 2048; r1 = S1[sx] * 2048; }
  dst[(size_t)by * bw + bx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

// The whole pyramid in ONE launch.  A level is resized from the level below, so eight dependent launches used to build it (~60 us of
// stream time, 8 launch overheads on the host).  Here every workgroup owns one 32x32 tile of one level and recomputes, in LDS, the
// regions of the levels below that feed it, cascading up from the source image: level 0 region -> level 1 region -> ... -> the
// tile.  Every pixel of every level is computed with the same fixed-point formula from the same exact inputs as in the
// level-by-level version (k_resize_border), so the result is bit-identical; the recomputation (a level-7 tile needs a ~130x130
// source window) costs a few microseconds of an otherwise idle GPU.  The 19-px REFLECT_101 border is written by the owner of
// the mirrored interior pixel.
struct PyrDesc {
  int n_levels;                 // levels of the pyramid (0: this launch form is not used)
  int base, lv0, lv1;           // this launch builds levels [lv0, lv1) from the image of level `base` (0: the source image)
  int w[8], h[8];
  double sx[8], sy[8];          // scale of level l relative to level l-1 (inv of cv::resize's inv_scale), [0] unused
  int tile_off[9];              // first workgroup of every level
  int tiles_x[8];
};
constexpr int kPyrRegion = 96;    // max edge of a cascaded region (checked on the host at create time)

__device__ __forceinline__ void resize_src(int d, double scale, int slen, int& s0, int& s1, int& c0, int& c1) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int si = (int)floorf(f);
  f -= si;
  bool edge = false;
  if (si < 0) { f = 0; si = 0; }
  if (si + 1 >= slen) { edge = true; f = 0; si = slen - 1; }
  c0 = max(-32768, min(32767, (int)rintf((1.f - f) * 2048))); c1 = max(-32768, min(32767, (int)rintf(f * 2048)));
  s0 = si; s1 = edge ? si : si + 1;
  if (edge) { c0 = 2048; c1 = 0; }
}
// vertical taps of cv::resize are NOT collapsed at the border: both rows are clamped (k_resize_border does the same)
__device__ __forceinline__ void resize_src_y(int d, double scale, int slen, int& s0, int& s1, int& c0, int& c1) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int si = (int)floorf(f);
  f -= si;
  c0 = max(-32768, min(32767, (int)rintf((1.f - f) * 2048))); c1 = max(-32768, min(32767, (int)rintf(f * 2048)));
  s0 = min(max(si, 0), slen - 1); s1 = min(max(si + 1, 0), slen - 1);
}

__global__ __launch_bounds__(256) void k_pyramid_all(const uint8_t* __restrict__ src, int sstride, PyrDesc P, const LevelDesc* __restrict__ levels, uint8_t* __restrict__ pyr) {
  __shared__ uint8_t bufA[kPyrRegion * kPyrRegion];
  __shared__ uint8_t bufB[kPyrRegion * kPyrRegion];
  __shared__ int rx0[8], rx1[8], ry0[8], ry1[8];
  int lvl = P.lv0;
  while (lvl + 1 < P.lv1 && (int)blockIdx.x >= P.tile_off[lvl + 1]) ++lvl;
  const int t = blockIdx.x - P.tile_off[lvl], ty = t / P.tiles_x[lvl], tx = t - ty * P.tiles_x[lvl];
  if (threadIdx.x == 0) {
    // inclusive pixel ranges needed at every level, from the tile down to the source
    int x0 = tx * 32, x1 = min(x0 + 31, P.w[lvl] - 1), y0 = ty * 32, y1 = min(y0 + 31, P.h[lvl] - 1);
    rx0[lvl] = x0; rx1[lvl] = x1; ry0[lvl] = y0; ry1[lvl] = y1;
    for (int k = lvl; k > P.base; --k) {
      int a0, a1, c0, c1, b0, b1;
      resize_src(x0, P.sx[k], P.w[k - 1], a0, a1, c0, c1);
      resize_src(x1, P.sx[k], P.w[k - 1], b0, b1, c0, c1);
      x0 = a0; x1 = b1;
      resize_src_y(y0, P.sy[k], P.h[k - 1], a0, a1, c0, c1);
      resize_src_y(y1, P.sy[k], P.h[k - 1], b0, b1, c0, c1);
      y0 = a0; y1 = b1;
      rx0[k - 1] = x0; rx1[k - 1] = x1; ry0[k - 1] = y0; ry1[k - 1] = y1;
    }
  }
  __syncthreads();
  // region of the base level from its image (the source, or a level an earlier launch wrote)
  uint8_t* cur = bufA;
  uint8_t* nxt = bufB;
  {
    const int bs = P.base;
    const int w0 = rx1[bs] - rx0[bs] + 1, h0 = ry1[bs] - ry0[bs] + 1;
    for (int i = threadIdx.x; i < w0 * h0; i += 256) { const int yy = i / w0, xx = i - yy * w0; cur[yy * kPyrRegion + xx] = src[(size_t)(ry0[bs] + yy) * sstride + rx0[bs] + xx]; }
  }
  __syncthreads();
  for (int k = P.base + 1; k <= lvl; ++k) {
    const int wk = rx1[k] - rx0[k] + 1, hk = ry1[k] - ry0[k] + 1;
    const int ox = rx0[k - 1], oy = ry0[k - 1];
    for (int i = threadIdx.x; i < wk * hk; i += 256) {
      const int yy = i / wk, xx = i - yy * wk;
      int sx0, sx1, a0, a1, sy0, sy1, b0, b1;
      resize_src(rx0[k] + xx, P.sx[k], P.w[k - 1], sx0, sx1, a0, a1);
      resize_src_y(ry0[k] + yy, P.sy[k], P.h[k - 1], sy0, sy1, b0, b1);
      const uint8_t* S0 = cur + (sy0 - oy) * kPyrRegion - ox;
      const uint8_t* S1 = cur + (sy1 - oy) * kPyrRegion - ox;
      const int r0 = S0[sx0] * a0 + S0[sx1] * a1, r1 = S1[sx0] * a0 + S1[sx1] * a1;
      nxt[yy * kPyrRegion + xx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
    }
    __syncthreads();
    uint8_t* tmp = cur; cur = nxt; nxt = tmp;
  }
  // the tile (interior) + the border pixels that mirror its pixels (REFLECT_101: p <-> -p and p <-> 2(len-1) - p, 19 px wide)
  const LevelDesc L = levels[lvl];
  uint8_t* dst = pyr + L.off;
  const int wt = rx1[lvl] - rx0[lvl] + 1, ht = ry1[lvl] - ry0[lvl] + 1;
  for (int i = threadIdx.x; i < wt * ht; i += 256) {
    const int yy = i / wt, xx = i - yy * wt;
    const int x = rx0[lvl] + xx, y = ry0[lvl] + yy;
    const uint8_t v = cur[yy * kPyrRegion + xx];
    int xs[2] = {x, 0}, ys[2] = {y, 0};
    bool mx = false, my = false;
    if (x >= 1 && x <= kEdge) { xs[1] = -x; mx = true; } else if (x <= L.w - 2 && x >= L.w - 1 - kEdge) { xs[1] = 2 * (L.w - 1) - x; mx = true; }
    if (y >= 1 && y <= kEdge) { ys[1] = -y; my = true; } else if (y <= L.h - 2 && y >= L.h - 1 - kEdge) { ys[1] = 2 * (L.h - 1) - y; my = true; }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if ((a == 0 || my) && (b == 0 || mx)) dst[(size_t)(ys[a] + kEdge) * L.bw + xs[b] + kEdge] = v;
  }
}

// ------------------------------------------------------------------------------------ K4
__device__ __forceinline__ int fast_score_lds(const uint8_t* roi, int stride, int x, int y, int t) {
  const uint8_t* c = roi + y * stride + x;
  const int v = c[0];
  int r[16];
  r[0] = c[3 * stride]; r[1] = c[3 * stride + 1]; r[2] = c[2 * stride + 2]; r[3] = c[stride + 3];
  r[4] = c[3]; r[5] = c[-stride + 3]; r[6] = c[-2 * stride + 2]; r[7] = c[-3 * stride + 1];
  r[8] = c[-3 * stride]; r[9] = c[-3 * stride - 1]; r[10] = c[-2 * stride - 2]; r[11] = c[-stride - 3];
  r[12] = c[-3]; r[13] = c[stride - 3]; r[14] = c[2 * stride - 2]; r[15] = c[3 * stride - 1];
  // cheap necessary condition (opposite pairs): a 9-arc contains one pixel of every opposite pair
  int best = -1000;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    int mb = 1000, md = 1000;
#pragma unroll
    for (int k = 0; k < 9; ++k) { const int q = r[(s + k) & 15]; mb = min(mb, q - v); md = min(md, v - q); }
    best = max(best, max(mb, md));
  }
  return best > t ? best - 1 : 0;
}

// One workgroup per cell.  out_cnt[cell], out_pack[cell*kCellCap + k] = x | y<<12 | score<<24 (level coords rel. (16,16))
__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t* __restrict__ pyr, const LevelDesc* __restrict__ levels,
                                                    const CellDesc* __restrict__ cells, int ini_th, int min_th,
                                                    int* __restrict__ out_cnt, uint32_t* __restrict__ out_pack) {
  __shared__ uint8_t roi[kMaxCellDim * kMaxCellDim];
  __shared__ uint8_t sc[kMaxCellDim * kMaxCellDim];
  __shared__ int s_wave[4], s_total;
  const CellDesc C = cells[blockIdx.x];
  const LevelDesc L = levels[C.level];
  const int w = C.x1 - C.x0, h = C.y1 - C.y0, tid = threadIdx.x;
  const uint8_t* img = pyr + L.off_inner;     // interior origin, row stride L.bw
  for (int i = tid; i < w * h; i += 256) { const int y = i / w, x = i - y * w; roi[y * kMaxCellDim + x] = img[(size_t)(C.y0 + y) * L.bw + (C.x0 + x)]; }
  __syncthreads();
  const int npix = w * h;
  const int per = (npix + 255) / 256;          // contiguous raster chunk per thread -> ordered compaction
  int t = ini_th;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = tid; i < npix; i += 256) {
      const int y = i / w, x = i - y * w;
      int s = 0;
      if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3 && w >= 7 && h >= 7) s = fast_score_lds(roi, kMaxCellDim, x, y, t);
      sc[y * kMaxCellDim + x] = (uint8_t)s;
    }
    __syncthreads();
    // NMS (strict > against the 8 neighbours) + count in this thread's raster chunk
    unsigned keep = 0;
    int cnt = 0;
    for (int k = 0; k < per; ++k) {
      const int i = tid * per + k;
      if (i >= npix) break;
      const int y = i / w, x = i - y * w;
      const int s = sc[y * kMaxCellDim + x];
      if (s == 0) continue;
      const uint8_t* p = sc + y * kMaxCellDim + x;
      if (s > p[-1] && s > p[1] && s > p[-kMaxCellDim - 1] && s > p[-kMaxCellDim] && s > p[-kMaxCellDim + 1] &&
          s > p[kMaxCellDim - 1] && s > p[kMaxCellDim] && s > p[kMaxCellDim + 1]) { keep |= 1u << k; ++cnt; }
    }
    // workgroup exclusive scan of cnt
    const int lane = tid & 63, wv = tid >> 6;
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    if (tid == 0) { int a = 0; for (int q = 0; q < 4; ++q) { const int v = s_wave[q]; s_wave[q] = a; a += v; } s_total = a; }
    __syncthreads();
    const int total = s_total;
    if (total > 0 || pass == 1) {
      int pos = s_wave[wv] + incl - cnt;
      for (int k = 0; k < per; ++k) {
        if (!(keep & (1u << k))) continue;
        const int i = tid * per + k;
        const int y = i / w, x = i - y * w;
        if (pos < kCellCap)
          out_pack[(size_t)blockIdx.x * kCellCap + pos] = (uint32_t)(x + C.addx) | ((uint32_t)(y + C.addy) << 12) | ((uint32_t)sc[y * kMaxCellDim + x] << 24);
        ++pos;
      }
      if (tid == 0) out_cnt[blockIdx.x] = total;
      return;
    }
    t = min_th;          // vKeysCell.empty() -> retry the whole cell with minThFAST
    __syncthreads();
  }
}

// umax of the circular patch (ORBextractor.cc:443-458), filled on the host
struct UMax { int v[kHalfPatch + 2]; };

// cv::fastAtan2, degrees
__device__ __forceinline__ float fast_atan2_dev(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI);
  const float p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) { c = ay / (ax + (float)2.2204460492503131e-16); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
  else { c = ax / (ay + (float)2.2204460492503131e-16); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// Compaction to dense arrays + K6 angle for every candidate: one wave per candidate (31 rows over the lanes).
// dense: x,y (float, relative to (16,16)), resp, angle, level
__global__ __launch_bounds__(256) void k_compact_angle(const uint8_t* __restrict__ pyr, const LevelDesc* __restrict__ levels,
                                                       const CellDesc* __restrict__ cells, const int* __restrict__ cnt, const int* __restrict__ cell_level,
                                                       int* __restrict__ level_cnt, const uint32_t* __restrict__ pack, int ncells, UMax um,
                                                       float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oresp,
                                                       float* __restrict__ oang, int* __restrict__ olevel) {
  const int cell = blockIdx.x;
  const int n = min(cnt[cell], kCellCap);
  const int lvl = cells[cell].level;
  const LevelDesc L = levels[lvl];
  const uint8_t* img = pyr + L.off_inner;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // dense offset of this cell = number of candidates in the cells before it (every workgroup sums its own prefix: ~1100 counts,
  // L2-resident - cheaper than a separate single-workgroup scan kernel in front of this one); workgroup 0 also writes the header
  // (per-level counts + total) the host fetches next to the candidates
  __shared__ int s_part[4], s_lvl[16];
  int base;
  {
    int a = 0;
    for (int c = threadIdx.x; c < cell; c += 256) a += min(cnt[c], kCellCap);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if (lane == 0) s_part[wv] = a;
    if (cell == 0 && threadIdx.x < 16) s_lvl[threadIdx.x] = 0;
    __syncthreads();
    base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (cell == 0) {
      for (int c = threadIdx.x; c < ncells; c += 256) { const int v = min(cnt[c], kCellCap); if (v) atomicAdd(&s_lvl[cell_level[c]], v); }
      __syncthreads();
      if (threadIdx.x < 16) level_cnt[threadIdx.x] = s_lvl[threadIdx.x];
      if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < 16; ++q) t += s_lvl[q]; level_cnt[16] = t; }
    }
  }
  for (int k = wv; k < n; k += 4) {
    const uint32_t pk = pack[(size_t)cell * kCellCap + k];
    const int x = (int)(pk & 0xfff), y = (int)((pk >> 12) & 0xfff), s = (int)(pk >> 24);
    const int cx = x + (kEdge - 3), cy = y + (kEdge - 3);       // cvRound of integer-valued coordinates
    // rows v = -15..15 -> lanes 0..30
    int m10 = 0, m01 = 0;
    if (lane < 31) {
      const int v = lane - kHalfPatch;
      const int d = um.v[v < 0 ? -v : v];
      const uint8_t* row = img + (size_t)(cy + v) * L.bw + cx;
      int rs = 0;
      for (int u = -d; u <= d; ++u) { const int p = row[u]; m10 += u * p; rs += p; }
      m01 = v * rs;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { m10 += __shfl_down(m10, off, 64); m01 += __shfl_down(m01, off, 64); }
    if (lane == 0) {
      const int o = base + k;
      ox[o] = (float)x; oy[o] = (float)y; oresp[o] = (float)s; olevel[o] = lvl;
      oang[o] = fast_atan2_dev((float)m01, (float)m10);
    }
  }
}

// ------------------------------------------------------------------------------------ K7
// 7x7 Gaussian, sigma 2, 8-bit fixed-point separable (OpenCV 3.4.0 path), REFLECT_101 at the interior edge.
struct Blur7 { int k[7]; };
// ------------------------------------------------------------------------------------ K8
// computeOrbDescriptor (reference src/ORBextractor.cc:97-136): for each of the 256 learned point pairs, rotated by the
// keypoint angle, compare two pixels of the blurred level image: bit = I(p0) < I(p1); pixel index =
// cvRound(x*b + y*a)*step + cvRound(x*a - y*b) around cvRound(kp) with a = cos, b = sin (fp32 products and sums, no
// contraction).  One wave per keypoint: lane i evaluates tests 4i..4i+3 (one 16-byte load of its slice of the pattern),
// neighbouring lanes merge their nibbles into a byte, even lanes write the 32 bytes.
// cos/sin: evaluated in double from +,-,* only and rounded to float (= correctly rounded cosf/sinf up to double-rounding
// ties); the oracle runs the same operations, so the descriptor BITS agree although ocml and glibc sinf/cosf do not.
struct OrbPattern { int4 q[64]; };

__device__ __forceinline__ void sincos_exact_dev(float angle, float* s_out, float* c_out) {
  const double x = (double)angle;
  const int k = (int)(x * 0.63661977236758138 + 0.5);
  const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
  const double z = r * r;
  const double sp = r * (1.0 + z * (-1.0 / 6 + z * (1.0 / 120 + z * (-1.0 / 5040 + z * (1.0 / 362880 + z * (-1.0 / 39916800 + z * (1.0 / 6227020800.0 +
                    z * (-1.0 / 1307674368000.0 + z * (1.0 / 355687428096000.0)))))))));
  const double cp = 1.0 + z * (-1.0 / 2 + z * (1.0 / 24 + z * (-1.0 / 720 + z * (1.0 / 40320 + z * (-1.0 / 3628800 + z * (1.0 / 479001600 +
                    z * (-1.0 / 87178291200.0 + z * (1.0 / 20922789888000.0))))))));
  double sn, cs;
  switch (k & 3) {
    case 0: sn = sp; cs = cp; break;
    case 1: sn = cp; cs = -sp; break;
    case 2: sn = -sp; cs = -cp; break;
    default: sn = -cp; cs = sp; break;
  }
  *s_out = (float)sn; *c_out = (float)cs;
}

__global__ __launch_bounds__(256) void k_orb_desc(const uint8_t* __restrict__ blur, const LevelDesc* __restrict__ levels, const int* __restrict__ sel, int n,
                                                  const float* __restrict__ dx, const float* __restrict__ dy, const float* __restrict__ dang,
                                                  const int* __restrict__ dlvl, const OrbPattern* __restrict__ pat, uint8_t* __restrict__ desc) {
  const int kp = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (kp >= n) return;
  const int id = sel[kp];
  const LevelDesc L = levels[dlvl[id]];
  const int cx = (int)dx[id] + (kEdge - 3), cy = (int)dy[id] + (kEdge - 3);       // cvRound of integer-valued level coordinates
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float a, b;
  sincos_exact_dev(dang[id] * factorPI, &b, &a);
  const uint8_t* center = blur + L.off_blur + (size_t)cy * L.w + cx;
  const int4 q = pat->q[lane];                      // 16 signed bytes: (x0,y0,x1,y1) of 4 tests
  const int words[4] = {q.x, q.y, q.z, q.w};
  int nib = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float x0 = (float)(signed char)(words[t] & 0xff), y0 = (float)(signed char)((words[t] >> 8) & 0xff);
    const float x1 = (float)(signed char)((words[t] >> 16) & 0xff), y1 = (float)(signed char)((words[t] >> 24) & 0xff);
    const int t0 = center[__float2int_rn(x0 * b + y0 * a) * L.w + __float2int_rn(x0 * a - y0 * b)];
    const int t1 = center[__float2int_rn(x1 * b + y1 * a) * L.w + __float2int_rn(x1 * a - y1 * b)];
    nib |= (t0 < t1) << t;
  }
  const int hi = __shfl_down(nib, 1, 64);
  if ((lane & 1) == 0) desc[(size_t)kp * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
}

// all pyramid levels in ONE launch: blockIdx.x -> (level, tile) through the per-level tile offsets
struct BlurTiles { int off[17]; int tiles_x[16]; };
__global__ __launch_bounds__(256) void k_blur7_all(const uint8_t* __restrict__ pyr, const LevelDesc* __restrict__ levels, int n_levels, BlurTiles bt, Blur7 K,
                                                   uint8_t* __restrict__ blur) {
  int lvl = 0;
  while (lvl + 1 < n_levels && (int)blockIdx.x >= bt.off[lvl + 1]) ++lvl;
  const LevelDesc L = levels[lvl];
  const int t = blockIdx.x - bt.off[lvl], by = t / bt.tiles_x[lvl], bx = t - by * bt.tiles_x[lvl];
  const uint8_t* __restrict__ src = pyr + L.off_inner;
  uint8_t* __restrict__ dst = blur + L.off_blur;
  const int w = L.w, h = L.h, sstride = L.bw;
  __shared__ uint8_t tile[38][40];
  __shared__ int hbuf[38][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8 threads, tile 32x32
  const int x0 = bx * 32, y0 = by * 32;
  for (int i = threadIdx.x; i < 38 * 38; i += 256) {
    const int yy = i / 38, xx = i - yy * 38;
    tile[yy][xx] = src[(size_t)reflect101(y0 + yy - 3, h) * sstride + reflect101(x0 + xx - 3, w)];
  }
  __syncthreads();
  for (int yy = ty; yy < 38; yy += 8) {
    int a = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) a += K.k[i] * tile[yy][tx + i];
    hbuf[yy][tx] = a;
  }
  __syncthreads();
  for (int yy = ty; yy < 32; yy += 8) {
    const int x = x0 + tx, y = y0 + yy;
    if (x >= w || y >= h) continue;
    int a = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) a += K.k[i] * hbuf[yy + i][tx];
    const int v = (a + (1 << 15)) >> 16;
    dst[(size_t)y * w + x] = (uint8_t)min(255, max(0, v));
  }
}

// --------------------------------------------------------------------------- host: quadtree (K5)
struct QNode {
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  int b = 0, e = 0;         // keys = idx[b, e): every node owns a contiguous run of the shared index array
  int prev = -1, next = -1; // intrusive list (front = head)
  bool noMore = false;
};

// DistributeOctTree (reference src/ORBextractor.cc:528-752, DivideNode :470-526) with an index-based
// intrusive list (same traversal/insert order as the reference's std::list: children are pushed to the
// FRONT, parents erased).  Ties in the "expand largest first" phase are broken by node creation order
// (the reference breaks them by heap address).  Allocation-free in steady state: the keys
// of a node are a run of one index array, DivideNode is a stable 4-way partition of that run
// (children keep the parent's key order, like the reference's push_back loops), buffers are reused
// across levels and frames.
class QuadTree {
 public:
  // candidates as SoA (x, y relative to the level's (minX, minY); resp)
  void run(const float* cx, const float* cy, const float* cresp, int ncand, int minX, int maxX, int minY, int maxY, int N, std::vector<int>& out) {
    out.clear();
    if (ncand <= 0) return;
    x = cx; y = cy; resp = cresp;
    nodes.clear(); head = tail = -1; count = 0;
    idx.resize(ncand); tmp.resize(ncand); cls.resize(ncand);
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
    const float hX = (float)(maxX - minX) / nIni;
    // initial nodes: stable bucket sort of the candidates by column strip
    ini_cnt.assign(nIni + 1, 0);
    for (int k = 0; k < ncand; ++k) { const int q = (int)(x[k] / hX); cls[k] = (uint8_t)q; ini_cnt[q + 1]++; }
    for (int i = 0; i < nIni; ++i) ini_cnt[i + 1] += ini_cnt[i];
    ini_fill.assign(ini_cnt.begin(), ini_cnt.end() - 1);
    for (int k = 0; k < ncand; ++k) idx[ini_fill[cls[k]]++] = k;
    for (int i = 0; i < nIni; ++i) {
      QNode n;
      n.ULx = (int)(hX * (float)i); n.ULy = 0; n.URx = (int)(hX * (float)(i + 1)); n.URy = 0;
      n.BLx = n.ULx; n.BLy = maxY - minY; n.BRx = n.URx; n.BRy = maxY - minY;
      n.b = ini_cnt[i]; n.e = ini_cnt[i + 1];
      push_back(n);
    }
    for (int it = head; it != -1;) {
      const int nx = nodes[it].next;
      const int sz = nodes[it].e - nodes[it].b;
      if (sz == 1) nodes[it].noMore = true;
      else if (sz == 0) erase(it);
      it = nx;
    }
    bool finish = false;
    while (!finish) {
      int prevSize = count;
      int nToExpand = 0;
      sizeAndNode.clear();
      for (int it = head; it != -1;) {
        if (nodes[it].noMore) { it = nodes[it].next; continue; }
        const int nx = nodes[it].next;
        divide_and_add(it, &nToExpand);
        erase(it);
        it = nx;
      }
      if (count >= N || count == prevSize) finish = true;
      else if (count + nToExpand * 3 > N) {
        while (!finish) {
          prevSize = count;
          prevSN.swap(sizeAndNode);
          sizeAndNode.clear();
          std::sort(prevSN.begin(), prevSN.end());
          for (int j = (int)prevSN.size() - 1; j >= 0; --j) {
            divide_and_add(prevSN[j].second, nullptr);
            erase(prevSN[j].second);
            if (count >= N) break;
          }
          if (count >= N || count == prevSize) finish = true;
        }
      }
    }
    for (int it = head; it != -1; it = nodes[it].next) {
      const QNode& n = nodes[it];
      int best = idx[n.b];
      float mr = resp[best];
      for (int k = n.b + 1; k < n.e; ++k) if (resp[idx[k]] > mr) { best = idx[k]; mr = resp[best]; }
      out.push_back(best);
    }
  }

 private:
  const float *x = nullptr, *y = nullptr, *resp = nullptr;
  std::vector<int> idx, tmp, ini_cnt, ini_fill;
  std::vector<uint8_t> cls;
  std::vector<QNode> nodes;
  std::vector<std::pair<int, int>> sizeAndNode, prevSN;   // (size, node id); id order == creation order
  int head = -1, tail = -1, count = 0;
  int push_back(const QNode& n) {
    nodes.push_back(n);
    const int id = (int)nodes.size() - 1;
    nodes[id].prev = tail; nodes[id].next = -1;
    if (tail != -1) nodes[tail].next = id; else head = id;
    tail = id; ++count;
    return id;
  }
  int push_front(const QNode& n) {
    nodes.push_back(n);
    const int id = (int)nodes.size() - 1;
    nodes[id].prev = -1; nodes[id].next = head;
    if (head != -1) nodes[head].prev = id; else tail = id;
    head = id; ++count;
    return id;
  }
  void erase(int id) {
    const QNode n = nodes[id];
    if (n.prev != -1) nodes[n.prev].next = n.next; else head = n.next;
    if (n.next != -1) nodes[n.next].prev = n.prev; else tail = n.prev;
    --count;
  }
  void divide_and_add(int id, int* nToExpand) {
    QNode c[4];
    const QNode n = nodes[id];
    const int halfX = (int)std::ceil((float)(n.URx - n.ULx) / 2), halfY = (int)std::ceil((float)(n.BRy - n.ULy) / 2);
    QNode &n1 = c[0], &n2 = c[1], &n3 = c[2], &n4 = c[3];
    n1.ULx = n.ULx; n1.ULy = n.ULy; n1.URx = n.ULx + halfX; n1.URy = n.ULy; n1.BLx = n.ULx; n1.BLy = n.ULy + halfY; n1.BRx = n.ULx + halfX; n1.BRy = n.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = n.URx; n2.URy = n.URy; n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = n.URx; n2.BRy = n.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy; n3.BLx = n.BLx; n3.BLy = n.BLy; n3.BRx = n1.BRx; n3.BRy = n.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy; n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = n.BRx; n4.BRy = n.BRy;
    int cnt[4] = {0, 0, 0, 0};
    const float sx = (float)n1.URx, sy = (float)n1.BRy;      // kp.pt.x < n1.UR.x compares float with int -> float
    for (int k = n.b; k < n.e; ++k) {
      const int ki = idx[k];
      const int q = (x[ki] < sx) ? ((y[ki] < sy) ? 0 : 2) : ((y[ki] < sy) ? 1 : 3);
      cls[k] = (uint8_t)q; ++cnt[q];
    }
    int cur[4];
    cur[0] = n.b; cur[1] = cur[0] + cnt[0]; cur[2] = cur[1] + cnt[1]; cur[3] = cur[2] + cnt[2];
    for (int q = 0; q < 4; ++q) { c[q].b = cur[q]; c[q].e = cur[q] + cnt[q]; c[q].noMore = cnt[q] == 1; }
    for (int k = n.b; k < n.e; ++k) tmp[cur[cls[k]]++] = idx[k];
    std::memcpy(idx.data() + n.b, tmp.data() + n.b, sizeof(int) * (size_t)(n.e - n.b));
    for (int q = 0; q < 4; ++q) {
      if (cnt[q] == 0) continue;
      const int nid = push_front(c[q]);
      if (cnt[q] > 1) {
        if (nToExpand) ++*nToExpand;
        sizeAndNode.push_back(std::make_pair(cnt[q], nid));
      }
    }
  }
};


}  // namespace vdo

using namespace vdo;

struct vdo_orb {
  vdo_ctx* ctx = nullptr;
  vdo_orb_params prm{};
  int w = 0, h = 0, ncells = 0;
  std::vector<LevelDesc> levels;
  std::vector<CellDesc> cells;
  std::vector<int> cell_level, nfeat;
  std::vector<float> scale;
  UMax um{};
  Blur7 blur{};
  PyrDesc pyr{};                 // cascaded pyramid launches (k_pyramid_all); n_levels == 0: level-by-level launches
  PyrDesc pyr2{};                // second cascade (upper levels from the top level of the first); lv1 == lv0: not used
  // device
  std::vector<void*> allocs;
  uint8_t *d_src = nullptr, *d_pyr = nullptr, *d_blur = nullptr;
  LevelDesc* d_levels = nullptr; CellDesc* d_cells = nullptr; int* d_cell_level = nullptr;
  int *d_cnt = nullptr, *d_offs = nullptr, *d_level_cnt = nullptr; uint32_t* d_pack = nullptr;
  float *d_x = nullptr, *d_y = nullptr, *d_resp = nullptr, *d_ang = nullptr; int* d_lvl = nullptr;
  int64_t pyr_bytes = 0, blur_bytes = 0;
  int dense_cap = 0;
  // K8 (on request): dense ids of the keypoints of the last extraction, device pattern / selection / descriptor rows
  std::vector<int> sel_dense;
  OrbPattern* d_pat = nullptr; int* d_sel = nullptr; uint8_t* d_desc = nullptr; int desc_cap = 0;
  // pinned staging: [32 ints: level counts, total][5][kSpecCand] — header and candidates arrive with ONE sync
  float* h_pin = nullptr;
  float* h_over = nullptr; size_t h_over_n = 0;      // overflow staging when a frame has more than kSpecCand candidates
  // host view of the last extraction (pointers into the pinned staging)
  const float *hx = nullptr, *hy = nullptr, *hresp = nullptr, *hang = nullptr; std::vector<int> hlevel_cnt;
  int n_cand = 0;
  QuadTree qt[16];                                    // one per level: buffers reused across frames
  std::vector<int> sel[16];
  std::unique_ptr<LevelPool> pool;                    // helpers for the per-level quadtrees (VDO_ORB_THREADS, default 3)
  double ms_device = 0, ms_tree = 0;                  // last extraction: launch..sync, host quadtree
  bool begun = false;                                 // between vdo_orb_extract_begin and _end
  hipEvent_t ev_cand = nullptr;                       // candidates are on the host (the blur stage runs behind it)
  std::chrono::steady_clock::time_point t_begin;
};

extern "C" int vdo_orb_pyramid_launches(const vdo_orb* o) {
  if (!o) return set_error(VDO_ERR_INVALID, "null handle");
  return o->pyr.n_levels ? (o->pyr2.lv1 > o->pyr2.lv0 ? 2 : 1) : (int)o->levels.size();
}

extern "C" int vdo_orb_destroy(vdo_orb* o) {
  if (!o) return VDO_OK;
  if (o->ctx) ctx_bind(o->ctx);
  for (void* p : o->allocs) hipFree(p);
  if (o->h_pin) hipHostFree(o->h_pin);
  if (o->ev_cand) hipEventDestroy(o->ev_cand);
  if (o->h_over) hipHostFree(o->h_over);
  delete o;
  return VDO_OK;
}

extern "C" int vdo_orb_create(vdo_ctx* ctx, const vdo_orb_params* prm, int w, int h, vdo_orb** out) {
  if (!ctx || !prm || !out || w < 64 || h < 64 || prm->n_levels < 1 || prm->n_levels > 16 || !(prm->scale_factor > 1.f))
    return set_error(VDO_ERR_INVALID, "vdo_orb_create: bad argument");
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  vdo_orb* o = new vdo_orb();
  o->ctx = ctx; o->prm = *prm; o->w = w; o->h = h;
  const int NL = prm->n_levels;
  // scale tables, level sizes (ORBextractor ctor :403-419, ComputePyramid :1116-1117)
  o->scale.resize(NL);
  o->scale[0] = 1.0f;
  for (int i = 1; i < NL; ++i) o->scale[i] = o->scale[i - 1] * prm->scale_factor;
  o->levels.resize(NL);
  int64_t off = 0, boff = 0;
  for (int l = 0; l < NL; ++l) {
    const float inv = 1.0f / o->scale[l];
    LevelDesc& L = o->levels[l];
    L.w = (int)lrintf((float)w * inv); L.h = (int)lrintf((float)h * inv);
    L.bw = L.w + 2 * kEdge; L.bh = L.h + 2 * kEdge;
    L.off = off; L.off_inner = off + (int64_t)kEdge * L.bw + kEdge; L.off_blur = boff;
    off += (int64_t)L.bw * L.bh;
    boff += (int64_t)L.w * L.h;
    if (L.w < 2 * kEdge + 8 || L.h < 2 * kEdge + 8) { delete o; return set_error(VDO_ERR_UNSUPPORTED, "pyramid level %d too small (%dx%d)", l, L.w, L.h); }
  }
  o->pyr_bytes = off; o->blur_bytes = boff;
  // features per level (:424-435)
  o->nfeat.resize(NL);
  {
    float factor = 1.0f / prm->scale_factor;
    float nd = prm->n_features * (1 - factor) / (1 - (float)std::pow((double)factor, (double)NL));
    int sum = 0;
    for (int l = 0; l < NL - 1; ++l) { o->nfeat[l] = (int)lrintf(nd); sum += o->nfeat[l]; nd *= factor; }
    o->nfeat[NL - 1] = std::max(prm->n_features - sum, 0);
  }
  // FAST cells (:760-796)
  for (int l = 0; l < NL; ++l) {
    const LevelDesc& L = o->levels[l];
    const float W = 30;
    const int minBX = kEdge - 3, minBY = minBX, maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols <= 0 || nRows <= 0) continue;
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    for (int i = 0; i < nRows; ++i) {
      const float iniY = (float)(minBY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBY - 3) continue;
      if (maxY > maxBY) maxY = (float)maxBY;
      for (int j = 0; j < nCols; ++j) {
        const float iniX = (float)(minBX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBX - 6) continue;
        if (maxX > maxBX) maxX = (float)maxBX;
        CellDesc c{l, (int)iniX, (int)iniY, (int)maxX, (int)maxY, j * wCell, i * hCell, 0};
        if (c.x1 - c.x0 > kMaxCellDim || c.y1 - c.y0 > kMaxCellDim) { delete o; return set_error(VDO_ERR_INTERNAL, "FAST cell larger than %d px", kMaxCellDim); }
        o->cells.push_back(c);
        o->cell_level.push_back(l);
      }
    }
  }
  o->ncells = (int)o->cells.size();
  // fused pyramid: possible when there are at most 8 levels and every cascaded region fits the LDS buffers
  if (NL <= 8 && !std::getenv("VDO_ORB_PYRAMID_LAUNCHES")) {
    // Two cascades: levels [0, split] from the source image, levels (split, NL) from level `split` - a tile of the top level
    // would otherwise recompute a ~140 x 140 source window through seven levels (58 us for the one-launch form, 8 launches x
    // ~6 us level by level; two launches: the deepest cascade is four levels)
    const int split = NL > 5 ? 4 : NL - 1;
    PyrDesc& P = o->pyr;
    P.n_levels = NL; P.base = 0; P.lv0 = 0; P.lv1 = split + 1;
    for (int l = 0; l < NL; ++l) {
      P.w[l] = o->levels[l].w; P.h[l] = o->levels[l].h;
      P.sx[l] = l ? 1. / ((double)o->levels[l].w / o->levels[l - 1].w) : 1.0;
      P.sy[l] = l ? 1. / ((double)o->levels[l].h / o->levels[l - 1].h) : 1.0;
      P.tiles_x[l] = (P.w[l] + 31) / 32;
    }
    PyrDesc& Q = o->pyr2;
    Q = P;
    Q.base = split; Q.lv0 = split + 1; Q.lv1 = NL;
    for (PyrDesc* D : {&P, &Q}) {
      int tot = 0;
      for (int l = 0; l <= NL; ++l) D->tile_off[l] = 0;
      for (int l = D->lv0; l < D->lv1; ++l) { D->tile_off[l] = tot; tot += D->tiles_x[l] * ((D->h[l] + 31) / 32); }
      for (int l = D->lv1; l <= NL; ++l) D->tile_off[l] = tot;
    }
    // widest region any tile needs at any level below it: the kernel's own cascade (same arithmetic) over every tile column / row
    auto src_range = [](int d0, int d1, double scale, int slen, bool clamp_both, int& s0, int& s1) {
      auto tap = [&](int d, int& lo, int& hi) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int si = (int)std::floor(f);
        if (clamp_both) { lo = std::min(std::max(si, 0), slen - 1); hi = std::min(std::max(si + 1, 0), slen - 1); return; }
        if (si < 0) si = 0;
        if (si + 1 >= slen) { lo = hi = slen - 1; return; }
        lo = si; hi = si + 1;
      };
      int a, b;
      tap(d0, s0, a); tap(d1, b, s1);
    };
    int worst = 32;
    for (const PyrDesc* D : {&P, &Q}) {
      for (int l = std::max(D->lv0, D->base + 1); l < D->lv1; ++l) {
        for (int axis = 0; axis < 2; ++axis) {
          const int len = axis ? D->h[l] : D->w[l];
          for (int t0 = 0; t0 < len; t0 += 32) {
            int d0 = t0, d1 = std::min(t0 + 31, len - 1);
            for (int k = l; k > D->base; --k) {
              int s0, s1;
              src_range(d0, d1, axis ? D->sy[k] : D->sx[k], axis ? D->h[k - 1] : D->w[k - 1], axis == 1, s0, s1);
              d0 = s0; d1 = s1;
              worst = std::max(worst, d1 - d0 + 1);
            }
          }
        }
      }
    }
    if (worst > kPyrRegion) P.n_levels = 0;
  }
  // umax (:443-458)
  {
    int* umax = o->um.v;
    int v, v0;
    const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) umax[v] = (int)lrint(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) { while (umax[v0] == umax[v0 + 1]) ++v0; umax[v] = v0; ++v0; }
  }
  // Gaussian kernel 7, sigma 2 -> 8-bit fixed point
  {
    double kd[7], sum = 0;
    for (int i = 0; i < 7; ++i) { const double x = i - 3; kd[i] = std::exp(-0.5 / 4.0 * x * x); sum += kd[i]; }
    for (int i = 0; i < 7; ++i) o->blur.k[i] = (int)lrint((double)(float)(kd[i] / sum) * 256);
  }
  hipStream_t s = ctx->stream;
  auto dev = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) return nullptr; o->allocs.push_back(p); return p; };
  o->dense_cap = o->ncells * kCellCap;
  o->d_src = (uint8_t*)dev((size_t)w * h * 4);
  o->d_pyr = (uint8_t*)dev(o->pyr_bytes); o->d_blur = (uint8_t*)dev(o->blur_bytes);
  o->d_levels = (LevelDesc*)dev(sizeof(LevelDesc) * NL); o->d_cells = (CellDesc*)dev(sizeof(CellDesc) * o->ncells);
  o->d_cell_level = (int*)dev(4 * (size_t)o->ncells);
  o->d_pat = (OrbPattern*)dev(sizeof(OrbPattern));
  o->d_cnt = (int*)dev(4 * (size_t)o->ncells); o->d_offs = (int*)dev(4 * ((size_t)o->ncells + 1)); o->d_level_cnt = (int*)dev(4 * 32);
  o->d_pack = (uint32_t*)dev(4 * (size_t)o->dense_cap);
  o->d_x = (float*)dev(4 * (size_t)o->dense_cap * 5);       // x | y | resp | angle | level: rows of one allocation (single strided D2H)
  if (o->d_x) {
    o->d_y = o->d_x + (size_t)o->dense_cap; o->d_resp = o->d_x + 2 * (size_t)o->dense_cap; o->d_ang = o->d_x + 3 * (size_t)o->dense_cap;
    o->d_lvl = (int*)(o->d_x + 4 * (size_t)o->dense_cap);
  }
  if (hipHostMalloc((void**)&o->h_pin, 4 * (32 + 5 * (size_t)kSpecCand)) != hipSuccess) o->h_pin = nullptr;
  for (void* p : o->allocs) if (!p) { vdo_orb_destroy(o); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  if (!o->d_lvl || !o->h_pin) { vdo_orb_destroy(o); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  hipMemcpyAsync(o->d_levels, o->levels.data(), sizeof(LevelDesc) * NL, hipMemcpyHostToDevice, s);
  hipMemcpyAsync(o->d_cells, o->cells.data(), sizeof(CellDesc) * o->ncells, hipMemcpyHostToDevice, s);
  hipMemcpyAsync(o->d_cell_level, o->cell_level.data(), 4 * (size_t)o->ncells, hipMemcpyHostToDevice, s);
  static_assert(sizeof(kOrbPattern31) == sizeof(OrbPattern), "256 tests x 4 signed bytes");
  hipMemcpyAsync(o->d_pat, kOrbPattern31, sizeof(OrbPattern), hipMemcpyHostToDevice, s);
  if (hipStreamSynchronize(s) != hipSuccess) { vdo_orb_destroy(o); return set_error(VDO_ERR_NO_DEVICE, "orb upload failed"); }
  if (hipEventCreateWithFlags(&o->ev_cand, hipEventDisableTiming) != hipSuccess) { vdo_orb_destroy(o); return set_error(VDO_ERR_NO_DEVICE, "hipEventCreate failed"); }
  {
    const char* e = std::getenv("VDO_ORB_THREADS");
    const int nw = e ? std::atoi(e) : 3;
    if (nw > 0) o->pool.reset(new LevelPool(std::min(nw, 15)));
  }
  *out = o;
  return VDO_OK;
}

// Device part of operator(): pyramid, blur, FAST cells, compaction + angles.  gray_dev: device, row stride `stride`.
static int orb_device_stage(vdo_orb* o, const uint8_t* gray_dev, int stride) {
  hipStream_t s = o->ctx->stream;
  const int NL = o->prm.n_levels;
  const dim3 tb(32, 8);
  if (o->pyr.n_levels) {
    hipLaunchKernelGGL(k_pyramid_all, dim3(o->pyr.tile_off[o->pyr.n_levels]), dim3(256), 0, s, gray_dev, stride, o->pyr, (const LevelDesc*)o->d_levels, o->d_pyr);
    if (o->pyr2.lv1 > o->pyr2.lv0) {
      const LevelDesc& B = o->levels[o->pyr2.base];
      hipLaunchKernelGGL(k_pyramid_all, dim3(o->pyr2.tile_off[o->pyr2.n_levels]), dim3(256), 0, s, (const uint8_t*)(o->d_pyr + B.off_inner), B.bw, o->pyr2, (const LevelDesc*)o->d_levels, o->d_pyr);
    }
  } else {
    {
      const LevelDesc& L = o->levels[0];
      hipLaunchKernelGGL(k_border_copy, dim3((L.bw + 31) / 32, (L.bh + 7) / 8), tb, 0, s, gray_dev, L.w, L.h, stride, o->d_pyr + L.off, L.bw, L.bh);
    }
    for (int l = 1; l < NL; ++l) {
      const LevelDesc &P = o->levels[l - 1], &L = o->levels[l];
      const double sx = 1. / ((double)L.w / P.w), sy = 1. / ((double)L.h / P.h);
      hipLaunchKernelGGL(k_resize_border, dim3((L.bw + 31) / 32, (L.bh + 7) / 8), tb, 0, s, (const uint8_t*)(o->d_pyr + P.off_inner), P.w, P.h, P.bw,
                         o->d_pyr + L.off, L.w, L.h, sx, sy);
    }
  }
  hipLaunchKernelGGL(k_fast_cells, dim3(o->ncells), dim3(256), 0, s, (const uint8_t*)o->d_pyr, (const LevelDesc*)o->d_levels, (const CellDesc*)o->d_cells,
                     o->prm.ini_th, o->prm.min_th, o->d_cnt, o->d_pack);
  hipLaunchKernelGGL(k_compact_angle, dim3(o->ncells), dim3(256), 0, s, (const uint8_t*)o->d_pyr, (const LevelDesc*)o->d_levels, (const CellDesc*)o->d_cells,
                     (const int*)o->d_cnt, (const int*)o->d_cell_level, o->d_level_cnt, (const uint32_t*)o->d_pack, o->ncells, o->um, o->d_x, o->d_y, o->d_resp, o->d_ang, o->d_lvl);
  return VDO_OK;
}

// K7: the reference blurs every non-empty level (result unused there since BRIEF is commented out, F1).  Nothing downstream
// waits for it, so it is queued BEHIND the copy of the candidates: the host starts the quadtrees ~45 us earlier.
static void orb_blur_stage(vdo_orb* o) {
  hipStream_t s = o->ctx->stream;
  BlurTiles bt{};
  int tot = 0;
  for (int l = 0; l < o->prm.n_levels; ++l) {
    const LevelDesc& L = o->levels[l];
    bt.off[l] = tot; bt.tiles_x[l] = (L.w + 31) / 32;
    tot += bt.tiles_x[l] * ((L.h + 31) / 32);
  }
  bt.off[o->prm.n_levels] = tot;
  hipLaunchKernelGGL(k_blur7_all, dim3(tot), dim3(256), 0, s, (const uint8_t*)o->d_pyr, (const LevelDesc*)o->d_levels, o->prm.n_levels, bt, o->blur, o->d_blur);
}

// operator() in two halves: _begin queues the device stage (pyramid, FAST cells, compaction + angles, blur) and the copy of the
// candidates on the extractor's stream and returns; _end waits for them and runs the quadtrees (K5).  vdo_orb_extract = both.
extern "C" int vdo_orb_extract_begin(vdo_orb* o, const uint8_t* gray, int stride, int src_is_device) {
  if (!o || !gray) return set_error(VDO_ERR_INVALID, "vdo_orb_extract_begin: null argument");
  int rc = ctx_bind(o->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = o->ctx->stream;
  const uint8_t* src = gray;
  int sstride = stride;
  if (!src_is_device) {
    // (a 2-D copy from pageable memory is executed row by row - 3 ms for a KITTI image; contiguous rows go up as one 1-D copy)
    if (stride == o->w) hipMemcpyAsync(o->d_src, gray, (size_t)o->w * o->h, hipMemcpyHostToDevice, s);
    else hipMemcpy2DAsync(o->d_src, o->w, gray, stride, o->w, o->h, hipMemcpyHostToDevice, s);
    src = o->d_src; sstride = o->w;
  }
  o->t_begin = std::chrono::steady_clock::now();
  orb_device_stage(o, src, sstride);
  // candidates -> host: header (level counts + total) and the first kSpecCand columns of the 5 candidate rows
  int* hdr = (int*)o->h_pin;
  float* rows = o->h_pin + 32;
  const int spec = std::min(kSpecCand, o->dense_cap);
  hipMemcpyAsync(hdr, o->d_level_cnt, 4 * 32, hipMemcpyDeviceToHost, s);
  hipMemcpy2DAsync(rows, 4 * (size_t)spec, o->d_x, 4 * (size_t)o->dense_cap, 4 * (size_t)spec, 5, hipMemcpyDeviceToHost, s);
  hipEventRecord(o->ev_cand, s);
  orb_blur_stage(o);
  o->begun = true;
  return VDO_OK;
}

extern "C" int vdo_orb_extract(vdo_orb* o, const uint8_t* gray, int stride, int src_is_device, vdo_keypoints* out) {
  int rc = vdo_orb_extract_begin(o, gray, stride, src_is_device);
  return rc != VDO_OK ? rc : vdo_orb_extract_end(o, out);
}

extern "C" int vdo_orb_extract_end(vdo_orb* o, vdo_keypoints* out) {
  if (!o || !out) return set_error(VDO_ERR_INVALID, "vdo_orb_extract_end: null argument");
  if (!o->begun) return set_error(VDO_ERR_INVALID, "vdo_orb_extract_end without vdo_orb_extract_begin");
  o->begun = false;
  int rc = ctx_bind(o->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = o->ctx->stream;
  const auto t_begin = o->t_begin;
  int* hdr = (int*)o->h_pin;
  float* rows = o->h_pin + 32;
  const int spec = std::min(kSpecCand, o->dense_cap);
  if (hipEventSynchronize(o->ev_cand) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "orb device stage failed: %s", hipGetErrorString(hipGetLastError()));
  const int total = hdr[16];
  o->n_cand = total;
  o->hlevel_cnt.assign(hdr, hdr + 16);
  size_t pitch = (size_t)spec;
  if (total > spec) {           // rare: fetch everything into a (grown on demand) pinned overflow buffer
    if (o->h_over_n < (size_t)total) {
      if (o->h_over) hipHostFree(o->h_over);
      o->h_over = nullptr; o->h_over_n = 0;
      if (hipHostMalloc((void**)&o->h_over, 4 * 5 * (size_t)total) != hipSuccess) return set_error(VDO_ERR_OOM, "hipHostMalloc failed");
      o->h_over_n = (size_t)total;
    }
    hipMemcpy2DAsync(o->h_over, 4 * (size_t)total, o->d_x, 4 * (size_t)o->dense_cap, 4 * (size_t)total, 5, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "orb D2H failed");
    rows = o->h_over; pitch = (size_t)total;
  }
  o->hx = rows; o->hy = rows + pitch; o->hresp = rows + 2 * pitch; o->hang = rows + 3 * pitch;
  const auto t_dev = std::chrono::steady_clock::now();
  // K5 on the host: one quadtree per level (candidates of a level are contiguous: cells are level-major), levels in parallel
  const int NL = o->prm.n_levels;
  int lvl_pos[17];
  lvl_pos[0] = 0;
  for (int l = 0; l < NL; ++l) lvl_pos[l + 1] = lvl_pos[l] + o->hlevel_cnt[l];
  const int minB = kEdge - 3;
  auto level_task = [&](int l) {
    const LevelDesc& L = o->levels[l];
    const int pos = lvl_pos[l];
    o->qt[l].run(o->hx + pos, o->hy + pos, o->hresp + pos, o->hlevel_cnt[l], minB, L.w - kEdge + 3, minB, L.h - kEdge + 3, o->nfeat[l], o->sel[l]);
  };
  if (o->pool) o->pool->run(NL, level_task);
  else for (int l = 0; l < NL; ++l) level_task(l);
  int n = 0;
  o->sel_dense.clear();
  for (int l = 0; l < NL; ++l) {
    const int pos = lvl_pos[l];
    const float *cx = o->hx + pos, *cy = o->hy + pos, *cr = o->hresp + pos, *ca = o->hang + pos;
    const int patch = (int)(kPatch * o->scale[l]);
    for (int id : o->sel[l]) {
      o->sel_dense.push_back(pos + id);
      if (n >= out->capacity) return set_error(VDO_ERR_INVALID, "vdo_orb_extract: keypoint capacity %d too small", out->capacity);
      float x = cx[id] + minB, y = cy[id] + minB;
      if (l != 0) { x = x * o->scale[l]; y = y * o->scale[l]; }
      out->x[n] = x; out->y[n] = y; out->response[n] = cr[id]; out->angle[n] = ca[id];
      out->octave[n] = l; out->size[n] = (float)patch;
      ++n;
    }
  }
  out->n = n;
  const auto t_end = std::chrono::steady_clock::now();
  o->ms_device = std::chrono::duration<double, std::milli>(t_dev - t_begin).count();
  o->ms_tree = std::chrono::duration<double, std::milli>(t_end - t_dev).count();
  return VDO_OK;
}

// K8: descriptors of the keypoints of the last extraction, in their output order.  desc32: host, [n][32].
extern "C" int vdo_orb_descriptors(vdo_orb* o, uint8_t* desc32, int capacity_rows) {
  if (!o || !desc32) return set_error(VDO_ERR_INVALID, "vdo_orb_descriptors: null argument");
  const int n = (int)o->sel_dense.size();
  if (n > capacity_rows) return set_error(VDO_ERR_INVALID, "vdo_orb_descriptors: %d keypoints, room for %d", n, capacity_rows);
  if (n == 0) return VDO_OK;
  int rc = ctx_bind(o->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = o->ctx->stream;
  if (n > o->desc_cap) {
    const int cap = std::max(n, 4096);
    int* ns = nullptr; uint8_t* nd = nullptr;
    if (hipMalloc((void**)&ns, 4 * (size_t)cap) != hipSuccess || hipMalloc((void**)&nd, 32 * (size_t)cap) != hipSuccess) { if (ns) hipFree(ns); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
    hipStreamSynchronize(s);
    auto drop = [&](void* p) { if (!p) return; hipFree(p); o->allocs.erase(std::remove(o->allocs.begin(), o->allocs.end(), p), o->allocs.end()); };
    drop(o->d_sel); drop(o->d_desc);
    o->d_sel = ns; o->d_desc = nd; o->desc_cap = cap;
    o->allocs.push_back(ns); o->allocs.push_back(nd);
  }
  // the blurred levels of this extraction are queued on the same stream (orb_blur_stage), the dense candidate arrays are still resident
  hipMemcpyAsync(o->d_sel, o->sel_dense.data(), 4 * (size_t)n, hipMemcpyHostToDevice, s);
  hipLaunchKernelGGL(k_orb_desc, dim3((n + 3) / 4), dim3(256), 0, s, (const uint8_t*)o->d_blur, (const LevelDesc*)o->d_levels, (const int*)o->d_sel, n,
                     (const float*)o->d_x, (const float*)o->d_y, (const float*)o->d_ang, (const int*)o->d_lvl, (const OrbPattern*)o->d_pat, o->d_desc);
  hipMemcpyAsync(desc32, o->d_desc, 32 * (size_t)n, hipMemcpyDeviceToHost, s);
  if (hipStreamSynchronize(s) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "orb descriptors: %s", hipGetErrorString(hipGetLastError()));
  return VDO_OK;
}

// operator() with both of its outputs: keypoints and (desc32 != NULL) the 32-byte descriptors, [out->capacity][32]
extern "C" int vdo_orb_extract_desc(vdo_orb* o, const uint8_t* gray, int stride, int src_is_device, vdo_keypoints* out, uint8_t* desc32) {
  int rc = vdo_orb_extract(o, gray, stride, src_is_device, out);
  if (rc != VDO_OK || !desc32) return rc;
  return vdo_orb_descriptors(o, desc32, out->capacity);
}

// Wall time of the last vdo_orb_extract: [0] launch of the device stage .. candidates on the host, [1] host quadtree (K5)
extern "C" int vdo_orb_last_timing(vdo_orb* o, double ms[2]) {
  if (!o || !ms) return set_error(VDO_ERR_INVALID, "null argument");
  ms[0] = o->ms_device; ms[1] = o->ms_tree;
  return VDO_OK;
}

extern "C" int vdo_orb_level_info(vdo_orb* o, int level, int* w, int* h, int* n_features, int* n_candidates) {
  if (!o || level < 0 || level >= o->prm.n_levels) return set_error(VDO_ERR_INVALID, "bad level");
  if (w) *w = o->levels[level].w;
  if (h) *h = o->levels[level].h;
  if (n_features) *n_features = o->nfeat[level];
  if (n_candidates) *n_candidates = o->hlevel_cnt.empty() ? 0 : o->hlevel_cnt[level];
  return VDO_OK;
}

// mvImagePyramid[level] with its 19-px border: (w+38) x (h+38) bytes
extern "C" int vdo_orb_get_pyramid(vdo_orb* o, int level, uint8_t* out_bordered) {
  if (!o || !out_bordered || level < 0 || level >= o->prm.n_levels) return set_error(VDO_ERR_INVALID, "bad argument");
  int rc = ctx_bind(o->ctx);
  if (rc != VDO_OK) return rc;
  const LevelDesc& L = o->levels[level];
  if (hipMemcpy(out_bordered, o->d_pyr + L.off, (size_t)L.bw * L.bh, hipMemcpyDeviceToHost) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "D2H failed");
  return VDO_OK;
}

extern "C" int vdo_orb_get_blurred(vdo_orb* o, int level, uint8_t* out) {
  if (!o || !out || level < 0 || level >= o->prm.n_levels) return set_error(VDO_ERR_INVALID, "bad argument");
  int rc = ctx_bind(o->ctx);
  if (rc != VDO_OK) return rc;
  int64_t boff = 0;
  for (int l = 0; l < level; ++l) boff += (int64_t)o->levels[l].w * o->levels[l].h;
  const LevelDesc& L = o->levels[level];
  hipStreamSynchronize(o->ctx->stream);                 // the blur stage runs behind the extraction's own synchronisation point
  if (hipMemcpy(out, o->d_blur + boff, (size_t)L.w * L.h, hipMemcpyDeviceToHost) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "D2H failed");
  return VDO_OK;
}

// FAST candidates of the last extraction for one level (reference push order; relative to (16,16))
extern "C" int vdo_orb_get_candidates(vdo_orb* o, int level, float* x, float* y, float* resp, float* angle, int cap, int* n) {
  if (!o || !n || level < 0 || level >= o->prm.n_levels || o->hlevel_cnt.empty()) return set_error(VDO_ERR_INVALID, "bad argument / no extraction yet");
  int pos = 0;
  for (int l = 0; l < level; ++l) pos += o->hlevel_cnt[l];
  const int cnt = o->hlevel_cnt[level];
  *n = cnt;
  for (int k = 0; k < cnt && k < cap; ++k) {
    if (x) x[k] = o->hx[pos + k];
    if (y) y[k] = o->hy[pos + k];
    if (resp) resp[k] = o->hresp[pos + k];
    if (angle) angle[k] = o->hang[pos + k];
  }
  return VDO_OK;
}

extern "C" int vdo_depth_preprocess(vdo_ctx* ctx, float* depth, int64_t n, float bf, float factor, int is_device) {
  if (!ctx || !depth || n <= 0) return set_error(VDO_ERR_INVALID, "bad argument");
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ctx->stream;
  float* d = depth;
  if (!is_device) {
    if (hipMalloc((void**)&d, 4 * (size_t)n) != hipSuccess) return set_error(VDO_ERR_OOM, "hipMalloc failed");
    hipMemcpyAsync(d, depth, 4 * (size_t)n, hipMemcpyHostToDevice, s);
  }
  hipLaunchKernelGGL(k_depth_preprocess, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, n, bf, factor);
  if (!is_device) {
    hipMemcpyAsync(depth, d, 4 * (size_t)n, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    hipFree(d);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "depth preprocess: %s", hipGetErrorString(e));
  return VDO_OK;
}

extern "C" int vdo_rgb2gray(vdo_ctx* ctx, const uint8_t* rgb, int64_t n_pixels, int channels, int rgb_order, uint8_t* gray) {
  if (!ctx || !rgb || !gray || n_pixels <= 0 || (channels != 3 && channels != 4)) return set_error(VDO_ERR_INVALID, "bad argument");
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ctx->stream;
  uint8_t *ds = nullptr, *dd = nullptr;
  if (hipMalloc((void**)&ds, (size_t)n_pixels * channels) != hipSuccess || hipMalloc((void**)&dd, (size_t)n_pixels) != hipSuccess) { if (ds) hipFree(ds); return set_error(VDO_ERR_OOM, "hipMalloc failed"); }
  hipMemcpyAsync(ds, rgb, (size_t)n_pixels * channels, hipMemcpyHostToDevice, s);
  hipLaunchKernelGGL(k_rgb2gray, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, s, (const uint8_t*)ds, n_pixels, channels, rgb_order, dd);
  hipMemcpyAsync(gray, dd, (size_t)n_pixels, hipMemcpyDeviceToHost, s);
  hipError_t e = hipStreamSynchronize(s);
  hipFree(ds); hipFree(dd);
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "rgb2gray: %s", hipGetErrorString(e));
  return VDO_OK;
}
