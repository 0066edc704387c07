// TEST INFRASTRUCTURE — CPU oracle (see vdo_oracle.h).  Front-end of the hot path:
//   K1  depth preprocess                 src/Tracking.cc:180-204
//   K2  RGB->gray                        src/Tracking.cc:209-222  (cv::cvtColor, OpenCV 3.4 fixed point)
//   K3  ORB pyramid                      src/ORBextractor.cc:1112-1137 (cv::resize INTER_LINEAR 8u, copyMakeBorder REFLECT_101)
//   K4  FAST-9/16 + NMS per 30-px cell   src/ORBextractor.cc:754-818   (cv::FAST(..., true))
//   K5  quadtree distribution            src/ORBextractor.cc:470-752
//   K6  intensity-centroid angle         src/ORBextractor.cc:66-93, 443-468 (cv::fastAtan2)
//   K7  7x7 sigma=2 Gaussian blur        src/ORBextractor.cc:1083-1084 (cv::GaussianBlur 8u)
//   K9  static keypoint filter           src/Frame.cc:100-128,178-194
//   K10 semi-dense object sampling       src/Frame.cc:201-228
// PARITY UNPINNED for K2/K3/K4/K6/K7: those semantics live in OpenCV 3.4.0, which is neither
// vendored in /root/reference nor installed here; they are restated from the published
// algorithms (fixed-point 11-bit bilinear resize, FAST corner score = max over the 16 arcs of
// the minimum absolute difference minus one, 3x3 strict NMS, 8-bit fixed-point separable blur,
// degree polynomial fastAtan2, cvRound = round-half-even).  K5's tie order depends on heap
// addresses in the reference; ties are broken here by node creation order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <vector>

#include "vdo_oracle.h"

namespace {

inline int cv_round_f(float v) { return (int)lrintf(v); }   // round-half-even (default FP mode)
inline int cv_round_d(double v) { return (int)lrint(v); }

const int EDGE_THRESHOLD = 19, PATCH_SIZE = 31, HALF_PATCH_SIZE = 15;

struct Img { int w = 0, h = 0; std::vector<uint8_t> d; uint8_t at(int y, int x) const { return d[(size_t)y * w + x]; } };

int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * (len - 1) - p; }
  return p;
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1 (imgproc/resize.cpp, generic path)
void resize_linear_8u(const Img& src, Img& dst) {
  const int sw = src.w, sh = src.h, dw = dst.w, dh = dst.h;
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(2 * (size_t)dw), ibeta(2 * (size_t)dh);
  int xmax = dw;
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)std::floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) { xmax = std::min(xmax, dx); fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    ialpha[2 * dx] = (short)std::max(-32768, std::min(32767, cv_round_f((1.f - fx) * 2048)));
    ialpha[2 * dx + 1] = (short)std::max(-32768, std::min(32767, cv_round_f(fx * 2048)));
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)std::floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[2 * dy] = (short)std::max(-32768, std::min(32767, cv_round_f((1.f - fy) * 2048)));
    ibeta[2 * dy + 1] = (short)std::max(-32768, std::min(32767, cv_round_f(fy * 2048)));
  }
  std::vector<int> row0(dw), row1(dw);
  auto hresize = [&](int sy, std::vector<int>& out) {
    const uint8_t* S = &src.d[(size_t)sy * sw];
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = xofs[dx];
      if (dx < xmax) out[dx] = S[sx] * ialpha[2 * dx] + S[sx + 1] * ialpha[2 * dx + 1];
      else out[dx] = S[sx] * 2048;
    }
  };
  for (int dy = 0; dy < dh; ++dy) {
    const int sy0 = std::min(std::max(yofs[dy], 0), sh - 1), sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
    hresize(sy0, row0);
    hresize(sy1, row1);
    const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    for (int dx = 0; dx < dw; ++dx)
      dst.d[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
  }
}

void level_sizes(const vdo_orb_params& p, int w, int h, std::vector<int>& ws, std::vector<int>& hs, std::vector<float>& scale) {
  ws.resize(p.n_levels); hs.resize(p.n_levels); scale.resize(p.n_levels);
  scale[0] = 1.0f;
  for (int i = 1; i < p.n_levels; ++i) scale[i] = scale[i - 1] * p.scale_factor;     // mvScaleFactor
  for (int l = 0; l < p.n_levels; ++l) {
    const float inv = 1.0f / scale[l];                                             // mvInvScaleFactor
    ws[l] = cv_round_f((float)w * inv);
    hs[l] = cv_round_f((float)h * inv);
  }
}

void features_per_level(const vdo_orb_params& p, std::vector<int>& n) {   // ORBextractor.cc:424-435
  n.resize(p.n_levels);
  float factor = 1.0f / p.scale_factor;
  float nd = p.n_features * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.n_levels));
  int sum = 0;
  for (int l = 0; l < p.n_levels - 1; ++l) { n[l] = cv_round_f(nd); sum += n[l]; nd *= factor; }
  n[p.n_levels - 1] = std::max(p.n_features - sum, 0);
}

void build_pyramid(const uint8_t* gray, int w, int h, const vdo_orb_params& p, std::vector<Img>& lv) {
  std::vector<int> ws, hs; std::vector<float> sc;
  level_sizes(p, w, h, ws, hs, sc);
  lv.resize(p.n_levels);
  lv[0].w = w; lv[0].h = h; lv[0].d.assign(gray, gray + (size_t)w * h);
  for (int l = 1; l < p.n_levels; ++l) {
    lv[l].w = ws[l]; lv[l].h = hs[l]; lv[l].d.resize((size_t)ws[l] * hs[l]);
    resize_linear_8u(lv[l - 1], lv[l]);
  }
}

// cv::FAST(roi, kps, threshold, true) — TYPE_9_16.  Returns keypoints (x,y,score) in raster order.
struct Cand { float x, y, resp; };
const int RING[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int fast_score(const Img& im, int x, int y, int t) {   // 0 = not a corner, else cornerScore (>= t)
  const int v = im.at(y, x);
  // cv::FAST's high-speed rejection (a necessary condition, the result is unchanged): an arc of 9 contiguous ring pixels contains
  // one pixel of every opposite pair (k, k + 8), so a corner needs every pair to have a pixel brighter than v + t, or every pair
  // to have one darker than v - t.  Pairs are visited in OpenCV's order 0/8, 4/12, 2/10, 6/14, then the odd ones; ~90 % of the
  // pixels leave after two pairs.
  {
    static const int order[8] = {0, 4, 2, 6, 1, 3, 5, 7};
    int d = 3;                                           // bit 0: still possibly "brighter" corner, bit 1: "darker"
    for (int q = 0; q < 8 && d; ++q) {
      const int k = order[q];
      const int a = im.at(y + RING[k][1], x + RING[k][0]), b = im.at(y + RING[k + 8][1], x + RING[k + 8][0]);
      d &= ((a > v + t || b > v + t) ? 1 : 0) | ((a < v - t || b < v - t) ? 2 : 0);
    }
    if (!d) return 0;
  }
  int r[16];
  for (int k = 0; k < 16; ++k) r[k] = im.at(y + RING[k][1], x + RING[k][0]);
  int best = -1000;
  for (int s = 0; s < 16; ++s) {
    int mb = 1000, md = 1000;
    for (int k = 0; k < 9; ++k) { const int q = r[(s + k) & 15]; mb = std::min(mb, q - v); md = std::min(md, v - q); }
    best = std::max(best, std::max(mb, md));
  }
  if (best <= t) return 0;
  return best - 1 == 0 ? 0 : best - 1;   // (uchar) score; best > t >= 0 so best-1 >= t; a score of 0 only if t==0 and best==1
}

void fast_roi(const Img& im, int x0, int y0, int x1, int y1, int t, std::vector<Cand>& out) {
  out.clear();
  const int w = x1 - x0, h = y1 - y0;
  if (w < 7 || h < 7) return;
  std::vector<int> score((size_t)w * h, 0);
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) score[(size_t)y * w + x] = fast_score(im, x0 + x, y0 + y, t);
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) {
      const int s = score[(size_t)y * w + x];
      if (!s) continue;     // thresholds used by the reference are 20 and 7, so a corner's score is >= 7
      const int* sp = &score[(size_t)y * w + x];
      if (s > sp[-1] && s > sp[1] && s > sp[-w - 1] && s > sp[-w] && s > sp[-w + 1] && s > sp[w - 1] && s > sp[w] && s > sp[w + 1])
        out.push_back({(float)x, (float)y, (float)s});
    }
}

// ORBextractor::ComputeKeyPointsOctTree, FAST part (ORBextractor.cc:760-818): candidates of one level
// in the reference's push order, coordinates relative to (minBorderX, minBorderY).
void level_candidates(const Img& im, const vdo_orb_params& p, std::vector<Cand>& out) {
  out.clear();
  const float W = 30;
  const int minBX = EDGE_THRESHOLD - 3, minBY = minBX, maxBX = im.w - EDGE_THRESHOLD + 3, maxBY = im.h - EDGE_THRESHOLD + 3;
  const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
  const int nCols = (int)(width / W), nRows = (int)(height / W);
  if (nCols <= 0 || nRows <= 0) return;
  const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
  std::vector<Cand> cell;
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
      fast_roi(im, (int)iniX, (int)iniY, (int)maxX, (int)maxY, p.ini_th, cell);
      if (cell.empty()) fast_roi(im, (int)iniX, (int)iniY, (int)maxX, (int)maxY, p.min_th, cell);
      for (Cand c : cell) { c.x += j * wCell; c.y += i * hCell; out.push_back(c); }
    }
  }
}

// ---- quadtree (ExtractorNode / DistributeOctTree) ------------------------------------------------
struct Node {
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  std::vector<Cand> keys;
  bool noMore = false;
  int id = 0;                                  // creation order (tie-break instead of the heap address)
  std::list<Node>::iterator lit;
};

void divide(const Node& n, Node& n1, Node& n2, Node& n3, Node& n4) {
  const int halfX = (int)std::ceil((float)(n.URx - n.ULx) / 2), halfY = (int)std::ceil((float)(n.BRy - n.ULy) / 2);
  n1.ULx = n.ULx; n1.ULy = n.ULy; n1.URx = n.ULx + halfX; n1.URy = n.ULy; n1.BLx = n.ULx; n1.BLy = n.ULy + halfY; n1.BRx = n.ULx + halfX; n1.BRy = n.ULy + halfY;
  n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = n.URx; n2.URy = n.URy; n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = n.URx; n2.BRy = n.ULy + halfY;
  n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy; n3.BLx = n.BLx; n3.BLy = n.BLy; n3.BRx = n1.BRx; n3.BRy = n.BLy;
  n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy; n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = n.BRx; n4.BRy = n.BRy;
  for (const Cand& kp : n.keys) {
    if (kp.x < n1.URx) { if (kp.y < n1.BRy) n1.keys.push_back(kp); else n3.keys.push_back(kp); }
    else if (kp.y < n1.BRy) n2.keys.push_back(kp);
    else n4.keys.push_back(kp);
  }
  if (n1.keys.size() == 1) n1.noMore = true;
  if (n2.keys.size() == 1) n2.noMore = true;
  if (n3.keys.size() == 1) n3.noMore = true;
  if (n4.keys.size() == 1) n4.noMore = true;
}

void distribute(const std::vector<Cand>& in, int minX, int maxX, int minY, int maxY, int N, std::vector<Cand>& out) {
  out.clear();
  if (in.empty()) return;
  const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
  const float hX = (float)(maxX - minX) / nIni;
  std::list<Node> nodes;
  std::vector<Node*> ini(nIni);
  int next_id = 0;
  for (int i = 0; i < nIni; ++i) {
    Node ni;
    ni.ULx = (int)(hX * (float)i); ni.ULy = 0; ni.URx = (int)(hX * (float)(i + 1)); ni.URy = 0;
    ni.BLx = ni.ULx; ni.BLy = maxY - minY; ni.BRx = ni.URx; ni.BRy = maxY - minY;
    ni.id = next_id++;
    nodes.push_back(ni);
    ini[i] = &nodes.back();
  }
  for (const Cand& kp : in) ini[(int)(kp.x / hX)]->keys.push_back(kp);
  for (auto it = nodes.begin(); it != nodes.end();) {
    if (it->keys.size() == 1) { it->noMore = true; ++it; }
    else if (it->keys.empty()) it = nodes.erase(it);
    else ++it;
  }
  bool finish = false;
  typedef std::pair<int, Node*> SP;
  auto sp_less = [](const SP& a, const SP& b) { return a.first != b.first ? a.first < b.first : a.second->id < b.second->id; };
  std::vector<SP> sizeAndNode;
  auto add_children = [&](Node (&c)[4], int* nToExpand) {
    for (int k = 0; k < 4; ++k) {
      if (c[k].keys.empty()) continue;
      c[k].id = next_id++;
      nodes.push_front(c[k]);
      if (c[k].keys.size() > 1) {
        if (nToExpand) ++*nToExpand;
        sizeAndNode.push_back(SP((int)c[k].keys.size(), &nodes.front()));
        nodes.front().lit = nodes.begin();
      }
    }
  };
  while (!finish) {
    int prevSize = (int)nodes.size();
    auto it = nodes.begin();
    int nToExpand = 0;
    sizeAndNode.clear();
    while (it != nodes.end()) {
      if (it->noMore) { ++it; continue; }
      Node c[4];
      divide(*it, c[0], c[1], c[2], c[3]);
      add_children(c, &nToExpand);
      it = nodes.erase(it);
    }
    if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
    else if ((int)nodes.size() + nToExpand * 3 > N) {
      while (!finish) {
        prevSize = (int)nodes.size();
        std::vector<SP> prev = sizeAndNode;
        sizeAndNode.clear();
        std::sort(prev.begin(), prev.end(), sp_less);
        for (int j = (int)prev.size() - 1; j >= 0; --j) {
          Node c[4];
          divide(*prev[j].second, c[0], c[1], c[2], c[3]);
          add_children(c, nullptr);
          nodes.erase(prev[j].second->lit);
          if ((int)nodes.size() >= N) break;
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
      }
    }
  }
  for (const Node& nd : nodes) {
    const Cand* best = &nd.keys[0];
    float mr = best->resp;
    for (size_t k = 1; k < nd.keys.size(); ++k) if (nd.keys[k].resp > mr) { best = &nd.keys[k]; mr = nd.keys[k].resp; }
    out.push_back(*best);
  }
}

// umax table (ORBextractor.cc:443-458)
void make_umax(int* umax) {
  int v, v0;
  const int vmax = (int)std::floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
  const int vmin = (int)std::ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(std::sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

// cv::fastAtan2 (core/mathfuncs_core, scalar path), degrees
float fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI);
  const float p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
  const float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) { c = ay / (ax + (float)2.2204460492503131e-16); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
  else { c = ax / (ay + (float)2.2204460492503131e-16); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

float ic_angle(const Img& im, float px, float py, const int* umax) {
  int m01 = 0, m10 = 0;
  const int cx = cv_round_f(px), cy = cv_round_f(py);
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m10 += u * im.at(cy, cx + u);
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int vsum = 0;
    const int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      const int vp = im.at(cy + v, cx + u), vm = im.at(cy - v, cx + u);
      vsum += (vp - vm);
      m10 += u * (vp + vm);
    }
    m01 += v * vsum;
  }
  return fast_atan2((float)m01, (float)m10);
}

void gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst);

#include "orb_pattern.inc"

// sin and cos of a float angle in [0, 2 pi] (what computeOrbDescriptor takes from cosf/sinf, src/ORBextractor.cc:101-102),
// evaluated in double from +,-,* only - Cody-Waite reduction by pi/2 and Taylor polynomials on |r| <= pi/4 (truncation
// < 1e-17) - and rounded to float: the correctly rounded float result except for double-rounding ties (~1e-9 of the
// arguments), and the SAME BITS as the GPU build, which runs the same operations (device and glibc cosf differ).
void sincos_exact(float angle, float* s_out, float* c_out) {
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

// computeOrbDescriptor (src/ORBextractor.cc:97-136): 256 rotated pair tests on the blurred level image.
// `img` = the blurred clone of mvImagePyramid[level] (contiguous, step = w); (px, py) level coordinates; angle in degrees.
void orb_descriptor(const uint8_t* img, int w, float px, float py, float angle_deg, uint8_t* desc) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  const float angle = angle_deg * factorPI;
  float a, b;
  sincos_exact(angle, &b, &a);                      // a = cos, b = sin
  const uint8_t* center = img + (size_t)cv_round_f(py) * w + cv_round_f(px);
  const signed char* pat = kRefBitPattern31;
  for (int i = 0; i < 32; ++i, pat += 32) {
    int val = 0;
    for (int k = 0; k < 8; ++k) {
      const float x0 = (float)pat[4 * k], y0 = (float)pat[4 * k + 1], x1 = (float)pat[4 * k + 2], y1 = (float)pat[4 * k + 3];
      const int t0 = center[cv_round_f(x0 * b + y0 * a) * w + cv_round_f(x0 * a - y0 * b)];
      const int t1 = center[cv_round_f(x1 * b + y1 * a) * w + cv_round_f(x1 * a - y1 * b)];
      val |= (t0 < t1) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

}  // namespace

extern "C" void vdo_oracle_depth_preprocess(float* depth, int64_t n, float bf, float factor) {
  for (int64_t i = 0; i < n; ++i) {
    if (depth[i] < 0) depth[i] = 0;
    else depth[i] = bf / (depth[i] / factor);
  }
}

extern "C" void vdo_oracle_rgb2gray(const uint8_t* rgb, int64_t n_pixels, int channels, int rgb_order, uint8_t* gray) {
  // cvtColor 8u: (R*4899 + G*9617 + B*1868 + 8192) >> 14
  for (int64_t i = 0; i < n_pixels; ++i) {
    const uint8_t* p = rgb + i * channels;
    const int r = rgb_order ? p[0] : p[2], g = p[1], b = rgb_order ? p[2] : p[0];
    gray[i] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14);
  }
}

extern "C" int vdo_oracle_orb_level_sizes(const vdo_orb_params* p, int w, int h, int32_t* ws, int32_t* hs, int32_t* nfeat) {
  std::vector<int> a, b, n; std::vector<float> s;
  level_sizes(*p, w, h, a, b, s);
  features_per_level(*p, n);
  for (int l = 0; l < p->n_levels; ++l) { ws[l] = a[l]; hs[l] = b[l]; if (nfeat) nfeat[l] = n[l]; }
  return 0;
}

// levels_out: concatenated bordered level images, level l is (w_l+38) x (h_l+38), row-major
extern "C" int vdo_oracle_orb_pyramid(const uint8_t* gray, int w, int h, const vdo_orb_params* p, uint8_t* levels_out) {
  std::vector<Img> lv;
  build_pyramid(gray, w, h, *p, lv);
  size_t off = 0;
  for (const Img& im : lv) {
    const int bw = im.w + 2 * EDGE_THRESHOLD, bh = im.h + 2 * EDGE_THRESHOLD;
    for (int y = 0; y < bh; ++y)
      for (int x = 0; x < bw; ++x)
        levels_out[off + (size_t)y * bw + x] = im.at(reflect101(y - EDGE_THRESHOLD, im.h), reflect101(x - EDGE_THRESHOLD, im.w));
    off += (size_t)bw * bh;
  }
  return 0;
}

// FAST candidates of one level (cell order, raster inside a cell); x,y relative to (16,16). Returns count.
extern "C" int vdo_oracle_orb_fast_level(const uint8_t* gray, int w, int h, const vdo_orb_params* p, int level,
                                         float* x, float* y, float* resp, int cap) {
  std::vector<Img> lv;
  build_pyramid(gray, w, h, *p, lv);
  std::vector<Cand> c;
  level_candidates(lv[level], *p, c);
  for (int i = 0; i < (int)c.size() && i < cap; ++i) { x[i] = c[i].x; y[i] = c[i].y; resp[i] = c[i].resp; }
  return (int)c.size();
}

// KAT hooks for the golden vectors of tools/pin_reference: cv::FAST(img, kps, thr, true) on a whole image (x, y, score in raster
// order) and cv::fastAtan2
extern "C" int vdo_oracle_fast_image(const uint8_t* img, int w, int h, int thr, float* x, float* y, float* resp, int cap) {
  Img im; im.w = w; im.h = h; im.d.assign(img, img + (size_t)w * h);
  std::vector<Cand> c;
  fast_roi(im, 0, 0, w, h, thr, c);
  for (int i = 0; i < (int)c.size() && i < cap; ++i) { x[i] = c[i].x; y[i] = c[i].y; resp[i] = c[i].resp; }
  return (int)c.size();
}
extern "C" float vdo_oracle_fast_atan2(float y, float x) { return fast_atan2(y, x); }
// the two remaining OpenCV primitives of ORBextractor.cc as stand-alone hooks: oracle/ref/ (the reference's ORBextractor.cc compiled
// verbatim against a mini-cv shim) forwards cv::resize / cv::copyMakeBorder here, so that build and this file differ ONLY in the
// first-party logic
extern "C" void vdo_oracle_resize_linear_8u(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  Img s, d; s.w = sw; s.h = sh; s.d.assign(src, src + (size_t)sw * sh); d.w = dw; d.h = dh; d.d.resize((size_t)dw * dh);
  resize_linear_8u(s, d);
  std::memcpy(dst, d.d.data(), d.d.size());
}
extern "C" int vdo_oracle_border_reflect101(int p, int len) { return reflect101(p, len); }

// ORBextractor::operator(): keypoints of all levels (level-0 coordinates), returns count.  desc (nullable): [cap][32] rotated
// BRIEF of every keypoint on the blurred level image - the call the reference has commented out (src/ORBextractor.cc:1083-1091)
extern "C" int vdo_oracle_orb_extract_desc(const uint8_t* gray, int w, int h, const vdo_orb_params* p,
                                           float* kx, float* ky, float* kresp, float* kangle, int32_t* koct, float* ksize, int cap, uint8_t* desc) {
  std::vector<Img> lv;
  build_pyramid(gray, w, h, *p, lv);
  std::vector<int> ws, hs, nf; std::vector<float> sc;
  level_sizes(*p, w, h, ws, hs, sc);
  features_per_level(*p, nf);
  int umax[HALF_PATCH_SIZE + 2];
  make_umax(umax);
  int n = 0;
  for (int l = 0; l < p->n_levels; ++l) {
    std::vector<Cand> c, sel;
    level_candidates(lv[l], *p, c);
    const int minB = EDGE_THRESHOLD - 3;
    distribute(c, minB, lv[l].w - EDGE_THRESHOLD + 3, minB, lv[l].h - EDGE_THRESHOLD + 3, nf[l], sel);
    const int patch = (int)(PATCH_SIZE * sc[l]);
    std::vector<uint8_t> working;
    if (desc && !sel.empty()) { working.resize((size_t)lv[l].w * lv[l].h); gaussian_blur7(lv[l].d.data(), lv[l].w, lv[l].h, working.data()); }
    for (const Cand& k : sel) {
      if (n >= cap) return -1;
      const float x = k.x + minB, y = k.y + minB;
      const float ang = ic_angle(lv[l], x, y, umax);
      if (desc) orb_descriptor(working.data(), lv[l].w, x, y, ang, desc + 32 * (size_t)n);
      float ox = x, oy = y;
      if (l != 0) { ox = x * sc[l]; oy = y * sc[l]; }
      kx[n] = ox; ky[n] = oy; kresp[n] = k.resp; kangle[n] = ang; koct[n] = l; ksize[n] = (float)patch;
      ++n;
    }
  }
  return n;
}

extern "C" int vdo_oracle_orb_extract(const uint8_t* gray, int w, int h, const vdo_orb_params* p,
                                      float* kx, float* ky, float* kresp, float* kangle, int32_t* koct, float* ksize, int cap) {
  return vdo_oracle_orb_extract_desc(gray, w, h, p, kx, ky, kresp, kangle, koct, ksize, cap, nullptr);
}

// KAT hooks for the descriptor stage
extern "C" void vdo_oracle_sincos_exact(float angle, float* s, float* c) { sincos_exact(angle, s, c); }
extern "C" void vdo_oracle_orb_descriptor(const uint8_t* blurred, int w, float px, float py, float angle_deg, uint8_t* desc32) {
  orb_descriptor(blurred, w, px, py, angle_deg, desc32);
}
extern "C" const signed char* vdo_oracle_orb_pattern(void) { return kRefBitPattern31; }

// cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8UC1 (OpenCV 3.4.0 path:
// separable filter with 8-bit fixed-point kernels, (sum + 2^15) >> 16)
extern "C" void vdo_oracle_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) { gaussian_blur7(src, w, h, dst); }
namespace {
void gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) {
  double kd[7], sum = 0;
  for (int i = 0; i < 7; ++i) { const double x = i - 3; kd[i] = std::exp(-0.5 / (2.0 * 2.0) * x * x); sum += kd[i]; }
  int k[7];
  for (int i = 0; i < 7; ++i) k[i] = cv_round_d((double)(float)(kd[i] / sum) * 256);
  std::vector<int> tmp((size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int i = 0; i < 7; ++i) s += k[i] * src[(size_t)y * w + reflect101(x + i - 3, w)];
      tmp[(size_t)y * w + x] = s;
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int i = 0; i < 7; ++i) s += k[i] * tmp[(size_t)reflect101(y + i - 3, h) * w + x];
      const int v = (s + (1 << 15)) >> 16;
      dst[(size_t)y * w + x] = (uint8_t)std::min(255, std::max(0, v));
    }
}
}  // namespace

// Frame::Frame static filter (UseSampleFea == 0 branch): returns count; outputs indexed by kept order
extern "C" int vdo_oracle_frame_static_filter(int n, const float* kx, const float* ky, const int32_t* koct,
                                              const int32_t* mask, const float* depth, const float* flow /*[h][w][2]*/,
                                              int w, int h, float th_depth,
                                              int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const int x = (int)kx[i], y = (int)ky[i];
    if (mask[(size_t)y * w + x] != 0) continue;
    const float d = depth[(size_t)y * w + x];
    if (d > th_depth || d <= 0) continue;
    const float fxe = flow[2 * ((size_t)y * w + x)], fye = flow[2 * ((size_t)y * w + x) + 1];
    if (fxe != 0 && fye != 0) {
      if (kx[i] + fxe < w && ky[i] + fye < h && kx[i] < w && ky[i] < h) {
        keep_idx[m] = i; corr_x[m] = kx[i] + fxe; corr_y[m] = ky[i] + fye; flow_x[m] = fxe; flow_y[m] = fye;
        const float dd = depth[(size_t)((int)ky[i]) * w + (int)kx[i]];
        depth_out[m] = dd > 0 ? dd : -1.f;
        ++m;
      }
    }
  }
  return m;
}

// Frame::Frame static filter, UseSampleFea == 1 branch (src/Frame.cc:132-166)
extern "C" int vdo_oracle_frame_static_filter_sampled(int n, const float* kx, const float* ky, const int32_t* mask, const float* depth, const float* flow,
                                                      int w, int h, float th_depth,
                                                      int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const int x = (int)kx[i], y = (int)ky[i];
    if (mask[(size_t)y * w + x] != 0) continue;
    if (depth[(size_t)y * w + x] > th_depth || depth[(size_t)y * w + x] <= 0) continue;
    const float fxe = flow[2 * ((size_t)y * w + x)], fye = flow[2 * ((size_t)y * w + x) + 1];
    if (fxe != 0 && fye != 0) {
      if (kx[i] + fxe < w && ky[i] + fye < h && kx[i] + fxe > 0 && ky[i] + fye > 0) {
        keep_idx[m] = i; corr_x[m] = kx[i] + fxe; corr_y[m] = ky[i] + fye; flow_x[m] = fxe; flow_y[m] = fye;
        const float dd = depth[(size_t)((int)ky[i]) * w + (int)kx[i]];      // depth gather (:178-194)
        depth_out[m] = dd > 0 ? dd : -1.f;
        ++m;
      }
    }
  }
  return m;
}

// Frame::SampleKeyPoints (src/Frame.cc:672-737) with cv::RNG(seed)
extern "C" int vdo_oracle_sample_keypoints(int rows, int cols, unsigned long long seed, float* x_out, float* y_out) {
  struct Rng {
    unsigned long long state;
    unsigned next() { state = (unsigned long long)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
  } rng{seed ? seed : 0xffffffffULL};
  const int N = 3000, n_div = 20;
  std::vector<std::vector<std::pair<float, float> > > grid(n_div * n_div);
  const int x_step = cols / n_div, y_step = rows / n_div;
  int key_num = 0;
  while (key_num < N) {
    for (int i = 0; i < n_div; ++i) {
      for (int j = 0; j < n_div; ++j) {
        const float x = rng.uniform(i * x_step, (i + 1) * x_step);
        const float y = rng.uniform(j * y_step, (j + 1) * y_step);
        if (x >= cols || y >= rows || x <= 0 || y <= 0) continue;
        grid[i * n_div + j].push_back(std::make_pair(x, y));
        key_num = key_num + 1;
        if (key_num >= N) break;
      }
      if (key_num >= N) break;
    }
  }
  int n = 0;
  for (size_t c = 0; c < grid.size(); ++c)
    for (size_t k = 0; k < grid[c].size(); ++k) { x_out[n] = grid[c][k].first; y_out[n] = grid[c][k].second; ++n; }
  return n;
}

// Frame::Frame semi-dense object sampling (stride 4, raster order)
extern "C" int vdo_oracle_frame_object_sample(const int32_t* mask, const float* depth, const float* flow, int w, int h,
                                              float th_depth_obj, int step, int cap,
                                              float* key_x, float* key_y, float* corr_x, float* corr_y,
                                              float* flow_x, float* flow_y, float* depth_out, int32_t* label) {
  int m = 0;
  for (int i = 0; i < h; i += step)
    for (int j = 0; j < w; j += step) {
      const size_t o = (size_t)i * w + j;
      if (mask[o] != 0 && depth[o] < th_depth_obj && depth[o] > 0) {
        const float fx = flow[2 * o], fy = flow[2 * o + 1];
        if (j + fx < w && j + fx > 0 && i + fy < h && i + fy > 0) {
          if (m >= cap) return -1;
          flow_x[m] = fx; flow_y[m] = fy; corr_x[m] = j + fx; corr_y[m] = i + fy;
          key_x[m] = (float)j; key_y[m] = (float)i; depth_out[m] = depth[o]; label[m] = mask[o];
          ++m;
        }
      }
    }
  return m;
}
