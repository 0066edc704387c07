// TEST INFRASTRUCTURE - oracle/_ref/libref_orb.so: the reference's OWN /root/reference/src/ORBextractor.cc, compiled verbatim from
// where it lies (textually included below so that its file-static functions - IC_Angle, computeOrbDescriptor, computeDescriptors -
// are reachable), against the mini-cv shim of oracle/ref/shim/ whose five OpenCV primitives are the oracle's restatements.
// Pins the FIRST-PARTY logic of SURVEY.md §8 rows a2-a6 (and the level arithmetic of a1): src/ORBextractor.cc:399-459 (budget per
// level, umax), :470-752 (quadtree incl. std::list order and the (size, pointer) sort), :754-842 (cell grid, 20 -> 7 fallback),
// :66-93 (IC_Angle), :97-136 (computeOrbDescriptor), :1035-1110 (operator(), key-point rescale), :1112-1137 (pyramid ROIs).
// Not a product file; nothing of the reference is copied into this repository: the recipe (Makefile) names the path.
// Heap addresses.  DistributeOctTree sorts (size, ExtractorNode*) pairs (src/ORBextractor.cc:673): nodes of equal size are ordered by
// their HEAP ADDRESS, so the reference's key points differ from run to run of the same binary on the same image (seen here with glibc's
// malloc: 2509 / 2507 key points for one image in one process).  To have something to compare with, this library
// allocates from a bump arena (operator new below, bound to this library only by -Bsymbolic; reset at every entry point): addresses
// grow in allocation order, so "by pointer" = "by creation order" - the rule the oracle and the product implement.
#include <cstddef>
#include <cstdlib>
#include <new>
#include <sys/mman.h>
namespace ref_arena {
static char* base = nullptr; static size_t used = 0; static const size_t kSize = (size_t)8 << 30;
inline void reset() {      // start over with ZERO pages, like a fresh process (the reference reads a few never-initialised members)
  if (base && used) madvise(base, (used + 4095) & ~(size_t)4095, MADV_DONTNEED);
  used = 0;
}
inline void* take(size_t n) {
  if (!base) { base = (char*)mmap(nullptr, kSize, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0); if (base == (char*)MAP_FAILED) abort(); }
  n = (n + 15) & ~(size_t)15;
  if (used + n > kSize) abort();
  void* p = base + used; used += n; return p;
}
}
void* operator new(std::size_t n) { return ref_arena::take(n ? n : 1); }
void* operator new[](std::size_t n) { return ref_arena::take(n ? n : 1); }
void operator delete(void*) noexcept {}
void operator delete[](void*) noexcept {}
void operator delete(void*, std::size_t) noexcept {}
void operator delete[](void*, std::size_t) noexcept {}

#ifndef REF_ORB_SRC
#error "compile with -DREF_ORB_SRC='\"/root/reference/src/ORBextractor.cc\"' (oracle/ref/Makefile)"
#endif
#include REF_ORB_SRC

#include <cstdint>

namespace {
// access to the protected stages, for the per-level outputs and for the descriptor call the reference has commented out (:1091)
struct Probe : public VDO_SLAM::ORBextractor {
  using VDO_SLAM::ORBextractor::ORBextractor;
  void pyramid(const cv::Mat& im) { ComputePyramid(im); }
  void keypoints(std::vector<std::vector<cv::KeyPoint>>& all) { ComputeKeyPointsOctTree(all); }
  const std::vector<cv::Point>& pat() const { return pattern; }
  const std::vector<int>& budget() const { return mnFeaturesPerLevel; }
  const std::vector<int>& um() const { return umax; }
};
cv::Mat wrap(const uint8_t* gray, int w, int h) {
  cv::Mat m(h, w, CV_8UC1);
  std::memcpy(m.data, gray, (size_t)w * h);
  return m;
}
}  // namespace

struct vdo_orb_params { int32_t n_features; float scale_factor; int32_t n_levels, ini_th, min_th; };   // = oracle/vdo_oracle.h

// ORBextractor::operator() exactly as the reference runs it: key points of all levels in level-0 coordinates
extern "C" int vdo_ref_orb_extract(const uint8_t* gray, int w, int h, const vdo_orb_params* p,
                                   float* kx, float* ky, float* kresp, float* kangle, int32_t* koct, float* ksize, int cap) {
  ref_arena::reset();
  VDO_SLAM::ORBextractor ex(p->n_features, p->scale_factor, p->n_levels, p->ini_th, p->min_th);
  std::vector<cv::KeyPoint> k; cv::Mat desc;
  ex(wrap(gray, w, h), cv::Mat(), k, desc);
  if ((int)k.size() > cap) return -1;
  for (size_t i = 0; i < k.size(); ++i) { kx[i] = k[i].pt.x; ky[i] = k[i].pt.y; kresp[i] = k[i].response; kangle[i] = k[i].angle; koct[i] = k[i].octave; ksize[i] = k[i].size; }
  return (int)k.size();
}

// the same stages one by one + computeDescriptors on the blurred clone (the call at src/ORBextractor.cc:1091, un-commented):
// desc [cap][32]; lx / ly = level coordinates the descriptor was taken at
extern "C" int vdo_ref_orb_extract_desc(const uint8_t* gray, int w, int h, const vdo_orb_params* p,
                                        float* kx, float* ky, float* kresp, float* kangle, int32_t* koct, float* ksize, int cap, uint8_t* desc32) {
  ref_arena::reset();
  Probe ex(p->n_features, p->scale_factor, p->n_levels, p->ini_th, p->min_th);
  ex.pyramid(wrap(gray, w, h));
  std::vector<std::vector<cv::KeyPoint>> all;
  ex.keypoints(all);
  const std::vector<float> sc = ex.GetScaleFactors();
  int n = 0;
  for (int l = 0; l < p->n_levels; ++l) {
    std::vector<cv::KeyPoint>& k = all[l];
    if (k.empty()) continue;
    cv::Mat working = ex.mvImagePyramid[l].clone();
    cv::GaussianBlur(working, working, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
    cv::Mat d;
    VDO_SLAM::computeDescriptors(working, k, d, ex.pat());
    for (size_t i = 0; i < k.size(); ++i, ++n) {
      if (n >= cap) return -1;
      std::memcpy(desc32 + 32 * (size_t)n, d.ptr((int)i), 32);
      cv::KeyPoint q = k[i];
      if (l != 0) q.pt *= sc[l];
      kx[n] = q.pt.x; ky[n] = q.pt.y; kresp[n] = q.response; kangle[n] = q.angle; koct[n] = q.octave; ksize[n] = q.size;
    }
  }
  return n;
}

// per-level facts: sizes of mvImagePyramid, mnFeaturesPerLevel, umax[16]
extern "C" int vdo_ref_orb_level_facts(int w, int h, const vdo_orb_params* p, int32_t* ws, int32_t* hs, int32_t* nfeat, int32_t* umax16) {
  ref_arena::reset();
  Probe ex(p->n_features, p->scale_factor, p->n_levels, p->ini_th, p->min_th);
  std::vector<uint8_t> z((size_t)w * h, 0);
  ex.pyramid(wrap(z.data(), w, h));
  for (int l = 0; l < p->n_levels; ++l) { ws[l] = ex.mvImagePyramid[l].cols; hs[l] = ex.mvImagePyramid[l].rows; nfeat[l] = ex.budget()[l]; }
  for (int i = 0; i < 16; ++i) umax16[i] = ex.um()[i];
  return 0;
}

// bordered pyramid levels as ComputePyramid leaves them: level l is (w_l + 38) x (h_l + 38), concatenated
extern "C" int vdo_ref_orb_pyramid(const uint8_t* gray, int w, int h, const vdo_orb_params* p, uint8_t* levels_out) {
  ref_arena::reset();
  Probe ex(p->n_features, p->scale_factor, p->n_levels, p->ini_th, p->min_th);
  ex.pyramid(wrap(gray, w, h));
  size_t off = 0;
  for (int l = 0; l < p->n_levels; ++l) {
    const cv::Mat& m = ex.mvImagePyramid[l];
    const int bw = m.cols + 2 * VDO_SLAM::EDGE_THRESHOLD, bh = m.rows + 2 * VDO_SLAM::EDGE_THRESHOLD;
    const uchar* base = m.data - (size_t)VDO_SLAM::EDGE_THRESHOLD * m.step - VDO_SLAM::EDGE_THRESHOLD;
    for (int y = 0; y < bh; ++y) std::memcpy(levels_out + off + (size_t)y * bw, base + (size_t)y * m.step, (size_t)bw);
    off += (size_t)bw * bh;
  }
  return 0;
}

// one key point through the reference's IC_Angle and computeOrbDescriptor (KAT hooks)
extern "C" float vdo_ref_ic_angle(const uint8_t* img, int w, int h, float px, float py) {
  ref_arena::reset();
  Probe ex(1000, 1.2f, 8, 20, 7);
  return VDO_SLAM::IC_Angle(wrap(img, w, h), cv::Point2f(px, py), ex.um());
}
extern "C" void vdo_ref_orb_descriptor(const uint8_t* blurred, int w, int h, float px, float py, float angle_deg, uint8_t* desc32) {
  ref_arena::reset();
  Probe ex(1000, 1.2f, 8, 20, 7);
  cv::KeyPoint k(px, py, 31.f, angle_deg);
  VDO_SLAM::computeOrbDescriptor(k, wrap(blurred, w, h), &ex.pat()[0], desc32);
}
