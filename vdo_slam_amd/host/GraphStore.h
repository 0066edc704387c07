// GraphStore — the factor-graph input of the batch optimisers as flat SoA arrays that grow by plain appends, one frame at
// a time (second half).  It holds what Track() saves into the Map ("Save Graph Structure",
// reference src/Tracking.cc:1031-1159) - static / dynamic features with depth and world point, camera poses, rigid
// motions + labels - without the Map's one-heap-object-per-3-D-point layout (vector<vector<cv::Mat>>), and it is what the
// graph builder of Optimizer reads directly: no conversion pass, no linear searches (the reference looks every
// observation up inside its tracklet, src/Optimizer.cc:1456-1463, 1643-1664).  A reference-format Map is materialised
// from it only on request (FramePipeline::SyncMap).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace VDO_SLAM {

class Map;

struct FeatureBlock {                         // features of all frames, frame i = [off[i], off[i+1])
  std::vector<int64_t> off{0};
  std::vector<float> u, v, d, xyz;            // pixel, depth, world point (3 per feature)
  int frames() const { return (int)off.size() - 1; }
  int count(int i) const { return (int)(off[i + 1] - off[i]); }
  void append(size_t n, const float* x, const float* y, const float* depth, const float* xw) {
    u.insert(u.end(), x, x + n); v.insert(v.end(), y, y + n); d.insert(d.end(), depth, depth + n);
    xyz.insert(xyz.end(), xw, xw + 3 * n);
    off.push_back(off.back() + (int64_t)n);
  }
  void clear() { off.assign(1, 0); u.clear(); v.clear(); d.clear(); xyz.clear(); }
  // Address space for `n_features` up front (pages are only touched as the arrays fill): without it a doubling std::vector
  // re-allocates and copies several MB now and then - a multi-millisecond frame in the middle of a sequence.
  void reserve(size_t n_features) { u.reserve(n_features); v.reserve(n_features); d.reserve(n_features); xyz.reserve(3 * n_features); off.reserve(4096); }
};

struct TrackList {                            // tracklets, flat: track t = pairs [off[t], off[t+1]) of (frame, feature)
  std::vector<int32_t> off{0}, frame, feat, obj;   // obj: object id per track (dynamic tracklets only)
  int size() const { return (int)off.size() - 1; }
};

struct GraphStore {
  FeatureBlock sta, dyn;
  std::vector<float> cam, cam_rf;             // [F][16] T_wc, and the copy the full batch refines (vmCameraPose / _RF)
  std::vector<int64_t> rm_off{0};             // transition i (frame i -> i+1): motions [rm_off[i], rm_off[i+1]); entry 0 = camera motion
  std::vector<float> rm, rm_rf;               // [..][16]                                              (vmRigidMotion / _RF)
  std::vector<int32_t> rm_label;              //                                                       (vnRMLabel)
  int frames() const { return (int)(cam.size() / 16); }
  int transitions() const { return (int)rm_off.size() - 1; }
  int n_motions(int i) const { return (int)(rm_off[i + 1] - rm_off[i]); }
  void add_camera(const float* Twc) { cam.insert(cam.end(), Twc, Twc + 16); cam_rf.insert(cam_rf.end(), Twc, Twc + 16); }
  void add_motions(int n, const float* H16, const int32_t* labels) {
    rm.insert(rm.end(), H16, H16 + 16 * (size_t)n); rm_rf.insert(rm_rf.end(), H16, H16 + 16 * (size_t)n);
    rm_label.insert(rm_label.end(), labels, labels + n);
    rm_off.push_back(rm_off.back() + n);
  }
  void clear() { sta.clear(); dyn.clear(); cam.clear(); cam_rf.clear(); rm_off.assign(1, 0); rm.clear(); rm_rf.clear(); rm_label.clear(); }
};

// Map <-> store (host/Optimizer.cc): the reference-format Map as input / output of the same builder
void StoreFromMap(const Map& m, GraphStore& s, TrackList& sta, TrackList& dyn);
void StoreToMap(const GraphStore& s, const TrackList& sta, const TrackList& dyn, Map& m);

}  // namespace VDO_SLAM
