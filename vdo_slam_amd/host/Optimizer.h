// Optimizer — the static optimisation entry points of the reference (include/Optimizer.h:25-32)
// with the g2o graph + solve replaced by libvdo_hip.
#pragma once
#include <vector>

#include "Frame.h"
#include "GraphStore.h"
#include "Map.h"
#include "minicv.h"

namespace VDO_SLAM {

using std::vector;

class Optimizer {
 public:
  int static PoseOptimizationNew(Frame* pCurFrame, Frame* pLastFrame, vector<int>& TemperalMatch);
  int static PoseOptimizationFlow2Cam(Frame* pCurFrame, Frame* pLastFrame, vector<int>& TemperalMatch);
  cv::Mat static PoseOptimizationObjMot(Frame* pCurFrame, Frame* pLastFrame, const vector<int>& ObjId, std::vector<int>& InlierID);
  cv::Mat static PoseOptimizationFlow2(Frame* pCurFrame, Frame* pLastFrame, const vector<int>& ObjId, std::vector<int>& InlierID);
  void static FullBatchOptimization(Map* pMap, const cv::Mat Calib_K);
  void static PartialBatchOptimization(Map* pMap, const cv::Mat Calib_K, const int WINDOW_SIZE);
  // the same two on the flat GraphStore the pipeline appends to frame by frame (no Map, no per-point cv::Mat): the graph is
  // built straight into the SoA arrays of vdo_ba_create; results are written back into the store (cam_rf / rm_rf / xyz for
  // the full batch; cam / rm[.][0] / sta.xyz for the window)
  void static FullBatchOptimization(GraphStore& store, const TrackList& sta_tracks, const TrackList& dyn_tracks, const float K4[4]);
  void static PartialBatchOptimization(GraphStore& store, const TrackList& sta_tracks, const float K4[4], const int WINDOW_SIZE);
  cv::Mat static Get3DinWorld(const cv::KeyPoint& Feats2d, const float& Dpts, const cv::Mat& Calib_K, const cv::Mat& CameraPose);
  cv::Mat static Get3DinCamera(const cv::KeyPoint& Feats2d, const float& Dpts, const cv::Mat& Calib_K);
  // LM statistics of the last batch optimisation (the reference only prints them, :1769,:1935)
  static vdo_lm_stats last_batch_stats;
};

}  // namespace VDO_SLAM
