// Map — plain public vectors: the factor-graph input format of the batch optimisers
// (reference include/Map.h:35-84).  Only the members the hot path reads/writes are kept.
#pragma once
#include <utility>
#include <vector>

#include "minicv.h"

namespace VDO_SLAM {

class Map {
 public:
  // static features / depths / 3-D points per frame, tracklets (frame id, feature id)
  std::vector<std::vector<cv::KeyPoint> > vpFeatSta;
  std::vector<std::vector<float> > vfDepSta;
  std::vector<std::vector<cv::Mat> > vp3DPointSta;
  std::vector<std::vector<std::pair<int, int> > > TrackletSta;
  // dynamic features
  std::vector<std::vector<cv::KeyPoint> > vpFeatDyn;
  std::vector<std::vector<float> > vfDepDyn;
  std::vector<std::vector<cv::Mat> > vp3DPointDyn;
  std::vector<std::vector<std::pair<int, int> > > TrackletDyn;
  std::vector<int> nObjID;
  // poses (T_wc) and rigid motions ([0] = camera motion, [1..] = objects) + refined copies
  std::vector<cv::Mat> vmCameraPose, vmCameraPose_RF;
  std::vector<cv::Mat> vmCameraPose_GT;                    // Converter::toInvMatrix(mCurrentFrame.mTcw_gt) per frame (src/Tracking.cc:320-328, 1113-1115)
  std::vector<std::vector<cv::Mat> > vmRigidMotion, vmRigidMotion_RF;
  std::vector<std::vector<int> > vnRMLabel;
};

}  // namespace VDO_SLAM
