// System / Tracking — the reference's entry API (include/System.h:42-53, include/Tracking.h) as thin shells around
// FramePipeline: settings file, colour conversion, the in-place depth conversion the caller sees, ground-truth rows as the
// gate of the object tracker, Map bookkeeping and the batch optimisations.  Visualisation (imTraj), metric printing and
// SaveResults' text formats are out of scope; the hot path is entirely behind FramePipeline.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "FramePipeline.h"
#include "Map.h"
#include "minicv.h"

namespace VDO_SLAM {

class System;

class Tracking {
 public:
  Tracking(System* pSys, Map* pMap, const std::string& strSettingPath, const int sensor);
  ~Tracking();
  // Tracking::GrabImageRGBD (src/Tracking.cc:164-314): imD is converted in place (raw disparity*factor -> metres), maskSEM
  // receives the labels UpdateMask recovered; returns mTcw.clone() (4x4 CV_32F)
  cv::Mat GrabImageRGBD(const cv::Mat& imRGB, cv::Mat& imD, const cv::Mat& imFlow, const cv::Mat& maskSEM, const cv::Mat& mTcw_gt,
                        const std::vector<std::vector<float> >& vObjPose_gt, const double& timestamp, cv::Mat& imTraj, const int& nImage);
  FramePipeline* pipeline() { return pipe_.get(); }
  int f_id = 0, StopFrame = 0;
  cv::Mat mK;

 private:
  Map* mpMap;
  std::map<std::string, double> cfg_;
  bool mbRGB = true;
  int mTestData = 2;                  // ChooseData: 1 OMD, 2 KITTI, 3 VirtualKITTI
  float mbf = 0, mDepthMapFactor = 1;
  vdo_ctx* ctx_[4] = {nullptr, nullptr, nullptr, nullptr};
  std::unique_ptr<FramePipeline> pipe_;
  std::vector<uint8_t> gray_;
  bool have_frame_ = false;
};

class System {
 public:
  enum eSensor { MONOCULAR = 0, STEREO = 1, RGBD = 2 };
  System(const std::string& strSettingsFile, const eSensor sensor);
  ~System();
  cv::Mat TrackRGBD(const cv::Mat& im, cv::Mat& depthmap, const cv::Mat& flowmap, const cv::Mat& masksem, const cv::Mat& mTcw_gt,
                    const std::vector<std::vector<float> >& vObjPose_gt, const double& timestamp, cv::Mat& imTraj, const int& nImage);
  void SaveResults(const std::string& filename);   // camera trajectory (T_wc rows), before and after the batch optimisation
  Map* map();                          // brought up to date from the pipeline's GraphStore on access
  Tracking* tracker() { return mpTracker; }

 private:
  eSensor mSensor;
  Map* mpMap;
  Tracking* mpTracker;
};

}  // namespace VDO_SLAM
