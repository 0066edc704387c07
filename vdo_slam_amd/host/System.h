// System / Tracking — the reference's entry API (include/System.h:42-53, include/Tracking.h) as thin shells around
// FramePipeline: settings file, colour conversion, the in-place depth conversion the caller sees, ground-truth rows as the
// gate of the object tracker, Map bookkeeping, the batch optimisations and SaveResults' five text files (src/System.cc:75-100).  Visualisation (imTraj)
// and metric printing are out of scope; the hot path is entirely behind FramePipeline.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "Frame.h"
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

  // ---- public state of the reference class (include/Tracking.h:116-198).  Scalar configuration / progress members are kept up to date by
  // every call.  The per-frame containers live in HBM / in FramePipeline's flat arrays; SyncFrameState() materialises them in the reference's
  // form on request (as System::map() does for the Map): mCurrentFrame - mTcw, the renewed static set (mvStatKeysTmp, mvStatDepthTmp,
  // mvStat3DPointTmp, mvCorres, mvFlowNext, N_s_tmp), the renewed object set (mvObjKeys, mvObjDepth, mvObj3DPoint, mvObjCorres, mvObjFlowNext,
  // vSemObjLabel, vObjLabel) and the per-object vectors (nSemPosition, nModLabel, bObjStat, vObjMod) as Track() leaves them
  // (src/Tracking.cc:2780-2812, 2984-2991, 836-933) -, the semi-dense samples mvTmpObj*, max_id, and the converted depth map / repaired mask
  // (mDepthMap, mSegMap).  mImGray / mFlowMap are the caller's own buffers; TemperalMatch* and repro_e are locals of Track() here.
  void SyncFrameState();
  Frame mCurrentFrame;
  cv::Mat mDepthMap, mSegMap;
  std::vector<cv::KeyPoint> mvTmpObjKeys, mvTmpObjCorres;
  std::vector<float> mvTmpObjDepth;
  std::vector<int> mvTmpSemObjLabel;
  std::vector<cv::Point2f> mvTmpObjFlowNext;
  int max_id = 1;
  enum eTrackingState { NO_IMAGES_YET = 0, NOT_INITIALIZED = 1, OK = 2 };
  eTrackingState mState = NO_IMAGES_YET, mLastProcessedState = NO_IMAGES_YET;
  enum eDataState { OMD = 1, KITTI = 2, VirtualKITTI = 3 };
  eDataState mTestData = KITTI;        // ChooseData
  int mSensor = 2;                     // System::RGBD
  int f_id = 0, StopFrame = 0;
  bool bLocalBatch = true, bGlobalBatch = true, bJoint = true, bFrame2Frame = false, bFirstFrame = true;
  int nWINDOW_SIZE = 0, nOVERLAP_SIZE = 0, nMaxTrackPointBG = 0, nMaxTrackPointOBJ = 0, nUseSampleFea = 0;
  float fSFMgThres = 0, fSFDsThres = 0;
  std::vector<float> all_timing;       // per frame: ms of this TrackRGBD call (the reference keeps five clock() buckets per frame)
  cv::Mat mK, mDistCoef;
  float mbf = 0, mThDepth = 0, mThDepthObj = 0, mDepthMapFactor = 1;

 private:
  Map* mpMap;
  std::map<std::string, double> cfg_;
  bool mbRGB = true;
  vdo_ctx* ctx_[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};     // front-end / camera LM / object LMs / helper thread / ORB thread
  std::unique_ptr<FramePipeline> pipe_;
  std::vector<uint8_t> gray_;
  bool have_frame_ = false;
  cv::Mat mOriginInv;                  // ground-truth pose of the first frame (src/Tracking.cc:319-323)
};

class System {
 public:
  enum eSensor { MONOCULAR = 0, STEREO = 1, RGBD = 2 };
  System(const std::string& strSettingsFile, const eSensor sensor);
  ~System();
  cv::Mat TrackRGBD(const cv::Mat& im, cv::Mat& depthmap, const cv::Mat& flowmap, const cv::Mat& masksem, const cv::Mat& mTcw_gt,
                    const std::vector<std::vector<float> >& vObjPose_gt, const double& timestamp, cv::Mat& imTraj, const int& nImage);
  // The reference's result files (src/System.cc:66-198), `filename` being the path PREFIX as there: initial_stereo_new.txt,
  // refined_stereo_new.txt, cam_pose_gt_stereo.txt (frame id + the 16 entries of T_wc, fixed, 9 digits) and the object motions per
  // transition in the same row format.  The reference writes the object motions in the BODY frame of the ground-truth object pose
  // (obj_mot_stereo_new.txt / _rf_new.txt / obj_mot_gt.txt / obj_centre.txt: ground-truth object-pose parsing, out of scope);
  // here they are written in the WORLD frame under names of their own: obj_mot_world_new.txt / obj_mot_world_rf_new.txt.
  void SaveResults(const std::string& filename);
  Map* map();                          // brought up to date from the pipeline's GraphStore on access
  Tracking* tracker() { return mpTracker; }

 private:
  eSensor mSensor;
  Map* mpMap;
  Tracking* mpTracker;
};

}  // namespace VDO_SLAM
