// Frame — reference include/Frame.h:35-257 reduced to the members on the hot path.  The
// constructor runs ORB + the static filter + the semi-dense object sampling on the GPU.
#pragma once
#include <vector>

#include "ORBextractor.h"
#include "minicv.h"

namespace VDO_SLAM {

class Frame {
 public:
  Frame() {}
  Frame(const cv::Mat& imGray, const cv::Mat& imDepth, const cv::Mat& imFlow, const cv::Mat& maskSEM, const double& timeStamp,
        ORBextractor* extractor, cv::Mat& K, cv::Mat& distCoef, const float& bf, const float& thDepth, const float& thDepthObj,
        const int& UseSampleFea);
  void ExtractORB(int flag, const cv::Mat& im);
  void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); }
  // Frame.cc:484-555: back-projection of a tracked feature into the world frame (Rwl * x3Dc + twl in fp32 - cv::gemm's float fast path for the
  // product, its double-accumulating generic path for twl = -Rlw^T tlw).  `addnoise`: the reference perturbs the depth with cv::RNG(time(NULL)) Gaussian noise on this path
  // (irreproducible by construction, and only reached with bJoint == false); the mirror uses the measured depth.
  cv::Mat UnprojectStereoStat(const int& i, const bool& addnoise);
  cv::Mat UnprojectStereoObject(const int& i, const bool& addnoise);
  cv::Mat ObtainFlowDepthObject(const int& i, const bool& addnoise);
  cv::Mat ObtainFlowDepthCamera(const int& i, const bool& addnoise);

  ORBextractor* mpORBextractorLeft = nullptr;
  double mTimeStamp = 0;
  cv::Mat mK, mDistCoef;
  static float fx, fy, cx, cy, invfx, invfy;
  float mbf = 0, mThDepth = 0, mThDepthObj = 0;
  int N = 0;
  std::vector<cv::KeyPoint> mvKeys;
  cv::Mat mDescriptors;
  // background
  int N_s = 0, N_s_tmp = 0;
  std::vector<cv::KeyPoint> mvStatKeys, mvStatKeysTmp, mvCorres;
  std::vector<float> mvStatDepth, mvStatDepthTmp;
  std::vector<cv::Mat> mvStat3DPointTmp;         // 3x1 CV_32F, world frame
  std::vector<cv::Point2f> mvFlowNext;
  // objects
  std::vector<cv::KeyPoint> mvObjKeys, mvObjCorres;
  std::vector<float> mvObjDepth;
  std::vector<cv::Point2f> mvObjFlowNext;
  std::vector<int> vSemObjLabel, vObjLabel;
  std::vector<cv::Mat> mvObj3DPoint;             // 3x1 CV_32F, world frame
  // per object of the frame (include/Frame.h:150-160)
  std::vector<bool> bObjStat;
  std::vector<cv::Mat> vObjMod;                  // 4x4 CV_32F
  std::vector<int> nModLabel, nSemPosition;
  cv::Mat mInitModel;
  cv::Mat mTcw;
  static long unsigned int nNextId;
  long unsigned int mnId = 0;
  static bool mbInitialComputations;
};

}  // namespace VDO_SLAM
