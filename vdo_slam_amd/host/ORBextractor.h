// ORBextractor — same public interface as the reference (include/ORBextractor.h:33-99);
// the body calls the HIP front-end through the C-ABI (vdo_orb_*).
#pragma once
#include <vector>

#include "host_context.h"
#include "minicv.h"

namespace VDO_SLAM {

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
  ~ORBextractor();
  // Compute the ORB keypoints on an image (mask ignored, descriptors left uninitialised — exactly
  // what the reference does, src/ORBextractor.cc:1066,1091).
  void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);
  int inline GetLevels() { return nlevels; }
  float inline GetScaleFactor() { return (float)scaleFactor; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }
  std::vector<cv::Mat> mvImagePyramid;   // level interiors (row stride = width + 38, as ROIs of the bordered images)
  // rotated BRIEF (computeOrbDescriptor, src/ORBextractor.cc:97-136): off = the reference as shipped (call commented out,
  // rows uninitialised); on = `descriptors` holds the 32-byte rows the commented-out call would have produced
  bool mbComputeDescriptors = false;

 protected:
  int nfeatures;
  double scaleFactor;
  int nlevels, iniThFAST, minThFAST;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  vdo_orb* mOrb = nullptr;
  int mW = 0, mH = 0;
  std::vector<cv::Mat> mBordered;
};

}  // namespace VDO_SLAM
