// TEST INFRASTRUCTURE - stand-in for the reference's include/Optimizer.h in the _ref build: the same eight statics (include/Optimizer.h:25-32 plus
// the two back-projection helpers), without the g2o include.  Their bodies (src/Optimizer.cc) ARE g2o code and cannot be compiled here; the _ref
// build supplies glue that hands the same correspondences to the oracle's restatement of that g2o code (oracle/ref/ref_track_entry.cc).
#ifndef VDO_REF_OPTIMIZER_STUB_H_
#define VDO_REF_OPTIMIZER_STUB_H_
#include "Map.h"
#include "Frame.h"
namespace VDO_SLAM {
using namespace std;
class Optimizer {
 public:
  int static PoseOptimizationNew(Frame* pCurFrame, Frame* pLastFrame, vector<int>& TemperalMatch);
  int static PoseOptimizationFlow2Cam(Frame* pCurFrame, Frame* pLastFrame, vector<int>& TemperalMatch);
  cv::Mat static PoseOptimizationObjMot(Frame* pCurFrame, Frame* pLastFrame, const vector<int>& ObjId, std::vector<int>& InlierID);
  cv::Mat static PoseOptimizationFlow2(Frame* pCurFrame, Frame* pLastFrame, const vector<int>& ObjId, std::vector<int>& InlierID);
  void static FullBatchOptimization(Map* pMap, const cv::Mat Calib_K);
  void static PartialBatchOptimization(Map* pMap, const cv::Mat Calib_K, const int WINDOW_SIZE);
  cv::Mat static Get3DinWorld(const cv::KeyPoint& Feats2d, const float& Dpts, const cv::Mat& Calib_K, const cv::Mat& CameraPose);
  cv::Mat static Get3DinCamera(const cv::KeyPoint& Feats2d, const float& Dpts, const cv::Mat& Calib_K);
};
}  // namespace VDO_SLAM
#endif
