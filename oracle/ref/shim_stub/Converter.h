// TEST INFRASTRUCTURE - stand-in for the reference's include/Converter.h in the _ref build.  The real header pulls in Eigen and g2o (absent); of its
// functions the sources compiled here (Tracking.cc, Frame.cc) call only toInvMatrix, which is restated next to the reference's own statements
// (src/Converter.cc:151-166) in oracle/ref/ref_track_entry.cc.
#ifndef VDO_REF_CONVERTER_STUB_H_
#define VDO_REF_CONVERTER_STUB_H_
#include <opencv2/core/core.hpp>
#include <Eigen/Dense>
namespace VDO_SLAM {
class Converter {
 public:
  static cv::Mat toInvMatrix(const cv::Mat& T);
  static Eigen::Matrix<double, 4, 4> toMatrix4d(const cv::Mat& cvMat4);      // (visualisation only: Tracking::DrawSparseFlowBirdeye, never called)
};
}  // namespace VDO_SLAM
#endif
