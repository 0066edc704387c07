// Converter — fp32 cv::Mat <-> fp64 marshalling at the optimiser boundary
// (reference include/Converter.h, src/Converter.cc:25-35, 37-59, 98-135, 151-166).
#pragma once
#include "minicv.h"

namespace VDO_SLAM {

class Converter {
 public:
  // T (4x4 CV_32F) -> 16 doubles row-major; the SE3Quat / Isometry conversion of
  // Converter::toSE3Quat happens inside libvdo_hip (kernels take the float-valued matrix).
  static void toDouble16(const cv::Mat& T, double out[16]) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = T.at<float>(i, j);
  }
  static cv::Mat toCvMat(const double T[16]) {               // Converter::toCvMat(Matrix4d) — double -> float
    cv::Mat m(4, 4, cv::CV_32F);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.at<float>(i, j) = (float)T[4 * i + j];
    return m;
  }
  static cv::Mat toCvSE3(const double R[9], const double t[3]) {   // Converter::toCvSE3
    cv::Mat m = cv::Mat::eye(4, 4, cv::CV_32F);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m.at<float>(i, j) = (float)R[3 * i + j]; m.at<float>(i, 3) = (float)t[i]; }
    return m;
  }
  // Converter::toInvMatrix (:151-166).  t_inv = -R.t() * t: a cv::gemm with a transposed operand -> OpenCV 3.4's generic GEMMSingleMul<float, double>:
  // the dot product accumulated in double (k ascending), times alpha = -1, one rounding to float
  static cv::Mat toInvMatrix(const cv::Mat& T) {
    cv::Mat Ti = cv::Mat::eye(4, 4, cv::CV_32F);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) Ti.at<float>(i, j) = T.at<float>(j, i);
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += (double)T.at<float>(k, i) * (double)T.at<float>(k, 3);
      Ti.at<float>(i, 3) = (float)(s * -1.0);
    }
    return Ti;
  }
};

}  // namespace VDO_SLAM
