// Frame host class (reference src/Frame.cc:61-260): ORB extraction, static-keypoint filter with
// flow correspondences and depth gather, semi-dense object sampling — on the GPU via the C-ABI.
#include "Frame.h"

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>

namespace VDO_SLAM {

long unsigned int Frame::nNextId = 0;
bool Frame::mbInitialComputations = true;
float Frame::cx, Frame::cy, Frame::fx, Frame::fy, Frame::invfx, Frame::invfy;

namespace {
// The reference has no error channel; it ends the process only for unreadable settings / a wrong sensor (src/System.cc:35-39, 55-59).
// A failure of the GPU path is not one of those: it surfaces as an exception that the flat hooks (host_capi.cc, System.cc) turn into
// a return code, so that a host that loaded this library through an FFI survives it.
[[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("VDO_SLAM::Frame: ") + what + ": " + vdo_last_error()); }
vdo_frame_images* g_imgs = nullptr;
int g_w = 0, g_h = 0;
vdo_frame_images* images_for(int w, int h) {
  if (!g_imgs || g_w != w || g_h != h) {
    if (g_imgs) vdo_frame_images_destroy(g_imgs);
    if (vdo_frame_images_create(HostContext(), w, h, &g_imgs) != VDO_OK) { g_imgs = nullptr; fail("frame images"); }
    g_w = w; g_h = h;
  }
  return g_imgs;
}
}  // namespace

void Frame::ExtractORB(int flag, const cv::Mat& im) {
  if (flag == 0) (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
}

Frame::Frame(const cv::Mat& imGray, const cv::Mat& imDepth, const cv::Mat& imFlow, const cv::Mat& maskSEM, const double& timeStamp,
             ORBextractor* extractor, cv::Mat& K, cv::Mat& distCoef, const float& bf, const float& thDepth, const float& thDepthObj,
             const int& UseSampleFea)
    : mpORBextractorLeft(extractor), mTimeStamp(timeStamp), mK(K.clone()), mDistCoef(distCoef.clone()), mbf(bf), mThDepth(thDepth), mThDepthObj(thDepthObj) {
  mnId = nNextId++;
  ExtractORB(0, imGray);
  N = (int)mvKeys.size();
  if (mvKeys.empty()) return;
  if (UseSampleFea != 0) {
    throw std::runtime_error("VDO_SLAM::Frame: UseSampleFeature=1 runs through FramePipeline (PipelineParams.use_sample_feature), not through this constructor");
  }
  vdo_frame_images* imgs = images_for(imGray.cols, imGray.rows);
  if (vdo_frame_images_upload(imgs, (const float*)imDepth.data, (const float*)imFlow.data, (const int32_t*)maskSEM.data) != VDO_OK) fail("GPU call");
  // ---- background features: Frame.cc:100-128 + :178-194
  std::vector<float> kx(N), ky(N);
  for (int i = 0; i < N; ++i) { kx[i] = mvKeys[i].pt.x; ky[i] = mvKeys[i].pt.y; }
  std::vector<int32_t> keep(N);
  std::vector<float> cxv(N), cyv(N), fxv(N), fyv(N), dv(N);
  int m = 0;
  if (vdo_frame_static_filter(imgs, N, kx.data(), ky.data(), mThDepth, keep.data(), cxv.data(), cyv.data(), fxv.data(), fyv.data(), dv.data(), &m) != VDO_OK) fail("GPU call");
  for (int i = 0; i < m; ++i) {
    const cv::KeyPoint& k = mvKeys[keep[i]];
    mvStatKeysTmp.push_back(k);
    mvCorres.push_back(cv::KeyPoint(cxv[i], cyv[i], 0, 0, 0, k.octave, -1));
    mvFlowNext.push_back(cv::Point2f(fxv[i], fyv[i]));
    mvStatDepthTmp.push_back(dv[i]);
  }
  N_s_tmp = m;
  // ---- semi-dense object features: Frame.cc:201-228
  const int cap = ((imGray.cols + 3) / 4) * ((imGray.rows + 3) / 4);
  std::vector<float> o[7];
  for (auto& v : o) v.resize(cap);
  std::vector<int32_t> lab(cap);
  int n_obj = 0;
  if (vdo_frame_object_sample(imgs, mThDepthObj, 4, cap, o[0].data(), o[1].data(), o[2].data(), o[3].data(), o[4].data(), o[5].data(), o[6].data(), lab.data(), &n_obj) != VDO_OK) fail("GPU call");
  for (int i = 0; i < n_obj; ++i) {
    mvObjFlowNext.push_back(cv::Point2f(o[4][i], o[5][i]));
    mvObjCorres.push_back(cv::KeyPoint(o[2][i], o[3][i], 0, 0, 0, -1));
    mvObjKeys.push_back(cv::KeyPoint(o[0][i], o[1][i], 0, 0, 0, -1));
    mvObjDepth.push_back(o[6][i]);
    vSemObjLabel.push_back(lab[i]);
  }
  if (mbInitialComputations) {
    fx = K.at<float>(0, 0); fy = K.at<float>(1, 1); cx = K.at<float>(0, 2); cy = K.at<float>(1, 2);
    invfx = 1.0f / fx; invfy = 1.0f / fy;
    mbInitialComputations = false;
  }
}

namespace {
// Rwl * x3Dc + twl with Rwl = Rlw^T (a copy), twl = -Rlw^T tlw (src/Frame.cc:506-511).  twl: cv::gemm with a transposed operand = the generic path,
// accumulated in double, one rounding; Rwl * x3Dc + twl: untransposed and 3 wide = cv::gemm's small-matrix fast path, in float, left to right
// (csrc/tracking_shared.hpp backproject: the same two rules on the device)
cv::Mat unproject_world(float u, float v, float z, const cv::Mat& Tcw) {
  const float x3[3] = {(u - Frame::cx) * z * Frame::invfx, (v - Frame::cy) * z * Frame::invfy, z};
  cv::Mat o(3, 1, cv::CV_32F);
  for (int i = 0; i < 3; ++i) {
    double t = 0;
    for (int k = 0; k < 3; ++k) t += (double)(-Tcw.at<float>(k, i)) * (double)Tcw.at<float>(k, 3);
    const float twl = (float)t;
    const float r = Tcw.at<float>(0, i) * x3[0] + Tcw.at<float>(1, i) * x3[1] + Tcw.at<float>(2, i) * x3[2];
    o.at<float>(i) = r + twl;
  }
  return o;
}
}  // namespace

cv::Mat Frame::UnprojectStereoStat(const int& i, const bool&) {
  const float z = mvStatDepth[i];
  if (z > 0) return unproject_world(mvStatKeys[i].pt.x, mvStatKeys[i].pt.y, z, mTcw);
  std::cout << "found a depth value < 0 ..." << std::endl;
  return cv::Mat();
}
cv::Mat Frame::UnprojectStereoObject(const int& i, const bool&) {
  const float z = mvObjDepth[i];
  if (z > 0) return unproject_world(mvObjKeys[i].pt.x, mvObjKeys[i].pt.y, z, mTcw);
  std::cout << "found a depth value < 0 ..." << std::endl;
  return cv::Mat();
}

// Frame.cc:617-670 (addnoise is never set on the bJoint path)
cv::Mat Frame::ObtainFlowDepthObject(const int& i, const bool&) {
  const float z = mvObjDepth[i];
  if (z > 0) { cv::Mat m(3, 1, cv::CV_32F); m.at<float>(0) = mvObjFlowNext[i].x; m.at<float>(1) = mvObjFlowNext[i].y; m.at<float>(2) = z; return m; }
  std::cout << "found a depth value < 0 ..." << std::endl;
  return cv::Mat();
}
cv::Mat Frame::ObtainFlowDepthCamera(const int& i, const bool&) {
  const float z = mvStatDepth[i];
  if (z > 0) { cv::Mat m(3, 1, cv::CV_32F); m.at<float>(0) = mvFlowNext[i].x; m.at<float>(1) = mvFlowNext[i].y; m.at<float>(2) = z; return m; }
  std::cout << "found a depth value < 0 ..." << std::endl;
  return cv::Mat();
}

}  // namespace VDO_SLAM
