// ORBextractor host class: reference interface (include/ORBextractor.h, src/ORBextractor.cc:399-459,
// 1035-1110); pyramid / FAST / quadtree / orientation / blur run in libvdo_hip (vdo_orb_*).
#include "ORBextractor.h"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace VDO_SLAM {

static void die(const char* what) {   // the reference has no error channel (it exits); a GPU failure surfaces as an exception the flat hooks turn into a return code
  throw std::runtime_error(std::string("VDO_SLAM::ORBextractor: ") + what + ": " + vdo_last_error());
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
  mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
  mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    mvScaleFactor[i] = mvScaleFactor[i - 1] * (float)scaleFactor;
    mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
  }
  mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
  for (int i = 0; i < nlevels; i++) {
    mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
    mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
  }
  mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor() { if (mOrb) vdo_orb_destroy(mOrb); }

void ORBextractor::operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors) {
  if (image.empty()) return;
  if (!mOrb || image.cols != mW || image.rows != mH) {
    if (mOrb) vdo_orb_destroy(mOrb);
    vdo_orb_params p{nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST};
    if (vdo_orb_create(HostContext(), &p, image.cols, image.rows, &mOrb) != VDO_OK) die("vdo_orb_create");
    mW = image.cols; mH = image.rows;
  }
  const int cap = nfeatures + 256;
  std::vector<float> x(cap), y(cap), r(cap), a(cap), s(cap);
  std::vector<int32_t> o(cap);
  vdo_keypoints out{cap, 0, x.data(), y.data(), r.data(), a.data(), s.data(), o.data()};
  if (vdo_orb_extract(mOrb, image.data, (int)image.step, 0, &out) != VDO_OK) die("vdo_orb_extract");
  keypoints.clear();
  keypoints.reserve(out.n);
  for (int i = 0; i < out.n; ++i) keypoints.push_back(cv::KeyPoint(x[i], y[i], s[i], a[i], r[i], o[i]));
  // descriptors: allocated; written only when mbComputeDescriptors is set (the reference's computeDescriptors call is
  // commented out, :1091, so its rows stay uninitialised - the default here too)
  if (out.n) descriptors.create(out.n, 32, cv::CV_8UC1); else descriptors = cv::Mat();
  if (out.n && mbComputeDescriptors && vdo_orb_descriptors(mOrb, descriptors.data, out.n) != VDO_OK) die("vdo_orb_descriptors");
  // mvImagePyramid: public member of the reference class; level interiors as views into the bordered images
  mBordered.resize(nlevels);
  for (int l = 0; l < nlevels; ++l) {
    int w, h;
    vdo_orb_level_info(mOrb, l, &w, &h, nullptr, nullptr);
    mBordered[l].create(h + 38, w + 38, cv::CV_8UC1);
    if (vdo_orb_get_pyramid(mOrb, l, mBordered[l].data) != VDO_OK) die("vdo_orb_get_pyramid");
    cv::Mat roi(h, w, cv::CV_8UC1, mBordered[l].data + 19 * mBordered[l].step + 19);
    roi.step = mBordered[l].step;
    mvImagePyramid[l] = roi;
  }
}

}  // namespace VDO_SLAM
