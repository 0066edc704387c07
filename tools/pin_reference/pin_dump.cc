// Pin harness (tools/pin_reference): runs the REAL reference - halajun/VDO_SLAM with OpenCV 3.4.0, Eigen3, CSparse and its
// vendored g2o - on the committed inputs of tests/golden/inputs/ and writes golden outputs into tests/golden/.
// tests/test_golden.py then checks the CPU oracle (oracle/) against them; until this has been run somewhere, every
// "parity green" of this repository means HIP == oracle with the oracle unpinned.
// Raw little-endian files, layouts documented in tests/golden/README.md and mirrored by tools/pin_reference/make_inputs.py.
//   pin_dump orb    <gray.u8> <w> <h> <out_dir>     ORBextractor(2500, 1.2, 8, 20, 7): pyramid, keypoints, blur; cvtColor, fastAtan2
//   pin_dump pnp    <case.bin> <out.bin>            cv::solvePnPRansac exactly as Tracking::GetInitModelCam calls it
//   pin_dump flow2  <case.bin> <out.bin>            Optimizer::PoseOptimizationFlow2Cam / PoseOptimizationFlow2
//   pin_dump batch  <map.bin> <out_dir>             Optimizer::FullBatchOptimization (+ the .g2o dumps it writes)
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <opencv2/calib3d/calib3d.hpp>
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>

#include "Converter.h"
#include "Frame.h"
#include "Map.h"
#include "ORBextractor.h"
#include "Optimizer.h"

using namespace VDO_SLAM;

namespace {

std::vector<char> slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { std::cerr << "cannot read " << path << std::endl; std::exit(2); }
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
struct Reader {
  std::vector<char> buf; size_t pos = 0;
  explicit Reader(const std::string& p) : buf(slurp(p)) {}
  template <class T> T one() { T v; std::memcpy(&v, buf.data() + pos, sizeof(T)); pos += sizeof(T); return v; }
  template <class T> std::vector<T> many(size_t n) { std::vector<T> v(n); if (n) std::memcpy(v.data(), buf.data() + pos, n * sizeof(T)); pos += n * sizeof(T); return v; }
};
struct Writer {
  std::ofstream f;
  explicit Writer(const std::string& p) : f(p, std::ios::binary) { if (!f) { std::cerr << "cannot write " << p << std::endl; std::exit(2); } }
  template <class T> void one(T v) { f.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
  template <class T> void many(const T* p, size_t n) { f.write(reinterpret_cast<const char*>(p), n * sizeof(T)); }
};
cv::Mat mat44(const float* p) { cv::Mat m(4, 4, CV_32F); std::memcpy(m.data, p, 64); return m; }
void put44(Writer& w, const cv::Mat& m) { cv::Mat c; m.convertTo(c, CV_32F); c = c.clone(); w.many(reinterpret_cast<const float*>(c.data), 16); }

// ---- ORB front-end: src/ORBextractor.cc as a whole + the OpenCV calls around it
int cmd_orb(int argc, char** argv) {
  const std::string in = argv[2]; const int w = std::atoi(argv[3]), h = std::atoi(argv[4]); const std::string out = argv[5];
  std::vector<char> px = slurp(in);
  cv::Mat gray(h, w, CV_8UC1, px.data());
  ORBextractor orb(2500, 1.2f, 8, 20, 7);                      // example/kitti-0000-0013.yaml:70-75
  std::vector<cv::KeyPoint> kps; cv::Mat desc;
  orb(gray, cv::Mat(), kps, desc);
  { Writer wr(out + "/orb_pyramid.bin");                        // int32 n_levels; per level int32 w, h, then the interior pixels (no border)
    wr.one<int32_t>((int32_t)orb.mvImagePyramid.size());
    for (const cv::Mat& lv : orb.mvImagePyramid) {
      wr.one<int32_t>(lv.cols); wr.one<int32_t>(lv.rows);
      for (int y = 0; y < lv.rows; ++y) wr.many(lv.ptr<uint8_t>(y), lv.cols);
    } }
  { Writer wr(out + "/orb_keypoints.bin");                      // int32 n; n x (float x, y, response, angle, size; int32 octave)
    wr.one<int32_t>((int32_t)kps.size());
    for (const cv::KeyPoint& k : kps) { wr.one<float>(k.pt.x); wr.one<float>(k.pt.y); wr.one<float>(k.response); wr.one<float>(k.angle); wr.one<float>(k.size); wr.one<int32_t>(k.octave); } }
  { cv::Mat b = orb.mvImagePyramid[0].clone();                  // src/ORBextractor.cc:1083-1084
    cv::GaussianBlur(b, b, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
    Writer wr(out + "/orb_blur_level0.bin");
    wr.one<int32_t>(b.cols); wr.one<int32_t>(b.rows);
    for (int y = 0; y < b.rows; ++y) wr.many(b.ptr<uint8_t>(y), b.cols); }
  { // cv::FAST on the whole level-0 interior at both thresholds (src/ORBextractor.cc:798-804 calls it per 30-px cell): x, y, score
    for (int thr : {20, 7}) {
      std::vector<cv::KeyPoint> f;
      cv::FAST(orb.mvImagePyramid[0], f, thr, true);
      Writer wr(out + "/fast_level0_thr" + std::to_string(thr) + ".bin");
      wr.one<int32_t>((int32_t)f.size());
      for (const cv::KeyPoint& k : f) { wr.one<float>(k.pt.x); wr.one<float>(k.pt.y); wr.one<float>(k.response); } } }
  { // cvtColor(RGB2GRAY) on a gradient cube (src/Tracking.cc:209-222) and fastAtan2 on a grid (src/ORBextractor.cc:92)
    cv::Mat rgb(64, 64, CV_8UC3);
    for (int y = 0; y < 64; ++y) for (int x = 0; x < 64; ++x) rgb.at<cv::Vec3b>(y, x) = cv::Vec3b((uint8_t)(x * 4 + 1), (uint8_t)(y * 4 + 2), (uint8_t)((x * y) & 255));
    cv::Mat g; cv::cvtColor(rgb, g, CV_RGB2GRAY);
    Writer wr(out + "/cvtcolor_rgb2gray_64x64.bin"); wr.many(g.data, 64 * 64);
    Writer wa(out + "/fastatan2_grid.bin");                     // 41 x 41 floats: fastAtan2(y, x) for y, x in -20..20
    for (int y = -20; y <= 20; ++y) for (int x = -20; x <= 20; ++x) wa.one<float>(cv::fastAtan2((float)y, (float)x)); }
  return 0;
}

// ---- cv::solvePnPRansac as called by Tracking::GetInitModelCam / GetInitModelObj (src/Tracking.cc:1636-1660)
int cmd_pnp(int argc, char** argv) {
  Reader r(argv[2]);
  const int n = r.one<int32_t>();
  std::vector<float> K4 = r.many<float>(4), X = r.many<float>(3 * (size_t)n), uv = r.many<float>(2 * (size_t)n);
  std::vector<cv::Point2f> cur_2d(n); std::vector<cv::Point3f> pre_3d(n);
  for (int i = 0; i < n; ++i) { cur_2d[i] = cv::Point2f(uv[2 * i], uv[2 * i + 1]); pre_3d[i] = cv::Point3f(X[3 * i], X[3 * i + 1], X[3 * i + 2]); }
  cv::Mat camera_mat = cv::Mat::zeros(3, 3, CV_64FC1), distCoeffs = cv::Mat::zeros(1, 4, CV_64FC1);
  camera_mat.at<double>(0, 0) = K4[0]; camera_mat.at<double>(1, 1) = K4[1]; camera_mat.at<double>(0, 2) = K4[2]; camera_mat.at<double>(1, 2) = K4[3]; camera_mat.at<double>(2, 2) = 1.0;
  cv::Mat Rvec(3, 1, CV_64FC1), Tvec(3, 1, CV_64FC1), d(3, 3, CV_64FC1), inliers;
  cv::solvePnPRansac(pre_3d, cur_2d, camera_mat, distCoeffs, Rvec, Tvec, false, 500, 0.4, 0.98, inliers, cv::SOLVEPNP_AP3P);
  cv::Rodrigues(Rvec, d);
  Writer w(argv[3]);                                            // double R[9] row-major, double t[3], int32 n_inliers, int32 idx[n_inliers]
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) w.one<double>(d.at<double>(i, j));
  for (int i = 0; i < 3; ++i) w.one<double>(Tvec.at<double>(i, 0));
  w.one<int32_t>(inliers.rows);
  for (int i = 0; i < inliers.rows; ++i) w.one<int32_t>(inliers.at<int>(i));
  return 0;
}

// ---- Optimizer::PoseOptimizationFlow2Cam (is_object = 0) / PoseOptimizationFlow2 (is_object = 1)   src/Optimizer.cc:2333-2542, 2755-2972
int cmd_flow2(int argc, char** argv) {
  Reader r(argv[2]);
  const int n = r.one<int32_t>(), is_object = r.one<int32_t>();
  std::vector<float> K4 = r.many<float>(4), Tl = r.many<float>(16), T0 = r.many<float>(16), key = r.many<float>(2 * (size_t)n), depth = r.many<float>(n), flow = r.many<float>(2 * (size_t)n);
  Frame::fx = K4[0]; Frame::fy = K4[1]; Frame::cx = K4[2]; Frame::cy = K4[3]; Frame::invfx = 1.0f / K4[0]; Frame::invfy = 1.0f / K4[1];
  Frame last, cur;
  last.SetPose(mat44(Tl.data()));
  std::vector<cv::KeyPoint> keys(n); std::vector<cv::Point2f> fl(n); std::vector<int> ids(n);
  for (int i = 0; i < n; ++i) { keys[i] = cv::KeyPoint(key[2 * i], key[2 * i + 1], 0, 0, 0, -1); fl[i] = cv::Point2f(flow[2 * i], flow[2 * i + 1]); ids[i] = i; }
  Writer w(argv[3]);                                            // int32 n_inliers, float pose[16], int32 inlier[n] (0/1), float cur_keys[n][2]
  std::vector<int32_t> inl(n, 0);
  if (!is_object) {
    last.mvStatKeys = keys; last.mvStatDepth = depth; last.mvFlowNext = fl;
    cur.mvStatKeys = keys;                                     // overwritten for the inliers (last key + refined flow)
    cur.SetPose(mat44(T0.data()));                             // "initial with camera pose" (:2357)
    std::vector<int> tm = ids;
    const int good = Optimizer::PoseOptimizationFlow2Cam(&cur, &last, tm);
    for (int i = 0; i < n; ++i) inl[i] = tm[i] != -1;
    w.one<int32_t>(good); put44(w, cur.mTcw); w.many(inl.data(), n);
    for (int i = 0; i < n; ++i) { w.one<float>(cur.mvStatKeys[i].pt.x); w.one<float>(cur.mvStatKeys[i].pt.y); }
  } else {
    last.mvObjKeys = keys; last.mvObjDepth = depth; last.mvObjFlowNext = fl;
    cur.mvObjKeys = keys; cur.vObjLabel.assign(n, 1);
    cur.mInitModel = mat44(T0.data());                         // (:2779)
    std::vector<int> in_id;
    cv::Mat pose = Optimizer::PoseOptimizationFlow2(&cur, &last, ids, in_id);
    for (int id : in_id) inl[id] = 1;
    w.one<int32_t>((int32_t)in_id.size()); put44(w, pose); w.many(inl.data(), n);
    for (int i = 0; i < n; ++i) { w.one<float>(cur.mvObjKeys[i].pt.x); w.one<float>(cur.mvObjKeys[i].pt.y); }
  }
  return 0;
}

// ---- Optimizer::FullBatchOptimization on a Map read from a flat dump (tools/pin_reference/make_inputs.py: write_map)   src/Optimizer.cc:1232-2175
int cmd_batch(int argc, char** argv) {
  Reader r(argv[2]);
  const std::string out = argv[3];
  Map map;
  const int F = r.one<int32_t>();
  std::vector<float> K9 = r.many<float>(9);
  cv::Mat K(3, 3, CV_32F); std::memcpy(K.data, K9.data(), 36);
  auto pt3 = [](const float* p) { cv::Mat m(3, 1, CV_32F); std::memcpy(m.data, p, 12); return m; };
  for (int i = 0; i < F; ++i) {
    std::vector<float> T = r.many<float>(16);
    map.vmCameraPose.push_back(mat44(T.data()));
    for (int dyn = 0; dyn < 2; ++dyn) {
      const int m = r.one<int32_t>();
      std::vector<float> uv = r.many<float>(2 * (size_t)m), dd = r.many<float>(m), xw = r.many<float>(3 * (size_t)m);
      std::vector<cv::KeyPoint> kp(m); std::vector<cv::Mat> p3(m);
      for (int j = 0; j < m; ++j) { kp[j] = cv::KeyPoint(uv[2 * j], uv[2 * j + 1], 0, 0, 0, -1); p3[j] = pt3(xw.data() + 3 * j); }
      (dyn ? map.vpFeatDyn : map.vpFeatSta).push_back(kp); (dyn ? map.vfDepDyn : map.vfDepSta).push_back(dd); (dyn ? map.vp3DPointDyn : map.vp3DPointSta).push_back(p3);
    }
  }
  for (int dyn = 0; dyn < 2; ++dyn) {
    const int nt = r.one<int32_t>();
    for (int t = 0; t < nt; ++t) {
      const int len = r.one<int32_t>();
      std::vector<int32_t> pr = r.many<int32_t>(2 * (size_t)len);
      std::vector<std::pair<int, int> > tr(len);
      for (int k = 0; k < len; ++k) tr[k] = std::make_pair(pr[2 * k], pr[2 * k + 1]);
      (dyn ? map.TrackletDyn : map.TrackletSta).push_back(tr);
    }
    if (dyn) { std::vector<int32_t> ob = r.many<int32_t>(nt); map.nObjID.assign(ob.begin(), ob.end()); }
  }
  for (int i = 0; i + 1 < F; ++i) {
    const int nm = r.one<int32_t>();
    std::vector<cv::Mat> mots(nm); std::vector<int> labs(nm);
    for (int j = 0; j < nm; ++j) { std::vector<float> T = r.many<float>(16); mots[j] = mat44(T.data()); }
    std::vector<int32_t> lb = r.many<int32_t>(nm);
    labs.assign(lb.begin(), lb.end());
    map.vmRigidMotion.push_back(mots); map.vnRMLabel.push_back(labs);
  }
  if (chdir(out.c_str()) != 0) { std::cerr << "cannot enter " << out << std::endl; return 2; }   // the .g2o dumps land in the working directory (:1934-1936)
  Optimizer::FullBatchOptimization(&map, K);
  Writer w("batch_refined.bin");                                // int32 F; F x float[16] camera poses (RF); per transition int32 n, n x float[16] motions (RF)
  w.one<int32_t>(F);
  for (int i = 0; i < F; ++i) put44(w, map.vmCameraPose_RF[i]);
  for (int i = 0; i + 1 < F; ++i) { w.one<int32_t>((int32_t)map.vmRigidMotion_RF[i].size()); for (const cv::Mat& m : map.vmRigidMotion_RF[i]) put44(w, m); }
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  const std::string c = argc > 1 ? argv[1] : "";
  if (c == "orb" && argc == 6) return cmd_orb(argc, argv);
  if (c == "pnp" && argc == 4) return cmd_pnp(argc, argv);
  if (c == "flow2" && argc == 4) return cmd_flow2(argc, argv);
  if (c == "batch" && argc == 4) return cmd_batch(argc, argv);
  std::cerr << "usage: pin_dump orb <gray.u8> <w> <h> <out_dir> | pnp <case.bin> <out.bin> | flow2 <case.bin> <out.bin> | batch <map.bin> <out_dir>" << std::endl;
  return 1;
}
