#include "System.h"

#include <stdexcept>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>

#include "Converter.h"
#include "Optimizer.h"

namespace VDO_SLAM {

namespace {
// The settings files of the reference are flat "key: value" YAML 1.0 (example/kitti-0000-0013.yaml): that subset is read here
// (cv::FileStorage is not available on this side).  Returns false when the file cannot be opened.
bool read_settings(const std::string& path, std::map<std::string, double>& out) {
  std::ifstream f(path.c_str());
  if (!f.is_open()) return false;
  std::string line;
  while (std::getline(f, line)) {
    const size_t h = line.find('#');
    if (h != std::string::npos) line = line.substr(0, h);
    if (line.empty() || line[0] == '%') continue;
    const size_t c = line.find(':');
    if (c == std::string::npos) continue;
    std::string key = line.substr(0, c), val = line.substr(c + 1);
    while (!key.empty() && (key.back() == ' ' || key.back() == '\t')) key.pop_back();
    char* end = nullptr;
    const double v = std::strtod(val.c_str(), &end);
    if (end != val.c_str()) out[key] = v;
  }
  return true;
}
double get(const std::map<std::string, double>& m, const char* k, double dflt = 0) { auto it = m.find(k); return it == m.end() ? dflt : it->second; }
// Everything Tracking::Tracking reads from the settings file (src/Tracking.cc:53-161), with the reference's types: float / int
// conversions of cv::FileNode (a missing key reads as 0, e.g. Camera.k3 in example/kitti-0018-0020.yaml; fps 0 -> 30), plus
// Camera.width / Camera.height, which the reference's yaml files carry and this side sizes its device images from.
struct TrackingSettings {
  float fx, fy, cx, cy, k1, k2, p1, p2, k3, bf, fps;
  int rgb, n_features; float scale_factor; int n_levels, ini_th, min_th, data_code;
  float th_depth_bg, th_depth_obj, depth_map_factor;
  int max_track_bg, max_track_obj; float sf_mg_thres, sf_ds_thres; int window_size, overlap_size, use_sample_feature, width, height;
};
bool read_tracking_settings(const std::string& path, TrackingSettings& t, std::map<std::string, double>* raw = nullptr) {
  std::map<std::string, double> c;
  if (!read_settings(path, c)) return false;
  t.fx = (float)get(c, "Camera.fx"); t.fy = (float)get(c, "Camera.fy"); t.cx = (float)get(c, "Camera.cx"); t.cy = (float)get(c, "Camera.cy");
  t.k1 = (float)get(c, "Camera.k1"); t.k2 = (float)get(c, "Camera.k2"); t.p1 = (float)get(c, "Camera.p1"); t.p2 = (float)get(c, "Camera.p2"); t.k3 = (float)get(c, "Camera.k3");
  t.bf = (float)get(c, "Camera.bf");
  t.fps = (float)get(c, "Camera.fps"); if (t.fps == 0) t.fps = 30;
  t.rgb = (int)get(c, "Camera.RGB");
  t.n_features = (int)get(c, "ORBextractor.nFeatures"); t.scale_factor = (float)get(c, "ORBextractor.scaleFactor"); t.n_levels = (int)get(c, "ORBextractor.nLevels");
  t.ini_th = (int)get(c, "ORBextractor.iniThFAST"); t.min_th = (int)get(c, "ORBextractor.minThFAST");
  t.data_code = (int)get(c, "ChooseData");
  t.th_depth_bg = (float)get(c, "ThDepthBG"); t.th_depth_obj = (float)get(c, "ThDepthOBJ"); t.depth_map_factor = (float)get(c, "DepthMapFactor");
  t.max_track_bg = (int)get(c, "MaxTrackPointBG"); t.max_track_obj = (int)get(c, "MaxTrackPointOBJ");
  t.sf_mg_thres = (float)get(c, "SFMgThres"); t.sf_ds_thres = (float)get(c, "SFDsThres");
  t.window_size = (int)get(c, "WINDOW_SIZE"); t.overlap_size = (int)get(c, "OVERLAP_SIZE"); t.use_sample_feature = (int)get(c, "UseSampleFeature");
  t.width = (int)get(c, "Camera.width"); t.height = (int)get(c, "Camera.height");
  if (raw) *raw = c;
  return true;
}
}  // namespace

Tracking::Tracking(System*, Map* pMap, const std::string& strSettingPath, const int) : mpMap(pMap) {
  TrackingSettings t{};
  if (!read_tracking_settings(strSettingPath, t, &cfg_)) { std::cerr << "Failed to open settings file at: " << strSettingPath << std::endl; std::exit(-1); }
  mK = cv::Mat::eye(3, 3, cv::CV_32F);
  mK.at<float>(0, 0) = t.fx; mK.at<float>(1, 1) = t.fy; mK.at<float>(0, 2) = t.cx; mK.at<float>(1, 2) = t.cy;
  mbf = t.bf;
  mbRGB = t.rgb != 0;
  mDepthMapFactor = t.depth_map_factor;
  mTestData = t.data_code == 1 ? OMD : t.data_code == 3 ? VirtualKITTI : KITTI;   // include/Tracking.h:129-133, src/Tracking.cc:115-128
  // distortion (src/Tracking.cc:66-77): the reference's three settings files are distortion-free, and its Frame only undistorts when k1 != 0
  // (src/Frame.cc:375-379); a calibrated distortion is outside what this path implements - refuse it instead of ignoring it
  mDistCoef = cv::Mat::zeros(t.k3 != 0 ? 5 : 4, 1, cv::CV_32F);
  mDistCoef.at<float>(0, 0) = t.k1; mDistCoef.at<float>(1, 0) = t.k2; mDistCoef.at<float>(2, 0) = t.p1; mDistCoef.at<float>(3, 0) = t.p2;
  if (t.k3 != 0) mDistCoef.at<float>(4, 0) = t.k3;
  if (t.k1 != 0 || t.k2 != 0 || t.p1 != 0 || t.p2 != 0 || t.k3 != 0) { std::cerr << "settings: lens distortion (Camera.k1..k3, p1, p2) is not supported by this build" << std::endl; std::exit(-1); }
  PipelineParams p{};
  p.width = t.width; p.height = t.height;
  p.K4[0] = t.fx; p.K4[1] = t.fy; p.K4[2] = t.cx; p.K4[3] = t.cy;
  p.bf = mbf; p.depth_map_factor = mDepthMapFactor;
  p.th_depth_bg = t.th_depth_bg; p.th_depth_obj = t.th_depth_obj;
  p.max_track_bg = t.max_track_bg; p.max_track_obj = t.max_track_obj;
  p.sf_mg_thres = t.sf_mg_thres; p.sf_ds_thres = t.sf_ds_thres;
  p.n_features = t.n_features; p.n_levels = t.n_levels; p.ini_th = t.ini_th; p.min_th = t.min_th; p.scale_factor = t.scale_factor;
  p.build_lm = 1; p.defer_objects = 0;               // TrackRGBD returns with the frame complete, like the reference
  p.use_sample_feature = t.use_sample_feature; p.sample_seed = 1;
  p.pnp_refit = 1;                                   // solvePnPRansac's EPnP re-estimation (OpenCV 3.4)
  p.window_size = t.window_size; p.overlap_size = t.overlap_size;
  mThDepth = p.th_depth_bg; mThDepthObj = p.th_depth_obj;
  nWINDOW_SIZE = p.window_size; nOVERLAP_SIZE = p.overlap_size; nMaxTrackPointBG = p.max_track_bg; nMaxTrackPointOBJ = p.max_track_obj;
  nUseSampleFea = p.use_sample_feature; fSFMgThres = p.sf_mg_thres; fSFDsThres = p.sf_ds_thres;
  mState = NO_IMAGES_YET;
  if (p.width <= 0 || p.height <= 0) { std::cerr << "settings: Camera.width / Camera.height missing" << std::endl; std::exit(-1); }
  const char* dev = std::getenv("VDO_DEVICE");
  for (int k = 0; k < 5; ++k)
    if (vdo_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &ctx_[k]) != VDO_OK) throw std::runtime_error(std::string("VDO_SLAM::Tracking: no HIP device: ") + vdo_last_error());   // (GPU failures throw; only settings / sensor errors exit, as in the reference)
  pipe_.reset(new FramePipeline(ctx_[0], ctx_[1], p, ctx_[2], ctx_[3], ctx_[4]));
  if (!pipe_->ok()) throw std::runtime_error(std::string("VDO_SLAM::Tracking: FramePipeline: ") + vdo_last_error());
  pipe_->AttachMap(mpMap);
}

Tracking::~Tracking() {
  pipe_.reset();
  for (int k = 0; k < 5; ++k) if (ctx_[k]) vdo_ctx_destroy(ctx_[k]);
}

cv::Mat Tracking::GrabImageRGBD(const cv::Mat& imRGB, cv::Mat& imD, const cv::Mat& imFlow, const cv::Mat& maskSEM, const cv::Mat& mTcw_gt,
                                const std::vector<std::vector<float> >& vObjPose_gt, const double&, cv::Mat&, const int& nImage) {
  StopFrame = nImage - 1;
  if (!have_frame_) f_id = 0;
  const auto t_call = std::chrono::steady_clock::now();
  mLastProcessedState = mState;
  if (mState == NO_IMAGES_YET) mState = NOT_INITIALIZED;
  // The device images were sized from Camera.width / Camera.height of the settings file: every input must have exactly that
  // size, the reference's element types (src/System.h:45-51) and contiguous rows - anything else would be read out of bounds.
  // The reference has no error channel: an empty Mat + a message on stderr.
  {
    const int W = pipe_->params().width, H = pipe_->params().height;
    auto bad = [&](const cv::Mat& m, const char* name, int depth, int ch_lo, int ch_hi) {
      const bool ok = !m.empty() && m.rows == H && m.cols == W && m.depth() == depth && m.channels() >= ch_lo && m.channels() <= ch_hi &&
                      m.step == (size_t)m.cols * m.elemSize();
      if (!ok) std::cerr << "VDO_SLAM::Tracking::GrabImageRGBD: " << name << " is " << m.cols << "x" << m.rows << " (type " << m.type() << ", step " << m.step
                         << "), expected a continuous " << W << "x" << H << " image of the settings file's Camera.width/height" << std::endl;
      return !ok;
    };
    if (bad(imRGB, "imRGB", cv::CV_8U, 1, 4) || imRGB.channels() == 2 || bad(imD, "imD", cv::CV_32F, 1, 1) || bad(imFlow, "imFlow", cv::CV_32F, 2, 2) ||
        bad(maskSEM, "maskSEM", cv::CV_32S, 1, 1))
      return cv::Mat();
  }
  const int64_t n = (int64_t)imRGB.rows * imRGB.cols;
  // colour -> grey (cvtColor CV_RGB2GRAY / CV_BGR2GRAY, Tracking.cc:209-222); the caller's image is not touched
  const uint8_t* gray = imRGB.data;
  if (imRGB.channels() >= 3) {
    gray_.resize((size_t)n);
    if (vdo_rgb2gray(ctx_[0], imRGB.data, n, imRGB.channels(), mbRGB ? 1 : 0, gray_.data()) != VDO_OK) { std::cerr << vdo_last_error() << std::endl; return cv::Mat(); }
    gray = gray_.data();
  }
  // ground-truth rows gate the object tracker (label = row[1], Tracking.cc:332-336, 791-841)
  std::vector<int> labels;
  for (const auto& row : vObjPose_gt) if (row.size() > 1) labels.push_back((int)row[1]);
  pipe_->SetObjectGate(labels.data(), (int)labels.size());
  // K1 (Tracking.cc:180-204): OMD and KITTI convert disparity*factor to metres - on the device, on the uploaded map; the caller's
  // imD, which the reference converts in place, receives the converted map back.  VirtualKITTI only clamps negative values.
  bool metric = false;
  if (mTestData != OMD && mTestData != KITTI) {
    float* d = (float*)imD.data;
    for (int64_t i = 0; i < n; ++i) if (d[i] < 0) d[i] = 0;
    metric = true;
  }
  FrameCounts fc{};
  static const bool trace_slow = std::getenv("VDO_PIPE_TRACE_SLOW") != nullptr;      // (debug: where a call of > 5 ms went)
  auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  const double ms_pre = since(t_call);
  if (pipe_->StepHost(gray, (const float*)imD.data, (const float*)imFlow.data, (const int32_t*)maskSEM.data, metric, &fc, (float*)imD.data) != 0) return cv::Mat();
  const double ms_step = since(t_call);
  if (!metric && !pipe_->DepthConvertedOnHost() && pipe_->DownloadDepth((float*)imD.data) != 0) return cv::Mat();
  const double ms_depth = since(t_call);
  if (fc.n_recovered_masks > 0) pipe_->DownloadMask((int32_t*)maskSEM.data);      // UpdateMask writes through the shared header (Tracking.cc:3049-3068)
  if (trace_slow && since(t_call) > 5.0)
    std::fprintf(stderr, "[slow TrackRGBD f=%d] pre %.2f step %.2f depth %.2f mask(%d) %.2f ms\n", f_id, ms_pre, ms_step, ms_depth, fc.n_recovered_masks, since(t_call));
  // ground-truth camera pose of the frame relative to the first one, as Map::vmCameraPose_GT keeps it (src/Tracking.cc:319-328, 1113-1115;
  // Initialization() sets the first frame's to the identity, :1255-1256) - bookkeeping for SaveResults only
  if (!mTcw_gt.empty() && mTcw_gt.rows == 4 && mTcw_gt.cols == 4 && mTcw_gt.depth() == cv::CV_32F) {
    cv::Mat Tgt = cv::Mat::eye(4, 4, cv::CV_32F);
    if (!have_frame_) mOriginInv = mTcw_gt.clone();
    else Tgt = Converter::toInvMatrix(mTcw_gt) * mOriginInv;
    mpMap->vmCameraPose_GT.push_back(Converter::toInvMatrix(Tgt));
  }
  have_frame_ = true;
  // full batch optimisation after the last frame, KITTI only (Tracking.cc:1189-1210: `bGlobalBatch && mTestData==KITTI`)
  if (f_id == StopFrame && f_id > 1 && bGlobalBatch && mTestData == KITTI) {
    if (pipe_->FullBatchOptimization() != 0) return cv::Mat();     // graph built straight from the pipeline's GraphStore
  }
  ++f_id;
  mState = OK; bFirstFrame = false; bFrame2Frame = true;
  all_timing.push_back(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_call).count());
  cv::Mat Tcw(4, 4, cv::CV_32F);
  std::memcpy(Tcw.data, pipe_->Tcw_out_, 64);
  return Tcw;
}

// The reference's per-frame containers, materialised from the pipeline's flat arrays (see System.h).  Ends a pending object stage first.
void Tracking::SyncFrameState() {
  FramePipeline& P = *pipe_;
  P.Flush();
  Frame& F = mCurrentFrame;
  F.mTcw = cv::Mat(4, 4, cv::CV_32F);
  std::memcpy(F.mTcw.data, P.Tcw_out_, 64);
  auto point3 = [](const float* p) { cv::Mat m(3, 1, cv::CV_32F); m.at<float>(0) = p[0]; m.at<float>(1) = p[1]; m.at<float>(2) = p[2]; return m; };
  {   // RenewFrameInfo, static part (src/Tracking.cc:2780-2812)
    const FramePipeline::StaSet& S = P.StaticSet();
    const size_t n = S.x.size();
    F.N_s_tmp = (int)n;
    F.mvStatKeysTmp.resize(n); F.mvCorres.resize(n); F.mvFlowNext.resize(n); F.mvStatDepthTmp = S.d; F.mvStat3DPointTmp.resize(n);
    for (size_t i = 0; i < n; ++i) {
      F.mvStatKeysTmp[i] = cv::KeyPoint(S.x[i], S.y[i], 0, 0, 0, -1);
      F.mvCorres[i] = cv::KeyPoint(S.cx[i], S.cy[i], 0, 0, 0, -1);
      F.mvFlowNext[i] = cv::Point2f(S.fx[i], S.fy[i]);
      F.mvStat3DPointTmp[i] = point3(S.xyz.data() + 3 * i);
    }
  }
  {   // RenewFrameInfo, objects (:2984-2991)
    const FramePipeline::ObjSet& O = P.ObjectSet();
    const size_t n = O.x.size();
    F.mvObjKeys.resize(n); F.mvObjCorres.resize(n); F.mvObjFlowNext.resize(n); F.mvObjDepth = O.d; F.mvObj3DPoint.resize(n);
    F.vSemObjLabel.assign(O.sem.begin(), O.sem.end()); F.vObjLabel.assign(O.label.begin(), O.label.end());
    for (size_t i = 0; i < n; ++i) {
      F.mvObjKeys[i] = cv::KeyPoint(O.x[i], O.y[i], 0, 0, 0, -1);
      F.mvObjCorres[i] = cv::KeyPoint(O.cx[i], O.cy[i], 0, 0, 0, -1);
      F.mvObjFlowNext[i] = cv::Point2f(O.fx[i], O.fy[i]);
      F.mvObj3DPoint[i] = point3(O.xyz.data() + 3 * i);
    }
  }
  {   // per object of the frame (:836-933)
    const std::vector<int32_t>&sp = P.ObjSemPosition(), &ml = P.ObjModLabel();
    const std::vector<uint8_t>& st = P.ObjStat();
    const std::vector<float>& Hm = P.ObjMod();
    const size_t n = sp.size();
    F.nSemPosition.assign(sp.begin(), sp.end()); F.nModLabel.assign(ml.begin(), ml.end());
    F.bObjStat.resize(n); F.vObjMod.resize(n);
    for (size_t a = 0; a < n; ++a) {
      F.bObjStat[a] = a < st.size() && st[a] != 0;
      F.vObjMod[a] = cv::Mat::eye(4, 4, cv::CV_32F);
      if (16 * (a + 1) <= Hm.size()) std::memcpy(F.vObjMod[a].data, Hm.data() + 16 * a, 64);
    }
  }
  {   // the semi-dense samples of the frame (Frame::Frame :201-228 -> mvTmpObj*, :2925-2983 reads them)
    const FramePipeline::ObjSet& T = P.ObjectSamples();
    const size_t n = std::min<size_t>((size_t)std::max(P.NumObjectSamples(), 0), T.x.size());
    mvTmpObjKeys.resize(n); mvTmpObjCorres.resize(n); mvTmpObjFlowNext.resize(n);
    mvTmpObjDepth.assign(T.d.begin(), T.d.begin() + n); mvTmpSemObjLabel.assign(T.sem.begin(), T.sem.begin() + n);
    for (size_t i = 0; i < n; ++i) {
      mvTmpObjKeys[i] = cv::KeyPoint(T.x[i], T.y[i], 0, 0, 0, -1);
      mvTmpObjCorres[i] = cv::KeyPoint(T.cx[i], T.cy[i], 0, 0, 0, -1);
      mvTmpObjFlowNext[i] = cv::Point2f(T.fx[i], T.fy[i]);
    }
  }
  max_id = P.MaxId();
  if (have_frame_) {                   // K1's metric depth map and the mask as UpdateMask left it
    const int W = P.params().width, H = P.params().height;
    mDepthMap.create(H, W, cv::CV_32F); mSegMap.create(H, W, cv::CV_32SC1);
    P.DownloadDepth((float*)mDepthMap.data);
    P.DownloadMask((int32_t*)mSegMap.data);
  }
}

System::System(const std::string& strSettingsFile, const eSensor sensor) : mSensor(sensor) {
  std::ifstream f(strSettingsFile.c_str());
  if (!f.is_open()) { std::cerr << "Failed to open settings file at: " << strSettingsFile << std::endl; std::exit(-1); }
  mpMap = new Map();
  mpTracker = new Tracking(this, mpMap, strSettingsFile, mSensor);
}

System::~System() { delete mpTracker; delete mpMap; }

cv::Mat System::TrackRGBD(const cv::Mat& im, cv::Mat& depthmap, const cv::Mat& flowmap, const cv::Mat& masksem, const cv::Mat& mTcw_gt,
                          const std::vector<std::vector<float> >& vObjPose_gt, const double& timestamp, cv::Mat& imTraj, const int& nImage) {
  if (mSensor != RGBD) { std::cerr << "ERROR: you called TrackRGBD but input sensor was not set to RGBD." << std::endl; std::exit(-1); }
  return mpTracker->GrabImageRGBD(im, depthmap, flowmap, masksem, mTcw_gt, vObjPose_gt, timestamp, imTraj, nImage);
}

// The Map (reference format) is materialised from the pipeline's flat store when somebody looks at it.
Map* System::map() { mpTracker->pipeline()->SyncMap(); return mpMap; }

void System::SaveResults(const std::string& filename) {
  map();
  auto row = [](std::ofstream& o, const cv::Mat& T) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) o << T.at<float>(r, c) << " ";
    o << 0.0 << " " << 0.0 << " " << 0.0 << " " << 1.0 << std::endl;
  };
  const int start_frame = 0;
  auto poses = [&](const char* name, const std::vector<cv::Mat>& P) {      // src/System.cc:125-178
    std::ofstream o((filename + name).c_str(), std::ios::trunc);
    for (size_t i = 0; i < P.size(); ++i) { o << start_frame + i << " " << std::fixed << std::setprecision(9); row(o, P[i]); }
  };
  poses("initial_stereo_new.txt", mpMap->vmCameraPose);
  poses("refined_stereo_new.txt", mpMap->vmCameraPose_RF);
  poses("cam_pose_gt_stereo.txt", mpMap->vmCameraPose_GT);
  auto motions = [&](const char* name, const std::vector<std::vector<cv::Mat> >& M) {   // row format of src/System.cc:85-100, world frame (System.h)
    std::ofstream o((filename + name).c_str(), std::ios::trunc);
    for (size_t i = 0; i < M.size(); ++i)
      for (size_t j = 1; j < M[i].size(); ++j) { o << start_frame + i + 1 << " " << mpMap->vnRMLabel[i][j] << " " << std::fixed << std::setprecision(9); row(o, M[i][j]); }
  };
  motions("obj_mot_world_new.txt", mpMap->vmRigidMotion);
  motions("obj_mot_world_rf_new.txt", mpMap->vmRigidMotion_RF);
}

}  // namespace VDO_SLAM

// ---- flat hooks (tests / Python) ----------------------------------------------------------------------------------------
extern "C" {
// The settings reader alone (no GPU): the 31 values of TrackingSettings in declaration order as doubles.  0, or -1 when the file
// cannot be opened.  tests/test_settings_yaml.py reads the reference's three example/*.yaml through it.
int host_settings_read(const char* path, double* out31) {
  VDO_SLAM::TrackingSettings t{};
  if (!VDO_SLAM::read_tracking_settings(path, t)) return -1;
  const double v[31] = {t.fx, t.fy, t.cx, t.cy, t.k1, t.k2, t.p1, t.p2, t.k3, t.bf, t.fps, (double)t.rgb, (double)t.n_features, t.scale_factor, (double)t.n_levels,
                        (double)t.ini_th, (double)t.min_th, (double)t.data_code, t.th_depth_bg, t.th_depth_obj, t.depth_map_factor, (double)t.max_track_bg,
                        (double)t.max_track_obj, t.sf_mg_thres, t.sf_ds_thres, (double)t.window_size, (double)t.overlap_size, (double)t.use_sample_feature,
                        (double)t.width, (double)t.height, 0.0};
  for (int i = 0; i < 31; ++i) out31[i] = v[i];
  return 0;
}
// (the C++ classes keep the reference's behaviour - exit(-1) on an unreadable settings file, no error channel; these hooks are
// reached through ctypes, so they check first and turn failures into return codes instead of ending the host process)
VDO_SLAM::System* host_system_create(const char* settings) {
  {
    std::ifstream f(settings);
    if (!f.is_open()) { std::fprintf(stderr, "host_system_create: cannot open %s\n", settings); return nullptr; }
  }
  {   // what Tracking::Tracking would exit(-1) on - lens distortion (not supported by this build), no image size - refused here with a message instead
    VDO_SLAM::TrackingSettings t;
    std::map<std::string, double> raw;
    if (!VDO_SLAM::read_tracking_settings(settings, t, &raw)) { std::fprintf(stderr, "host_system_create: cannot parse %s\n", settings); return nullptr; }
    if (t.k1 != 0 || t.k2 != 0 || t.p1 != 0 || t.p2 != 0 || t.k3 != 0) { std::fprintf(stderr, "host_system_create: lens distortion (Camera.k1..k3, p1, p2) is not supported by this build\n"); return nullptr; }
    if (raw.count("Camera.width") == 0 || raw.count("Camera.height") == 0 || raw["Camera.width"] <= 0 || raw["Camera.height"] <= 0) {
      std::fprintf(stderr, "host_system_create: Camera.width / Camera.height missing in %s\n", settings); return nullptr;
    }
  }
  {
    const char* dev = std::getenv("VDO_DEVICE");
    vdo_ctx* probe = nullptr;
    if (vdo_ctx_create(dev ? std::atoi(dev) : 0, nullptr, &probe) != VDO_OK) { std::fprintf(stderr, "host_system_create: %s\n", vdo_last_error()); return nullptr; }
    vdo_ctx_destroy(probe);
  }
  try { return new VDO_SLAM::System(settings, VDO_SLAM::System::RGBD); }
  catch (const std::exception& e) { std::fprintf(stderr, "host_system_create: %s\n", e.what()); return nullptr; }
}
void host_system_destroy(VDO_SLAM::System* s) { delete s; }
// one TrackRGBD call on host images: im (h x w x channels u8), depth (in/out f32), flow (f32 x2), mask (in/out i32), ground-truth
// object rows [n_rows][row_len]; Tcw_out 16 floats.  Returns 0, -1 when the tracker returned an empty pose, -2 on a GPU failure.
int host_system_track(VDO_SLAM::System* s, const unsigned char* im, int channels, float* depth, const float* flow, int* mask, int w, int h,
                      const float* obj_rows, int n_rows, int row_len, int n_images, float* Tcw_out) {
  cv::Mat I(h, w, VDO_CV_MAKETYPE(cv::CV_8U, channels), (void*)im), D(h, w, cv::CV_32FC1, depth), Fl(h, w, cv::CV_32FC2, (void*)flow), M(h, w, cv::CV_32SC1, mask);
  cv::Mat gt = cv::Mat::eye(4, 4, cv::CV_32F), traj;
  std::vector<std::vector<float> > rows(n_rows);
  for (int i = 0; i < n_rows; ++i) rows[i].assign(obj_rows + (size_t)i * row_len, obj_rows + (size_t)(i + 1) * row_len);
  try {
    cv::Mat T = s->TrackRGBD(I, D, Fl, M, gt, rows, 0.0, traj, n_images);
    if (T.empty()) return -1;
    std::memcpy(Tcw_out, T.data, 64);
  } catch (const std::exception& e) { std::fprintf(stderr, "host_system_track: %s\n", e.what()); return -2; }
  return 0;
}
// Tracking::SyncFrameState() + a flat copy of what it filled (tests): what = 0 static set [10][n] (x y cx cy fx fy depth X Y Z), 1 object set
// [12][n] (... + vSemObjLabel, vObjLabel), 2 per object [n][19] (nSemPosition, nModLabel, bObjStat, vObjMod), 3 samples [8][n]
// (x y cx cy fx fy depth label), 4 scalars (max_id, mTcw).  Returns n (rows filled only if cap allows), -1 on a bad argument.
int host_system_frame_state(VDO_SLAM::System* s, int what, float* out, int cap) {
  try {
    VDO_SLAM::Tracking* T = s->tracker();
    T->SyncFrameState();
    const VDO_SLAM::Frame& F = T->mCurrentFrame;
    if (what == 0) {
      const int n = F.N_s_tmp;
      if (out && cap >= 10 * n)
        for (int i = 0; i < n; ++i) {
          const float v[10] = {F.mvStatKeysTmp[i].pt.x, F.mvStatKeysTmp[i].pt.y, F.mvCorres[i].pt.x, F.mvCorres[i].pt.y, F.mvFlowNext[i].x, F.mvFlowNext[i].y, F.mvStatDepthTmp[i],
                               F.mvStat3DPointTmp[i].at<float>(0), F.mvStat3DPointTmp[i].at<float>(1), F.mvStat3DPointTmp[i].at<float>(2)};
          for (int k = 0; k < 10; ++k) out[(size_t)k * n + i] = v[k];
        }
      return n;
    }
    if (what == 1) {
      const int n = (int)F.mvObjKeys.size();
      if (out && cap >= 12 * n)
        for (int i = 0; i < n; ++i) {
          const float v[12] = {F.mvObjKeys[i].pt.x, F.mvObjKeys[i].pt.y, F.mvObjCorres[i].pt.x, F.mvObjCorres[i].pt.y, F.mvObjFlowNext[i].x, F.mvObjFlowNext[i].y, F.mvObjDepth[i],
                               F.mvObj3DPoint[i].at<float>(0), F.mvObj3DPoint[i].at<float>(1), F.mvObj3DPoint[i].at<float>(2), (float)F.vSemObjLabel[i], (float)F.vObjLabel[i]};
          for (int k = 0; k < 12; ++k) out[(size_t)k * n + i] = v[k];
        }
      return n;
    }
    if (what == 2) {
      const int n = (int)F.nSemPosition.size();
      if (out && cap >= 19 * n)
        for (int a = 0; a < n; ++a) {
          float* o = out + 19 * (size_t)a;
          o[0] = (float)F.nSemPosition[a]; o[1] = (float)F.nModLabel[a]; o[2] = F.bObjStat[a] ? 1.f : 0.f;
          std::memcpy(o + 3, F.vObjMod[a].data, 64);
        }
      return n;
    }
    if (what == 3) {
      const int n = (int)T->mvTmpObjKeys.size();
      if (out && cap >= 8 * n)
        for (int i = 0; i < n; ++i) {
          const float v[8] = {T->mvTmpObjKeys[i].pt.x, T->mvTmpObjKeys[i].pt.y, T->mvTmpObjCorres[i].pt.x, T->mvTmpObjCorres[i].pt.y, T->mvTmpObjFlowNext[i].x, T->mvTmpObjFlowNext[i].y,
                              T->mvTmpObjDepth[i], (float)T->mvTmpSemObjLabel[i]};
          for (int k = 0; k < 8; ++k) out[(size_t)k * n + i] = v[k];
        }
      return n;
    }
    if (what == 4) {
      if (out && cap >= 17) { out[0] = (float)T->max_id; std::memcpy(out + 1, F.mTcw.data, 64); }
      return 1;
    }
  } catch (const std::exception& e) { std::fprintf(stderr, "host_system_frame_state: %s\n", e.what()); }
  return -1;
}
// throughput mode of the shell: the object stage of a frame ends inside the next TrackRGBD call (same results, the object motions of
// frame k become visible with frame k+1; the final batch optimisation / SaveResults / map() flush it).  Off by default: the
// reference has everything done when TrackRGBD returns.
void host_system_set_defer(VDO_SLAM::System* s, int on) { s->tracker()->pipeline()->SetDeferObjects(on != 0); }
void host_system_timing(VDO_SLAM::System* s, double* ms11) { for (int i = 0; i < 11; ++i) ms11[i] = s->tracker()->pipeline()->ms_[i]; }
int host_system_flush(VDO_SLAM::System* s) { return s->tracker()->pipeline()->Flush(nullptr); }
int host_system_motions(VDO_SLAM::System* s, int cap, int* sem_label, float* H16) {
  const auto& m = s->tracker()->pipeline()->motions_;
  const int n = std::min(cap, (int)m.size());
  for (int a = 0; a < n; ++a) { sem_label[a] = m[a].sem_label; std::memcpy(H16 + 16 * a, m[a].H, 64); }
  return (int)m.size();
}
int host_system_refined_poses(VDO_SLAM::System* s, int cap, float* Twc16) {
  const auto& P = s->map()->vmCameraPose_RF;
  const int n = std::min(cap, (int)P.size());
  for (int i = 0; i < n; ++i) std::memcpy(Twc16 + 16 * i, P[i].data, 64);
  return (int)P.size();
}
void host_system_save(VDO_SLAM::System* s, const char* path) { s->SaveResults(path); }
// flat copy of the Map (reference format, include/Map.h:35-84) after System::map() has brought it up to date - the layout of oracle/ref's vdo_ref_system_map_export: what = 0
// vmCameraPose [F][16], 1 vmCameraPose_RF, 2 vmRigidMotion [n][16], 3 vmRigidMotion_RF, 4 vnRMLabel [n], 5 vp3DPointSta [n][3], 6 vp3DPointDyn [n][3], 7 motions per frame [F-1]
long host_system_map_export(VDO_SLAM::System* s, int what, float* out, long cap) {
  try {
    VDO_SLAM::Map* m = s->map();
    long n = 0;
    auto put = [&](float v) { if (out && n < cap) out[n] = v; ++n; };
    auto put_mat = [&](const cv::Mat& M) { const float* p = (const float*)M.data; for (int i = 0; i < M.rows * M.cols; ++i) put(p[i]); };
    if (what == 0 || what == 1) { for (const cv::Mat& T : (what ? m->vmCameraPose_RF : m->vmCameraPose)) put_mat(T); }
    else if (what == 2 || what == 3) { for (const auto& fr : (what == 3 ? m->vmRigidMotion_RF : m->vmRigidMotion)) for (const cv::Mat& T : fr) put_mat(T); }
    else if (what == 4) { for (const auto& fr : m->vnRMLabel) for (int l : fr) put((float)l); }
    else if (what == 5 || what == 6) { for (const auto& fr : (what == 6 ? m->vp3DPointDyn : m->vp3DPointSta)) for (const cv::Mat& X : fr) put_mat(X); }
    else if (what == 7) { for (const auto& fr : m->vmRigidMotion) put((float)fr.size()); }
    else return -1;
    return n;
  } catch (const std::exception& e) { std::fprintf(stderr, "host_system_map_export: %s\n", e.what()); return -1; }
}
// the tracklets Track() has built so far (GetStaticTrack / GetDynamicTrackNew, src/Tracking.cc:2201-2421): which = 0 static, 1 dynamic; off == NULL -> sizes only
int host_pipeline_tracks(VDO_SLAM::FramePipeline* fp, int which, int64_t* sizes2, int32_t* off, int32_t* frame, int32_t* feat, int32_t* obj);
int host_system_tracks(VDO_SLAM::System* s, int which, int64_t* sizes2, int32_t* off, int32_t* frame, int32_t* feat, int32_t* obj) {
  return host_pipeline_tracks(s->tracker()->pipeline(), which, sizes2, off, frame, feat, obj);
}
}
