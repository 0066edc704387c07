// extern "C" hooks that let the Python test-suite drive the C++ host classes of this directory
// (ORBextractor / Frame / Optimizer with the reference's signatures).  Not part of the product
// ABI (that is include/vdo_slam_hip.h); they only marshal flat arrays into Map / Frame objects.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "Converter.h"
#include "Frame.h"
#include "Map.h"
#include "ORBextractor.h"
#include "Optimizer.h"

using namespace VDO_SLAM;

extern "C" {

struct host_map_flat {
  int n_frames;
  const float* K;                                    // 3x3
  const float* cam_pose;                             // [F][16]
  const int* sta_cnt; const float *sta_uv, *sta_d, *sta_xw;
  int n_tr_sta; const int *tr_sta_len, *tr_sta_pairs;
  const int* dyn_cnt; const float *dyn_uv, *dyn_d, *dyn_xw;
  int n_tr_dyn; const int *tr_dyn_len, *tr_dyn_pairs, *obj_of_dyn;
  const int* rm_cnt; const float* rm; const int* rm_label;
};

static cv::Mat mat44(const float* p) { cv::Mat m(4, 4, cv::CV_32F); std::memcpy(m.data, p, 64); return m; }
static cv::Mat mat31(const float* p) { cv::Mat m(3, 1, cv::CV_32F); std::memcpy(m.data, p, 12); return m; }

int host_batch_optimization(const host_map_flat* f, int partial_window, float* cam_pose_out, float* rm_out,
                            float* sta_xw_out, float* dyn_xw_out, vdo_lm_stats* st) {
  Map map;
  const int F = f->n_frames;
  cv::Mat K(3, 3, cv::CV_32F);
  std::memcpy(K.data, f->K, 36);
  size_t so = 0, dof = 0, ro = 0;
  map.vpFeatSta.resize(F); map.vfDepSta.resize(F); map.vp3DPointSta.resize(F);
  map.vpFeatDyn.resize(F); map.vfDepDyn.resize(F); map.vp3DPointDyn.resize(F);
  for (int i = 0; i < F; ++i) {
    map.vmCameraPose.push_back(mat44(f->cam_pose + 16 * i));
    for (int j = 0; j < f->sta_cnt[i]; ++j, ++so) {
      map.vpFeatSta[i].push_back(cv::KeyPoint(f->sta_uv[2 * so], f->sta_uv[2 * so + 1], 0));
      map.vfDepSta[i].push_back(f->sta_d[so]);
      map.vp3DPointSta[i].push_back(mat31(f->sta_xw + 3 * so));
    }
    for (int j = 0; j < f->dyn_cnt[i]; ++j, ++dof) {
      map.vpFeatDyn[i].push_back(cv::KeyPoint(f->dyn_uv[2 * dof], f->dyn_uv[2 * dof + 1], 0));
      map.vfDepDyn[i].push_back(f->dyn_d[dof]);
      map.vp3DPointDyn[i].push_back(mat31(f->dyn_xw + 3 * dof));
    }
    if (i < F - 1) {
      std::vector<cv::Mat> mots; std::vector<int> labs;
      for (int j = 0; j < f->rm_cnt[i]; ++j, ++ro) { mots.push_back(mat44(f->rm + 16 * ro)); labs.push_back(f->rm_label[ro]); }
      map.vmRigidMotion.push_back(mots); map.vnRMLabel.push_back(labs);
    }
  }
  map.vmCameraPose_RF = map.vmCameraPose; map.vmRigidMotion_RF = map.vmRigidMotion;
  size_t po = 0;
  for (int t = 0; t < f->n_tr_sta; ++t) {
    std::vector<std::pair<int, int> > tr;
    for (int k = 0; k < f->tr_sta_len[t]; ++k, ++po) tr.push_back(std::make_pair(f->tr_sta_pairs[2 * po], f->tr_sta_pairs[2 * po + 1]));
    map.TrackletSta.push_back(tr);
  }
  po = 0;
  for (int t = 0; t < f->n_tr_dyn; ++t) {
    std::vector<std::pair<int, int> > tr;
    for (int k = 0; k < f->tr_dyn_len[t]; ++k, ++po) tr.push_back(std::make_pair(f->tr_dyn_pairs[2 * po], f->tr_dyn_pairs[2 * po + 1]));
    map.TrackletDyn.push_back(tr);
    map.nObjID.push_back(f->obj_of_dyn[t]);
  }
  try {
    if (partial_window > 0) Optimizer::PartialBatchOptimization(&map, K, partial_window);
    else Optimizer::FullBatchOptimization(&map, K);
  } catch (const std::exception& e) { std::fprintf(stderr, "host_batch_optimization: %s\n", e.what()); return -1; }
  if (st) *st = Optimizer::last_batch_stats;
  so = dof = ro = 0;
  for (int i = 0; i < F; ++i) {
    const cv::Mat& T = partial_window > 0 ? map.vmCameraPose[i] : map.vmCameraPose_RF[i];
    std::memcpy(cam_pose_out + 16 * i, T.data, 64);
    for (int j = 0; j < f->sta_cnt[i]; ++j, ++so) std::memcpy(sta_xw_out + 3 * so, map.vp3DPointSta[i][j].data, 12);
    for (int j = 0; j < f->dyn_cnt[i]; ++j, ++dof) std::memcpy(dyn_xw_out + 3 * dof, map.vp3DPointDyn[i][j].data, 12);
    if (i < F - 1)
      for (int j = 0; j < f->rm_cnt[i]; ++j, ++ro) {
        const cv::Mat& M = partial_window > 0 ? map.vmRigidMotion[i][j] : map.vmRigidMotion_RF[i][j];
        std::memcpy(rm_out + 16 * ro, M.data, 64);
      }
  }
  return 0;
}

// Frame::Frame through the host classes (after GrabImageRGBD's depth preprocessing).  Returns N (ORB keypoints).
int host_frame(const unsigned char* gray, float* depth_raw_inout, const float* flow, const int* mask, int w, int h,
               float bf, float depth_factor, float th_bg, float th_obj,
               float* kx, float* ky, int* koct, int cap,
               int* n_stat, float* stat_corr /*[cap][2]*/, float* stat_depth, int* n_obj, float* obj_key /*[capo][2]*/, int* obj_label, int capo) {
  try {
  static ORBextractor* orb = nullptr;
  if (!orb) orb = new ORBextractor(2500, 1.2f, 8, 20, 7);
  if (vdo_depth_preprocess(HostContext(), depth_raw_inout, (int64_t)w * h, bf, depth_factor, 0) != VDO_OK) return -1;   // Tracking.cc:180-204
  cv::Mat G(h, w, cv::CV_8UC1, (void*)gray), D(h, w, cv::CV_32FC1, depth_raw_inout), Fl(h, w, cv::CV_32FC2, (void*)flow), M(h, w, cv::CV_32SC1, (void*)mask);
  cv::Mat K = cv::Mat::eye(3, 3, cv::CV_32F), dist = cv::Mat::zeros(4, 1, cv::CV_32F);
  K.at<float>(0, 0) = 721.5377f; K.at<float>(1, 1) = 721.5377f; K.at<float>(0, 2) = 609.5593f; K.at<float>(1, 2) = 172.854f;
  Frame fr(G, D, Fl, M, 0.0, orb, K, dist, bf, th_bg, th_obj, 0);
  for (int i = 0; i < fr.N && i < cap; ++i) { kx[i] = fr.mvKeys[i].pt.x; ky[i] = fr.mvKeys[i].pt.y; koct[i] = fr.mvKeys[i].octave; }
  *n_stat = fr.N_s_tmp;
  for (int i = 0; i < fr.N_s_tmp && i < cap; ++i) { stat_corr[2 * i] = fr.mvCorres[i].pt.x; stat_corr[2 * i + 1] = fr.mvCorres[i].pt.y; stat_depth[i] = fr.mvStatDepthTmp[i]; }
  *n_obj = (int)fr.mvObjKeys.size();
  for (int i = 0; i < *n_obj && i < capo; ++i) { obj_key[2 * i] = fr.mvObjKeys[i].pt.x; obj_key[2 * i + 1] = fr.mvObjKeys[i].pt.y; obj_label[i] = fr.vSemObjLabel[i]; }
  return fr.N;
  } catch (const std::exception& e) { std::fprintf(stderr, "host_frame: %s\n", e.what()); return -1; }
}

// Optimizer::PoseOptimizationFlow2Cam through the host classes.  Tcw_last / Tcw_init: 4x4 float row-major.
int host_pose_optimization_flow2cam(int n, const float* last_xy, const float* flow, const float* depth, const float* Tcw_last,
                                    const float* Tcw_init, float* Tcw_out, int* match_out, float* cur_xy_out) {
  Frame last, cur;
  Frame::fx = 721.5377f; Frame::fy = 721.5377f; Frame::cx = 609.5593f; Frame::cy = 172.854f;
  last.mTcw = mat44(Tcw_last); cur.mTcw = mat44(Tcw_init);
  std::vector<int> match(n);
  for (int i = 0; i < n; ++i) {
    last.mvStatKeys.push_back(cv::KeyPoint(last_xy[2 * i], last_xy[2 * i + 1], 0));
    last.mvFlowNext.push_back(cv::Point2f(flow[2 * i], flow[2 * i + 1]));
    last.mvStatDepth.push_back(depth[i]);
    cur.mvStatKeys.push_back(cv::KeyPoint(last_xy[2 * i] + flow[2 * i], last_xy[2 * i + 1] + flow[2 * i + 1], 0));
    match[i] = i;
  }
  int inl;
  try { inl = Optimizer::PoseOptimizationFlow2Cam(&cur, &last, match); } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return -1; }
  std::memcpy(Tcw_out, cur.mTcw.data, 64);
  for (int i = 0; i < n; ++i) { match_out[i] = match[i]; cur_xy_out[2 * i] = cur.mvStatKeys[i].pt.x; cur_xy_out[2 * i + 1] = cur.mvStatKeys[i].pt.y; }
  return inl;
}

// Optimizer::PoseOptimizationNew through the host classes: the last frame's static keys + depths are back-projected with its
// pose (UnprojectStereoStat), the current frame's keys are the observations.  Returns the inlier count.
int host_pose_optimization_new(int n, const float* last_xy, const float* depth, const float* cur_xy, const float* Tcw_last, const float* Tcw_init,
                               float* Tcw_out, int* match_out) {
  Frame last, cur;
  Frame::fx = 721.5377f; Frame::fy = 721.5377f; Frame::cx = 609.5593f; Frame::cy = 172.854f; Frame::invfx = 1.0f / Frame::fx; Frame::invfy = 1.0f / Frame::fy;
  last.mTcw = mat44(Tcw_last); cur.mTcw = mat44(Tcw_init);
  std::vector<int> match(n);
  for (int i = 0; i < n; ++i) {
    last.mvStatKeys.push_back(cv::KeyPoint(last_xy[2 * i], last_xy[2 * i + 1], 0));
    last.mvStatDepth.push_back(depth[i]);
    cur.mvStatKeys.push_back(cv::KeyPoint(cur_xy[2 * i], cur_xy[2 * i + 1], 0));
    match[i] = i;
  }
  int inl = -1;
  try { inl = Optimizer::PoseOptimizationNew(&cur, &last, match); } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return -1; }
  std::memcpy(Tcw_out, cur.mTcw.data, 64);
  for (int i = 0; i < n; ++i) match_out[i] = match[i];
  return inl;
}

// Optimizer::PoseOptimizationObjMot through the host classes.  Returns the number of inliers; H_out = the returned motion.
int host_pose_optimization_objmot(int n, const float* last_xy, const float* depth, const float* cur_xy, const float* Tcw_last, const float* Tcw_cur,
                                  const float* init_model, float* H_out, int* inlier_flag, int* obj_label_out) {
  Frame last, cur;
  Frame::fx = 721.5377f; Frame::fy = 721.5377f; Frame::cx = 609.5593f; Frame::cy = 172.854f; Frame::invfx = 1.0f / Frame::fx; Frame::invfy = 1.0f / Frame::fy;
  last.mTcw = mat44(Tcw_last); cur.mTcw = mat44(Tcw_cur); cur.mInitModel = mat44(init_model);
  std::vector<int> ids(n), inliers;
  for (int i = 0; i < n; ++i) {
    last.mvObjKeys.push_back(cv::KeyPoint(last_xy[2 * i], last_xy[2 * i + 1], 0));
    last.mvObjDepth.push_back(depth[i]);
    cur.mvObjKeys.push_back(cv::KeyPoint(cur_xy[2 * i], cur_xy[2 * i + 1], 0));
    cur.vObjLabel.push_back(7);
    ids[i] = i;
  }
  cv::Mat H;
  try { H = Optimizer::PoseOptimizationObjMot(&cur, &last, ids, inliers); } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return -1; }
  std::memcpy(H_out, H.data, 64);
  for (int i = 0; i < n; ++i) { inlier_flag[i] = 0; obj_label_out[i] = cur.vObjLabel[i]; }
  for (int id : inliers) inlier_flag[id] = 1;
  return (int)inliers.size();
}

}  // extern "C"
