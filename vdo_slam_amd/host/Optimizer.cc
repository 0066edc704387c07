// Optimizer host class: same static entry points as the reference (include/Optimizer.h:25-32).
// The graph CONSTRUCTION follows the reference builders (src/Optimizer.cc:1259-1766 full batch,
// :44-637 partial batch, :2356-2442 / :2778-2865 per-frame) but fills the SoA structures of the
// C-ABI instead of allocating g2o vertices/edges; optimisation and gating run in libvdo_hip.
#include "Optimizer.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <iostream>

#include "Converter.h"

namespace VDO_SLAM {

vdo_lm_stats Optimizer::last_batch_stats;

namespace {

void die(const char* what) {   // the reference has no error channel (it exits); a GPU failure surfaces as an exception the flat hooks turn into a return code
  throw std::runtime_error(std::string("VDO_SLAM::Optimizer: ") + what + ": " + vdo_last_error());
}

// Eigen::Quaterniond(Matrix3d) + normalisation (+ sign fix when `positive_w`), then back to a
// rotation matrix: what Converter::toSE3Quat -> SE3Quat::operator Isometry3d produces.
void quat_roundtrip(const double R[9], bool positive_w, double Ro[9]) {
  double x, y, z, w;
  double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0); w = 0.5 * t; t = 0.5 / t;
    x = (R[7] - R[5]) * t; y = (R[2] - R[6]) * t; z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    double c[3];
    c[i] = 0.5 * t; t = 0.5 / t;
    w = (R[3 * k + j] - R[3 * j + k]) * t;
    c[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    c[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    x = c[0]; y = c[1]; z = c[2];
  }
  if (positive_w && w < 0) { x = -x; y = -y; z = -z; w = -w; }
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Ro[0] = 1 - (tyy + tzz); Ro[1] = txy - twz; Ro[2] = txz + twy;
  Ro[3] = txy + twz; Ro[4] = 1 - (txx + tzz); Ro[5] = tyz - twx;
  Ro[6] = txz - twy; Ro[7] = tyz + twx; Ro[8] = 1 - (txx + tyy);
}

// cv::Mat (4x4 float) -> VertexSE3 estimate / EdgeSE3 measurement (12 doubles) via toSE3Quat
void to_iso12(const cv::Mat& T, double out[12]) {
  double R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = T.at<float>(i, j);
  quat_roundtrip(R, true, out);
  out[9] = T.at<float>(0, 3); out[10] = T.at<float>(1, 3); out[11] = T.at<float>(2, 3);
}
// VertexSE3::getEstimateData (toVectorQT) -> q.matrix() -> Converter::toCvSE3 (write-back, :2094-2122)
cv::Mat iso12_to_cv(const double p[12]) {
  double Ro[9];
  quat_roundtrip(p, false, Ro);
  return Converter::toCvSE3(Ro, p + 9);
}

struct GraphBuilder {
  std::vector<double> pose, point, eb_z[3], eb_w, et_z[3], et_w, ep_z, ep_info, pr_z, pr_info;
  std::vector<int32_t> eb_pose, eb_point, et_p1, et_p2, et_pose, ep_i, ep_j, pr_pose;
  int add_pose(const cv::Mat& T) { double p[12]; to_iso12(T, p); pose.insert(pose.end(), p, p + 12); return (int)(pose.size() / 12) - 1; }
  int add_point(const cv::Mat& Xw) { for (int k = 0; k < 3; ++k) point.push_back((double)Xw.at<float>(k)); return (int)(point.size() / 3) - 1; }
  void add_eb(int cam, int pt, const cv::Mat& Xc, double w) {
    eb_pose.push_back(cam); eb_point.push_back(pt);
    for (int k = 0; k < 3; ++k) eb_z[k].push_back((double)Xc.at<float>(k));
    eb_w.push_back(w);
  }
  void add_et(int p1, int p2, int h, double w) {
    et_p1.push_back(p1); et_p2.push_back(p2); et_pose.push_back(h);
    for (int k = 0; k < 3; ++k) et_z[k].push_back(0.0);
    et_w.push_back(w);
  }
  static void push_info(std::vector<double>& v, double s) { for (int i = 0; i < 36; ++i) v.push_back(i % 7 == 0 ? s : 0.0); }
  void add_ep(int i, int j, const cv::Mat& Z, double s) {
    ep_i.push_back(i); ep_j.push_back(j);
    double z[12]; to_iso12(Z, z); ep_z.insert(ep_z.end(), z, z + 12);
    push_info(ep_info, s);
  }
  void add_prior(int v, const cv::Mat& Z, double s) {
    pr_pose.push_back(v);
    double z[12]; to_iso12(Z, z); pr_z.insert(pr_z.end(), z, z + 12);
    push_info(pr_info, s);
  }
  // run LM on the GPU; returns refined poses / points
  void optimize(double huber, int max_it, double gain, std::vector<double>& pose_out, std::vector<double>& point_out, vdo_lm_stats* st) {
    const int Eb = (int)eb_pose.size(), Et = (int)et_p1.size();
    std::vector<double> ebz(3 * (size_t)Eb), etz(3 * (size_t)Et);
    for (int k = 0; k < 3; ++k) { std::copy(eb_z[k].begin(), eb_z[k].end(), ebz.begin() + (size_t)k * Eb); std::copy(et_z[k].begin(), et_z[k].end(), etz.begin() + (size_t)k * Et); }
    vdo_ba_graph g{};
    g.n_pose = (int)(pose.size() / 12); g.n_point = (int)(point.size() / 3); g.n_eb = Eb; g.n_et = Et;
    g.n_ep = (int)ep_i.size(); g.n_prior = (int)pr_pose.size();
    g.pose = pose.data(); g.point = point.data();
    g.eb_pose = eb_pose.data(); g.eb_point = eb_point.data(); g.eb_z = ebz.data(); g.eb_w = eb_w.data();
    g.et_p1 = et_p1.data(); g.et_p2 = et_p2.data(); g.et_pose = et_pose.data(); g.et_z = etz.data(); g.et_w = et_w.data();
    g.ep_i = ep_i.data(); g.ep_j = ep_j.data(); g.ep_z = ep_z.data(); g.ep_info = ep_info.data();
    g.pr_pose = pr_pose.data(); g.pr_z = pr_z.data(); g.pr_info = pr_info.data();
    g.huber_eb = g.huber_et = g.huber_ep = huber;
    vdo_ba* ba = nullptr;
    if (vdo_ba_create(HostContext(), &g, &ba) != VDO_OK) die("vdo_ba_create");
    vdo_lm_options o{max_it, gain, std::getenv("VDO_VERBOSE") ? 1 : 0, 0, 0.0, 0};
    if (vdo_ba_optimize(ba, &o, st) != VDO_OK) die("vdo_ba_optimize");
    pose_out.resize(pose.size()); point_out.resize(point.size());
    if (vdo_ba_get_estimates(ba, pose_out.data(), point_out.data()) != VDO_OK) die("vdo_ba_get_estimates");
    vdo_ba_destroy(ba);
  }
};

cv::Mat point_to_cv(const double* p) {   // Converter::toCvMat(Matrix<double,3,1>)
  cv::Mat m(3, 1, cv::CV_32F);
  for (int k = 0; k < 3; ++k) m.at<float>(k) = (float)p[k];
  return m;
}

// position of every feature inside its tracklet (the reference searches linearly, :1456-1463)
void label_tracks(const std::vector<std::vector<std::pair<int, int> > >& tracks, std::vector<std::vector<int> >& lab, std::vector<std::vector<int> >& pos) {
  for (int t = 0; t < (int)tracks.size(); ++t) {
    if (tracks[t].size() < 3) continue;     // track length >= 3 (:1275-1296)
    for (int k = 0; k < (int)tracks[t].size(); ++k) { lab[tracks[t][k].first][tracks[t][k].second] = t; pos[tracks[t][k].first][tracks[t][k].second] = k; }
  }
}

}  // namespace

// ------------------------------------------------------------------------------- per frame
static void fill_flow2(vdo_flow2_problem& p, Frame* pCurFrame, Frame* pLastFrame, const cv::Mat& Init, double info_prior, int max_it) {
  p.K[0] = pCurFrame->fx; p.K[1] = pCurFrame->fy; p.K[2] = pCurFrame->cx; p.K[3] = pCurFrame->cy;
  const cv::Mat Twl = Converter::toInvMatrix(pLastFrame->mTcw);    // Rwl = Rlw^T, twl = -Rlw^T tlw in fp32 (:2414-2420)
  Converter::toDouble16(Twl, p.Twl);
  Converter::toDouble16(Init, p.T0);
  p.info_flow = 0.1; p.info_prior = info_prior;
  p.huber_delta = (double)std::sqrt(0.04f);     // const float deltaMono = sqrt(rp_thres)
  p.chi2_gate = (double)0.04f;
  p.max_iterations = max_it; p.ref_quirks = 1;
}

int Optimizer::PoseOptimizationFlow2Cam(Frame* pCurFrame, Frame* pLastFrame, vector<int>& TemperalMatch) {
  const int N = (int)TemperalMatch.size();
  std::vector<double> obs(2 * (size_t)N), flo(2 * (size_t)N), dep(N);
  for (int i = 0; i < N; ++i) {
    const int m = TemperalMatch[i];
    obs[2 * i] = pLastFrame->mvStatKeys[m].pt.x; obs[2 * i + 1] = pLastFrame->mvStatKeys[m].pt.y;
    flo[2 * i] = pLastFrame->mvFlowNext[m].x; flo[2 * i + 1] = pLastFrame->mvFlowNext[m].y;    // ObtainFlowDepthCamera
    dep[i] = pLastFrame->mvStatDepth[m];
  }
  if (N < 3) return 0;                                                // nInitialCorrespondences<3 (:2449-2450)
  vdo_flow2_problem p{};
  p.n = N; p.obs = obs.data(); p.flow = flo.data(); p.depth = dep.data();
  fill_flow2(p, pCurFrame, pLastFrame, pCurFrame->mTcw, 0.3, 100);
  vdo_flow2_result r;
  std::vector<double> fnew(2 * (size_t)N);
  std::vector<uint8_t> inl(N);
  if (vdo_flow2_optimize(HostContext(), &p, &r, fnew.data(), inl.data()) != VDO_OK) die("vdo_flow2_optimize");
  pCurFrame->SetPose(Converter::toCvMat(r.T));
  for (int i = 0; i < N; ++i) {
    if (!inl[i]) { TemperalMatch[i] = -1; continue; }                 // :2488-2493
    const int m = TemperalMatch[i];
    pCurFrame->mvStatKeys[m].pt.x = (float)(pLastFrame->mvStatKeys[m].pt.x + fnew[2 * i]);        // :2527-2532
    pCurFrame->mvStatKeys[m].pt.y = (float)(pLastFrame->mvStatKeys[m].pt.y + fnew[2 * i + 1]);
  }
  std::cout << "(Camera) inliers number/total numbers: " << r.n_inliers << "/" << N << std::endl;
  return r.n_inliers;
}

cv::Mat Optimizer::PoseOptimizationFlow2(Frame* pCurFrame, Frame* pLastFrame, const vector<int>& ObjId, std::vector<int>& InlierID) {
  const int N = (int)ObjId.size();
  if (N < 3) return cv::Mat::eye(4, 4, cv::CV_32F);                   // :2872-2873
  std::vector<double> obs(2 * (size_t)N), flo(2 * (size_t)N), dep(N);
  for (int i = 0; i < N; ++i) {
    const int m = ObjId[i];
    obs[2 * i] = pLastFrame->mvObjKeys[m].pt.x; obs[2 * i + 1] = pLastFrame->mvObjKeys[m].pt.y;
    flo[2 * i] = pLastFrame->mvObjFlowNext[m].x; flo[2 * i + 1] = pLastFrame->mvObjFlowNext[m].y;   // ObtainFlowDepthObject
    dep[i] = pLastFrame->mvObjDepth[m];
  }
  vdo_flow2_problem p{};
  p.n = N; p.obs = obs.data(); p.flow = flo.data(); p.depth = dep.data();
  fill_flow2(p, pCurFrame, pLastFrame, pCurFrame->mInitModel, 0.5, 200);
  vdo_flow2_result r;
  std::vector<double> fnew(2 * (size_t)N);
  std::vector<uint8_t> inl(N);
  if (vdo_flow2_optimize(HostContext(), &p, &r, fnew.data(), inl.data()) != VDO_OK) die("vdo_flow2_optimize");
  InlierID.clear();
  for (int i = 0; i < N; ++i) {
    if (!inl[i]) continue;
    const int m = ObjId[i];
    pCurFrame->mvObjKeys[m].pt.x = (float)(pLastFrame->mvObjKeys[m].pt.x + fnew[2 * i]);            // :2949-2950
    pCurFrame->mvObjKeys[m].pt.y = (float)(pLastFrame->mvObjKeys[m].pt.y + fnew[2 * i + 1]);
    InlierID.push_back(m);
  }
  std::cout << "(Object) inliers number/total numbers: " << r.n_inliers << "/" << N << std::endl;
  return Converter::toCvMat(r.T);
}

// ------------------------------------------------------------------------------- batch
void Optimizer::FullBatchOptimization(Map* pMap, const cv::Mat Calib_K) {
  const int N = (int)pMap->vpFeatSta.size();
  const auto& StaTracks = pMap->TrackletSta;
  const auto& DynTracks = pMap->TrackletDyn;
  std::vector<std::vector<int> > labS(N), posS(N), mkS(N), labD(N), posD(N), mkD(N);
  for (int i = 0; i < N; ++i) {
    labS[i].assign(pMap->vpFeatSta[i].size(), -1); posS[i] = labS[i]; mkS[i] = labS[i];
    labD[i].assign(pMap->vpFeatDyn[i].size(), -1); posD[i] = labD[i]; mkD[i] = labD[i];
  }
  label_tracks(StaTracks, labS, posS);
  label_tracks(DynTracks, labD, posD);
  std::vector<std::vector<int> > VertexID(N > 0 ? N - 1 : 0);
  for (int i = 0; i < N - 1; ++i) VertexID[i].assign(pMap->vnRMLabel[i].size(), -1);

  // information (float sigma^2 as in :1330-1335) and Huber delta (:1352)
  const float sigma2_cam = 0.001f, sigma2_3d_sta = 80, sigma2_obj_smo = 0.001f, sigma2_obj = 100, sigma2_3d_dyn = 80;
  const float deltaHuber = 0.0001f;
  const cv::Mat IDENT = cv::Mat::eye(4, 4, cv::CV_32F);
  GraphBuilder G;
  int PreFrame = -1;
  for (int i = 0; i < N; ++i) {
    const int cam = G.add_pose(pMap->vmCameraPose[i]);
    if (i == 0) G.add_prior(cam, pMap->vmCameraPose[i], 100000.0);                                  // :1364-1373
    else {
      VertexID[i - 1][0] = cam;
      G.add_ep(PreFrame, cam, pMap->vmRigidMotion[i - 1][0], 1.0 / sigma2_cam);                      // :1383-1399
    }
    // static features (:1404-1524)
    for (int j = 0; j < (int)labS[i].size(); ++j) {
      if (labS[i][j] == -1) continue;
      const int tr = labS[i][j], ps = posS[i][j];
      int pt;
      if (ps == 0) pt = G.add_point(pMap->vp3DPointSta[i][j]);
      else pt = mkS[StaTracks[tr][ps - 1].first][StaTracks[tr][ps - 1].second];
      if (pt < 0) continue;
      G.add_eb(cam, pt, Get3DinCamera(pMap->vpFeatSta[i][j], pMap->vfDepSta[i][j], Calib_K), 1.0 / sigma2_3d_sta);
      mkS[i][j] = pt;
    }
    // dynamic features + object motions (:1530-1747)
    if (i == 0) {
      for (int j = 0; j < (int)labD[i].size(); ++j) {
        if (labD[i][j] == -1) continue;
        const int pt = G.add_point(pMap->vp3DPointDyn[i][j]);
        G.add_eb(cam, pt, Get3DinCamera(pMap->vpFeatDyn[i][j], pMap->vfDepDyn[i][j], Calib_K), 1.0 / sigma2_3d_dyn);
        mkD[i][j] = pt;
      }
    } else {
      const int nmot = (int)pMap->vmRigidMotion[i - 1].size();
      std::vector<int> ObjUniqueID(nmot > 0 ? nmot - 1 : 0, -1);
      for (int j = 1; j < nmot; ++j) {
        const int mv = G.add_pose(IDENT);                                                           // motions start at identity (:1581)
        if (i > 2) {                                                                                // smoothness (:1593-1622)
          int TraceID = -1;
          for (int k = 0; k < (int)pMap->vnRMLabel[i - 2].size(); ++k)
            if (pMap->vnRMLabel[i - 2][k] == pMap->vnRMLabel[i - 1][j]) { TraceID = k; break; }
          if (TraceID != -1 && VertexID[i - 2][TraceID] >= 0) G.add_ep(VertexID[i - 2][TraceID], mv, IDENT, 1.0 / sigma2_obj_smo);
        }
        ObjUniqueID[j - 1] = mv;
        VertexID[i - 1][j] = mv;
      }
      for (int j = 0; j < (int)labD[i].size(); ++j) {
        if (labD[i][j] == -1) continue;
        const int tr = labD[i][j], ps = posD[i][j];
        int ObjPositionID = -1;
        for (int k = 1; k < (int)pMap->vnRMLabel[i - 1].size(); ++k)
          if (pMap->vnRMLabel[i - 1][k] == pMap->nObjID[tr]) { ObjPositionID = ObjUniqueID[k - 1]; break; }
        if (ObjPositionID == -1 && ps != 0) continue;
        const int pt = G.add_point(pMap->vp3DPointDyn[i][j]);
        G.add_eb(cam, pt, Get3DinCamera(pMap->vpFeatDyn[i][j], pMap->vfDepDyn[i][j], Calib_K), 1.0 / sigma2_3d_dyn);
        if (ps != 0) {
          const int prev = mkD[DynTracks[tr][ps - 1].first][DynTracks[tr][ps - 1].second];
          if (prev >= 0) G.add_et(prev, pt, ObjPositionID, 1.0 / sigma2_obj);      // (the reference would dereference a null vertex here)
        }
        mkD[i][j] = pt;
      }
    }
    PreFrame = cam;
  }
  std::vector<double> pose, point;
  G.optimize((double)deltaHuber, 300, 1e-4, pose, point, &last_batch_stats);                        // optimize(300), gain 1e-4
  // write back (:2094-2172)
  if ((int)pMap->vmCameraPose_RF.size() < N) pMap->vmCameraPose_RF = pMap->vmCameraPose;
  if ((int)pMap->vmRigidMotion_RF.size() < N - 1) pMap->vmRigidMotion_RF = pMap->vmRigidMotion;
  for (int i = 0; i < N - 1; ++i)
    for (int j = 0; j < (int)VertexID[i].size(); ++j) {
      if (VertexID[i][j] < 0) continue;
      const cv::Mat T = iso12_to_cv(&pose[12 * (size_t)VertexID[i][j]]);
      if (j == 0) pMap->vmCameraPose_RF[i + 1] = T; else pMap->vmRigidMotion_RF[i][j] = T;
    }
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < (int)mkS[i].size(); ++j) if (mkS[i][j] != -1) pMap->vp3DPointSta[i][j] = point_to_cv(&point[3 * (size_t)mkS[i][j]]);
    for (int j = 0; j < (int)mkD[i].size(); ++j) if (mkD[i][j] != -1) pMap->vp3DPointDyn[i][j] = point_to_cv(&point[3 * (size_t)mkD[i][j]]);
  }
}

void Optimizer::PartialBatchOptimization(Map* pMap, const cv::Mat Calib_K, const int WINDOW_SIZE) {
  const int N = (int)pMap->vpFeatSta.size();
  const auto& StaTracks = pMap->TrackletSta;
  std::vector<std::vector<int> > labS(N), posS(N), mkS(N);
  for (int i = 0; i < N; ++i) { labS[i].assign(pMap->vpFeatSta[i].size(), -1); posS[i] = labS[i]; mkS[i] = labS[i]; }
  label_tracks(StaTracks, labS, posS);
  const float sigma2_cam = 0.0001f, sigma2_3d_sta = 16;          // :190-195 (STATIC_ONLY = true, :211)
  const float deltaHuber = 0.0001f;
  const int Start = N - WINDOW_SIZE;
  std::vector<int> camID(N, -1);
  GraphBuilder G;
  int PreFrame = -1;
  for (int i = Start; i < N; ++i) {
    const int cam = G.add_pose(pMap->vmCameraPose[i]);
    if (i == Start && N == WINDOW_SIZE) G.add_prior(cam, pMap->vmCameraPose[i], 1.0 / 0.0000001);  // :227-236
    camID[i] = cam;
    if (i != Start) G.add_ep(PreFrame, cam, pMap->vmRigidMotion[i - 1][0], 1.0 / sigma2_cam);
    for (int j = 0; j < (int)labS[i].size(); ++j) {
      if (labS[i][j] == -1) continue;
      const int tr = labS[i][j], ps = posS[i][j];
      int pt;
      if (ps == 0) pt = G.add_point(pMap->vp3DPointSta[i][j]);
      else {
        const int pf = StaTracks[tr][ps - 1].first;
        pt = pf >= Start ? mkS[pf][StaTracks[tr][ps - 1].second] : -1;   // tracks that started before the window are skipped (:341-344)
      }
      if (pt < 0) continue;
      G.add_eb(cam, pt, Get3DinCamera(pMap->vpFeatSta[i][j], pMap->vfDepSta[i][j], Calib_K), 1.0 / sigma2_3d_sta);
      mkS[i][j] = pt;
    }
    PreFrame = cam;
  }
  std::vector<double> pose, point;
  G.optimize((double)deltaHuber, 100, 1e-3, pose, point, &last_batch_stats);                        // optimize(100), gain 1e-3 (:182,:807)
  for (int i = Start; i < N; ++i) {                                                                 // :1055-1068
    pMap->vmCameraPose[i] = iso12_to_cv(&pose[12 * (size_t)camID[i]]);
    if (i > Start) pMap->vmRigidMotion[i - 1][0] = Converter::toInvMatrix(pMap->vmCameraPose[i - 1]) * pMap->vmCameraPose[i];
  }
  for (int i = Start; i < N; ++i)
    for (int j = 0; j < (int)mkS[i].size(); ++j) if (mkS[i][j] != -1) pMap->vp3DPointSta[i][j] = point_to_cv(&point[3 * (size_t)mkS[i][j]]);
}

// :2974-3013 (fp32 arithmetic)
cv::Mat Optimizer::Get3DinCamera(const cv::KeyPoint& Feats2d, const float& Dpts, const cv::Mat& Calib_K) {
  const float invfx = 1.0f / Calib_K.at<float>(0, 0), invfy = 1.0f / Calib_K.at<float>(1, 1);
  const float cx = Calib_K.at<float>(0, 2), cy = Calib_K.at<float>(1, 2);
  const float z = Dpts, x = (Feats2d.pt.x - cx) * z * invfx, y = (Feats2d.pt.y - cy) * z * invfy;
  cv::Mat m(3, 1, cv::CV_32F);
  m.at<float>(0) = x; m.at<float>(1) = y; m.at<float>(2) = z;
  return m;
}
cv::Mat Optimizer::Get3DinWorld(const cv::KeyPoint& Feats2d, const float& Dpts, const cv::Mat& Calib_K, const cv::Mat& CameraPose) {
  const cv::Mat x3D = Get3DinCamera(Feats2d, Dpts, Calib_K);
  cv::Mat o(3, 1, cv::CV_32F);
  for (int i = 0; i < 3; ++i) {
    float s = 0;
    for (int k = 0; k < 3; ++k) s += CameraPose.at<float>(i, k) * x3D.at<float>(k);
    o.at<float>(i) = s + CameraPose.at<float>(i, 3);
  }
  return o;
}

}  // namespace VDO_SLAM
