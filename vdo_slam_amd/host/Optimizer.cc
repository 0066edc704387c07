// Optimizer host class: same static entry points as the reference (include/Optimizer.h:25-32).
// The graph CONSTRUCTION follows the reference builders (src/Optimizer.cc:1259-1766 full batch,
// :44-637 partial batch, :2356-2442 / :2778-2865 per-frame) but fills the SoA structures of the
// C-ABI instead of allocating g2o vertices/edges; optimisation and gating run in libvdo_hip.
#include "Optimizer.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <iostream>

#include <cstring>

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

// 4x4 float (row-major) -> VertexSE3 estimate / EdgeSE3 measurement (12 doubles) via toSE3Quat
void to_iso12(const float* T, double out[12]) {
  double R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = T[4 * i + j];
  quat_roundtrip(R, true, out);
  out[9] = T[3]; out[10] = T[7]; out[11] = T[11];
}
// VertexSE3::getEstimateData (toVectorQT) -> q.matrix() -> Converter::toCvSE3 (write-back, :2094-2122)
void iso12_to_f16(const double p[12], float* T) {
  double Ro[9];
  quat_roundtrip(p, false, Ro);
  const cv::Mat M = Converter::toCvSE3(Ro, p + 9);
  std::memcpy(T, M.data, 64);
}
const float kIdent16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

struct GraphBuilder {
  std::vector<double> pose, point, eb_z[3], eb_w, et_z[3], et_w, ep_z, ep_info, pr_z, pr_info;
  std::vector<int32_t> eb_pose, eb_point, et_p1, et_p2, et_pose, ep_i, ep_j, pr_pose;
  int add_pose(const float* T) { double p[12]; to_iso12(T, p); pose.insert(pose.end(), p, p + 12); return (int)(pose.size() / 12) - 1; }
  int add_point(const float* Xw) { for (int k = 0; k < 3; ++k) point.push_back((double)Xw[k]); return (int)(point.size() / 3) - 1; }
  // measurement = Optimizer::Get3DinCamera(feature, depth, K) (:2997-3013, fp32 arithmetic)
  void add_eb(int cam, int pt, float u, float v, float z, const float* K4, double w) {
    eb_pose.push_back(cam); eb_point.push_back(pt);
    const float invfx = 1.0f / K4[0], invfy = 1.0f / K4[1];
    const float x = (u - K4[2]) * z * invfx, y = (v - K4[3]) * z * invfy;
    eb_z[0].push_back((double)x); eb_z[1].push_back((double)y); eb_z[2].push_back((double)z);
    eb_w.push_back(w);
  }
  void add_et(int p1, int p2, int h, double w) {
    et_p1.push_back(p1); et_p2.push_back(p2); et_pose.push_back(h);
    for (int k = 0; k < 3; ++k) et_z[k].push_back(0.0);
    et_w.push_back(w);
  }
  static void push_info(std::vector<double>& v, double s) { for (int i = 0; i < 36; ++i) v.push_back(i % 7 == 0 ? s : 0.0); }
  void add_ep(int i, int j, const float* Z, double s) {
    ep_i.push_back(i); ep_j.push_back(j);
    double z[12]; to_iso12(Z, z); ep_z.insert(ep_z.end(), z, z + 12);
    push_info(ep_info, s);
  }
  void add_prior(int v, const float* Z, double s) {
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
    static const bool trace = std::getenv("VDO_BATCH_TRACE") != nullptr;      // (debug: where a batch optimisation spends its time)
    const auto t0 = std::chrono::steady_clock::now();
    if (vdo_ba_create(HostContext(), &g, &ba) != VDO_OK) die("vdo_ba_create");
    const auto t1 = std::chrono::steady_clock::now();
    vdo_lm_options o{max_it, gain, std::getenv("VDO_VERBOSE") ? std::max(1, std::atoi(std::getenv("VDO_VERBOSE"))) : 0, 0, 0.0, 0};      // (VDO_VERBOSE=2: every trial with its PCG iterations)
    if (vdo_ba_optimize(ba, &o, st) != VDO_OK) die("vdo_ba_optimize");
    const auto t2 = std::chrono::steady_clock::now();
    pose_out.resize(pose.size()); point_out.resize(point.size());
    if (vdo_ba_get_estimates(ba, pose_out.data(), point_out.data()) != VDO_OK) die("vdo_ba_get_estimates");
    const auto t3 = std::chrono::steady_clock::now();
    vdo_ba_destroy(ba);
    if (trace) {
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      std::fprintf(stderr, "[batch] P %d L %d Eb %d Et %d: create %.2f ms, optimize %.2f ms (%d its, %d trials), estimates %.2f ms, destroy %.2f ms\n", g.n_pose, g.n_point, Eb, Et,
                   ms(t0, t1), ms(t1, t2), st ? st->iterations : -1, st ? st->total_trials : -1, ms(t2, t3), ms(t3, std::chrono::steady_clock::now()));
    }
  }
};

// tracklet and position inside it of every feature (global feature index = off[frame] + feature); tracks shorter than 3 are
// not used (:1275-1296).  One pass over the pairs - the reference searches every observation linearly (:1456-1463).
void label_tracks(const TrackList& T, const FeatureBlock& F, std::vector<int32_t>& lab, std::vector<int32_t>& pos) {
  lab.assign((size_t)F.off.back(), -1); pos.assign((size_t)F.off.back(), -1);
  for (int t = 0; t < T.size(); ++t) {
    if (T.off[t + 1] - T.off[t] < 3) continue;
    for (int q = T.off[t]; q < T.off[t + 1]; ++q) {
      const int64_t g = F.off[T.frame[q]] + T.feat[q];
      lab[(size_t)g] = t; pos[(size_t)g] = q - T.off[t];
    }
  }
}

// The same for a WINDOW of frames [first, ...): only their features are labelled, at index g - F.off[first]; a track whose last observation (a track's frames ascend)
// lies before the window is skipped whole.  What PartialBatchOptimization pays per window then follows the window, not the length of the sequence so far
// (label_tracks + the marker array over ALL features: 0.2 ms at frame 20, 0.6 ms at frame 148 of the bench sequence, ~3 ms at KITTI-0020's 837).
void label_tracks_window(const TrackList& T, const FeatureBlock& F, int first, std::vector<int32_t>& lab, std::vector<int32_t>& pos) {
  const int64_t base = F.off[first];
  lab.assign((size_t)(F.off.back() - base), -1); pos.assign((size_t)(F.off.back() - base), -1);
  for (int t = 0; t < T.size(); ++t) {
    if (T.off[t + 1] - T.off[t] < 3) continue;
    if (T.frame[T.off[t + 1] - 1] < first) continue;
    for (int q = T.off[t]; q < T.off[t + 1]; ++q) {
      if (T.frame[q] < first) continue;
      const int64_t g = F.off[T.frame[q]] + T.feat[q] - base;
      lab[(size_t)g] = t; pos[(size_t)g] = q - T.off[t];
    }
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

// ---- the non-joint variants (bJoint == false): unary reprojection edges, g2o replaced by vdo_pose_optimize
// Optimizer::PoseOptimizationNew (src/Optimizer.cc:2177-2331): camera pose from the current static keys vs the last frame's
// back-projected points (EdgeSE3ProjectXYZOnlyPose, information I2, Huber sqrt(0.01), 100 iterations, chi2 gate 0.01)
int Optimizer::PoseOptimizationNew(Frame* pCurFrame, Frame* pLastFrame, vector<int>& TemperalMatch) {
  const int N = (int)TemperalMatch.size();
  std::vector<double> obs(2 * (size_t)N), Xw(3 * (size_t)N);
  for (int i = 0; i < N; ++i) {
    const cv::KeyPoint& kpUn = pCurFrame->mvStatKeys[TemperalMatch[i]];
    obs[2 * i] = kpUn.pt.x; obs[2 * i + 1] = kpUn.pt.y;
    const cv::Mat X = pLastFrame->UnprojectStereoStat(TemperalMatch[i], 1);
    for (int k = 0; k < 3; ++k) Xw[3 * i + k] = X.empty() ? 0.0 : (double)X.at<float>(k);
  }
  if (N < 3) return 0;                                                // nInitialCorrespondences<3 (:2258-2259)
  const float rp_thres = 0.01f;
  vdo_pose_problem p{};
  p.n = N; p.kind = 0; p.obs = obs.data(); p.Xw = Xw.data();
  p.K[0] = pCurFrame->fx; p.K[1] = pCurFrame->fy; p.K[2] = pCurFrame->cx; p.K[3] = pCurFrame->cy;
  Converter::toDouble16(pCurFrame->mTcw, p.T0);
  p.huber_delta = (double)std::sqrt(rp_thres); p.chi2_gate = (double)rp_thres; p.max_iterations = 100;
  vdo_flow2_result r;
  std::vector<uint8_t> inl(N);
  if (vdo_pose_optimize(HostContext(), &p, &r, inl.data()) != VDO_OK) die("vdo_pose_optimize");
  int nBad = 0;
  for (int i = 0; i < N; ++i) if (!inl[i]) { TemperalMatch[i] = -1; ++nBad; }                       // :2287-2293
  pCurFrame->SetPose(Converter::toCvMat(r.T));
  std::cout << "(Camera) inliers number/total numbers: " << N - nBad << "/" << N << std::endl;
  return N - nBad;
}

// Optimizer::PoseOptimizationObjMot (:2544-2753): world-frame object motion H from the current object keys vs the last frame's
// back-projected object points through P = K * Tcw (EdgeSE3ProjectXYZOnlyObjMotion, no robust kernel, 200 iterations)
cv::Mat Optimizer::PoseOptimizationObjMot(Frame* pCurFrame, Frame* pLastFrame, const vector<int>& ObjId, std::vector<int>& InlierID) {
  const int N = (int)ObjId.size();
  if (N < 3) return cv::Mat::eye(4, 4, cv::CV_32F);                    // :2665-2666
  std::vector<double> obs(2 * (size_t)N), Xw(3 * (size_t)N);
  for (int i = 0; i < N; ++i) {
    const cv::KeyPoint& kpUn = pCurFrame->mvObjKeys[ObjId[i]];
    obs[2 * i] = kpUn.pt.x; obs[2 * i + 1] = kpUn.pt.y;
    const cv::Mat X = pLastFrame->UnprojectStereoObject(ObjId[i], 0);
    for (int k = 0; k < 3; ++k) Xw[3 * i + k] = X.empty() ? 0.0 : (double)X.at<float>(k);
  }
  const float rp_thres = 0.01f;
  vdo_pose_problem p{};
  p.n = N; p.kind = 1; p.obs = obs.data(); p.Xw = Xw.data();
  // PP = KK * toMatrix4d(mTcw) in double (:2604-2606)
  double T[16];
  Converter::toDouble16(pCurFrame->mTcw, T);
  const double KK[12] = {pCurFrame->fx, 0, pCurFrame->cx, 0, 0, pCurFrame->fy, pCurFrame->cy, 0, 0, 0, 1, 0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) { double a = 0; for (int k = 0; k < 4; ++k) a += KK[4 * i + k] * T[4 * k + j]; p.P[4 * i + j] = a; }
  const cv::Mat Init = Converter::toInvMatrix(pCurFrame->mTcw) * pCurFrame->mInitModel;             // :2590
  Converter::toDouble16(Init, p.T0);
  p.huber_delta = 0.0; p.chi2_gate = (double)rp_thres; p.max_iterations = 200;
  vdo_flow2_result r;
  std::vector<uint8_t> inl(N);
  if (vdo_pose_optimize(HostContext(), &p, &r, inl.data()) != VDO_OK) die("vdo_pose_optimize");
  InlierID.clear();
  int nBad = 0;
  if (pCurFrame->vObjLabel.size() < pCurFrame->mvObjKeys.size()) pCurFrame->vObjLabel.resize(pCurFrame->mvObjKeys.size(), 0);
  for (int i = 0; i < N; ++i) {                                        // :2735-2742
    if (inl[i]) InlierID.push_back(ObjId[i]);
    else { pCurFrame->vObjLabel[ObjId[i]] = -1; ++nBad; }
  }
  std::cout << "(OBJ)inliers number/total numbers: " << N - nBad << "/" << N << std::endl;
  return Converter::toCvMat(r.T);
}

// ------------------------------------------------------------------------------- batch
// The graph builders of the reference (src/Optimizer.cc:1259-1766 full, :44-637 partial) over the flat GraphStore: one pass
// over the frames, every observation finds its landmark through the per-feature (tracklet, position) labels, vertices and
// edges go straight into the SoA arrays vdo_ba_create takes.
void Optimizer::FullBatchOptimization(GraphStore& S, const TrackList& StaTracks, const TrackList& DynTracks, const float K4[4]) {
  const int N = S.frames();
  if (N <= 0) return;
  std::vector<int32_t> labS, posS, labD, posD;
  label_tracks(StaTracks, S.sta, labS, posS);
  label_tracks(DynTracks, S.dyn, labD, posD);
  std::vector<int32_t> mkS(labS.size(), -1), mkD(labD.size(), -1);
  std::vector<int32_t> VertexID((size_t)S.rm_off.back(), -1);            // vertex of every (transition, motion)

  // information (float sigma^2 as in :1330-1335) and Huber delta (:1352)
  const float sigma2_cam = 0.001f, sigma2_3d_sta = 80, sigma2_obj_smo = 0.001f, sigma2_obj = 100, sigma2_3d_dyn = 80;
  const float deltaHuber = 0.0001f;
  GraphBuilder G;
  int PreFrame = -1;
  for (int i = 0; i < N; ++i) {
    const int cam = G.add_pose(&S.cam[16 * (size_t)i]);
    if (i == 0) G.add_prior(cam, &S.cam[0], 100000.0);                                              // :1364-1373
    else if (i - 1 < S.transitions() && S.n_motions(i - 1) > 0) {
      VertexID[(size_t)S.rm_off[i - 1]] = cam;
      G.add_ep(PreFrame, cam, &S.rm[16 * (size_t)S.rm_off[i - 1]], 1.0 / sigma2_cam);              // :1383-1399
    }
    // static features (:1404-1524)
    for (int64_t g = S.sta.off[i]; g < S.sta.off[i + 1]; ++g) {
      if (labS[(size_t)g] == -1) continue;
      const int tr = labS[(size_t)g], ps = posS[(size_t)g];
      int pt;
      if (ps == 0) pt = G.add_point(&S.sta.xyz[3 * (size_t)g]);
      else { const int q = StaTracks.off[tr] + ps - 1; pt = mkS[(size_t)(S.sta.off[StaTracks.frame[q]] + StaTracks.feat[q])]; }
      if (pt < 0) continue;
      G.add_eb(cam, pt, S.sta.u[(size_t)g], S.sta.v[(size_t)g], S.sta.d[(size_t)g], K4, 1.0 / sigma2_3d_sta);
      mkS[(size_t)g] = pt;
    }
    // dynamic features + object motions (:1530-1747)
    if (i == 0) {
      for (int64_t g = S.dyn.off[0]; g < S.dyn.off[1]; ++g) {
        if (labD[(size_t)g] == -1) continue;
        const int pt = G.add_point(&S.dyn.xyz[3 * (size_t)g]);
        G.add_eb(cam, pt, S.dyn.u[(size_t)g], S.dyn.v[(size_t)g], S.dyn.d[(size_t)g], K4, 1.0 / sigma2_3d_dyn);
        mkD[(size_t)g] = pt;
      }
    } else if (i - 1 < S.transitions()) {
      const int64_t r0 = S.rm_off[i - 1];
      const int nmot = S.n_motions(i - 1);
      for (int j = 1; j < nmot; ++j) {
        const int mv = G.add_pose(kIdent16);                                                        // motions start at identity (:1581)
        if (i > 2) {                                                                                // smoothness (:1593-1622)
          const int64_t p0 = S.rm_off[i - 2];
          int TraceID = -1;
          for (int k = 0; k < S.n_motions(i - 2); ++k)
            if (S.rm_label[(size_t)(p0 + k)] == S.rm_label[(size_t)(r0 + j)]) { TraceID = k; break; }
          if (TraceID != -1 && VertexID[(size_t)(p0 + TraceID)] >= 0) G.add_ep(VertexID[(size_t)(p0 + TraceID)], mv, kIdent16, 1.0 / sigma2_obj_smo);
        }
        VertexID[(size_t)(r0 + j)] = mv;
      }
      for (int64_t g = S.dyn.off[i]; g < S.dyn.off[i + 1]; ++g) {
        if (labD[(size_t)g] == -1) continue;
        const int tr = labD[(size_t)g], ps = posD[(size_t)g];
        int ObjPositionID = -1;
        for (int k = 1; k < nmot; ++k)
          if (S.rm_label[(size_t)(r0 + k)] == DynTracks.obj[tr]) { ObjPositionID = VertexID[(size_t)(r0 + k)]; break; }
        if (ObjPositionID == -1 && ps != 0) continue;
        const int pt = G.add_point(&S.dyn.xyz[3 * (size_t)g]);
        G.add_eb(cam, pt, S.dyn.u[(size_t)g], S.dyn.v[(size_t)g], S.dyn.d[(size_t)g], K4, 1.0 / sigma2_3d_dyn);
        if (ps != 0) {
          const int q = DynTracks.off[tr] + ps - 1;
          const int prev = mkD[(size_t)(S.dyn.off[DynTracks.frame[q]] + DynTracks.feat[q])];
          if (prev >= 0) G.add_et(prev, pt, ObjPositionID, 1.0 / sigma2_obj);      // (the reference would dereference a null vertex here)
        }
        mkD[(size_t)g] = pt;
      }
    }
    PreFrame = cam;
  }
  std::vector<double> pose, point;
  G.optimize((double)deltaHuber, 300, 1e-4, pose, point, &last_batch_stats);                        // optimize(300), gain 1e-4
  // write back (:2094-2172): refined camera poses / motions into the *_RF copies, refined points in place
  if (S.cam_rf.size() != S.cam.size()) S.cam_rf = S.cam;
  if (S.rm_rf.size() != S.rm.size()) S.rm_rf = S.rm;
  for (int i = 0; i + 1 < N && i < S.transitions(); ++i)
    for (int j = 0; j < S.n_motions(i); ++j) {
      const int v = VertexID[(size_t)(S.rm_off[i] + j)];
      if (v < 0) continue;
      if (j == 0) iso12_to_f16(&pose[12 * (size_t)v], &S.cam_rf[16 * (size_t)(i + 1)]);
      else iso12_to_f16(&pose[12 * (size_t)v], &S.rm_rf[16 * (size_t)(S.rm_off[i] + j)]);
    }
  for (size_t g = 0; g < mkS.size(); ++g) if (mkS[g] != -1) for (int k = 0; k < 3; ++k) S.sta.xyz[3 * g + k] = (float)point[3 * (size_t)mkS[g] + k];
  for (size_t g = 0; g < mkD.size(); ++g) if (mkD[g] != -1) for (int k = 0; k < 3; ++k) S.dyn.xyz[3 * g + k] = (float)point[3 * (size_t)mkD[g] + k];
}

void Optimizer::PartialBatchOptimization(GraphStore& S, const TrackList& StaTracks, const float K4[4], const int WINDOW_SIZE) {
  const int N = S.frames();
  if (N < WINDOW_SIZE || WINDOW_SIZE <= 0) return;
  const auto t_begin = std::chrono::steady_clock::now();
  const int Start = N - WINDOW_SIZE;
  const int64_t base = S.sta.off[Start];                         // labels / markers of the window's features only (index g - base)
  std::vector<int32_t> labS, posS;
  label_tracks_window(StaTracks, S.sta, Start, labS, posS);
  std::vector<int32_t> mkS(labS.size(), -1);
  const float sigma2_cam = 0.0001f, sigma2_3d_sta = 16;          // :190-195 (STATIC_ONLY = true, :211)
  const float deltaHuber = 0.0001f;
  std::vector<int> camID(N, -1);
  GraphBuilder G;
  int PreFrame = -1;
  for (int i = Start; i < N; ++i) {
    const int cam = G.add_pose(&S.cam[16 * (size_t)i]);
    if (i == Start && N == WINDOW_SIZE) G.add_prior(cam, &S.cam[16 * (size_t)i], 1.0 / 0.0000001);  // :227-236
    camID[i] = cam;
    if (i != Start) G.add_ep(PreFrame, cam, &S.rm[16 * (size_t)S.rm_off[i - 1]], 1.0 / sigma2_cam);
    for (int64_t g = S.sta.off[i]; g < S.sta.off[i + 1]; ++g) {
      if (labS[(size_t)(g - base)] == -1) continue;
      const int tr = labS[(size_t)(g - base)], ps = posS[(size_t)(g - base)];
      int pt;
      if (ps == 0) pt = G.add_point(&S.sta.xyz[3 * (size_t)g]);
      else {
        const int q = StaTracks.off[tr] + ps - 1;
        const int pf = StaTracks.frame[q];
        pt = pf >= Start ? mkS[(size_t)(S.sta.off[pf] + StaTracks.feat[q] - base)] : -1;   // tracks that started before the window are skipped (:341-344)
      }
      if (pt < 0) continue;
      G.add_eb(cam, pt, S.sta.u[(size_t)g], S.sta.v[(size_t)g], S.sta.d[(size_t)g], K4, 1.0 / sigma2_3d_sta);
      mkS[(size_t)(g - base)] = pt;
    }
    PreFrame = cam;
  }
  std::vector<double> pose, point;
  const auto t_built = std::chrono::steady_clock::now();
  G.optimize((double)deltaHuber, 100, 1e-3, pose, point, &last_batch_stats);                        // optimize(100), gain 1e-3 (:182,:807)
  if (std::getenv("VDO_BATCH_TRACE"))
    std::fprintf(stderr, "[partial batch] frames %d: graph built in %.2f ms, optimised in %.2f ms\n", N, std::chrono::duration<double, std::milli>(t_built - t_begin).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_built).count());
  for (int i = Start; i < N; ++i) {                                                                 // :1055-1068
    iso12_to_f16(&pose[12 * (size_t)camID[i]], &S.cam[16 * (size_t)i]);
    if (i > Start) {                                                                                // vmRigidMotion[i-1][0] = toInvMatrix(pose[i-1]) * pose[i]  (fp32)
      const cv::Mat A(4, 4, cv::CV_32F, &S.cam[16 * (size_t)(i - 1)]), B(4, 4, cv::CV_32F, &S.cam[16 * (size_t)i]);
      const cv::Mat M = Converter::toInvMatrix(A) * B;
      std::memcpy(&S.rm[16 * (size_t)S.rm_off[i - 1]], M.data, 64);
    }
  }
  for (size_t g = 0; g < mkS.size(); ++g) if (mkS[g] != -1) for (int k = 0; k < 3; ++k) S.sta.xyz[3 * (g + (size_t)base) + k] = (float)point[3 * (size_t)mkS[g] + k];
}

// ---- the reference's signatures: the Map is read into a store, optimised, and receives the results (same builder)
void StoreFromMap(const Map& m, GraphStore& S, TrackList& sta, TrackList& dyn) {
  S.clear();
  const int N = (int)m.vpFeatSta.size();
  auto fill = [](FeatureBlock& F, const std::vector<cv::KeyPoint>& kp, const std::vector<float>& dep, const std::vector<cv::Mat>& pts) {
    const size_t n = kp.size();
    for (size_t j = 0; j < n; ++j) {
      F.u.push_back(kp[j].pt.x); F.v.push_back(kp[j].pt.y); F.d.push_back(dep[j]);
      for (int k = 0; k < 3; ++k) F.xyz.push_back(pts[j].at<float>(k));
    }
    F.off.push_back(F.off.back() + (int64_t)n);
  };
  for (int i = 0; i < N; ++i) {
    fill(S.sta, m.vpFeatSta[i], m.vfDepSta[i], m.vp3DPointSta[i]);
    fill(S.dyn, m.vpFeatDyn[i], m.vfDepDyn[i], m.vp3DPointDyn[i]);
    S.cam.insert(S.cam.end(), (const float*)m.vmCameraPose[i].data, (const float*)m.vmCameraPose[i].data + 16);
    const cv::Mat& rf = i < (int)m.vmCameraPose_RF.size() ? m.vmCameraPose_RF[i] : m.vmCameraPose[i];
    S.cam_rf.insert(S.cam_rf.end(), (const float*)rf.data, (const float*)rf.data + 16);
    if (i < (int)m.vmRigidMotion.size()) {
      for (size_t j = 0; j < m.vmRigidMotion[i].size(); ++j) {
        S.rm.insert(S.rm.end(), (const float*)m.vmRigidMotion[i][j].data, (const float*)m.vmRigidMotion[i][j].data + 16);
        const cv::Mat& r2 = (i < (int)m.vmRigidMotion_RF.size() && j < m.vmRigidMotion_RF[i].size()) ? m.vmRigidMotion_RF[i][j] : m.vmRigidMotion[i][j];
        S.rm_rf.insert(S.rm_rf.end(), (const float*)r2.data, (const float*)r2.data + 16);
        S.rm_label.push_back(m.vnRMLabel[i][j]);
      }
      S.rm_off.push_back(S.rm_off.back() + (int64_t)m.vmRigidMotion[i].size());
    }
  }
  auto tracks = [](const std::vector<std::vector<std::pair<int, int> > >& T, TrackList& L) {
    L.off.assign(1, 0); L.frame.clear(); L.feat.clear(); L.obj.clear();
    for (const auto& t : T) {
      for (const auto& pr : t) { L.frame.push_back(pr.first); L.feat.push_back(pr.second); }
      L.off.push_back((int32_t)L.frame.size());
    }
  };
  tracks(m.TrackletSta, sta); tracks(m.TrackletDyn, dyn);
  dyn.obj.assign(m.nObjID.begin(), m.nObjID.end());
}

void StoreToMap(const GraphStore& S, const TrackList& sta, const TrackList& dyn, Map& m) {
  const int N = S.frames();
  auto m44 = [](const float* p) { cv::Mat M(4, 4, cv::CV_32F); std::memcpy(M.data, p, 64); return M; };
  auto feats = [](const FeatureBlock& F, int i, std::vector<cv::KeyPoint>& kp, std::vector<float>& dep, std::vector<cv::Mat>& pts) {
    const int64_t a = F.off[i], n = F.off[i + 1] - a;
    kp.resize((size_t)n); dep.assign(F.d.begin() + a, F.d.begin() + a + n); pts.resize((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
      kp[(size_t)j] = cv::KeyPoint(F.u[(size_t)(a + j)], F.v[(size_t)(a + j)], 0);
      cv::Mat p(3, 1, cv::CV_32F);
      std::memcpy(p.data, &F.xyz[3 * (size_t)(a + j)], 12);
      pts[(size_t)j] = p;
    }
  };
  m.vpFeatSta.resize(N); m.vfDepSta.resize(N); m.vp3DPointSta.resize(N);
  m.vpFeatDyn.resize(N); m.vfDepDyn.resize(N); m.vp3DPointDyn.resize(N);
  m.vmCameraPose.resize(N); m.vmCameraPose_RF.resize(N);
  for (int i = 0; i < N; ++i) {
    feats(S.sta, i, m.vpFeatSta[i], m.vfDepSta[i], m.vp3DPointSta[i]);
    if (i < S.dyn.frames()) feats(S.dyn, i, m.vpFeatDyn[i], m.vfDepDyn[i], m.vp3DPointDyn[i]);
    m.vmCameraPose[i] = m44(&S.cam[16 * (size_t)i]); m.vmCameraPose_RF[i] = m44(&S.cam_rf[16 * (size_t)i]);
  }
  const int Tn = S.transitions();
  m.vmRigidMotion.resize(Tn); m.vmRigidMotion_RF.resize(Tn); m.vnRMLabel.resize(Tn);
  for (int i = 0; i < Tn; ++i) {
    const int n = S.n_motions(i);
    m.vmRigidMotion[i].resize(n); m.vmRigidMotion_RF[i].resize(n); m.vnRMLabel[i].resize(n);
    for (int j = 0; j < n; ++j) {
      m.vmRigidMotion[i][j] = m44(&S.rm[16 * (size_t)(S.rm_off[i] + j)]); m.vmRigidMotion_RF[i][j] = m44(&S.rm_rf[16 * (size_t)(S.rm_off[i] + j)]);
      m.vnRMLabel[i][j] = S.rm_label[(size_t)(S.rm_off[i] + j)];
    }
  }
  auto tracks = [](const TrackList& L, std::vector<std::vector<std::pair<int, int> > >& T) {
    T.assign(L.size(), {});
    for (int t = 0; t < L.size(); ++t)
      for (int q = L.off[t]; q < L.off[t + 1]; ++q) T[t].push_back(std::make_pair((int)L.frame[q], (int)L.feat[q]));
  };
  tracks(sta, m.TrackletSta); tracks(dyn, m.TrackletDyn);
  m.nObjID.assign(dyn.obj.begin(), dyn.obj.end());
}

namespace {
void K4_of(const cv::Mat& K, float K4[4]) { K4[0] = K.at<float>(0, 0); K4[1] = K.at<float>(1, 1); K4[2] = K.at<float>(0, 2); K4[3] = K.at<float>(1, 2); }
}  // namespace

void Optimizer::FullBatchOptimization(Map* pMap, const cv::Mat Calib_K) {
  GraphStore S; TrackList sta, dyn;
  StoreFromMap(*pMap, S, sta, dyn);
  float K4[4]; K4_of(Calib_K, K4);
  FullBatchOptimization(S, sta, dyn, K4);
  StoreToMap(S, sta, dyn, *pMap);
}

void Optimizer::PartialBatchOptimization(Map* pMap, const cv::Mat Calib_K, const int WINDOW_SIZE) {
  GraphStore S; TrackList sta, dyn;
  StoreFromMap(*pMap, S, sta, dyn);
  float K4[4]; K4_of(Calib_K, K4);
  PartialBatchOptimization(S, sta, K4, WINDOW_SIZE);
  StoreToMap(S, sta, dyn, *pMap);
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
