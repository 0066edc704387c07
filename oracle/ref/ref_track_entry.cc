// TEST INFRASTRUCTURE - oracle/_ref/libref_track.so: the reference's OWN per-frame front end - src/System.cc, src/Tracking.cc, src/Frame.cc,
// src/Map.cc and src/ORBextractor.cc, compiled verbatim from where they lie under /root/reference (oracle/ref/Makefile) against the mini-cv shim of
// oracle/ref/shim/ - behind a few C entry points.  This file is everything of that library that is NOT the reference's source:
//   * a bump allocator (the reference's quadtree orders nodes by heap address: see ref_orb_entry.cc);
//   * Converter::toInvMatrix, statement for statement as src/Converter.cc:151-166 has it (the real Converter.cc needs Eigen + g2o);
//   * the eight Optimizer statics (include/Optimizer.h:25-32).  Their bodies in src/Optimizer.cc are g2o code and cannot be compiled here; the glue
//     below collects the same correspondences, hands them to the ORACLE's restatement of that g2o code (oracle/flow_oracle.cpp) and writes the results
//     back the way src/Optimizer.cc:2333-2542 / :2755-2972 do.  The two batch optimisers are no-ops that count their calls (the schedule of
//     src/Tracking.cc:1165-1183 is what this build pins; the batch arithmetic has its own oracle);
//   * flat getters for tests.
// So: Track()'s control flow, GrabImageRGBD, Frame::Frame, GetSceneFlowObj, DynObjTracking, GetInitModelCam/Obj, RenewFrameInfo, UpdateMask,
// GetStaticTrack / GetDynamicTrackNew and "Save Graph Structure" run as the reference wrote them; OpenCV primitives and g2o are the oracle's.
// Not a product file; nothing of the reference is copied into this repository.
#include <cstddef>
#include <cstdlib>
#include <new>
#include <sys/mman.h>
namespace ref_arena {
static char* base = nullptr; static size_t used = 0, sys_begin = 0; static const size_t kSize = (size_t)256 << 30;
// A System starts on never-touched (zero) pages, like a fresh process - the reference reads a few never-initialised members - and address space is never
// reused: what g2o's type registrations and other static initialisers of this library allocated when it was loaded stays where it is (round 4 rewound the
// arena to 0 for every first System and wiped them - harmless while the library held no g2o).  The pages of a System go back when the last one is destroyed.
inline void mark() { sys_begin = used; }
inline void release() {
  const size_t a = (sys_begin + 4095) & ~(size_t)4095, b = used & ~(size_t)4095;
  if (base && b > a) madvise(base + a, b - a, MADV_DONTNEED);
  used = (used + 4095) & ~(size_t)4095;                  // (the next allocation starts on a page of its own: untouched)
}
inline void* take(size_t n) {
  if (!base) { base = (char*)mmap(nullptr, kSize, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0); if (base == (char*)MAP_FAILED) abort(); }
  n = (n + 15) & ~(size_t)15;
  if (used + n > kSize) abort();
  void* p = base + used; used += n; return p;
}
}
// (addresses grow in allocation order and are never reused: "by pointer" = "by creation order")
void* operator new(std::size_t n) { return ref_arena::take(n ? n : 1); }
void* operator new[](std::size_t n) { return ref_arena::take(n ? n : 1); }
void operator delete(void*) noexcept {}
void operator delete[](void*) noexcept {}
void operator delete(void*, std::size_t) noexcept {}
void operator delete[](void*, std::size_t) noexcept {}

// (standard and shim headers first: the access override below is for the reference's own class definitions only)
#include <list>
#include <mutex>
#include <set>
#include <thread>
#include "minicv_ref.hpp"
#include <Eigen/Core>
#include <cvplot/cvplot.h>
#define private public
#define protected public
#include "System.h"
#undef private
#undef protected
#include "Converter.h"
#include "Optimizer.h"

#include <cstdint>
#include <cstring>
#include <ctime>

#include "../vdo_oracle.h"

#ifndef REF_REAL_OPTIMIZER
namespace VDO_SLAM {

// src/Converter.cc:151-166, statement for statement
cv::Mat Converter::toInvMatrix(const cv::Mat& T) {
  cv::Mat T_inv = cv::Mat::eye(4, 4, CV_32F);
  const cv::Mat R = T.rowRange(0, 3).colRange(0, 3);
  const cv::Mat t = T.rowRange(0, 3).col(3);
  cv::Mat t_inv = -R.t() * t;
  cv::Mat R_inv = R.t();
  cv::Mat tmp_R = T_inv.rowRange(0, 3).colRange(0, 3);
  R_inv.copyTo(tmp_R);
  cv::Mat tmp_t = T_inv.rowRange(0, 3).col(3);
  t_inv.copyTo(tmp_t);
  return T_inv;
}
Eigen::Matrix<double, 4, 4> Converter::toMatrix4d(const cv::Mat& m) {
  Eigen::Matrix<double, 4, 4> M;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) M(i, j) = m.at<float>(i, j);
  return M;
}

// src/Optimizer.cc:2974-3013, statement for statement
cv::Mat Optimizer::Get3DinWorld(const cv::KeyPoint& Feats2d, const float& Dpts, const cv::Mat& Calib_K, const cv::Mat& CameraPose) {
  const float invfx = 1.0f / Calib_K.at<float>(0, 0);
  const float invfy = 1.0f / Calib_K.at<float>(1, 1);
  const float cx = Calib_K.at<float>(0, 2);
  const float cy = Calib_K.at<float>(1, 2);
  const float u = Feats2d.pt.x;
  const float v = Feats2d.pt.y;
  const float z = Dpts;
  const float x = (u - cx) * z * invfx;
  const float y = (v - cy) * z * invfy;
  cv::Mat x3D = (cv::Mat_<float>(3, 1) << x, y, z);
  const cv::Mat mRwc = CameraPose.rowRange(0, 3).colRange(0, 3);
  const cv::Mat mtwc = CameraPose.rowRange(0, 3).col(3);
  return mRwc * x3D + mtwc;
}
cv::Mat Optimizer::Get3DinCamera(const cv::KeyPoint& Feats2d, const float& Dpts, const cv::Mat& Calib_K) {
  const float invfx = 1.0f / Calib_K.at<float>(0, 0);
  const float invfy = 1.0f / Calib_K.at<float>(1, 1);
  const float cx = Calib_K.at<float>(0, 2);
  const float cy = Calib_K.at<float>(1, 2);
  const float u = Feats2d.pt.x;
  const float v = Feats2d.pt.y;
  const float z = Dpts;
  const float x = (u - cx) * z * invfx;
  const float y = (v - cy) * z * invfy;
  cv::Mat x3D = (cv::Mat_<float>(3, 1) << x, y, z);
  return x3D;
}

namespace {
int g_full_batch_calls = 0, g_partial_batch_calls = 0;
int g_last_cam_iterations = 0;

// what both joint optimisers hand to g2o (src/Optimizer.cc:2380-2443 / :2800-2866): per correspondence the last frame's key point, its
// measured flow and depth (ObtainFlowDepthCamera / ObtainFlowDepthObject: flow.x, flow.y, depth as floats), K, Twl = inverse of the last pose
// (Rwl = Rlw.t(), twl = -Rlw.t() * tlw through cv::Mat, then toMatrix3d / toVector3d), the initial estimate (toSE3Quat of a CV_32F matrix)
void fill_problem(vdo_flow2_problem& p, int n, std::vector<double>& obs, std::vector<double>& flow, std::vector<double>& depth, Frame* pCurFrame, Frame* pLastFrame,
                  const cv::Mat& Init, double info_prior, int max_it) {
  std::memset(&p, 0, sizeof p);
  p.n = n; p.obs = obs.data(); p.flow = flow.data(); p.depth = depth.data();
  p.K[0] = pCurFrame->fx; p.K[1] = pCurFrame->fy; p.K[2] = pCurFrame->cx; p.K[3] = pCurFrame->cy;
  const cv::Mat Rlw = pLastFrame->mTcw.rowRange(0, 3).colRange(0, 3);
  const cv::Mat Rwl = Rlw.t();
  const cv::Mat tlw = pLastFrame->mTcw.rowRange(0, 3).col(3);
  const cv::Mat twl = -Rlw.t() * tlw;
  for (int i = 0; i < 16; ++i) p.Twl[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) p.Twl[4 * i + j] = Rwl.at<float>(i, j); p.Twl[4 * i + 3] = twl.at<float>(i); }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) p.T0[4 * i + j] = Init.at<float>(i, j);
  p.info_flow = 0.1; p.info_prior = info_prior;
  const float rp_thres = 0.04f;
  p.huber_delta = (double)(float)sqrt(rp_thres);          // const float deltaMono = sqrt(rp_thres)
  p.chi2_gate = (double)rp_thres;                         // chi2Mono[0]
  p.max_iterations = max_it; p.ref_quirks = 1;
}
cv::Mat to_cv(const double T[16]) {                        // Converter::toCvMat(SE3Quat): double -> float per entry
  cv::Mat m(4, 4, CV_32F);
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.at<float>(i, j) = (float)T[4 * i + j];
  return m;
}
}  // namespace

// src/Optimizer.cc:2333-2542
int Optimizer::PoseOptimizationFlow2Cam(Frame* pCurFrame, Frame* pLastFrame, vector<int>& TemperalMatch) {
  const int N = (int)TemperalMatch.size();
  cv::Mat Init = pCurFrame->mTcw;
  std::vector<double> obs(2 * (size_t)std::max(N, 1)), flow(2 * (size_t)std::max(N, 1)), depth((size_t)std::max(N, 1));
  for (int i = 0; i < N; ++i) {
    const cv::Mat FloD = pLastFrame->ObtainFlowDepthCamera(TemperalMatch[i], 0);
    const cv::KeyPoint& kpUn = pLastFrame->mvStatKeys[TemperalMatch[i]];
    obs[2 * i] = kpUn.pt.x; obs[2 * i + 1] = kpUn.pt.y;
    flow[2 * i] = FloD.at<float>(0); flow[2 * i + 1] = FloD.at<float>(1); depth[i] = FloD.at<float>(2);
  }
  if (N < 3) return 0;
  vdo_flow2_problem p;
  fill_problem(p, N, obs, flow, depth, pCurFrame, pLastFrame, Init, 0.3, 100);
  double T[16];
  std::vector<double> fnew(2 * (size_t)N);
  std::vector<uint8_t> inl((size_t)N);
  vdo_lm_stats st;
  if (std::getenv("VDO_REF_DUMP")) {
    static int call = 0; char name[256]; std::snprintf(name, sizeof name, "%s/cam_%d.bin", std::getenv("VDO_REF_DUMP"), call++);
    FILE* f = std::fopen(name, "wb");
    if (f) { std::fwrite(&N, 4, 1, f); std::fwrite(obs.data(), 8, 2 * N, f); std::fwrite(flow.data(), 8, 2 * N, f); std::fwrite(depth.data(), 8, N, f); std::fwrite(p.Twl, 8, 16, f); std::fwrite(p.T0, 8, 16, f); std::fclose(f); }
  }
  const int n_in = vdo_oracle_flow2_optimize(&p, T, fnew.data(), inl.data(), &st);
  g_last_cam_iterations = st.iterations;
  pCurFrame->SetPose(to_cv(T));
  const std::vector<int> match = TemperalMatch;           // (the reference reads TemperalMatch[i] for the inliers only: those entries are untouched)
  for (int i = 0; i < N; ++i) {
    if (!inl[i]) { TemperalMatch[i] = -1; continue; }
    pCurFrame->mvStatKeys[match[i]].pt.x = pLastFrame->mvStatKeys[match[i]].pt.x + fnew[2 * i];
    pCurFrame->mvStatKeys[match[i]].pt.y = pLastFrame->mvStatKeys[match[i]].pt.y + fnew[2 * i + 1];
  }
  return n_in;
}

// src/Optimizer.cc:2755-2972
cv::Mat Optimizer::PoseOptimizationFlow2(Frame* pCurFrame, Frame* pLastFrame, const vector<int>& ObjId, std::vector<int>& InlierID) {
  const int N = (int)ObjId.size();
  cv::Mat Init = pCurFrame->mInitModel;
  std::vector<double> obs(2 * (size_t)std::max(N, 1)), flow(2 * (size_t)std::max(N, 1)), depth((size_t)std::max(N, 1));
  for (int i = 0; i < N; ++i) {
    const cv::Mat FloD = pLastFrame->ObtainFlowDepthObject(ObjId[i], 0);
    const cv::KeyPoint& kpUn = pLastFrame->mvObjKeys[ObjId[i]];
    obs[2 * i] = kpUn.pt.x; obs[2 * i + 1] = kpUn.pt.y;
    flow[2 * i] = FloD.at<float>(0); flow[2 * i + 1] = FloD.at<float>(1); depth[i] = FloD.at<float>(2);
  }
  if (N < 3) return cv::Mat::eye(4, 4, CV_32F);
  vdo_flow2_problem p;
  fill_problem(p, N, obs, flow, depth, pCurFrame, pLastFrame, Init, 0.5, 200);
  double T[16];
  std::vector<double> fnew(2 * (size_t)N);
  std::vector<uint8_t> inl((size_t)N);
  vdo_lm_stats st;
  vdo_oracle_flow2_optimize(&p, T, fnew.data(), inl.data(), &st);
  std::vector<int> output_inlier;
  for (int i = 0; i < N; ++i) {
    if (inl[i]) {
      pCurFrame->mvObjKeys[ObjId[i]].pt.x = pLastFrame->mvObjKeys[ObjId[i]].pt.x + fnew[2 * i];
      pCurFrame->mvObjKeys[ObjId[i]].pt.y = pLastFrame->mvObjKeys[ObjId[i]].pt.y + fnew[2 * i + 1];
      output_inlier.push_back(ObjId[i]);
    } else {
      pCurFrame->vObjLabel[ObjId[i]] = -1;
    }
  }
  InlierID = output_inlier;
  return to_cv(T);
}

// (bJoint is set to true at the top of every GrabImageRGBD, src/Tracking.cc:170: the non-joint optimisers are unreachable from TrackRGBD)
int Optimizer::PoseOptimizationNew(Frame*, Frame*, vector<int>&) { std::fprintf(stderr, "ref_track: PoseOptimizationNew is not reachable with bJoint = true\n"); std::abort(); }
cv::Mat Optimizer::PoseOptimizationObjMot(Frame*, Frame*, const vector<int>&, std::vector<int>&) { std::fprintf(stderr, "ref_track: PoseOptimizationObjMot is not reachable with bJoint = true\n"); std::abort(); }
void Optimizer::FullBatchOptimization(Map*, const cv::Mat) { ++g_full_batch_calls; }
void Optimizer::PartialBatchOptimization(Map*, const cv::Mat, const int) { ++g_partial_batch_calls; }

}  // namespace VDO_SLAM
#else
// libref_full.so: src/Converter.cc, src/Optimizer.cc and the vendored g2o are compiled verbatim beside this file (oracle/ref/Makefile) - no glue.
// The batch optimisers really run (src/Tracking.cc:1165-1183); their call counts are not observable from outside and read -1.
namespace VDO_SLAM { namespace { int g_full_batch_calls = -1, g_partial_batch_calls = -1, g_last_cam_iterations = -1; } }
#endif

using VDO_SLAM::System;
using VDO_SLAM::Tracking;
using VDO_SLAM::Frame;

namespace { int g_live_systems = 0; long g_fake_time = -1; }

// Frame::SampleKeyPoints seeds cv::RNG with time(NULL) (src/Frame.cc:684): the calls to time() from inside this library bind here (-Bsymbolic), so a
// test can give every frame the seed the oracle uses; -1 = the real clock
extern "C" time_t time(time_t* t) {
  time_t v;
  if (g_fake_time >= 0) v = (time_t)g_fake_time;
  else { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); v = ts.tv_sec; }
  if (t) *t = v;
  return v;
}
extern "C" void vdo_ref_set_time(long t) { g_fake_time = t; }

// The reference reads members it never initialises - e.g. Frame::N_s of the first frame sizes TemperalMatch (src/Tracking.cc:345; the RGB-D constructor
// of Frame does not set it) - and works because a fresh process hands it zero pages.  The Frame temporaries live on the STACK, whose contents depend on
// whatever ran before (another test, a signal handler of the HIP runtime when the product shares the process): every call into the reference therefore
// runs on a thread of its own with a freshly mapped, all-zero stack and every signal blocked - same inputs, same bytes, every time.
#include <functional>
#include <pthread.h>
#include <signal.h>
#include <execinfo.h>
#include <unistd.h>
#include <sys/syscall.h>
namespace {
void segv_backtrace(int sig, siginfo_t* si, void*) {      // VDO_REF_BT=1 (debug): which thread crashed where
  char msg[200];
  const int m = std::snprintf(msg, sizeof msg, "ref_track: fatal signal %d at address %p in thread %ld (main pid %d), backtrace:\n", sig, si ? si->si_addr : nullptr, (long)syscall(186), (int)getpid());
  if (write(2, msg, m) < 0) {}
  void* bt[64];
  const int n = backtrace(bt, 64);
  backtrace_symbols_fd(bt, n, 2);
  _exit(128 + sig);
}
void install_bt_handler() {
  { void* warm[4]; backtrace(warm, 4); }              // (loads libgcc now, not inside the handler)
  static char alt[1 << 16];
  stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0; sigaltstack(&ss, nullptr);
  struct sigaction sa; std::memset(&sa, 0, sizeof sa); sa.sa_sigaction = segv_backtrace; sa.sa_flags = SA_ONSTACK | SA_SIGINFO;
  sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGABRT, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr);
}
void* clean_stack_trampoline(void* p) {
  if (std::getenv("VDO_REF_BT")) {
    install_bt_handler();
    sigset_t un; sigemptyset(&un); sigaddset(&un, SIGSEGV); sigaddset(&un, SIGABRT); sigaddset(&un, SIGBUS); pthread_sigmask(SIG_UNBLOCK, &un, nullptr);
  }
  (*(std::function<void()>*)p)(); return nullptr;
}
void run_on_clean_stack(std::function<void()> fn) {
  const size_t kStack = (size_t)256 << 20;
  void* stk = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (stk == MAP_FAILED) { fn(); return; }
  pthread_attr_t at; pthread_attr_init(&at); pthread_attr_setstack(&at, stk, kStack);
  sigset_t all, old; sigfillset(&all); pthread_sigmask(SIG_BLOCK, &all, &old);      // (inherited by the new thread)
  pthread_t th;
  const int rc = pthread_create(&th, &at, clean_stack_trampoline, &fn);
  pthread_sigmask(SIG_SETMASK, &old, nullptr);
  if (rc == 0) pthread_join(th, nullptr); else fn();
  pthread_attr_destroy(&at);
  munmap(stk, kStack);
}
}  // namespace

extern "C" {
// System::System(settings, RGBD) (src/System.cc:22-48)
void* vdo_ref_system_create(const char* settings) {
  if (std::getenv("VDO_REF_BT")) install_bt_handler();
  if (g_live_systems == 0) ref_arena::mark();
  ++g_live_systems;
#ifndef REF_REAL_OPTIMIZER
  VDO_SLAM::g_full_batch_calls = VDO_SLAM::g_partial_batch_calls = 0;
#endif
  // (the reference keeps the intrinsics in class statics set by the FIRST Frame of the process, src/Frame.cc:26-30,240-254: one calibration per
  //  process; a test that builds a second System with other settings starts them over)
  Frame::mbInitialComputations = true; Frame::nNextId = 0;
  System* out = nullptr;
  run_on_clean_stack([&] { out = new System(settings, System::RGBD); });
  return out;
}
void vdo_ref_system_destroy(void* s) {      // (the reference never frees its Tracking / Map either: the pages of the arena go back instead)
  (void)s;
  if (g_live_systems > 0 && --g_live_systems == 0) ref_arena::release();
}

// One System::TrackRGBD call (include/System.h:45-51): im (h x w x channels u8), depth (in/out f32: raw -> metres), flow (f32 x 2), mask (in/out i32),
// ground-truth camera pose (4x4 f32) and object rows [n_rows][row_len]; Tcw_out 16 floats.
int vdo_ref_system_track(void* sp, const unsigned char* im, int channels, float* depth, const float* flow, int* mask, int w, int h, const float* Tcw_gt16,
                         const float* obj_rows, int n_rows, int row_len, double timestamp, int n_images, float* Tcw_out) {
  System* s = (System*)sp;
  cv::Mat I(h, w, CV_MAKETYPE(CV_8U, channels), (void*)im), D(h, w, CV_32FC1, depth), Fl(h, w, CV_32FC2, (void*)flow), M(h, w, CV_32SC1, mask);
  cv::Mat gt = cv::Mat::eye(4, 4, CV_32F), traj = cv::Mat::zeros(10, 10, CV_8UC3);
  if (Tcw_gt16) std::memcpy(gt.data, Tcw_gt16, 64);
  std::vector<std::vector<float> > rows(n_rows);
  for (int i = 0; i < n_rows; ++i) rows[i].assign(obj_rows + (size_t)i * row_len, obj_rows + (size_t)(i + 1) * row_len);
  cv::Mat T;
  run_on_clean_stack([&] { T = s->TrackRGBD(I, D, Fl, M, gt, rows, timestamp, traj, n_images); });
  if (T.empty()) return -1;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Tcw_out[4 * i + j] = T.at<float>(i, j);
  return 0;
}

// per-frame facts after the call: [0] mvKeys (ORB / samples), [1] N_s_tmp (renewed static set), [2] mvObjKeys (renewed object set), [3] objects (nSemPosition),
// [4] TemperalMatch_subset, [5] camera LM iterations (the glue's), [6] f_id, [7] max_id, [8] mvTmpObjKeys, [9] full batch calls, [10] partial batch calls,
// [11] static tracklets, [12] dynamic tracklets (sizes of mpMap->TrackletSta / TrackletDyn)
void vdo_ref_system_counts(void* sp, int32_t* out13) {
  Tracking* T = ((System*)sp)->mpTracker;
  const Frame& F = T->mCurrentFrame;
  out13[0] = (int)F.mvKeys.size(); out13[1] = F.N_s_tmp; out13[2] = (int)F.mvObjKeys.size(); out13[3] = (int)F.nSemPosition.size();
  out13[4] = (int)T->TemperalMatch_subset.size(); out13[5] = VDO_SLAM::g_last_cam_iterations; out13[6] = T->f_id; out13[7] = T->max_id;
  out13[8] = (int)T->mvTmpObjKeys.size(); out13[9] = VDO_SLAM::g_full_batch_calls; out13[10] = VDO_SLAM::g_partial_batch_calls;
  out13[11] = (int)((System*)sp)->mpMap->TrackletSta.size(); out13[12] = (int)((System*)sp)->mpMap->TrackletDyn.size();
}

// the same flat views as host_system_frame_state of the product's host library (vdo_slam_amd/host/System.cc): what = 0 static set [10][n]
// (x y cx cy fx fy depth X Y Z), 1 object set [12][n] (... + vSemObjLabel, vObjLabel), 2 per object [n][19] (nSemPosition, nModLabel, bObjStat, vObjMod),
// 3 samples [8][n] (x y cx cy fx fy depth label), 4 scalars (max_id, mTcw).  Returns n (rows filled only if cap allows).
int vdo_ref_system_frame_state(void* sp, int what, float* out, int cap) {
  Tracking* T = ((System*)sp)->mpTracker;
  const Frame& F = T->mCurrentFrame;
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
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) o[3 + 4 * i + j] = F.vObjMod[a].at<float>(i, j);
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
    if (out && cap >= 17) { out[0] = (float)T->max_id; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[1 + 4 * i + j] = F.mTcw.at<float>(i, j); }
    return 1;
  }
  return -1;
}

// tracklets of the Map (GetStaticTrack / GetDynamicTrackNew, src/Tracking.cc:2201-2421): which = 0 static, 1 dynamic.  With off == NULL returns the sizes.
int vdo_ref_system_tracks(void* sp, int which, int64_t* sizes2, int32_t* off, int32_t* frame, int32_t* feat, int32_t* obj) {
  VDO_SLAM::Map* m = ((System*)sp)->mpMap;
  const std::vector<std::vector<std::pair<int, int> > >& T = which ? m->TrackletDyn : m->TrackletSta;
  int64_t np = 0;
  for (size_t t = 0; t < T.size(); ++t) np += (int64_t)T[t].size();
  if (sizes2) { sizes2[0] = (int64_t)T.size(); sizes2[1] = np; }
  if (off) {
    int64_t k = 0;
    off[0] = 0;
    for (size_t t = 0; t < T.size(); ++t) {
      for (size_t q = 0; q < T[t].size(); ++q, ++k) { frame[k] = T[t][q].first; feat[k] = T[t][q].second; }
      off[t + 1] = (int32_t)k;
      if (which && obj) obj[t] = m->nObjID[t];
    }
  }
  return 0;
}

// flat copy of the Map the batch optimisers read and write (include/Map.h:35-84): what = 0 vmCameraPose [F][16], 1 vmCameraPose_RF, 2 vmRigidMotion (all frames, all entries)
// [n][16], 3 vmRigidMotion_RF, 4 vnRMLabel [n] (as floats), 5 vp3DPointSta [n][3], 6 vp3DPointDyn [n][3], 7 entries of vmRigidMotion per frame [F-1].  Returns the number of floats
// (out filled only when cap allows), -1 on a bad `what`.
long vdo_ref_system_map_export(void* sp, int what, float* out, long cap) {
  VDO_SLAM::Map* m = ((System*)sp)->mpMap;
  long n = 0;
  auto put = [&](float v) { if (out && n < cap) out[n] = v; ++n; };
  auto put_mat = [&](const cv::Mat& M) { for (int i = 0; i < M.rows; ++i) for (int j = 0; j < M.cols; ++j) put(M.at<float>(i, j)); };
  if (what == 0 || what == 1) { for (const cv::Mat& T : (what ? m->vmCameraPose_RF : m->vmCameraPose)) put_mat(T); }
  else if (what == 2 || what == 3) { for (const auto& fr : (what == 3 ? m->vmRigidMotion_RF : m->vmRigidMotion)) for (const cv::Mat& T : fr) put_mat(T); }
  else if (what == 4) { for (const auto& fr : m->vnRMLabel) for (int l : fr) put((float)l); }
  else if (what == 5 || what == 6) { for (const auto& fr : (what == 6 ? m->vp3DPointDyn : m->vp3DPointSta)) for (const cv::Mat& X : fr) put_mat(X); }
  else if (what == 7) { for (const auto& fr : m->vmRigidMotion) put((float)fr.size()); }
  else return -1;
  return n;
}

// the five clock() brackets of the frame (all_timing, src/Tracking.cc:230-243, 685-703, 868-1010, 1016-1026, 1370-1603), milliseconds
void vdo_ref_system_timing(void* sp, float* ms5) {
  Tracking* T = ((System*)sp)->mpTracker;
  for (int i = 0; i < 5; ++i) ms5[i] = i < (int)T->all_timing.size() ? T->all_timing[i] : 0.f;
}
}
