#include "FramePipeline.h"

#include "Optimizer.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <stdexcept>

namespace VDO_SLAM {

namespace {
// Converter::toInvMatrix (src/Converter.cc:151-166): t_inv = -R.t() * t is a cv::gemm with a TRANSPOSED operand (GEMM_1_T), which OpenCV 3.4 runs
// through its generic GEMMSingleMul<float, double>: the dot product accumulated in double, k ascending, times alpha = -1, ONE rounding to float.
// (The untransposed small products of this file - Tcw * H, R * x + t - take cv::gemm's 2..4-wide fast path instead, which works in float,
// left to right: those loops are written in float on purpose.)
void inv_rigid(const float* T, float* o) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) o[4 * i + j] = T[4 * j + i];
    double s = 0.0;
    for (int k = 0; k < 3; ++k) s += (double)T[4 * k + i] * (double)T[4 * k + 3];
    o[4 * i + 3] = (float)(s * -1.0);
  }
  o[12] = o[13] = o[14] = 0; o[15] = 1;
}
}  // namespace

// One persistent helper thread: run() hands it a job, wait() returns the job's result.  It polls (a job arrives every few
// hundred microseconds while a sequence is running; a condition variable's wake-up latency would eat the overlap) and backs
// off to yield/sleep when idle.
class FramePipeline::Worker {
 public:
  Worker() : th_([this] { loop(); }) {}
  ~Worker() { state_.store(3, std::memory_order_release); th_.join(); }
  void run(std::function<int()> job) { job_ = std::move(job); state_.store(1, std::memory_order_release); }
  int wait() {
    if (!busy()) return 0;
    while (state_.load(std::memory_order_acquire) != 2) std::this_thread::yield();
    state_.store(0, std::memory_order_relaxed);
    return rc_;
  }
  bool busy() const { const int s = state_.load(std::memory_order_acquire); return s == 1 || s == 2; }

 private:
  void loop() {
    unsigned idle = 0;
    for (;;) {
      const int s = state_.load(std::memory_order_acquire);
      if (s == 3) return;
      if (s == 1) { rc_ = job_(); state_.store(2, std::memory_order_release); idle = 0; continue; }
      if (++idle < 200000u) continue;                            // ~0.1 ms of polling, then be polite
      if (idle < 400000u) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
  std::function<int()> job_;
  int rc_ = 0;
  std::atomic<int> state_{0};     // 0 idle, 1 job posted, 2 job done, 3 quit
  std::thread th_;                // (last: started after the other members exist)
};


static void fill_flow2(vdo_flow2_problem& p, int n, const double* obs, const double* flow, const double* depth, const float* K4, const float* Tcw_last,
                       const double* T0, double info_prior, int max_it) {
  std::memset(&p, 0, sizeof p);
  p.n = n; p.obs = obs; p.flow = flow; p.depth = depth;
  for (int i = 0; i < 4; ++i) p.K[i] = K4[i];
  float Twl[16];
  inv_rigid(Tcw_last, Twl);                              // Converter::toInvMatrix(pLastFrame->mTcw)  (Optimizer.cc:2414-2420)
  for (int i = 0; i < 16; ++i) { p.Twl[i] = Twl[i]; p.T0[i] = T0[i]; }
  p.info_flow = 0.1; p.info_prior = info_prior; p.huber_delta = (double)std::sqrt(0.04f); p.chi2_gate = (double)0.04f;
  p.max_iterations = max_it; p.ref_quirks = 1;
}

#define VDO_TRY(call) do { if ((call) != VDO_OK) { std::fprintf(stderr, "FramePipeline: %s\n", vdo_last_error()); return -1; } } while (0)

FramePipeline::FramePipeline(vdo_ctx* ctx, vdo_ctx* ctx_lm, const PipelineParams& p, vdo_ctx* ctx_obj, vdo_ctx* ctx_worker, vdo_ctx* ctx_orb)
    : ctx_(ctx), ctx_lm_(ctx_lm), ctx_obj_(ctx_obj ? ctx_obj : ctx_lm), ctx_w_(ctx_worker ? ctx_worker : ctx), p_(p) {
  vdo_orb_params op{p.n_features, p.scale_factor, p.n_levels, p.ini_th, p.min_th};
  if (vdo_orb_create(ctx_orb ? ctx_orb : ctx, &op, p.width, p.height, &orb_) != VDO_OK) return;
  for (int k = 0; k < 2; ++k) if (vdo_frame_images_create(ctx, p.width, p.height, &img_[k]) != VDO_OK) return;
  if (vdo_tracks_create(0, &tr_sta_) != VDO_OK || vdo_tracks_create(1, &tr_dyn_) != VDO_OK) return;
  const int capk = std::max(p.n_features + 256, 3008);           // (SampleKeyPoints yields 3000)
  kx_.resize(capk); ky_.resize(capk); kr_.resize(capk); ka_.resize(capk); ks_.resize(capk); ko_.resize(capk);
  for (int i = 0; i < 16; ++i) Tcw_last_[i] = vel_[i] = (i % 5 == 0) ? 1.f : 0.f;
  if (p.build_lm) {
    const int32_t ccap = std::max(p.max_track_bg + 8, p.n_features + 256);      // frame 1 tracks every filtered ORB keypoint of frame 0 (Initialization)
    if (vdo_flow2_batch_reserve(ctx_lm, 1, &ccap, &lm_cam_) != VDO_OK) return;
    std::vector<int32_t> ocap(obj_slots_, obj_cap_);
    if (vdo_flow2_batch_reserve(ctx_obj_, obj_slots_, ocap.data(), &lm_obj_) != VDO_OK) return;
  }
  store_.sta.reserve((size_t)1 << 20); store_.dyn.reserve((size_t)1 << 22);      // ~800 frames of 1 200 static / 5 000 object features before a re-allocation
  if (ctx_worker) worker_.reset(new Worker());
  if (ctx_worker && ctx_orb) worker_orb_.reset(new Worker());
  ctx_orb_ = ctx_orb;
  orb_split_ = ctx_orb != nullptr;
  ok_ = true;
}

FramePipeline::~FramePipeline() {
  if (worker_) { worker_->wait(); worker_.reset(); }
  if (worker_orb_) { worker_orb_->wait(); worker_orb_.reset(); }
  if (orb_) vdo_orb_destroy(orb_);
  for (int k = 0; k < 2; ++k) if (img_[k]) vdo_frame_images_destroy(img_[k]);
  if (tr_sta_) vdo_tracks_destroy(tr_sta_);
  if (tr_dyn_) vdo_tracks_destroy(tr_dyn_);
  if (lm_cam_) vdo_flow2_batch_destroy(lm_cam_);
  if (lm_obj_) vdo_flow2_batch_destroy(lm_obj_);
}

// More accepted objects than slots, or an object with more correspondences than a slot holds: a larger batch replaces the
// current one (no launch of it is in flight here: the previous frame's object stage has been consumed).
int FramePipeline::ReserveObjectSlots(int n_objects, int max_points) {
  if (n_objects <= obj_slots_ && max_points <= obj_cap_) return 0;
  const int slots = std::max(obj_slots_, (n_objects + 3) / 4 * 4), cap = std::max(obj_cap_, (max_points + 2047) / 2048 * 2048);
  std::vector<int32_t> ocap(slots, cap);
  vdo_flow2_batch* nb = nullptr;
  VDO_TRY(vdo_flow2_batch_reserve(ctx_obj_, slots, ocap.data(), &nb));
  if (lm_obj_) vdo_flow2_batch_destroy(lm_obj_);
  lm_obj_ = nb; obj_slots_ = slots; obj_cap_ = cap;
  return 0;
}

// The camera stage of the NEXT frame: GetInitModelCam (RANSAC-P3P + EPnP refit against the motion model, Tracking.cc:1614-1715) and the launch
// of PoseOptimizationFlow2Cam (K16, Tracking.cc:690-700) on the LM stream.  Nothing in it reads the next frame's images: the 3-D points, key
// points, flow and depth are the last frame's (mLastFrame.mvStatKeys / mvCorres / mvFlowNext / mvStatDepth), the "current" key points are
// the correspondences the last frame's flow predicts, the motion model is mVelocity * last pose.  So it can start as soon as a frame's
// static stage is over - Step() calls it at its end (cam_ahead_), while the frame's object optimisations are still running, and the next
// Step() finds the camera pose computed or on its way; without that (first call, VDO_PIPE_NO_CAM_AHEAD) the next Step() calls it at its
// start.  Same inputs, same arithmetic, same results either way (tests/test_track_sequence_gpu.py runs both).
int FramePipeline::CameraStage() {
  const auto t0 = std::chrono::steady_clock::now();
  cam_run_ = nullptr; cam_n_pts_ = 0; cam_n_ransac_ = 0; cam_n_mm_ = 0;
  const int n_s = have_last_ ? (int)sta_.cx.size() : 0;
  if (have_last_ && n_s >= 4) {
    std::vector<double>& X = dcam_[0]; std::vector<double>& uvd = dcam_[1];
    X.resize(3 * (size_t)n_s); uvd.resize(2 * (size_t)n_s);
    for (int i = 0; i < n_s; ++i) {
      X[3 * i] = sta_.xyz[3 * i]; X[3 * i + 1] = sta_.xyz[3 * i + 1]; X[3 * i + 2] = sta_.xyz[3 * i + 2];
      uvd[2 * i] = sta_.cx[i]; uvd[2 * i + 1] = sta_.cy[i];
    }
    vdo_pnp_problem pp{n_s, X.data(), uvd.data(), {p_.K4[0], p_.K4[1], p_.K4[2], p_.K4[3]}, 500, 0.4, 0.98, p_.pnp_refit};
    vdo_pnp_result pr;
    inl_ransac_cam_.assign(n_s, 0);
    VDO_TRY(vdo_pnp_ransac(ctx_, &pp, &pr, inl_ransac_cam_.data()));
    // motion-model inliers (mVelocity * last pose), same 0.4 px gate; the larger set seeds the optimisation
    float MM[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float a = 0; for (int k = 0; k < 4; ++k) a += vel_[4 * i + k] * Tcw_last_[4 * k + j]; MM[4 * i + j] = a; }
    int mm = 0;
    inl_mm_cam_.assign(n_s, 0);
    for (int i = 0; i < n_s; ++i) {
      const float x = sta_.xyz[3 * i], y = sta_.xyz[3 * i + 1], z = sta_.xyz[3 * i + 2];
      const float xc = MM[0] * x + MM[1] * y + MM[2] * z + MM[3], yc = MM[4] * x + MM[5] * y + MM[6] * z + MM[7], invz = 1.0f / (MM[8] * x + MM[9] * y + MM[10] * z + MM[11]);
      const float u_ = sta_.cx[i] - (p_.K4[0] * xc * invz + p_.K4[2]), v_ = sta_.cy[i] - (p_.K4[1] * yc * invz + p_.K4[3]);
      if (std::sqrt(u_ * u_ + v_ * v_) < 0.4f) { inl_mm_cam_[i] = 1; ++mm; }
    }
    cam_n_ransac_ = pr.n_inliers; cam_n_mm_ = mm;
    if (lm_cam_) {
      // TemperalMatch_subset + initial pose: RANSAC model if it has more inliers than the motion model (Tracking.cc:1690-1712)
      const bool use_ransac = pr.n_inliers > mm;
      const std::vector<uint8_t>& flag = use_ransac ? inl_ransac_cam_ : inl_mm_cam_;
      double T0[16];
      for (int i = 0; i < 16; ++i) T0[i] = use_ransac ? (double)(float)pr.T[i] : (double)MM[i];     // iniTcw is a CV_32F Mat
      cam_subset_.clear();
      std::vector<double>&ob = dcam_[2], &fl = dcam_[3], &dp = dcam_[4];
      ob.clear(); fl.clear(); dp.clear();
      for (int i = 0; i < n_s; ++i) {
        if (!flag[i]) continue;
        cam_subset_.push_back(i);
        ob.push_back(sta_.x[i]); ob.push_back(sta_.y[i]); fl.push_back(sta_.fx[i]); fl.push_back(sta_.fy[i]); dp.push_back(sta_.d[i]);
      }
      vdo_flow2_problem fp;
      fill_flow2(fp, (int)cam_subset_.size(), ob.data(), fl.data(), dp.data(), p_.K4, Tcw_last_, T0, 0.3, 100);
      VDO_TRY(vdo_flow2_batch_set(lm_cam_, 0, &fp));        // (copied into the batch's pinned block: dcam_ is free again)
      cam_run_ = lm_cam_; cam_n_pts_ = fp.n;
      for (int i = 0; i < 16; ++i) Tcw_init_[i] = (float)T0[i];
    }
  } else if (lm_cam_) {
    for (int i = 0; i < 16; ++i) Tcw_init_[i] = Tcw_last_[i];
    VDO_TRY(vdo_flow2_batch_set(lm_cam_, 0, nullptr));
    if (have_last_) { cam_run_ = lm_cam_; cam_n_pts_ = 0; }
  }
  if (cam_run_) VDO_TRY(vdo_flow2_batch_run(cam_run_));      // on the LM stream
  ms_[0] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

int FramePipeline::Step(const uint8_t* d_gray, const float* d_depth_raw, const float* d_flow, const int32_t* d_mask,
                        vdo_flow2_batch* cam, vdo_flow2_batch* obj, int n_cam_pts, int n_obj_problems, FrameCounts* out) {
  if (!ok_) return -1;
  FrameCounts fc{};
  auto t_prev = std::chrono::steady_clock::now();
  const auto t_step0 = t_prev;
  static const bool ev_env = std::getenv("VDO_PIPE_EVENTS") != nullptr;
  ev_on_ = ev_env; ev_t0_ = t_step0;
  double ms0[12];
  for (int i = 0; i < 12; ++i) ms0[i] = ms_[i];
  static const bool trace_slow = std::getenv("VDO_PIPE_TRACE_SLOW") != nullptr;    // (debug: the sections of a step that took > 2.5 ms)
  auto tick = [&](int slot) { const auto t = std::chrono::steady_clock::now(); ms_[slot] += std::chrono::duration<double, std::milli>(t - t_prev).count(); t_prev = t; };
  vdo_frame_images *cur = img_[cur_], *last = img_[cur_ ^ 1];
  const int W = p_.width, H = p_.height;
  struct Join { Worker* w; ~Join() { if (w) w->wait(); } } join_guard{worker_.get()};      // never leave Step with the helper thread on its locals
  // ---- ORB (K3-K7) needs only the grey image: with a stream of its own its device stage starts now, under the camera stage
  vdo_keypoints kp{(int32_t)kx_.size(), 0, kx_.data(), ky_.data(), kr_.data(), ka_.data(), ks_.data(), ko_.data()};
  // K9 + K10 of the new image: only RenewFrameInfo needs them
  int n_new_s = 0, n_tmp = 0;
  std::vector<int32_t>& keep = i_[1];
  ObjSet& tmp = tmpb_[cur_];                            // K10: semi-dense sampling of this image (mvTmpObj*)
  const int cap_s = ((W + 3) / 4) * ((H + 3) / 4);
  auto size_filter_outputs = [&]() {
    keep.resize(std::max(kp.n, 1));
    for (int k = 2; k < 7; ++k) f_[k].resize(std::max(kp.n, 1));
  };
  auto size_sample_outputs = [&]() {
    tmp.x.resize(cap_s); tmp.y.resize(cap_s); tmp.cx.resize(cap_s); tmp.cy.resize(cap_s); tmp.fx.resize(cap_s); tmp.fy.resize(cap_s); tmp.d.resize(cap_s); tmp.sem.resize(cap_s);
  };
  // K10 alone, on the ORB thread's stream (second scratch set of the image set): nothing in this Step reads the samples - the object
  // stage of this frame does, in the next Step - so they leave the static stage and run behind UpdateMask on the thread that is free
  auto run_k10_on_orb = [&]() -> int {
    const auto t0 = std::chrono::steady_clock::now();
    size_sample_outputs();
    VDO_TRY(vdo_frame_object_sample_on(ctx_orb_, cur, p_.th_depth_obj, 4, cap_s, tmp.x.data(), tmp.y.data(), tmp.cx.data(), tmp.cy.data(), tmp.fx.data(), tmp.fy.data(), tmp.d.data(),
                                       tmp.sem.data(), &n_tmp));
    ms_[11] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
  };
  // (with a thread of its own the whole extraction - device stage, quadtrees, angles - leaves the main thread: nothing before the
  // static stage reads a keypoint)
  Join join_orb{nullptr};                                // declared after kp: joined before kp goes away on every path out of Step
  const int tag = f_id_ + 1;
  orb_ready_.store(0, std::memory_order_relaxed); objects_done_.store(0, std::memory_order_relaxed); mask_final_.store(0, std::memory_order_relaxed);   // (a Step that failed may have left this tag behind)
  // (destroyed BEFORE join_orb: whatever way Step is left, the ORB thread's waits below end)
  struct Release {
    std::atomic<int>*obj, *msk; int tag;
    ~Release() { for (std::atomic<int>* a : {obj, msk}) { const int v = a->load(); if (v != tag && v != -tag) a->store(-tag); } }
  } release{&objects_done_, &mask_final_, tag};
  const bool orb_pending = worker_orb_ && !p_.use_sample_feature;
  const bool tail_via_orb = orb_pending && pending_ && worker_;     // its job goes on with the tail of the last frame's object stage
  const bool k10_via_orb = orb_pending && have_last_ && worker_;    // ... and ends with K10 of this frame, behind UpdateMask
  if (orb_pending) {
    worker_orb_->run([this, &kp, &run_k10_on_orb, &fc, d_gray, W, tag, tail_via_orb, k10_via_orb]() -> int {
      const auto t0 = std::chrono::steady_clock::now();
      int rc = vdo_orb_extract(orb_, d_gray, W, host_inputs_ ? 0 : 1, &kp) == VDO_OK ? 0 : -1;
      ms_[1] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rc != 0) std::fprintf(stderr, "FramePipeline: %s\n", vdo_last_error());
      mark(kEvOrbDone);
      orb_ready_.store(rc == 0 ? tag : -tag, std::memory_order_release);
      if (tail_via_orb) {                                // the tail of the last frame's object stage, as soon as that stage is over
        int v;
        while ((v = objects_done_.load(std::memory_order_acquire)) != tag && v != -tag) std::this_thread::yield();
        if (v == tag) {
          const int rc2 = FinishObjectsTail(&fc);
          tail_done_.store(true, std::memory_order_release);
          if (rc2 != 0) rc = rc2;
        }
      }
      if (k10_via_orb) {                                 // K10 reads the mask UpdateMask may repair: behind it
        int v;
        while ((v = mask_final_.load(std::memory_order_acquire)) != tag && v != -tag) std::this_thread::yield();
        if (v == tag && run_k10_on_orb() != 0) rc = -1;
      }
      return rc;
    });
    join_orb.w = worker_orb_.get();
  } else if (orb_split_ && !p_.use_sample_feature) VDO_TRY(vdo_orb_extract_begin(orb_, d_gray, W, host_inputs_ ? 0 : 1));
  // the keypoints of this frame are ready - called by whoever reads them first: the static stage
  auto orb_join = [&]() -> int {
    if (!orb_pending) return 0;
    int v;
    while ((v = orb_ready_.load(std::memory_order_acquire)) != tag && v != -tag) std::this_thread::yield();
    fc.n_orb = kp.n;
    return v == tag ? 0 : -1;
  };
  // ---- deferred mode: the object stage of the PREVIOUS frame ends during this frame's camera stage + ORB front-end (nothing
  // there depends on the object set) - on the helper thread if there is one, else right after ORB on this thread
  bool fin_async = false;
  if (pending_ && worker_) {
    vdo_frame_images_set_ctx(img_obj_, ctx_w_);
    worker_->run([this, &fc] { return FinishObjects(&fc, true); });
    fin_async = true;
  }
  // ---- GrabImageRGBD: images, K1, UpdateMask (K15), propagation (K11)            Tracking.cc:180-305
  // Host inputs (System::TrackRGBD): 7.5 MB of pageable copies.  Only the depth map is needed at once (K1, K11); the flow and the mask are first
  // read behind the camera stage (UpdateMask, the static stage): they go up while the camera optimisation runs on its own stream (the
  // copies block this thread for ~0.15 ms it would otherwise spend waiting for that launch).  The converted depth map the caller is owed
  // (the reference converts imD in place) is read back while the object optimisations of the frame run, instead of after the frame.
  // (Converting the caller's copy on the host instead - the same two correctly rounded divisions per pixel - takes one thread 0.4 ms: measured, dropped.)
  const bool late_upload = host_inputs_ && !std::getenv("VDO_PIPE_SYNC_UPLOAD");
  depth_on_host_ = false;
  // ... and when the helper thread has nothing to do at this point (synchronous mode: no object stage of the last frame to finish), IT brings the
  // flow and the mask up, on its own stream, while this thread uploads the depth map and runs K1 / K11 on it (two pageable copies side by side:
  // ~55 GB/s instead of ~37)
  bool up_async = false;
  static const bool up_async_on = std::getenv("VDO_PIPE_NO_ASYNC_UPLOAD") == nullptr;
  if (up_async_on && late_upload && worker_ && !fin_async) {
    vdo_ctx* cw = ctx_w_;
    worker_->run([cw, cur, d_flow, d_mask]() -> int { return vdo_frame_images_upload_on(cw, cur, nullptr, d_flow, d_mask) == VDO_OK ? 0 : -1; });
    up_async = true;
  }
  if (late_upload) VDO_TRY(vdo_frame_images_upload(cur, d_depth_raw, nullptr, nullptr));
  else if (host_inputs_) VDO_TRY(vdo_frame_images_upload(cur, d_depth_raw, d_flow, d_mask));
  else VDO_TRY(vdo_frame_images_ingest_device(cur, d_depth_raw, d_flow, d_mask, p_.bf, p_.depth_map_factor, depth_metric_ ? 0 : 1));      // copies + K1, one launch
  if (host_inputs_ && !depth_metric_) VDO_TRY(vdo_frame_images_depth_preprocess(cur, p_.bf, p_.depth_map_factor));
  const int n_s = have_last_ ? (int)sta_.cx.size() : 0;
  std::vector<float>& stat_depth = f_[0]; std::vector<float>& obj_depth = f_[1]; std::vector<int32_t>& obj_sem = i_[0];
  stat_depth.assign(n_s, -1.f);
  // K11 (static): the depth under the propagated static keys (mCurrentFrame.mvStatDepth, src/Tracking.cc:259-281).  Nothing of this frame's
  // object chain reads it - RenewFrameInfo replaces it by the depth of the renewed set (:1040) - so it runs with the static stage (stage_static
  // below, the helper thread's stream) instead of standing, with its round trip to the host, in front of UpdateMask on this thread.
  if (!have_last_) VDO_TRY(vdo_ctx_synchronize(ctx_));
  tick(0); mark(kEvInputs);
  // ---- GetInitModelCam + PoseOptimizationFlow2Cam (K16): launched at the end of the LAST Step if the camera stage runs ahead (CameraStage)
  {
    vdo_flow2_batch* cam_ext = cam;
    if (!cam_ahead_ && CameraStage() != 0) return -1;
    cam_ahead_ = false;
    fc.n_ransac_cam = cam_n_ransac_; fc.n_motion_model_cam = cam_n_mm_;
    if (cam_run_ || !cam_ext) { cam = cam_run_; n_cam_pts = cam_n_pts_; }
    else VDO_TRY(vdo_flow2_batch_run(cam_ext));             // (a caller-supplied batch, build_lm = 0)
  }
  t_prev = std::chrono::steady_clock::now();               // (CameraStage books its own time)
  if (up_async) { if (worker_->wait() != 0) { std::fprintf(stderr, "FramePipeline: %s\n", vdo_last_error()); return -1; } }
  else if (late_upload) VDO_TRY(vdo_frame_images_upload(cur, nullptr, d_flow, d_mask));
  if (p_.use_sample_feature) {                           // Option II of Frame::Frame (src/Frame.cc:132-166): random samples instead of ORB
    int ns = 0;
    VDO_TRY(vdo_sample_keypoints(H, W, (uint64_t)(p_.sample_seed + f_id_), kp.capacity, kx_.data(), ky_.data(), &ns));
    kp.n = ns;
    fc.n_orb = kp.n;
    tick(1);
  } else if (orb_pending) {
    // (on its own thread)
  } else {
    if (orb_split_) VDO_TRY(vdo_orb_extract_end(orb_, &kp));
    else VDO_TRY(vdo_orb_extract(orb_, d_gray, W, host_inputs_ ? 0 : 1, &kp));
    fc.n_orb = kp.n;
    tick(1);
  }
  // K9 (+ K10 unless the ORB thread samples the objects) of the new image: one call, one synchronisation
  auto frame_filters = [&]() -> int {
    if (orb_join() != 0) return -1;
    size_filter_outputs();
    if (k10_via_orb) {
      VDO_TRY((p_.use_sample_feature ? vdo_frame_static_filter_sampled : vdo_frame_static_filter)(cur, kp.n, kx_.data(), ky_.data(), p_.th_depth_bg, keep.data(), f_[2].data(), f_[3].data(),
                                                                                                    f_[4].data(), f_[5].data(), f_[6].data(), &n_new_s));
    } else {
      size_sample_outputs();
      VDO_TRY(vdo_frame_filters(cur, kp.n, kx_.data(), ky_.data(), p_.th_depth_bg, p_.use_sample_feature ? 1 : 0, keep.data(), f_[2].data(), f_[3].data(),
                                f_[4].data(), f_[5].data(), f_[6].data(), &n_new_s,
                                p_.th_depth_obj, 4, cap_s, tmp.x.data(), tmp.y.data(), tmp.cx.data(), tmp.cy.data(), tmp.fx.data(), tmp.fy.data(), tmp.d.data(), tmp.sem.data(), &n_tmp));
      fc.n_object_samples = n_tmp;
    }
    fc.n_static_new = n_new_s;
    return 0;
  };
  tick(2);
  // ---- consume the camera result
  float Tcw[16];
  for (int i = 0; i < 16; ++i) Tcw[i] = Tcw_last_[i];
  inl_out_.assign(std::max(n_cam_pts, 1), 1);
  if (cam) {                                             // (the fetch is stream-ordered behind the kernel and synchronises once)
    vdo_flow2_result r;
    flow_out_.resize(2 * (size_t)std::max(n_cam_pts, 1));
    double* fo = flow_out_.data(); uint8_t* io = inl_out_.data();
    VDO_TRY(vdo_flow2_batch_fetch(cam, &r, &fo, &io));
    if (!(cam == lm_cam_ && n_cam_pts < 3)) for (int i = 0; i < 16; ++i) Tcw[i] = (float)r.T[i];
    else if (cam == lm_cam_ && have_last_) for (int i = 0; i < 16; ++i) Tcw[i] = Tcw_init_[i];      // < 3 matches: the pose stays at its initial value
    fc.n_cam_inliers = r.n_inliers; fc.cam_lm_iterations = r.iterations;
  }
  // current static keys: the propagated correspondences, moved to (last key + refined flow) for the LM inliers (Optimizer.cc:2527-2532)
  std::vector<float>&cur_sx = f_[9], &cur_sy = f_[10];
  std::vector<int32_t>& tm = i_[7];
  if (have_last_) {
    cur_sx = sta_.cx; cur_sy = sta_.cy;
    tm.assign(n_s, -1);
    if (cam && cam == lm_cam_) {
      for (size_t j = 0; j < cam_subset_.size(); ++j) {
        if (!inl_out_[j]) continue;
        const int i = cam_subset_[j];
        tm[i] = i;
        // (float key + DOUBLE refined flow, rounded once on the assignment: `pt.x = pLastFrame->mvStatKeys[..].pt.x + flow_new(0)`, src/Optimizer.cc:2529-2530)
        cur_sx[i] = (float)((double)sta_.x[i] + flow_out_[2 * j]); cur_sy[i] = (float)((double)sta_.y[i] + flow_out_[2 * j + 1]);
      }
    } else {
      for (int i = 0; i < n_s; ++i) tm[i] = inl_out_[n_cam_pts > 0 ? i % n_cam_pts : 0] ? i : -1;
    }
  }
  tick(3); mark(kEvCamFetched);
  // ---- the object set of the last frame: wait for its object stage (its tail - tracklets, Map - goes on behind)
  bool tail_async = false, tail_on_orb = false;
  if (fin_async) {
    const int rc = worker_->wait();
    vdo_frame_images_set_ctx(last, ctx_);
    if (rc != 0) return -1;
    if (tail_pending_ && tail_via_orb) {
      // the ORB thread takes the tail (it is told that the object stage is over), the helper thread goes straight to the static stage
      tail_done_.store(false, std::memory_order_relaxed);
      objects_done_.store(tag, std::memory_order_release);
      tail_async = true; tail_on_orb = true;
    } else if (tail_pending_) { worker_->run([this, &fc] { return FinishObjectsTail(&fc); }); tail_async = true; }
  } else if (pending_) {
    if (FinishObjects(&fc) != 0) return -1;
  }
  t_prev = std::chrono::steady_clock::now();
  // ---- UpdateMask (K15) + object part of the propagation (K11): they need the object set of the last frame
  const int n_o = have_last_ ? (int)obj_.cx.size() : 0;
  obj_depth.assign(n_o, 0.f); obj_sem.assign(n_o, 0);
  std::vector<float>& flow3d = f_[7];
  std::vector<int32_t>& olab = i_[2];
  if (have_last_) {
    // UpdateMask (K15) -> K11 (objects) -> GetSceneFlowObj (K13): one call, one synchronisation   Tracking.cc:2997-3068, 283-305, 1278-1364
    int rec = 0;
    flow3d.resize(3 * (size_t)std::max(n_o, 1));
    olab.assign(n_o, -2);
    VDO_TRY(vdo_object_chain(cur, last, n_o, obj_.sem.data(), obj_.cx.data(), obj_.cy.data(), p_.th_depth_obj, Tcw, obj_.x.data(), obj_.y.data(), obj_.d.data(),
                             Tcw_last_, p_.K4, &rec, obj_depth.data(), obj_sem.data(), flow3d.data(), olab.data()));
    fc.n_recovered_masks = rec;
  }
  // vdo_object_chain returns without a synchronisation when there are no object samples (n_o == 0): the asynchronous ingest of this frame's images
  // (vdo_frame_images_ingest_device on ctx_'s stream) must still be through before the static stage - another stream - and K10 on the ORB context read
  // the depth map, the flow and the mask (every vdo_frame_images call is host-synchronous for its readers)
  if (have_last_ && n_o == 0) VDO_TRY(vdo_ctx_synchronize(ctx_));
  mask_final_.store(tag, std::memory_order_release);      // (UpdateMask is through: K10 may sample the mask)
  tick(10); mark(kEvObjChain);
  StaSet nsta; ObjSet nobj;
  std::vector<int32_t> sta_asso, dyn_asso;
  // ---- K9 + K10 of the new image, RenewFrameInfo (static) (K14, K12), static tracklets: independent of the object chain
  // below (scene flow -> DynObjTracking -> object RANSAC -> object LMs) - on the helper thread if there is one
  vdo_ctx* ctx_f = worker_ ? ctx_w_ : ctx_;
  auto stage_static = [&]() -> int {
    auto tp = std::chrono::steady_clock::now();
    auto tk = [&](int slot) { const auto t = std::chrono::steady_clock::now(); ms_[slot] += std::chrono::duration<double, std::milli>(t - tp).count(); tp = t; };
    VDO_TRY(vdo_propagate_static(cur, n_s, sta_.cx.data(), sta_.cy.data(), stat_depth.data()));      // K11 (static), see above
    if (frame_filters() != 0) return -1;
    tk(2); mark(kEvFilters);
    const int cs = p_.max_track_bg + 2;
    nsta.x.resize(cs); nsta.y.resize(cs); nsta.cx.resize(cs); nsta.cy.resize(cs); nsta.fx.resize(cs); nsta.fy.resize(cs); nsta.d.resize(cs);
    sta_asso.resize(cs);
    int m = 0;
    // top-up source: every ORB keypoint, or - UseSampleFeature - the filtered samples mvStatKeysTmp (Tracking.cc:2718-2721)
    int n_src = kp.n;
    const float *src_x = kx_.data(), *src_y = ky_.data();
    if (p_.use_sample_feature) {
      f_[13].resize(std::max(n_new_s, 1)); f_[14].resize(std::max(n_new_s, 1));
      for (int i = 0; i < n_new_s; ++i) { f_[13][i] = kx_[keep[i]]; f_[14][i] = ky_[keep[i]]; }
      n_src = n_new_s; src_x = f_[13].data(); src_y = f_[14].data();
    }
    float Twc[16];
    inv_rigid(Tcw, Twc);
    nsta.xyz.resize(3 * (size_t)cs);
    // RenewFrameInfo (static) + Get3DinWorld (mvStat3DPointTmp) in one pass, one synchronisation
    VDO_TRY(vdo_renew_static_world(cur, n_s, tm.data(), cur_sx.data(), cur_sy.data(), n_src, src_x, src_y, p_.max_track_bg, p_.K4, Twc,
                                   nsta.x.data(), nsta.y.data(), nsta.cx.data(), nsta.cy.data(), nsta.fx.data(), nsta.fy.data(), sta_asso.data(), nsta.d.data(), nsta.xyz.data(), &m));
    for (auto* v : {&nsta.x, &nsta.y, &nsta.cx, &nsta.cy, &nsta.fx, &nsta.fy, &nsta.d}) v->resize(m);
    sta_asso.resize(m);
    nsta.xyz.resize(3 * (size_t)std::max(m, 1));
    // ---- static tracklets (incremental GetStaticTrack)                             Tracking.cc:2201-2300
    while (!tail_done_.load(std::memory_order_acquire)) std::this_thread::yield();      // (the tail of the last frame may be reading the static tracklets: windowed batch optimisation)
    VDO_TRY(vdo_tracks_add_frame(tr_sta_, m, sta_asso.data(), nullptr));
    tk(5); mark(kEvStaticDone);
    return 0;
  };
  // (declared after every local stage_static touches: on an early return this wait runs BEFORE those locals are destroyed)
  Join join_static{worker_.get()};
  if (tail_async && !tail_on_orb && worker_->wait() != 0) return -1;
  bool static_async = false;
  if (have_last_ && worker_) {
    vdo_frame_images_set_ctx(cur, ctx_w_);
    worker_->run(stage_static);
    static_async = true;
  }
  if (!have_last_) {
    // ---- Initialization(): the new features ARE the tracked set                   Tracking.cc:1215-1276
    if (obj) VDO_TRY(vdo_flow2_batch_run(obj));
    if (frame_filters() != 0) return -1;
    nsta.x.resize(n_new_s); nsta.y.resize(n_new_s);
    for (int i = 0; i < n_new_s; ++i) { nsta.x[i] = kx_[keep[i]]; nsta.y[i] = ky_[keep[i]]; }
    nsta.cx.assign(f_[2].begin(), f_[2].begin() + n_new_s); nsta.cy.assign(f_[3].begin(), f_[3].begin() + n_new_s);
    nsta.fx.assign(f_[4].begin(), f_[4].begin() + n_new_s); nsta.fy.assign(f_[5].begin(), f_[5].begin() + n_new_s);
    nsta.d.assign(f_[6].begin(), f_[6].begin() + n_new_s);
    nobj = tmp;                                         // (copy: the buffer keeps its capacity for later frames)
    n_tmp_ = n_tmp; tmp_idx_obj_ = cur_;                // (ObjectSamples(): the samples of this frame)
    for (auto* v : {&nobj.x, &nobj.y, &nobj.cx, &nobj.cy, &nobj.fx, &nobj.fy, &nobj.d}) v->resize(n_tmp);
    nobj.sem.resize(n_tmp); nobj.label.assign(n_tmp, -2);
    if (obj) VDO_TRY(vdo_ctx_synchronize(ctx_lm_));
    {
      const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      nsta.xyz.resize(3 * (size_t)std::max(n_new_s, 1)); nobj.xyz.resize(3 * (size_t)std::max(n_tmp, 1));
      VDO_TRY(vdo_get3d_world(ctx_, n_new_s, nsta.x.data(), nsta.y.data(), nsta.d.data(), p_.K4, I4, nsta.xyz.data()));   // Get3DinCamera
      VDO_TRY(vdo_get3d_world(ctx_, n_tmp, nobj.x.data(), nobj.y.data(), nobj.d.data(), p_.K4, I4, nobj.xyz.data()));
    }
  } else {
    // ---- GetSceneFlowObj (K13) + DynObjTracking                                   Tracking.cc:1278-1612
    vdo_dyn_obj_params dp{W, H, 25, 50, p_.sf_mg_thres, p_.sf_ds_thres, p_.th_depth_obj, f_id_};
    std::vector<int32_t>&off = i_[3], &idx = i_[4], &osem = i_[5], &omod = i_[6];
    off.assign(n_o + 2, 0); idx.resize(std::max(n_o, 1)); osem.resize(n_o + 1); omod.resize(n_o + 1);
    int n_objects = 0;
    VDO_TRY(vdo_dyn_obj_tracking(&dp, n_o, obj_sem.data(), olab.data(), obj_.cx.data(), obj_.cy.data(), obj_depth.data(), flow3d.data(), obj_.sem.data(),
                                 (int)last_sem_pos_.size(), last_sem_pos_.data(), last_mod_label_.data(), last_obj_stat_.data(), &max_id_,
                                 off.data(), idx.data(), osem.data(), omod.data(), &n_objects));
    fc.n_objects = n_objects;
    tick(4); mark(kEvDynObj);
    // ---- GetInitModelObj for every accepted object: one batched RANSAC call                   Tracking.cc:1717-1849
    if (n_objects > 0) {
      std::vector<double>& X = d_[0]; std::vector<double>& uvd = d_[1];
      X.resize(3 * (size_t)off[n_objects] + 3); uvd.resize(2 * (size_t)off[n_objects] + 2);
      std::vector<vdo_pnp_problem> pp(n_objects);
      std::vector<vdo_pnp_result> pr(n_objects);
      for (int a = 0; a < n_objects; ++a) {
        for (int q = off[a]; q < off[a + 1]; ++q) {
          const int id = idx[q];
          X[3 * q] = obj_.xyz[3 * id]; X[3 * q + 1] = obj_.xyz[3 * id + 1]; X[3 * q + 2] = obj_.xyz[3 * id + 2];
          uvd[2 * q] = obj_.cx[id]; uvd[2 * q + 1] = obj_.cy[id];
        }
        pp[a] = vdo_pnp_problem{off[a + 1] - off[a], X.data() + 3 * (size_t)off[a], uvd.data() + 2 * (size_t)off[a], {p_.K4[0], p_.K4[1], p_.K4[2], p_.K4[3]}, 500, 0.4, 0.98, p_.pnp_refit};
      }
      std::vector<uint8_t>& rin = inl_ransac_;
      rin.assign((size_t)off[n_objects] + 1, 0);
      std::vector<uint8_t*> rip(n_objects);
      for (int a = 0; a < n_objects; ++a) rip[a] = rin.data() + off[a];
      // ---- the motion model of an object that was there in the last frame: MotionModel = mCurrentFrame.mTcw * mLastFrame.vObjMod[PreObjID],
      // its 0.4 px inliers; RANSAC seeds the LM only if it has MORE inliers                       Tracking.cc:1767-1825
      // The count needs nothing of the RANSAC: it runs on this thread WHILE the device works on the hypotheses and the votes (vdo_pnp_ransac_batch_overlap).
      std::vector<uint8_t>& min_ = inl_mm_;
      min_.assign((size_t)off[n_objects] + 1, 0);
      obj_use_mm_.assign(n_objects, 0);
      obj_mm_.resize(16 * (size_t)n_objects);
      std::vector<int> mm_cnt(n_objects, -1);                 // -1: the object has no motion of the last frame
      struct MmCtx { FramePipeline* self; const int* off; const int32_t* idx; const int32_t* omod; const float* Tcw; int n_objects; std::vector<uint8_t>* min_; std::vector<int>* cnt; };
      MmCtx mmc{this, off.data(), idx.data(), omod.data(), Tcw, n_objects, &min_, &mm_cnt};
      auto mm_work = [](void* vp) {
        MmCtx& c = *static_cast<MmCtx*>(vp);
        FramePipeline& P = *c.self;
        for (int a = 0; a < c.n_objects; ++a) {
          int pre = -1;
          for (size_t i = 0; i < P.last_mod_label_.size(); ++i) if (P.last_mod_label_[i] == c.omod[a]) { pre = (int)i; break; }
          if (pre < 0 || 16 * (size_t)pre + 16 > P.last_obj_mod_.size()) continue;
          float* MM = P.obj_mm_.data() + 16 * (size_t)a;
          const float* Hl = P.last_obj_mod_.data() + 16 * (size_t)pre;
          for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float s = 0; for (int k = 0; k < 4; ++k) s += c.Tcw[4 * i + k] * Hl[4 * k + j]; MM[4 * i + j] = s; }
          int mm = 0;
          for (int q = c.off[a]; q < c.off[a + 1]; ++q) {
            const int id = c.idx[q];
            const float x = P.obj_.xyz[3 * id], y = P.obj_.xyz[3 * id + 1], z = P.obj_.xyz[3 * id + 2];
            const float xc = MM[0] * x + MM[1] * y + MM[2] * z + MM[3], yc = MM[4] * x + MM[5] * y + MM[6] * z + MM[7], invz = 1.0f / (MM[8] * x + MM[9] * y + MM[10] * z + MM[11]);
            const float u_ = P.obj_.cx[id] - (P.p_.K4[0] * xc * invz + P.p_.K4[2]), v_ = P.obj_.cy[id] - (P.p_.K4[1] * yc * invz + P.p_.K4[3]);
            if (std::sqrt(u_ * u_ + v_ * v_) < 0.4f) { (*c.min_)[q] = 1; ++mm; }
          }
          (*c.cnt)[a] = mm;
        }
      };
      static const bool mm_overlap = std::getenv("VDO_PIPE_NO_MM_OVERLAP") == nullptr;      // (A/B switch)
      if (mm_overlap) VDO_TRY(vdo_pnp_ransac_batch_overlap(ctx_, n_objects, pp.data(), pr.data(), rip.data(), +mm_work, &mmc));
      else { VDO_TRY(vdo_pnp_ransac_batch(ctx_, n_objects, pp.data(), pr.data(), rip.data())); mm_work(&mmc); }
      for (int a = 0; a < n_objects; ++a) fc.n_ransac_obj += pr[a].n_inliers;
      for (int a = 0; a < n_objects; ++a) {
        if (mm_cnt[a] < 0) continue;
        fc.n_mm_inliers_obj += mm_cnt[a];
        if (!(pr[a].n_inliers > mm_cnt[a])) { obj_use_mm_[a] = 1; ++fc.n_motion_model_obj; }
      }
      if (lm_obj_) {
        // per object: ObjIdTest_in = inliers of the chosen model; fewer than 50 -> the object is not tracked this frame (Tracking.cc:879)
        if ((int)obj_subsets_.size() < n_objects) obj_subsets_.resize(n_objects);      // (inner vectors keep their capacity from frame to frame)
        for (int a = 0; a < n_objects; ++a) obj_subsets_[a].clear();
        obj_stat_.assign(n_objects, 1);
        obj_buf_.resize(n_objects);
        int need_pts = 0;
        for (int a = 0; a < n_objects; ++a) {
          std::vector<int32_t>& sub = obj_subsets_[a];
          const std::vector<uint8_t>& flag = obj_use_mm_[a] ? min_ : rin;
          for (int q = off[a]; q < off[a + 1]; ++q) if (flag[q]) sub.push_back(idx[q]);
          // (the reference also sets vObjLabel = -1 outside the chosen set, Tracking.cc:1841-1846: RenewFrameInfo only reads the labels of LM
          // inliers, a subset of the chosen set, and then replaces vObjLabel, :2862,2991 - nothing observes it)
          bool gated = true;                                                            // ground truth in both frames (Tracking.cc:791-841)
          if (gate_on_) {
            gated = std::find(gate_cur_.begin(), gate_cur_.end(), osem[a]) != gate_cur_.end() && std::find(gate_last_.begin(), gate_last_.end(), osem[a]) != gate_last_.end();
            if (!gated) { sub.clear(); for (int q = off[a]; q < off[a + 1]; ++q) sub.push_back(idx[q]); }   // vnObjInlierID = ObjIdNew
          }
          if (!gated || (int)sub.size() < 50) obj_stat_[a] = 0;
          else need_pts = std::max(need_pts, (int)sub.size());
        }
        if (ReserveObjectSlots(n_objects, need_pts) != 0) return -1;                    // every object gets a slot, whatever its size
        for (int a = 0; a < n_objects; ++a) {
          const std::vector<int32_t>& sub = obj_subsets_[a];
          if (!obj_stat_[a]) { VDO_TRY(vdo_flow2_batch_set(lm_obj_, a, nullptr)); continue; }
          ObjBuf& B = obj_buf_[a];
          B.ob.clear(); B.fl.clear(); B.dp.clear();
          for (int id : sub) { B.ob.push_back(obj_.x[id]); B.ob.push_back(obj_.y[id]); B.fl.push_back(obj_.fx[id]); B.fl.push_back(obj_.fy[id]); B.dp.push_back(obj_.d[id]); }
          double T0[16];
          for (int i = 0; i < 16; ++i) T0[i] = obj_use_mm_[a] ? (double)obj_mm_[16 * (size_t)a + i] : (double)(float)pr[a].T[i];   // mInitModel (CV_32F)
          vdo_flow2_problem fp;
          fill_flow2(fp, (int)sub.size(), B.ob.data(), B.fl.data(), B.dp.data(), p_.K4, Tcw_last_, T0, 0.5, 200);
          VDO_TRY(vdo_flow2_batch_set(lm_obj_, a, &fp));
        }
        for (int a = n_objects; a < obj_slots_; ++a) VDO_TRY(vdo_flow2_batch_set(lm_obj_, a, nullptr));
        obj = lm_obj_; n_obj_problems = obj_slots_;
      }
    } else if (lm_obj_) {
      for (int a = 0; a < obj_slots_; ++a) VDO_TRY(vdo_flow2_batch_set(lm_obj_, a, nullptr));
      obj = nullptr;
    }
    tick(9); mark(kEvObjLmBuilt);
    // ---- object motions (K17) on the LM stream, RenewFrameInfo (static) meanwhile  Tracking.cc:932 || :2666-2805
    if (obj) VDO_TRY(vdo_flow2_batch_run(obj));
    mark(kEvObjLmLaunched);
    // ---- RenewFrameInfo (static) meanwhile                                         Tracking.cc:2666-2805
    if (static_async) {
      const int rc = worker_->wait();
      vdo_frame_images_set_ctx(cur, ctx_);
      if (rc != 0) return -1;
    } else if (stage_static() != 0) return -1;
    if ((tail_on_orb || k10_via_orb) && worker_orb_->wait() != 0) return -1;
    if (k10_via_orb) fc.n_object_samples = n_tmp;
    mark(kEvStaticJoined);
    t_prev = std::chrono::steady_clock::now();
    // the object stage (results of the LMs, RenewFrameInfo of the objects, dynamic tracklets) ends in FinishObjects():
    // right below, or - deferred mode - inside the next Step, after that frame's camera stage and ORB front-end
    n_objects_ = n_objects; obj_run_ = obj; n_obj_problems_ = n_obj_problems; n_tmp_ = n_tmp; img_obj_ = cur; f_id_obj_ = f_id_; tmp_idx_obj_ = cur_;
    std::memcpy(Tcw_obj_, Tcw, sizeof Tcw);
    pending_ = true;
  }
  tick(8);
  fc.n_static_tracked = (int)nsta.x.size();
  int64_t np = 0;
  vdo_tracks_size(tr_sta_, &fc.n_static_tracks, &np);
  if (keep_graph_) {                                     // "Save Graph Structure" (1), (5): static features and the camera pose of this frame
    store_.sta.append(nsta.x.size(), nsta.x.data(), nsta.y.data(), nsta.d.data(), nsta.xyz.data());
    float Twc_m[16];
    inv_rigid(Tcw, Twc_m);
    store_.add_camera(Twc_m);
    if (!have_last_) store_.dyn.append(nobj.x.size(), nobj.x.data(), nobj.y.data(), nobj.d.data(), nobj.xyz.data());
  }
  sta_ = std::move(nsta);
  if (!have_last_) { fc.n_object_tracked = (int)nobj.x.size(); obj_ = std::move(nobj); vdo_tracks_size(tr_dyn_, &fc.n_dynamic_tracks, &np); }
  {                                                      // mVelocity = mCurrentFrame.mTcw * LastTwc   (Tracking.cc:703-709)
    float Twl[16];
    inv_rigid(Tcw_last_, Twl);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float a = 0; for (int k = 0; k < 4; ++k) a += Tcw[4 * i + k] * Twl[4 * k + j]; vel_[4 * i + j] = a; }
  }
  inv_rigid(vel_, cam_motion_);                          // (6.1) CameraMotionTmp = toInvMatrix(mVelocity)
  std::memcpy(Tcw_last_, Tcw, sizeof Tcw);
  std::memcpy(Tcw_out_, Tcw, sizeof Tcw);
  cur_ ^= 1; have_last_ = true; ++f_id_;
  gate_last_ = gate_cur_;
  // the camera stage of the NEXT frame (needs nothing of its images): under the object optimisations of this one
  if (cam_ahead_on_ && lm_cam_ && !cam_ahead_) {
    if (CameraStage() != 0) return -1;
    cam_ahead_ = true;
    mark(kEvCamStageDone);
  }
  if (host_inputs_ && depth_inout_ && !depth_metric_ && pending_ && !p_.defer_objects) {      // (the object LMs of the frame are in flight: the copy engine is free)
    VDO_TRY(vdo_frame_images_download_depth(img_[cur_ ^ 1], depth_inout_));
    depth_on_host_ = true;
  }
  if (pending_ && !p_.defer_objects) { if (FinishObjects(&fc) != 0) return -1; mark(kEvObjDone); }
  mark(kEvStepEnd);
  if (trace_slow) {
    const double tot = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_step0).count();
    if (tot > 2.5) {
      std::fprintf(stderr, "[slow step f=%d] %.2f ms:", f_id_ - 1, tot);
      for (int i = 0; i < 12; ++i) std::fprintf(stderr, " [%d] %.2f", i, ms_[i] - ms0[i]);
      std::fprintf(stderr, "\n");
    }
  }
  if (out) *out = fc;
  return 0;
}


// The object stage of a frame: consume the object LMs (K17), RenewFrameInfo of the objects (K14, K12), dynamic tracklets.
// fc: the object-related counts of THAT frame (n_object_tracked, n_dynamic_tracks) are written into it.
int FramePipeline::FinishObjects(FrameCounts* fcp, bool defer_tail) {
  if (!pending_) return 0;
  FrameCounts dummy{};
  FrameCounts& fc = fcp ? *fcp : dummy;
  auto t_prev = std::chrono::steady_clock::now();
  auto tick = [&](int slot) { const auto t = std::chrono::steady_clock::now(); ms_[slot] += std::chrono::duration<double, std::milli>(t - t_prev).count(); t_prev = t; };
  const int n_objects = n_objects_, n_tmp = n_tmp_, n_obj_problems = n_obj_problems_;
  vdo_flow2_batch* obj = obj_run_;
  vdo_frame_images* cur = img_obj_;
  const float* Tcw = Tcw_obj_;
  std::vector<int32_t>&olab = i_[2], &off = i_[3], &idx = i_[4], &osem = i_[5], &omod = i_[6];
  ObjSet& tmp = tmpb_[tmp_idx_obj_];
  ObjSet nobj;
  std::vector<int32_t> dyn_asso;
  float Twc[16];
  inv_rigid(Tcw, Twc);
  motions_.clear();                                      // (a frame without tracked objects reports none)
  {
    // ---- consume the object results, RenewFrameInfo (objects)                      Tracking.cc:2806-2995
    std::vector<float>&cur_ox = f_[11], &cur_oy = f_[12];
    cur_ox = obj_.cx; cur_oy = obj_.cy;
    std::vector<uint8_t> stat(std::max(n_objects, 1), 1);
    std::vector<int32_t>*p_off = &off, *p_idx = &idx;
    if (obj && obj == lm_obj_) {
      // vnObjInlierID = LM inliers; current keys of the inliers move to (last key + refined flow); H = Tcw^-1 * (Tcw H)  (Tracking.cc:932-933)
      const int NS = n_obj_problems;                       // slots of the batch when it was launched
      std::vector<vdo_flow2_result>& rs = lm_rs_;           // (members: their buffers keep their capacity from frame to frame)
      std::vector<std::vector<double>>& fo = lm_fo_; std::vector<std::vector<uint8_t>>& io = lm_io_;
      std::vector<double*>& fop = lm_fop_; std::vector<uint8_t*>& iop = lm_iop_;
      rs.resize(NS); fop.resize(NS); iop.resize(NS);
      if ((int)fo.size() < NS) { fo.resize(NS); io.resize(NS); }
      for (int a = 0; a < NS; ++a) {
        const size_t na = (a < n_objects && obj_stat_[a]) ? obj_subsets_[a].size() : 0;
        fo[a].resize(2 * na + 2); io[a].resize(na + 1);
        fop[a] = fo[a].data(); iop[a] = io[a].data();
      }
      VDO_TRY(vdo_flow2_batch_fetch(obj, rs.data(), fop.data(), iop.data()));
      mark(kEvObjLmFetched);
      static const bool trace_obj = std::getenv("VDO_PIPE_TRACE_OBJ") != nullptr;
      if (trace_obj) {
        std::fprintf(stderr, "[obj lm f=%d]", f_id_obj_);
        for (int a = 0; a < n_objects; ++a)
          if (obj_stat_[a]) std::fprintf(stderr, " sem %d n %zu its %d trials %d inl %d |", osem[a], obj_subsets_[a].size(), rs[a].iterations, rs[a].trials, rs[a].n_inliers);
        std::fprintf(stderr, "\n");
      }
      inl_off_.assign(1, 0); inl_idx_.clear();
      float Twc_c[16];
      inv_rigid(Tcw, Twc_c);
      for (int a = 0; a < n_objects; ++a) {
        stat[a] = obj_stat_[a];
        if (stat[a]) {
          const std::vector<int32_t>& sub = obj_subsets_[a];
          for (size_t j = 0; j < sub.size(); ++j) {
            if (!io[a][j]) { olab[sub[j]] = -1; continue; }                     // outliers of the object optimisation (Optimizer.cc:2960-2966)
            inl_idx_.push_back(sub[j]);
            cur_ox[sub[j]] = (float)((double)obj_.x[sub[j]] + fo[a][2 * j]); cur_oy[sub[j]] = (float)((double)obj_.y[sub[j]] + fo[a][2 * j + 1]);      // (float + double, one rounding: src/Optimizer.cc:2949-2950)
          }
          ObjectMotion om; om.mod_label = omod[a]; om.sem_label = osem[a]; om.n_inliers = rs[a].n_inliers;
          for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float acc = 0; for (int k = 0; k < 4; ++k) acc += Twc_c[4 * i + k] * (float)rs[a].T[4 * k + j]; om.H[4 * i + j] = acc; }
          motions_.push_back(om);
        } else {
          for (int q = off[a]; q < off[a + 1]; ++q) inl_idx_.push_back(idx[q]);   // untracked object: vnObjInlierID = its point set (Tracking.cc:872-886)
        }
        inl_off_.push_back((int32_t)inl_idx_.size());
      }
      if (inl_idx_.empty()) inl_idx_.push_back(0);
      p_off = &inl_off_; p_idx = &inl_idx_;
    } else if (obj) {
      std::vector<vdo_flow2_result> rs(std::max(n_obj_problems, 1));
      VDO_TRY(vdo_flow2_batch_fetch(obj, rs.data(), nullptr, nullptr));
    }
    tick(6);
    const int cap_o = (*p_off)[n_objects] + n_tmp + 8;
    nobj.x.resize(cap_o); nobj.y.resize(cap_o); nobj.cx.resize(cap_o); nobj.cy.resize(cap_o); nobj.fx.resize(cap_o); nobj.fy.resize(cap_o); nobj.d.resize(cap_o);
    nobj.sem.resize(cap_o); nobj.label.resize(cap_o); dyn_asso.resize(cap_o);
    int mo = 0;
    nobj.xyz.resize(3 * (size_t)std::max(cap_o, 1));
    // RenewFrameInfo (objects) + mvObj3DPoint in one pass, one synchronisation
    VDO_TRY(vdo_renew_object_world(cur, n_objects, p_off->data(), p_idx->data(), stat.data(), osem.data(), omod.data(), cur_ox.data(), cur_oy.data(), olab.data(),
                                   n_tmp, tmp.x.data(), tmp.y.data(), tmp.d.data(), tmp.sem.data(), tmp.fx.data(), tmp.fy.data(), tmp.cx.data(), tmp.cy.data(),
                                   p_.max_track_obj, cap_o, p_.K4, Twc, nobj.x.data(), nobj.y.data(), nobj.d.data(), nobj.sem.data(), nobj.fx.data(), nobj.fy.data(),
                                   nobj.cx.data(), nobj.cy.data(), dyn_asso.data(), nobj.label.data(), nobj.xyz.data(), &mo));
    for (auto* v : {&nobj.x, &nobj.y, &nobj.cx, &nobj.cy, &nobj.fx, &nobj.fy, &nobj.d}) v->resize(mo);
    nobj.sem.resize(mo); nobj.label.resize(mo); dyn_asso.resize(mo);
    nobj.xyz.resize(3 * (size_t)std::max(mo, 1));
    tick(7); mark(kEvObjRenewed);
    last_sem_pos_.assign(osem.begin(), osem.begin() + n_objects);
    last_mod_label_.assign(omod.begin(), omod.begin() + n_objects);
    last_obj_stat_.assign(stat.begin(), stat.begin() + n_objects);
    // mLastFrame.vObjMod of the next frame: the motion of every object of this frame, identity where it was not tracked (Tracking.cc:836,884,933)
    last_obj_mod_.assign(16 * (size_t)n_objects, 0.f);
    for (int a = 0, m = 0; a < n_objects; ++a) {
      float* Hd = last_obj_mod_.data() + 16 * (size_t)a;
      if (obj && obj == lm_obj_ && stat[a] && m < (int)motions_.size()) std::memcpy(Hd, motions_[m++].H, 64);
      else Hd[0] = Hd[5] = Hd[10] = Hd[15] = 1.f;
    }
  }
  fc.n_object_tracked = (int)nobj.x.size();
  obj_ = std::move(nobj);                                // from here on the next frame's object chain (K15, K11, K13 ...) can start
  // ... and its inputs that are THIS frame's - the object set just renewed - go to the device now, under the tail of this frame and the start of the next
  // (vdo_object_chain_prestage: asynchronous; the chain of the next Step runs on ctx_'s stream, behind this copy)
  if (!p_.defer_objects && !obj_.cx.empty())
    VDO_TRY(vdo_object_chain_prestage(ctx_, (int)obj_.cx.size(), obj_.sem.data(), obj_.cx.data(), obj_.cy.data(), obj_.x.data(), obj_.y.data(), obj_.d.data()));
  dyn_asso_tail_ = std::move(dyn_asso);
  tail_has_lm_ = obj && obj == lm_obj_;
  pending_ = false; tail_pending_ = true;
  return defer_tail ? 0 : FinishObjectsTail(&fc);
}

int FramePipeline::FinishObjectsTail(FrameCounts* fcp) {
  if (!tail_pending_) return 0;
  FrameCounts dummy{};
  FrameCounts& fc = fcp ? *fcp : dummy;
  auto t_prev = std::chrono::steady_clock::now();
  auto tick = [&](int slot) { const auto t = std::chrono::steady_clock::now(); ms_[slot] += std::chrono::duration<double, std::milli>(t - t_prev).count(); t_prev = t; };
  const ObjSet& nobj = obj_;
  // ---- tracklets (incremental GetDynamicTrackNew)                                  Tracking.cc:2309-2421
  VDO_TRY(vdo_tracks_add_frame(tr_dyn_, (int)nobj.x.size(), dyn_asso_tail_.data(), nobj.label.data()));
  tick(8);
  int64_t np = 0;
  vdo_tracks_size(tr_dyn_, &fc.n_dynamic_tracks, &np);
  if (keep_graph_) {                                     // "Save Graph Structure" (2), (6): object features, rigid motions + labels
    store_.dyn.append(nobj.x.size(), nobj.x.data(), nobj.y.data(), nobj.d.data(), nobj.xyz.data());
    std::vector<float> mots(cam_motion_, cam_motion_ + 16); std::vector<int32_t> labs(1, 0);
    if (tail_has_lm_)
      for (const ObjectMotion& om : motions_) { mots.insert(mots.end(), om.H, om.H + 16); labs.push_back(om.mod_label); }
    store_.add_motions((int)labs.size(), mots.data(), labs.data());
    // ---- partial batch optimisation on the last window (local optimisation)      Tracking.cc:1165-1183
    const int Wn = p_.window_size, Ov = p_.overlap_size;
    if (Wn > 0 && Wn > Ov && (f_id_obj_ - Ov + 1) % (Wn - Ov) == 0 && f_id_obj_ >= Wn - 1 && store_.sta.frames() == f_id_obj_ + 1) {
      const auto t_gt0 = std::chrono::steady_clock::now();
      if (GetTracks(&tl_sta_, nullptr, f_id_obj_ + 1 - Wn) != 0) return -1;      // (the window's first frame: PartialBatchOptimization reads nothing of a track that ended before it)
      static const bool trace_batch = std::getenv("VDO_BATCH_TRACE") != nullptr;
      if (trace_batch) std::fprintf(stderr, "[partial batch] static tracklets flattened in %.2f ms (%zu tracks, %zu entries)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_gt0).count(), tl_sta_.off.size() - 1, tl_sta_.frame.size());
      try { Optimizer::PartialBatchOptimization(store_, tl_sta_, p_.K4, Wn); }
      catch (const std::exception& e) { std::fprintf(stderr, "FramePipeline: %s\n", e.what()); return -1; }
      ++n_partial_batches_;
    }
  }
  tail_pending_ = false;
  return 0;
}

int FramePipeline::StepHost(const uint8_t* gray, const float* depth, const float* flow, const int32_t* mask, bool depth_is_metric, FrameCounts* out, float* depth_inout) {
  host_inputs_ = true; depth_metric_ = depth_is_metric; depth_inout_ = depth_inout;
  const int rc = Step(gray, depth, flow, mask, nullptr, nullptr, 0, 0, out);
  host_inputs_ = false; depth_metric_ = false; depth_inout_ = nullptr;
  return rc;
}

int FramePipeline::DownloadMask(int32_t* mask_out) {
  VDO_TRY(vdo_frame_images_download_mask(img_[cur_ ^ 1], mask_out));      // (cur_ was flipped at the end of Step)
  return 0;
}

int FramePipeline::DownloadDepth(float* depth_out) {
  VDO_TRY(vdo_frame_images_download_depth(img_[cur_ ^ 1], depth_out));
  return 0;
}

int FramePipeline::FinalizeMap() {
  if (!map_) return -1;
  if (pending_ && FinishObjects(nullptr) != 0) return -1;
  return SyncMap();
}

// the reference-format Map (vector<vector<cv::Mat>> ...) rebuilt from the store + mpMap->TrackletSta = GetStaticTrack();
// mpMap->TrackletDyn = GetDynamicTrackNew()   (Tracking.cc:1065-1071)
int FramePipeline::SyncMap() {
  if (!map_) return -1;
  if (pending_ && FinishObjects(nullptr) != 0) return -1;
  if (GetTracks(&tl_sta_, &tl_dyn_) != 0) return -1;
  StoreToMap(store_, tl_sta_, tl_dyn_, *map_);
  return 0;
}

int FramePipeline::FullBatchOptimization() {
  if (pending_ && FinishObjects(nullptr) != 0) return -1;
  if (GetTracks(&tl_sta_, &tl_dyn_) != 0) return -1;
  try { Optimizer::FullBatchOptimization(store_, tl_sta_, tl_dyn_, p_.K4); }
  catch (const std::exception& e) { std::fprintf(stderr, "FramePipeline: %s\n", e.what()); return -1; }
  return 0;
}

// the tracklets kept incrementally (vdo_tracks_*) as flat lists
int FramePipeline::GetTracks(TrackList* sta, TrackList* dyn, int first_frame) {
  for (int which = 0; which < 2; ++which) {
    TrackList* L = which ? dyn : sta;
    if (!L) continue;
    vdo_tracks* t = which ? tr_dyn_ : tr_sta_;
    int nt = 0; int64_t np = 0;
    VDO_TRY(vdo_tracks_size(t, &nt, &np));
    L->off.assign((size_t)nt + 1, 0); L->frame.resize((size_t)std::max<int64_t>(np, 1)); L->feat.resize((size_t)std::max<int64_t>(np, 1)); L->obj.assign((size_t)std::max(nt, 1), 0);
    if (first_frame >= 0) {
      int64_t np2 = 0;
      VDO_TRY(vdo_tracks_get_since(t, first_frame, &nt, &np2, L->off.data(), L->frame.data(), L->feat.data(), which ? L->obj.data() : nullptr));
      np = np2;
      L->off.resize((size_t)nt + 1);
    } else VDO_TRY(vdo_tracks_get(t, L->off.data(), L->frame.data(), L->feat.data(), which ? L->obj.data() : nullptr));
    L->frame.resize((size_t)np); L->feat.resize((size_t)np); L->obj.resize((size_t)nt);
  }
  return 0;
}

}  // namespace VDO_SLAM

// ---- flat hooks for the Python bench / tests ------------------------------------------------------
using VDO_SLAM::FramePipeline;
using VDO_SLAM::FrameCounts;
using VDO_SLAM::PipelineParams;

extern "C" {
FramePipeline* host_pipeline_create(vdo_ctx* ctx, vdo_ctx* ctx_lm, const PipelineParams* p, vdo_ctx* ctx_obj, vdo_ctx* ctx_worker, vdo_ctx* ctx_orb) {
  FramePipeline* fp = new FramePipeline(ctx, ctx_lm, *p, ctx_obj, ctx_worker, ctx_orb);
  if (!fp->ok()) { delete fp; return nullptr; }
  return fp;
}
void host_pipeline_destroy(FramePipeline* fp) { delete fp; }
// accumulated wall ms per section since creation: [0] K1+K15+K11, [1] ORB, [2] K9+K10, [3] wait camera LM + fetch,
// [4] K13 + DynObjTracking, [5] RenewFrameInfo static + K12, [6] wait object LMs + fetch, [7] RenewFrameInfo objects + K12, [8] tracklets,
// [9] object RANSAC initialisers ([0] includes the camera one)
// [10] K15 + K11 (objects)
void host_pipeline_timing(FramePipeline* fp, double* ms11) { for (int i = 0; i < 11; ++i) ms11[i] = fp->ms_[i]; }
// VDO_PIPE_EVENTS=1: mean time after the start of its Step at which each milestone (FramePipeline::kEv*) was reached, -1 = never; reset != 0 clears
int host_pipeline_events(FramePipeline* fp, double* ms, int reset) {
  for (int i = 0; i < FramePipeline::kEvCount; ++i) { ms[i] = fp->ev_n_[i] ? fp->ev_ms_[i] / fp->ev_n_[i] : -1.0; if (reset) { fp->ev_ms_[i] = 0; fp->ev_n_[i] = 0; } }
  return FramePipeline::kEvCount;
}
int host_pipeline_flush(FramePipeline* fp, FrameCounts* out) { return fp->Flush(out); }
// The tracklets as Track() keeps them (GetStaticTrack / GetDynamicTrackNew, src/Tracking.cc:2201-2421), which = 0 static, 1 dynamic:
// with off == NULL returns the sizes (n_tracks, n_pairs); else fills off [n_tracks + 1], frame / feat [n_pairs], obj [n_tracks] (dynamic only).
int host_pipeline_tracks(FramePipeline* fp, int which, int64_t* sizes2, int32_t* off, int32_t* frame, int32_t* feat, int32_t* obj) {
  VDO_SLAM::TrackList L;
  if (fp->Flush(nullptr) != 0) return -1;          // a pending (deferred) object stage owns the dynamic tracklets of its frame
  if (fp->GetTracks(which ? nullptr : &L, which ? &L : nullptr) != 0) return -1;
  if (sizes2) { sizes2[0] = L.size(); sizes2[1] = (int64_t)L.frame.size(); }
  if (off) {
    std::copy(L.off.begin(), L.off.end(), off);
    std::copy(L.frame.begin(), L.frame.end(), frame);
    std::copy(L.feat.begin(), L.feat.end(), feat);
    if (which && obj) std::copy(L.obj.begin(), L.obj.end(), obj);
  }
  return 0;
}

// ---- Map: Track() -> Map -> Optimizer::FullBatchOptimization (tests / demos)
VDO_SLAM::Map* host_pipeline_attach_map(FramePipeline* fp) { VDO_SLAM::Map* m = new VDO_SLAM::Map(); fp->AttachMap(m); return m; }
void host_map_destroy(VDO_SLAM::Map* m) { delete m; }
int host_pipeline_finalize_map(FramePipeline* fp) { return fp->FinalizeMap(); }
void host_pipeline_keep_graph(FramePipeline* fp) { fp->KeepGraph(); }
// Optimizer::FullBatchOptimization straight from the store (no Map); the attached Map, if any, is brought up to date afterwards
int host_pipeline_full_batch(FramePipeline* fp, vdo_lm_stats* st) {
  if (fp->FullBatchOptimization() != 0) return -1;
  if (st) *st = VDO_SLAM::Optimizer::last_batch_stats;
  return 0;
}
// dims: [0] frames, [1] static features, [2] dynamic features, [3] transitions, [4] motions
void host_pipeline_store_dims(const FramePipeline* fp, int64_t* dims) {
  const VDO_SLAM::GraphStore& S = fp->store();
  dims[0] = S.frames(); dims[1] = S.sta.off.back(); dims[2] = S.dyn.off.back(); dims[3] = S.transitions(); dims[4] = S.rm_off.back();
}
// refined (refined != 0) or unrefined camera poses T_wc [frames][16] of the store
void host_pipeline_store_poses(const FramePipeline* fp, int refined, float* out) {
  const VDO_SLAM::GraphStore& S = fp->store();
  std::memcpy(out, (refined ? S.cam_rf : S.cam).data(), sizeof(float) * S.cam.size());
}
int host_pipeline_partial_batches(FramePipeline* fp) { return fp->n_partial_batches_; }
// dims: [0] frames, [1] static features, [2] dynamic features, [3] static tracklets, [4] their pairs, [5] dynamic tracklets, [6] their pairs, [7] rigid motions
void host_map_dims(const VDO_SLAM::Map* m, int* dims) {
  for (int i = 0; i < 8; ++i) dims[i] = 0;
  dims[0] = (int)m->vpFeatSta.size();
  for (const auto& f : m->vpFeatSta) dims[1] += (int)f.size();
  for (const auto& f : m->vpFeatDyn) dims[2] += (int)f.size();
  dims[3] = (int)m->TrackletSta.size(); for (const auto& t : m->TrackletSta) dims[4] += (int)t.size();
  dims[5] = (int)m->TrackletDyn.size(); for (const auto& t : m->TrackletDyn) dims[6] += (int)t.size();
  for (const auto& r : m->vmRigidMotion) dims[7] += (int)r.size();
}
// flat copy of the Map (arrays sized by host_map_dims); refined != 0: the *_RF poses / motions
void host_map_export(const VDO_SLAM::Map* m, int refined, float* cam_pose, int* sta_cnt, float* sta_uv, float* sta_d, float* sta_xw, int* tr_sta_len, int* tr_sta_pairs,
                     int* dyn_cnt, float* dyn_uv, float* dyn_d, float* dyn_xw, int* tr_dyn_len, int* tr_dyn_pairs, int* obj_of_dyn, int* rm_cnt, float* rm, int* rm_label) {
  const int F = (int)m->vpFeatSta.size();
  size_t so = 0, dof = 0, ro = 0, po = 0;
  for (int i = 0; i < F; ++i) {
    std::memcpy(cam_pose + 16 * i, (refined ? m->vmCameraPose_RF : m->vmCameraPose)[i].data, 64);
    sta_cnt[i] = (int)m->vpFeatSta[i].size();
    for (size_t j = 0; j < m->vpFeatSta[i].size(); ++j, ++so) {
      sta_uv[2 * so] = m->vpFeatSta[i][j].pt.x; sta_uv[2 * so + 1] = m->vpFeatSta[i][j].pt.y; sta_d[so] = m->vfDepSta[i][j];
      std::memcpy(sta_xw + 3 * so, m->vp3DPointSta[i][j].data, 12);
    }
    dyn_cnt[i] = (int)m->vpFeatDyn[i].size();
    for (size_t j = 0; j < m->vpFeatDyn[i].size(); ++j, ++dof) {
      dyn_uv[2 * dof] = m->vpFeatDyn[i][j].pt.x; dyn_uv[2 * dof + 1] = m->vpFeatDyn[i][j].pt.y; dyn_d[dof] = m->vfDepDyn[i][j];
      std::memcpy(dyn_xw + 3 * dof, m->vp3DPointDyn[i][j].data, 12);
    }
    if (i < (int)m->vmRigidMotion.size()) {
      const auto& R = refined ? m->vmRigidMotion_RF[i] : m->vmRigidMotion[i];
      rm_cnt[i] = (int)R.size();
      for (size_t j = 0; j < R.size(); ++j, ++ro) { std::memcpy(rm + 16 * ro, R[j].data, 64); rm_label[ro] = m->vnRMLabel[i][j]; }
    }
  }
  for (size_t t = 0; t < m->TrackletSta.size(); ++t) {
    tr_sta_len[t] = (int)m->TrackletSta[t].size();
    for (const auto& pr : m->TrackletSta[t]) { tr_sta_pairs[2 * po] = pr.first; tr_sta_pairs[2 * po + 1] = pr.second; ++po; }
  }
  po = 0;
  for (size_t t = 0; t < m->TrackletDyn.size(); ++t) {
    tr_dyn_len[t] = (int)m->TrackletDyn[t].size(); obj_of_dyn[t] = m->nObjID[t];
    for (const auto& pr : m->TrackletDyn[t]) { tr_dyn_pairs[2 * po] = pr.first; tr_dyn_pairs[2 * po + 1] = pr.second; ++po; }
  }
}
// Optimizer::FullBatchOptimization on the Map (GPU solve): refined poses / motions land in vmCameraPose_RF / vmRigidMotion_RF
int host_map_full_batch(VDO_SLAM::Map* m, const float* K9, vdo_lm_stats* st) {
  cv::Mat K(3, 3, cv::CV_32F);
  std::memcpy(K.data, K9, 36);
  try { VDO_SLAM::Optimizer::FullBatchOptimization(m, K); }
  catch (const std::exception& e) { std::fprintf(stderr, "host_map_full_batch: %s\n", e.what()); return -1; }
  if (st) *st = VDO_SLAM::Optimizer::last_batch_stats;
  return 0;
}
void host_pipeline_pose(FramePipeline* fp, float* Tcw16) { std::memcpy(Tcw16, fp->Tcw_out_, 64); }
int host_pipeline_motions(FramePipeline* fp, int cap, int* mod_label, int* sem_label, int* n_inliers, float* H16) {
  const int n = std::min(cap, (int)fp->motions_.size());
  for (int a = 0; a < n; ++a) { mod_label[a] = fp->motions_[a].mod_label; sem_label[a] = fp->motions_[a].sem_label; n_inliers[a] = fp->motions_[a].n_inliers; std::memcpy(H16 + 16 * a, fp->motions_[a].H, 64); }
  return (int)fp->motions_.size();
}
int host_pipeline_step(FramePipeline* fp, const uint8_t* d_gray, const float* d_depth_raw, const float* d_flow, const int32_t* d_mask,
                       vdo_flow2_batch* cam, vdo_flow2_batch* obj, int n_cam_pts, int n_obj_problems, FrameCounts* out) {
  return fp->Step(d_gray, d_depth_raw, d_flow, d_mask, cam, obj, n_cam_pts, n_obj_problems, out);
}
}
