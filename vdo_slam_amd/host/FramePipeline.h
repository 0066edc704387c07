// FramePipeline — the per-frame sequence of Tracking::GrabImageRGBD + Tracking::Track (reference
// src/Tracking.cc:164-314, 646-1276) over the C-ABI, with frame-to-frame state, minus what is not on the
// hot path (ground-truth bookkeeping, metrics, drawing).
//
//   depth preprocess (K1) -> static propagation (K11) -> GetInitModelCam (RANSAC-P3P | motion model) -> camera LM (K16, stream 2)
//        || ORB (K3-K7)
//   [deferred mode: the object stage of the previous frame ends here]
//   -> UpdateMask (K15) + object propagation (K11) -> scene flow (K13) + DynObjTracking -> GetInitModelObj (RANSAC per object)
//   -> object LMs (K17, one launch, stream 3) || Frame filters (K9, K10) + RenewFrameInfo static (K14, K12) + static tracklets
//   -> object stage: RenewFrameInfo objects (K14, K12) -> dynamic tracklets      (FinishObjects)
//
// The LM kernels run on their own contexts/streams so that they overlap the front-end work that does not depend on them
// (the ORB keypoints are first needed by RenewFrameInfo at the end of the frame, src/Tracking.cc:1168; nothing of the next
// frame's camera stage needs this frame's object results).
#pragma once
#include <chrono>
#include <cstdint>
#include <functional>
#include <memory>
#include <atomic>
#include <cstdlib>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "GraphStore.h"
#include "Map.h"

namespace VDO_SLAM {

struct PipelineParams {
  int width, height;
  float K4[4];                       // fx, fy, cx, cy
  float bf, depth_map_factor;        // Camera.bf, DepthMapFactor
  float th_depth_bg, th_depth_obj;   // ThDepthBG, ThDepthOBJ
  int max_track_bg, max_track_obj;   // MaxTrackPointBG, MaxTrackPointOBJ
  float sf_mg_thres, sf_ds_thres;    // SFMgThres, SFDsThres
  int n_features, n_levels, ini_th, min_th; float scale_factor;   // ORBextractor.*
  int build_lm;                      // 1: build the frame's pose problems from the chained correspondences (full Track()); 0: caller supplies them
  int defer_objects;                 // 1: Step() returns with the object LMs of the frame in flight; their results (object motions, renewed object
                                     //    set, dynamic tracklets) are consumed inside the next Step() - after that frame's camera stage and ORB
                                     //    front-end, which do not depend on them - or by Flush().  Same results, one frame of latency for the objects.
  int use_sample_feature;            // UseSampleFeature (omd.yaml): Frame::SampleKeyPoints (3000 random grid positions) instead of ORB, the static
  int sample_seed;                   //    filter's sampled branch, top-up from the filtered samples; cv::RNG seed of frame f = sample_seed + f
                                     //    (the reference seeds with time(NULL))
  int pnp_refit;                     // 1: GetInitModelCam/Obj receive what cv::solvePnPRansac returns since OpenCV 3.3 - the winning P3P model re-estimated on
                                     //    its inliers by EPnP (vdo_pnp_problem.refit); 0: the raw P3P hypothesis
  int window_size, overlap_size;     // WINDOW_SIZE / OVERLAP_SIZE: with a Map attached, Optimizer::PartialBatchOptimization runs on the last
                                     //    window_size frames whenever (f_id-overlap+1) % (window-overlap) == 0 && f_id >= window-1
                                     //    (src/Tracking.cc:1169-1183); 0 = never
};

struct FrameCounts { int n_orb, n_static_new, n_object_samples, n_static_tracked, n_object_tracked, n_objects, n_recovered_masks, n_static_tracks, n_dynamic_tracks,
                         n_ransac_cam, n_motion_model_cam, n_ransac_obj, n_cam_inliers, cam_lm_iterations,
                         n_mm_inliers_obj,        // sum over the re-tracked objects of the frame of their motion-model inliers (Tracking.cc:1784-1797)
                         n_motion_model_obj; };   // objects of the frame whose LM was seeded by the motion model (Tracking.cc:1803-1825: RANSAC did not have MORE inliers)

class FramePipeline {
 public:
  // ctx: front-end / tracking kernels; ctx_lm: camera pose problems; ctx_obj: object pose problems (NULL: ctx_lm) - three HIP streams.
  // ctx_worker (optional): a second HOST thread runs the stages that are independent of what the main thread is doing - the object
  // stage of the previous frame (deferred mode) next to this frame's camera stage + ORB, and K9/K10 + RenewFrameInfo (static) next
  // to the scene-flow / object-tracking / object-RANSAC chain - with this context (its own stream and scratch arena).
  // ctx_orb (optional): the ORB extractor gets this context (a stream of its own).  With ctx_worker as well, ORB (K3-K7: only the
  // grey image goes in, only RenewFrameInfo / the static filter read what comes out) runs on a THIRD host thread from the start of
  // Step() to the static stage; without ctx_worker its device stage is queued at the very start of Step()
  // (vdo_orb_extract_begin) and the quadtrees follow where ORB used to be (vdo_orb_extract_end).
  FramePipeline(vdo_ctx* ctx, vdo_ctx* ctx_lm, const PipelineParams& p, vdo_ctx* ctx_obj = nullptr, vdo_ctx* ctx_worker = nullptr, vdo_ctx* ctx_orb = nullptr);
  ~FramePipeline();
  // One frame.  d_* are DEVICE pointers of the raw inputs (gray u8, disparity*factor f32, flow 2xf32, mask i32).
  // cam / obj: the frame's pose problems (already resident); their results are fetched like Track() consumes them.
  int Step(const uint8_t* d_gray, const float* d_depth_raw, const float* d_flow, const int32_t* d_mask,
           vdo_flow2_batch* cam, vdo_flow2_batch* obj, int n_cam_pts, int n_obj_problems, FrameCounts* out);
  // The same with HOST pointers (the System::TrackRGBD shell): images are uploaded first; depth_is_metric: K1 was already applied.
  // depth_inout (optional, the caller's raw depth map): receives the converted map (metres; the reference converts imD in place) inside the
  // Step when that read-back can run under the frame's object optimisations - DepthConvertedOnHost() tells; otherwise the caller fetches it
  // with DownloadDepth.
  int StepHost(const uint8_t* gray, const float* depth, const float* flow, const int32_t* mask, bool depth_is_metric, FrameCounts* out, float* depth_inout = nullptr);
  bool DepthConvertedOnHost() const { return depth_on_host_; }
  // Objects the caller has ground truth for in the frame about to be given to Step (label = mask id).  The reference only
  // tracks an object whose label has a ground-truth row in BOTH the last and the current frame (src/Tracking.cc:791-841:
  // otherwise bObjStat = false and the object keeps its point set untouched).  Without a call every label is allowed.
  void SetObjectGate(const int* labels, int n) { gate_on_ = true; gate_cur_.assign(labels, labels + n); }
  // copy of the (possibly UpdateMask-modified) instance mask of the last frame given to Step
  int DownloadMask(int32_t* mask_out);
  // the depth map of the last frame given to Step after K1 (metres): what GrabImageRGBD leaves in the caller's imD
  int DownloadDepth(float* depth_out);
  // "Save Graph Structure" of Track() (src/Tracking.cc:1031-1159, Initialization :1238-1246): once KeepGraph() / AttachMap()
  // was called every frame appends its static / dynamic features, depths, 3-D points, camera pose and rigid motions (+ labels)
  // to a flat GraphStore (plain array appends: a few microseconds per frame) - what Optimizer::Full/PartialBatchOptimization
  // build their graph from directly.  A Map in the reference's format (one cv::Mat per 3-D point) is materialised from the
  // store on request only: SyncMap(), which FinalizeMap() calls after the last frame (Flush() first in deferred mode).
  void KeepGraph() { keep_graph_ = true; }
  void AttachMap(Map* m) { map_ = m; keep_graph_ = true; }
  int SyncMap();                       // store + tracklets -> the attached Map (complete rebuild)
  int FinalizeMap();
  int FullBatchOptimization();         // Optimizer::FullBatchOptimization on the store (tracklets read from the incremental builders)
  const GraphStore& store() const { return store_; }
  int n_partial_batches_ = 0;          // PartialBatchOptimization runs so far
  // Ends a pending object stage (deferred mode); fc (optional) receives its n_object_tracked / n_dynamic_tracks.
  int Flush(FrameCounts* fc = nullptr) { return FinishObjects(fc); }
  // the tracklets kept incrementally (GetStaticTrack / GetDynamicTrackNew) as flat lists; either pointer may be NULL
  int GetTracks(TrackList* sta, TrackList* dyn, int first_frame = -1);      // first_frame >= 0: only the tracks still observed in that frame or later (vdo_tracks_get_since)
  bool ok() const { return ok_; }
  // deferred object stage on / off between frames (a pending stage is consumed by the next Step or by Flush either way)
  void SetDeferObjects(bool on) { p_.defer_objects = on ? 1 : 0; }
  const PipelineParams& params() const { return p_; }
  struct ObjectMotion { int mod_label, sem_label, n_inliers; float H[16]; };   // H: world-frame motion of the object from the last to this frame
  std::vector<ObjectMotion> motions_;   // objects tracked in the last Step (build_lm mode)
  float Tcw_out_[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double ms_[12] = {0};              // accumulated wall time per section (see host_pipeline_timing)
  // VDO_PIPE_EVENTS=1 (debug / bench): when, relative to the start of its Step, each milestone of a frame was reached - summed over the
  // Steps since the last reset (host_pipeline_events).  Slots: kEv* below.
  enum { kEvInputs = 0, kEvCamFetched, kEvObjChain, kEvDynObj, kEvObjLmBuilt, kEvObjLmLaunched, kEvOrbDevice, kEvOrbDone, kEvFilters, kEvStaticDone,
         kEvStaticJoined, kEvCamStageDone, kEvObjLmFetched, kEvObjRenewed, kEvObjDone, kEvStepEnd, kEvCount };
  double ev_ms_[kEvCount] = {0}; long ev_n_[kEvCount] = {0};
  bool ev_on_ = false;
  std::chrono::steady_clock::time_point ev_t0_;
  void mark(int k) { if (ev_on_) { ev_ms_[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ev_t0_).count(); ++ev_n_[k]; } }

  // ---- the per-frame state the reference keeps in Tracking::mCurrentFrame / mLastFrame after Track() (RenewFrameInfo,
  // src/Tracking.cc:2780-2812, 2984-2991; per-object vectors :836-933), as views of the pipeline's own flat arrays.  Valid between
  // Steps, after Flush() in deferred mode; Tracking::SyncFrameState() turns them into the reference's containers.
  struct ObjSet { std::vector<float> x, y, cx, cy, fx, fy, d, xyz; std::vector<int32_t> sem, label; };   // mvObjKeys, mvObjCorres, mvObjFlowNext, mvObjDepth, mvObj3DPoint, vSemObjLabel, vObjLabel
  struct StaSet { std::vector<float> x, y, cx, cy, fx, fy, d, xyz; };                                      // mvStatKeysTmp, mvCorres, mvFlowNext, mvStatDepthTmp, mvStat3DPointTmp
  const StaSet& StaticSet() const { return sta_; }
  const ObjSet& ObjectSet() const { return obj_; }
  const ObjSet& ObjectSamples() const { return tmpb_[tmp_idx_obj_]; }        // mvTmpObjKeys / Corres / FlowNext / Depth / SemObjLabel of the last frame (first n_object_samples entries)
  int NumObjectSamples() const { return n_tmp_; }
  const std::vector<int32_t>& ObjSemPosition() const { return last_sem_pos_; }   // nSemPosition
  const std::vector<int32_t>& ObjModLabel() const { return last_mod_label_; }    // nModLabel
  const std::vector<uint8_t>& ObjStat() const { return last_obj_stat_; }          // bObjStat
  const std::vector<float>& ObjMod() const { return last_obj_mod_; }              // vObjMod: [n][16], world-frame motion, identity where not tracked
  int MaxId() const { return max_id_; }                                           // max_id

 private:
  int CameraStage();                        // GetInitModelCam + launch of the camera optimisation for the frame after the last one
  int FinishObjects(FrameCounts* fc, bool defer_tail = false);
  int FinishObjectsTail(FrameCounts* fc);   // dynamic tracklets, Map, windowed optimisation: nothing the next frame's object chain waits for
  bool tail_pending_ = false, tail_has_lm_ = false;
  std::vector<int32_t> dyn_asso_tail_;
  GraphStore store_;
  TrackList tl_sta_, tl_dyn_;
  bool keep_graph_ = false;
  bool host_inputs_ = false, depth_metric_ = false, gate_on_ = false;
  float* depth_inout_ = nullptr;      // StepHost: the caller's depth map to convert in place
  bool depth_on_host_ = false;
  std::vector<int> gate_cur_, gate_last_;
  Map* map_ = nullptr;
  int f_id_obj_ = 0;                  // frame id of the pending object stage
  float cam_motion_[16];              // Converter::toInvMatrix(mVelocity) of the frame whose object stage is pending
  class Worker;
  std::unique_ptr<Worker> worker_;
  std::unique_ptr<Worker> worker_orb_;    // ORB of the current frame, then the tail of the last frame's object stage (ctx_orb + ctx_worker)
  std::atomic<bool> tail_done_{true};
  bool orb_split_ = false;
  vdo_ctx *ctx_, *ctx_lm_, *ctx_obj_, *ctx_w_, *ctx_orb_ = nullptr;
  // object stage handed from Step() to FinishObjects()
  bool pending_ = false;
  int n_objects_ = 0, n_obj_problems_ = 0, n_tmp_ = 0;
  vdo_flow2_batch* obj_run_ = nullptr;
  vdo_frame_images* img_obj_ = nullptr;
  float Tcw_obj_[16];
  PipelineParams p_;
  vdo_orb* orb_ = nullptr;
  vdo_frame_images* img_[2] = {nullptr, nullptr};
  vdo_tracks *tr_sta_ = nullptr, *tr_dyn_ = nullptr;
  int cur_ = 0, f_id_ = 0;
  bool ok_ = false, have_last_ = false;
  int32_t max_id_ = 1;
  StaSet sta_;                        // last frame: static keys + their correspondences in the next image
  ObjSet tmpb_[2];                    // K10 output of a frame, by image-set index (the object stage of frame k-1 reads [k-1] while frame k fills [k])
  int tmp_idx_obj_ = 0;               // which of the two the pending object stage reads
  std::atomic<int> orb_ready_{0};     // +-(frame id + 1): keypoints and speculative K9 / K10 of that frame are there (negative: failed)
  std::atomic<int> objects_done_{0};  // +-(frame id + 1): the object stage the ORB thread's tail waits for is over (negative: no tail)
  std::atomic<int> mask_final_{0};    // +-(frame id + 1): UpdateMask of that frame is through (K10 on the ORB thread waits for it; negative: skip)
  ObjSet obj_;                        // last frame: object keys, correspondences, depth, semantic + motion labels
  std::vector<int32_t> last_sem_pos_, last_mod_label_; std::vector<uint8_t> last_obj_stat_;
  std::vector<float> last_obj_mod_;   // mLastFrame.vObjMod: 16 floats per object of the last frame (identity for an object that was not tracked, Tracking.cc:836,884)
  float Tcw_last_[16], vel_[16];      // last pose, mVelocity
  // scratch reused across frames
  std::vector<float> kx_, ky_, kr_, ka_, ks_; std::vector<int32_t> ko_;
  std::vector<float> f_[16]; std::vector<int32_t> i_[8];
  std::vector<double> flow_out_; std::vector<uint8_t> inl_out_, inl_ransac_;
  std::vector<double> d_[5];
  // build_lm mode
  // object-LM slots: one per accepted object of a frame, each with room for obj_cap_ correspondences.  The reference has no
  // limit on either (src/Tracking.cc:785-1001 loops over all objects, an untracked label carries all its step-4 samples), so
  // the batch is re-reserved (grown, never shrunk) when a frame needs more - rare, costs one allocation.
  int obj_slots_ = 8, obj_cap_ = 6000;
  int ReserveObjectSlots(int n_objects, int max_points);
  struct ObjBuf { std::vector<double> ob, fl, dp; };
  vdo_flow2_batch *lm_cam_ = nullptr, *lm_obj_ = nullptr;
  std::vector<int32_t> cam_subset_, inl_off_, inl_idx_;
  std::vector<std::vector<int32_t>> obj_subsets_;
  std::vector<vdo_flow2_result> lm_rs_; std::vector<std::vector<double>> lm_fo_; std::vector<std::vector<uint8_t>> lm_io_;   // fetch buffers of the object LMs
  std::vector<double*> lm_fop_; std::vector<uint8_t*> lm_iop_;
  std::vector<uint8_t> inl_mm_, obj_stat_, obj_use_mm_;   // obj_use_mm_[a]: the motion model seeds object a's LM
  std::vector<float> obj_mm_;                             // MotionModel of the frame's objects (16 floats each)
  std::vector<ObjBuf> obj_buf_;
  float Tcw_init_[16];
  // camera stage (CameraStage): launched ahead of its frame or at the start of its Step
  bool cam_ahead_ = false;
  bool cam_ahead_on_ = std::getenv("VDO_PIPE_NO_CAM_AHEAD") == nullptr;      // (read when the pipeline is built)
  vdo_flow2_batch* cam_run_ = nullptr;
  int cam_n_pts_ = 0, cam_n_ransac_ = 0, cam_n_mm_ = 0;
  std::vector<double> dcam_[5];
  std::vector<uint8_t> inl_ransac_cam_, inl_mm_cam_;
};

}  // namespace VDO_SLAM
