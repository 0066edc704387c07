/* vdo_slam_hip.h — C-ABI of libvdo_hip.so: the MI355X (gfx950) hot path of VDO-SLAM.
 *
 * The reference (halajun/VDO_SLAM) has no FFI/plugin boundary: its hot path sits behind
 * C++ class signatures in libObjSLAM.so.  This header is the one boundary
 * the new build introduces: host C++ that keeps the reference's class API
 * (vdo_slam_amd/host/: ORBextractor, Frame, Tracking, Optimizer, System) calls these
 * entry points; nothing else does.  Conventions:
 *   - every function returns int: 0 = ok, <0 = vdo_status error; never throws, never aborts
 *     (vdo_last_error() gives a thread-local message);
 *   - plain pointers + explicit sizes, no C++/torch types;
 *   - all device work is stream-ordered on the context's hipStream_t (which may be an
 *     externally owned stream, e.g. torch's current stream), so callers can bracket calls
 *     with their own events;
 *   - fp64 for everything the reference computes in g2o (number_t = double,
 *     dependencies/g2o/config.h:14-29), fp32/u8/i32 for image-side data (cv::Mat types).
 * There is no CPU fallback: without a HIP device every compute entry point returns
 * VDO_ERR_NO_DEVICE.
 */
#ifndef VDO_SLAM_HIP_H_
#define VDO_SLAM_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vdo_status {
  VDO_OK = 0,
  VDO_ERR_INVALID = -1,     /* bad argument / malformed graph                         */
  VDO_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure                    */
  VDO_ERR_OOM = -3,
  VDO_ERR_UNSUPPORTED = -4, /* structurally valid input outside the supported envelope */
  VDO_ERR_INTERNAL = -5
} vdo_status;

int vdo_version(void);               /* 100*major + minor */
const char* vdo_last_error(void);    /* thread-local, never NULL */

/* ---- context ---------------------------------------------------------------------------*/
typedef struct vdo_ctx vdo_ctx;
/* stream == NULL: the context creates (and owns) a non-blocking stream on `device`. */
int vdo_ctx_create(int device, void* hip_stream, vdo_ctx** out);
int vdo_ctx_destroy(vdo_ctx* ctx);
int vdo_ctx_synchronize(vdo_ctx* ctx);
int vdo_ctx_stream(vdo_ctx* ctx, void** hip_stream_out);   /* the hipStream_t all calls are ordered on */
/* Context with an owned stream restricted to the compute units [cu_first, cu_first+cu_count) (invert != 0:
 * to every CU except those).  The per-frame LM kernels are one latency-bound workgroup per problem: giving
 * them CUs of their own keeps the image kernels of the same frame (which run concurrently) off their SIMDs. */
int vdo_ctx_create_cu_mask(int device, int cu_first, int cu_count, int invert, vdo_ctx** out);

/* ---- batch dynamic bundle adjustment ------------------------------------------------------
 * Replaces the g2o calls inside Optimizer::FullBatchOptimization (reference
 * src/Optimizer.cc:1232-2175: graph at :1355-1766, optimize(300) at :1935) and
 * Optimizer::PartialBatchOptimization (:42-1230, optimize(100) at :807): i.e.
 * SparseOptimizer::initializeOptimization/optimize (g2o/core/sparse_optimizer.cpp:205-267,
 * 354-443), OptimizationAlgorithmLevenberg::solve (g2o/core/optimization_algorithm_levenberg.cpp:61-164),
 * BlockSolver::buildSystem (g2o/core/block_solver.hpp:502-560) and the linear solve
 * (g2o/solvers/linear_solver_csparse.h:108-144).
 *
 * Graph in SoA form.  Poses = g2o::VertexSE3 estimates (Isometry3) as 12 doubles:
 * R row-major (9) then t (3).  For bandwidth the binary edges should be sorted by eb_pose
 * and the ternary edges by et_pose (the reference inserts them frame by frame, which already
 * is camera-major); correctness does not depend on the order.
 * Structural requirement (checked): a point is p1 of at most one and p2 of at most one
 * ternary edge (dynamic tracks are chains, src/Optimizer.cc:1704-1741). */
typedef struct vdo_ba_graph {
  int32_t n_pose, n_point, n_eb, n_et, n_ep, n_prior;
  const double* pose;      /* [n_pose][12]  cameras (T_wc) and object motions (H)        */
  const double* point;     /* [n_point][3]  VertexPointXYZ (world)                       */
  /* EdgeSE3PointXYZ (g2o/types/edge_se3_pointxyz.cpp:99-140), information = w * I3 */
  const int32_t* eb_pose;  /* [n_eb] */
  const int32_t* eb_point; /* [n_eb] */
  const double* eb_z;      /* [3][n_eb] SoA: measured point in the camera frame           */
  const double* eb_w;      /* [n_eb] */
  /* LandmarkMotionTernaryEdge (g2o/types/types_dyn_slam3d.cpp:53-85), information = w * I3 */
  const int32_t* et_p1;    /* [n_et] point at frame k-1 */
  const int32_t* et_p2;    /* [n_et] point at frame k   */
  const int32_t* et_pose;  /* [n_et] motion vertex H    */
  const double* et_z;      /* [3][n_et] SoA measurement (zero in the reference)           */
  const double* et_w;      /* [n_et] */
  /* EdgeSE3 (g2o/types/edge_se3.cpp:77-104): odometry + motion smoothness */
  const int32_t* ep_i;     /* [n_ep] */
  const int32_t* ep_j;     /* [n_ep] */
  const double* ep_z;      /* [n_ep][12] measurement */
  const double* ep_info;   /* [n_ep][36] information, row-major */
  /* EdgeSE3Prior (g2o/types/edge_se3_prior.cpp:89-102), ParameterSE3Offset = identity */
  const int32_t* pr_pose;  /* [n_prior] */
  const double* pr_z;      /* [n_prior][12] */
  const double* pr_info;   /* [n_prior][36] */
  /* RobustKernelHuber delta per edge class (<=0: none); the prior has no kernel
   * (src/Optimizer.cc:1364-1373).  NB the reference keeps delta^2 in a float
   * (g2o/core/robust_kernel_impl.h:84); reproduced. */
  double huber_eb, huber_et, huber_ep;
} vdo_ba_graph;

/* One linearisation in block form (what BlockSolver::buildSystem leaves in Hpp/Hpl/Hll/b).
 * Host arrays, caller-allocated; NULL pointers are skipped. */
typedef struct vdo_ba_system {
  double* Hpp;     /* [n_pose][36]  diagonal blocks, row-major                           */
  double* bp;      /* [n_pose][6]                                                       */
  double* Hll;     /* [n_point][9]                                                      */
  double* bl;      /* [n_point][3]                                                      */
  double* Hpl_eb;  /* [18][n_eb] SoA 6x3 block (pose x point) of each binary edge (index r*3+c) */
  double* Hll_et;  /* [9][n_et]  SoA 3x3 block p1 x p2                                    */
  double* Hlp1_et; /* [18][n_et] SoA 3x6 block p1 x pose                                  */
  double* Hlp2_et; /* [18][n_et] SoA 3x6 block p2 x pose                                  */
  double* Hpp_ep;  /* [n_ep][36] 6x6 block pose_i x pose_j                                */
  double chi2;        /* activeChi2 */
  double robust_chi2; /* activeRobustChi2 (g2o/core/sparse_optimizer.cpp:102-114) */
} vdo_ba_system;

typedef struct vdo_lm_options {
  int32_t max_iterations;     /* SparseOptimizer::optimize(n): 300 full batch, 100 partial */
  double gain_threshold;      /* SparseOptimizerTerminateAction::setGainThreshold; <0 = not installed */
  int32_t verbose;
  int32_t solver;             /* 2: Schur complement solved matrix-free by PCG with the pose-chain preconditioner;
                               * 3: Schur complement assembled densely and factorised by a blocked Cholesky on the fp64 MFMA
                               *    units (6 n_pose <= 8192) - what g2o's LinearSolverDense / CSparse do on the reduced system;
                               * 0: auto = 3 when the EdgeSE3 graph is not a set of simple paths (loop closures, branches) or a
                               *    PCG solve needed more than 60 iterations, else 2                                      */
  double pcg_tolerance;       /* relative residual ||r||_M / ||b||_M; <=0 -> 1e-8 (1e-10 until round 6; g2o's own PCG: 1e-6) */
  int32_t pcg_max_iterations; /* <=0 -> 24 n_pose + 200, capped at 20000                   */
} vdo_lm_options;

#define VDO_LM_MAX_TRACE 512
typedef struct vdo_lm_stats {
  int32_t iterations;          /* outer iterations executed (return value of optimize())   */
  int32_t total_trials;        /* levenberg trials summed over iterations                  */
  int32_t stop_reason;         /* 0 max_iter, 1 LM terminate, 2 chi2 increase, 3 gain action, 4 fail */
  double initial_chi2, final_chi2, final_lambda;
  double chi2_trace[VDO_LM_MAX_TRACE];
  int32_t trials_trace[VDO_LM_MAX_TRACE];
  double ms_total, ms_linearize, ms_solve;
} vdo_lm_stats;

typedef struct vdo_ba vdo_ba;
/* Uploads the graph to HBM (SoA, resident until destroy) and builds the chain structure.
 * Limits (g2o has none; VDO_ERR_UNSUPPORTED names the offending track): a DYNAMIC landmark track - a chain of points linked by
 * LandmarkMotionTernaryEdges - is processed by ONE workgroup and must fit its tile: <= 256 points, <= 1536 edge incidences, <= 512 distinct
 * pose vertices (cameras + motions), and sum over those poses of ceil(observations / 6) <= 256: it may run over up to 256 frames (its
 * points bring a camera and a motion vertex each; 128 until round 6).  A STATIC point has no such limit since round 6: up to 256 pose vertices it lives in a tile
 * (a graph that holds such a track pays with fewer resident workgroups per CU - the tile kernels' LDS grows with the pose slots of the
 * largest tile - and has no dense solver beyond ~200 slots); beyond them it becomes a hub landmark with a workgroup of its own
 * (csrc/ba_hub.hip; such graphs are solved by the PCG, the dense solver refuses them). */
int vdo_ba_create(vdo_ctx* ctx, const vdo_ba_graph* g, vdo_ba** out);
int vdo_ba_destroy(vdo_ba* ba);
/* K18: `repeat` back-to-back linearisation sweeps (errors + Jacobians + Huber + block
 * accumulation) at the current estimate.  If ms_sweep != NULL it receives the mean
 * duration of ONE binary-edge sweep kernel measured with hipEvents on the ctx stream. */
int vdo_ba_linearize(vdo_ba* ba, int repeat, float* ms_sweep);
/* Measurement hook of the roofline leg (bench.py, tools/sweep_only.py; no counterpart in the reference - its g2o prints one
 * time per outer iteration, g2o/core/sparse_optimizer.cpp:406-418): `repeat` back-to-back runs, hipEvents on the ctx stream, of
 *   ms[0] the tile sweep kernel alone (k_sweep_tile<true, .>),
 *   ms[1] a whole linearisation as BlockSolver::buildSystem means it (g2o/core/block_solver.hpp:502-560): sweep + expansion of
 *         the pose blocks (k_finalize_pose) + pose-pose edges (k_posepose) + chi2 reduction,
 * and the layout figures the byte model of DESIGN.md 4.1 needs:
 *   dims[0] tiles, [1] (tile, pose-slot) pairs, [2] running sums per partial row (16 / 32), [3] max slots of a tile,
 *   [4] bytes read per EdgeSE3PointXYZ entry (key + measurement [+ weight]), [5] bytes read per ternary edge,
 *   [6] EdgeSE3PointXYZ ENTRIES of the tiles' edge blocks (>= the graph's edges: every tile holds its edges as a padded block of
 *       256 x (edges per thread) entries, thread-transposed, so that every load of a tile kernel is one contiguous row), [7] hub landmarks (static points whose observations do not fit a tile: a workgroup of its own each). */
int vdo_ba_profile_linearize(vdo_ba* ba, int repeat, float ms[2], int64_t dims[8]);
/* new (measurement): mean milliseconds of ONE Schur mat-vec launch (k_schur_tile<0>, the product B Hll^-1 B^T p of a CG iteration of the reduced-camera solve that
   replaces g2o's BlockSolver::solve / LinearSolverCSparse, g2o/core/block_solver.hpp:143-295), timed alone with events on the context's stream.  Call after vdo_ba_optimize. */
int vdo_ba_profile_schur(vdo_ba* ba, int repeat, float* ms);
/* One self-consistent linearisation in block form (BlockSolver::buildSystem) AT THE CURRENT ESTIMATE: the blocks of the last
 * vdo_ba_linearize when nothing moved the estimate since, else (after vdo_ba_optimize / vdo_ba_set_estimates) a fresh linearisation is
 * run first - collective on a sharded handle. */
int vdo_ba_download_system(vdo_ba* ba, vdo_ba_system* out);
/* Full Levenberg–Marquardt (control flow identical to the modified g2o). */
int vdo_ba_optimize(vdo_ba* ba, const vdo_lm_options* opt, vdo_lm_stats* stats);
int vdo_ba_get_estimates(vdo_ba* ba, double* pose_out /*[n_pose][12]*/, double* point_out /*[n_point][3]*/);
int vdo_ba_set_estimates(vdo_ba* ba, const double* pose, const double* point);
/* Multi-GPU (landmark-track shards).  One process per GPU; every rank creates its
 * vdo_ba from a SHARD graph: all pose vertices, all pose-pose edges and priors (replicated), and
 * the points it owns with every binary/ternary edge incident to them (vdo_ba_partition assigns
 * whole tracks to ranks).  `fn` must reduce `count` doubles in place at the DEVICE pointer `buf`
 * across ranks, stream-ordered on the ctx stream (op 0 = sum, 1 = max; e.g. torch.distributed
 * all_reduce over RCCL).  Exchanges: Hpp|bp|chi2 (42P+2 doubles) once per linearisation, the
 * block-Jacobi diagonal (21P+1) once per Levenberg trial, 6P doubles per PCG iteration, 3 scalars per
 * trial.  Every rank returns the same poses; points come back for the owned shard only. */
typedef int (*vdo_allreduce_fn)(void* user, void* device_buf, int64_t count, int op);
int vdo_ba_set_allreduce(vdo_ba* ba, vdo_allreduce_fn fn, void* user, int shard_rank);
/* The same exchanges issued by the library itself over RCCL (xGMI inside a node): ncclAllReduce in place on the solver's
 * device buffers, on the context's stream - no host callback in the loop.  Rank 0 draws a 128-byte id
 * (ncclGetUniqueId) which the host hands to every rank over any side channel; every rank then creates its communicator on the
 * context its vdo_ba uses and attaches it.  vdo_rccl_comm_stats: all-reduces issued so far and their payload bytes. */
typedef struct vdo_rccl_comm vdo_rccl_comm;
int vdo_rccl_unique_id(char id_out[128]);
int vdo_rccl_comm_create(vdo_ctx* ctx, const char id[128], int n_ranks, int rank, vdo_rccl_comm** out);
int vdo_rccl_comm_destroy(vdo_rccl_comm* comm);
int vdo_rccl_comm_stats(const vdo_rccl_comm* comm, int64_t* calls, int64_t* bytes);
int vdo_rccl_allreduce(vdo_rccl_comm* comm, double* device_buf, int64_t count, int op /* 0 sum, 1 max */);
int vdo_ba_set_rccl(vdo_ba* ba, vdo_rccl_comm* comm /* NULL: back to single GPU */);
/* Host-only (no GPU needed): owner rank of every point so that tracks (points linked by ternary
 * edges) stay together, shards are contiguous in first-observing-frame order and balanced by
 * incidence count. */
int vdo_ba_partition(const vdo_ba_graph* g, int world, int32_t* owner_of_point /*[n_point]*/);

/* ---- per-frame joint pose + optical-flow optimisation ------------------------------------------
 * Replaces the g2o calls inside Optimizer::PoseOptimizationFlow2Cam (reference
 * src/Optimizer.cc:2333-2542; camera, called from Tracking::Track src/Tracking.cc:697) and
 * Optimizer::PoseOptimizationFlow2 (:2755-2972; one call per object, src/Tracking.cc:932):
 * graph of 1 VertexSE3Expmap + N marginalised VertexSBAFlow, N EdgeSE3ProjectFlow2 (Huber) +
 * N EdgeFlowPrior, BlockSolver_6_3 + LinearSolverDense, optimize(100|200), chi2 gating.
 * The whole LM loop of a problem runs in one persistent workgroup; a batch (e.g. all objects
 * of a frame) is one kernel launch.  All pointers are HOST pointers; values are the
 * float->double conversions the reference performs in Converter (src/Converter.cc:25-35). */
typedef struct vdo_flow2_problem {
  int32_t n;              /* correspondences (TemperalMatch.size() / ObjId.size())          */
  const double* obs;      /* [n][2] last-frame pixel (kpUn.pt)                             */
  const double* flow;     /* [n][2] measured optical flow (initial estimate and prior)     */
  const double* depth;    /* [n]    depth of the last-frame pixel                          */
  double K[4];            /* fx, fy, cx, cy                                                */
  double Twl[16];         /* 4x4 row-major: last-frame camera-to-world                     */
  double T0[16];          /* 4x4 row-major initial estimate (mTcw / mInitModel)            */
  double info_flow;       /* 0.1  (Optimizer.cc:2405,2827)                                 */
  double info_prior;      /* 0.3 camera (:2440) / 0.5 object (:2863)                       */
  double huber_delta;     /* (double)sqrtf(0.04f) (:2371,2793)                             */
  double chi2_gate;       /* 0.04f (:2335,2757)                                            */
  int32_t max_iterations; /* 100 camera (:2455) / 200 object (:2878)                       */
  int32_t ref_quirks;     /* 1 = reproduce the BlockSolver_6_3 / 2-DoF aliasing;
                             0 = mathematically intended 2x2 Schur step (not parity-comparable) */
} vdo_flow2_problem;

typedef struct vdo_flow2_result {
  double T[16];           /* refined pose, 4x4 row-major (SE3Quat::to_homogeneous_matrix)  */
  int32_t n_inliers;      /* nInitialCorrespondences - nBad                                */
  int32_t iterations, trials, stop_reason;
  double initial_chi2, final_chi2, final_lambda;
} vdo_flow2_result;

typedef struct vdo_flow2_batch vdo_flow2_batch;
/* Upload n_problems problems (inputs become HBM-resident). */
int vdo_flow2_batch_create(vdo_ctx* ctx, int n_problems, const vdo_flow2_problem* probs, vdo_flow2_batch** out);
/* Per-frame use: slots of fixed capacity that are (re)defined every frame without re-allocating —
 * vdo_flow2_batch_reserve once, then per frame vdo_flow2_batch_set (inputs staged through pinned memory,
 * stream-ordered, no sync; problem == NULL empties a slot) -> run -> fetch. */
int vdo_flow2_batch_reserve(vdo_ctx* ctx, int n_problems, const int32_t* capacity, vdo_flow2_batch** out);
int vdo_flow2_batch_set(vdo_flow2_batch* batch, int k, const vdo_flow2_problem* problem);
/* One kernel launch: every problem is optimised from its initial estimate (stream-ordered, no sync). */
int vdo_flow2_batch_run(vdo_flow2_batch* batch);
/* results[n_problems]; flow_out[k] -> [n_k][2] refined flows; inlier_out[k] -> [n_k] (1 = inlier). Synchronises. */
int vdo_flow2_batch_fetch(vdo_flow2_batch* batch, vdo_flow2_result* results, double** flow_out, uint8_t** inlier_out);
int vdo_flow2_batch_destroy(vdo_flow2_batch* batch);
/* Convenience: create + run + fetch + destroy for a single problem. */
int vdo_flow2_optimize(vdo_ctx* ctx, const vdo_flow2_problem* p, vdo_flow2_result* result, double* flow_out, uint8_t* inlier_out);

/* ---- non-joint per-frame pose refinement (bJoint == false) -----------------------------------------
 * Replaces the g2o calls inside Optimizer::PoseOptimizationNew (reference src/Optimizer.cc:2177-2331;
 * camera, called from src/Tracking.cc:699) and Optimizer::PoseOptimizationObjMot (:2544-2753; per
 * object, src/Tracking.cc:936): 1 VertexSE3Expmap + n unary reprojection edges with information I2
 * (EdgeSE3ProjectXYZOnlyPose with Huber sqrt(0.01) / EdgeSE3ProjectXYZOnlyObjMotion without kernel),
 * BlockSolver_6_3 + LinearSolverDense, optimize(100 / 200), one classification round at chi2 > 0.01f.
 * Results use vdo_flow2_result. */
typedef struct vdo_pose_problem {
  int32_t n;              /* correspondences                                               */
  int32_t kind;           /* 0 EdgeSE3ProjectXYZOnlyPose (K) ; 1 EdgeSE3ProjectXYZOnlyObjMotion (P) */
  const double* obs;      /* [n][2] current-frame pixel (kpUn.pt)                          */
  const double* Xw;       /* [n][3] back-projected last-frame point (float -> double)      */
  double K[4];            /* fx, fy, cx, cy (kind 0)                                       */
  double P[12];           /* 3x4 row-major K*Tcw (kind 1, Optimizer.cc:2604-2606)          */
  double T0[16];          /* initial estimate: mTcw (kind 0) / Tcw^-1 * mInitModel (kind 1) */
  double huber_delta;     /* (double)sqrtf(0.01f) kind 0 (:2213) ; <= 0: no kernel (kind 1) */
  double chi2_gate;       /* 0.01f (:2181, :2546)                                          */
  int32_t max_iterations; /* 100 (:2269) / 200 (:2664)                                     */
  int32_t pad;
} vdo_pose_problem;

typedef struct vdo_pose_batch vdo_pose_batch;
int vdo_pose_batch_create(vdo_ctx* ctx, int n_problems, const vdo_pose_problem* probs, vdo_pose_batch** out);
int vdo_pose_batch_run(vdo_pose_batch* batch);       /* one kernel launch, stream-ordered, no sync */
int vdo_pose_batch_fetch(vdo_pose_batch* batch, vdo_flow2_result* results, uint8_t** inlier_out);
int vdo_pose_batch_destroy(vdo_pose_batch* batch);
int vdo_pose_optimize(vdo_ctx* ctx, const vdo_pose_problem* p, vdo_flow2_result* result, uint8_t* inlier_out);

/* ---- RANSAC initialiser of the per-frame pose problems -------------------------------
 * What Tracking::GetInitModelCam / GetInitModelObj (reference src/Tracking.cc:1614-1715, 1717-1849) get from
 * cv::solvePnPRansac(pre_3d, cur_2d, K, distCoeffs = 0, rvec, tvec, false, 500, 0.4, 0.98, inliers,
 * SOLVEPNP_AP3P): the sequential RANSAC of OpenCV 3.4 (cv::RNG subsets of 4 points, minimal P3P solve on 3 with
 * the 4th as tie-breaker, inliers = squared reprojection error <= thr^2, iteration budget shrunk by
 * RANSACUpdateNumIters) with every hypothesis solved (AP3P) and voted on the GPU at once and the loop replayed on the
 * host over the votes.  `refit` is a FLAG WORD (below): bit 0 applies OpenCV's final EPnP re-estimation on the inliers (what
 * solvePnPRansac returns since 3.3 and what the reference's LM is seeded with - the host classes set it); with the bit clear
 * the winning minimal hypothesis is returned.  Behaviour change of round 5 for callers that pass 0 or 1: the minimal solver
 * is AP3P with the hypotheses re-orthogonalised as cv::Rodrigues does (bit 1 restores Grunert's P3P of rounds 1-4), so
 * hypotheses, inlier sets and poses of noisy problems differ from the earlier rounds'; a batch that mixes the two solvers
 * returns VDO_ERR_INVALID.  T = [R|t] camera-from-world (what Rodrigues(rvec), tvec give). */
typedef struct vdo_pnp_problem {
  int32_t n;
  const double* X;          /* [n][3] 3-D points (pre_3d)                         */
  const double* uv;         /* [n][2] pixels (cur_2d)                             */
  double K[4];              /* fx, fy, cx, cy                                     */
  int32_t max_iterations;   /* 500                                                */
  double reproj_threshold;  /* 0.4 px                                             */
  double confidence;        /* 0.98                                               */
  int32_t refit;            /* bit 0: the winning model is re-estimated on its inliers by EPnP, as cv::solvePnPRansac does for the P3P
                             *    / AP3P kernels since OpenCV 3.3 (the inlier set stays the RANSAC one); clear: the minimal hypothesis.
                             * bit 1 (value 2): Grunert's P3P as the minimal solver (rounds 1-4) instead of AP3P - Ke & Roumeliotis in the
                             *    layout of OpenCV 3.4's ap3p.cpp, the solver the reference's calls name (SOLVEPNP_AP3P), the default.
                             *    All problems of a batch must name the same solver.                                                */
} vdo_pnp_problem;
typedef struct vdo_pnp_result {
  double T[16];             /* 4x4 row-major; identity when no model was found    */
  int32_t n_inliers;        /* inliers.rows                                       */
  int32_t iterations_run;   /* hypotheses the sequential loop would have examined */
  int32_t best_iteration;   /* index of the winning hypothesis (-1: none)         */
} vdo_pnp_result;
/* All problems of a frame (camera + every object) in one call: two launches, one synchronisation. */
int vdo_pnp_ransac_batch(vdo_ctx* ctx, int n_problems, const vdo_pnp_problem* probs, vdo_pnp_result* results, uint8_t** inlier_out);
/* The same with a hook: host_work(host_arg) is called once, on the calling thread, while the device runs the hypotheses and the votes - for host work of the
 * caller that does not depend on the result (GetInitModelObj's motion-model inlier count, src/Tracking.cc:1767-1800, depends on neither).  NULL: no hook. */
int vdo_pnp_ransac_batch_overlap(vdo_ctx* ctx, int n_problems, const vdo_pnp_problem* probs, vdo_pnp_result* results, uint8_t** inlier_out,
                                 void (*host_work)(void*), void* host_arg);
int vdo_pnp_ransac(vdo_ctx* ctx, const vdo_pnp_problem* p, vdo_pnp_result* result, uint8_t* inlier_out);

/* ---- ORB front-end ------------------------------------------------------------------------------
 * Replaces ORBextractor::ORBextractor (reference src/ORBextractor.cc:399-459) and
 * ORBextractor::operator() (:1035-1110): ComputePyramid (:1112-1137), ComputeKeyPointsOctTree
 * (:754-842: per-cell cv::FAST with threshold fallback, DistributeOctTree :528-752, IC_Angle
 * :66-93) and the per-level 7x7 GaussianBlur (:1083-1084).  Descriptors are NOT produced: the
 * reference never computes them (call commented out at :1091). */
typedef struct vdo_orb_params {
  int32_t n_features;     /* ORBextractor.nFeatures  2500 */
  float scale_factor;     /* ORBextractor.scaleFactor 1.2 */
  int32_t n_levels;       /* ORBextractor.nLevels    8    */
  int32_t ini_th;         /* ORBextractor.iniThFAST  20   */
  int32_t min_th;         /* ORBextractor.minThFAST  7    */
} vdo_orb_params;

/* SoA mirror of std::vector<cv::KeyPoint>; arrays are caller-allocated with `capacity` entries. */
typedef struct vdo_keypoints {
  int32_t capacity, n;
  float *x, *y, *response, *angle, *size;
  int32_t* octave;
} vdo_keypoints;

typedef struct vdo_orb vdo_orb;
int vdo_orb_create(vdo_ctx* ctx, const vdo_orb_params* prm, int width, int height, vdo_orb** out);
int vdo_orb_destroy(vdo_orb* orb);
/* gray: 8-bit single channel, row stride `stride` bytes; host pointer unless src_is_device. */
int vdo_orb_extract(vdo_orb* orb, const uint8_t* gray, int stride, int src_is_device, vdo_keypoints* out);
/* The same in two halves: _begin queues the device stage (K3, K4, K6, K7 + the copy of the candidates) on the extractor's stream
 * and returns at once, _end waits for it and runs the quadtrees (K5) - a caller overlaps other work with the device stage. */
int vdo_orb_extract_begin(vdo_orb* orb, const uint8_t* gray, int stride, int src_is_device);
int vdo_orb_extract_end(vdo_orb* orb, vdo_keypoints* out);
/* K8 - rotated BRIEF, the `_descriptors` output of ORBextractor::operator() (computeOrbDescriptor src/ORBextractor.cc:97-136,
 * pattern :139-397; the reference allocates the 32-byte rows but has the call commented out, :1083-1091).  vdo_orb_descriptors:
 * descriptors of the keypoints the last vdo_orb_extract / _end returned, same order, host buffer [capacity_rows][32];
 * vdo_orb_extract_desc = operator() with both outputs (desc32 nullable, [out->capacity][32]). */
int vdo_orb_descriptors(vdo_orb* orb, uint8_t* desc32, int capacity_rows);
int vdo_orb_extract_desc(vdo_orb* orb, const uint8_t* gray, int stride, int src_is_device, vdo_keypoints* out, uint8_t* desc32);
/* Inspection of the last extraction (mvImagePyramid is a public member of the reference class). */
int vdo_orb_level_info(vdo_orb* orb, int level, int* w, int* h, int* n_features, int* n_candidates);
int vdo_orb_get_pyramid(vdo_orb* orb, int level, uint8_t* out_bordered /* (w+38)*(h+38) */);
int vdo_orb_get_blurred(vdo_orb* orb, int level, uint8_t* out /* w*h */);
int vdo_orb_get_candidates(vdo_orb* orb, int level, float* x, float* y, float* response, float* angle, int cap, int* n);
/* Wall time of the last vdo_orb_extract: ms[0] device stage launch .. candidates on the host, ms[1] host quadtree. */
int vdo_orb_last_timing(vdo_orb* orb, double ms[2]);
/* Kernel launches that build the pyramid of one image: 2 (levels 0-4 cascaded in LDS from the image, the rest from level 4;
 * 1 with <= 5 levels) - possible with <= 8 levels whose cascaded source windows fit the LDS buffers - or n_levels (level by
 * level; also forced by the environment variable VDO_ORB_PYRAMID_LAUNCHES). */
int vdo_orb_pyramid_launches(const vdo_orb* orb);

/* K1: Tracking::GrabImageRGBD depth preprocessing (src/Tracking.cc:180-204), in place. */
int vdo_depth_preprocess(vdo_ctx* ctx, float* depth, int64_t n, float bf, float depth_map_factor, int is_device);
/* K2: cv::cvtColor(RGB2GRAY / BGR2GRAY) as called at src/Tracking.cc:209-222 (host in, host out). */
int vdo_rgb2gray(vdo_ctx* ctx, const uint8_t* rgb, int64_t n_pixels, int channels, int rgb_order, uint8_t* gray);

/* ---- Frame construction (src/Frame.cc:61-260) -------------------------------------------------*/
typedef struct vdo_frame_images vdo_frame_images;   /* HBM-resident depth (f32), flow (2 x f32), mask (i32) */
int vdo_frame_images_create(vdo_ctx* ctx, int width, int height, vdo_frame_images** out);
int vdo_frame_images_upload(vdo_frame_images* f, const float* depth, const float* flow, const int32_t* mask);
/* The same copies issued and awaited on the stream of `ctx` instead of the image set's own - for a host thread that brings the flow and the
 * mask of a frame up while another one already works on its depth map (the reference hands TrackRGBD host cv::Mat's, include/System.h:45-51). */
int vdo_frame_images_upload_on(vdo_ctx* ctx, vdo_frame_images* f, const float* depth, const float* flow, const int32_t* mask);
int vdo_frame_images_upload_device(vdo_frame_images* f, const float* depth_dev, const float* flow_dev, const int32_t* mask_dev);
int vdo_frame_images_depth_preprocess(vdo_frame_images* f, float bf, float depth_map_factor);   /* K1 in place, resident image */
/* vdo_frame_images_upload_device + (convert_depth != 0) vdo_frame_images_depth_preprocess as ONE launch (GrabImageRGBD's head,
 * src/Tracking.cc:180-204 + the cv::Mat headers it keeps): same bytes, same two divisions per depth pixel, a quarter of the stream operations. */
int vdo_frame_images_ingest_device(vdo_frame_images* f, const float* depth_dev, const float* flow_dev, const int32_t* mask_dev, float bf, float depth_map_factor,
                                   int convert_depth);
int vdo_frame_images_destroy(vdo_frame_images* f);
/* Later calls on these images run on `ctx` (its stream and scratch arena) instead of the context they were created with.
 * Every call on vdo_frame_images is host-synchronous, so the switch needs no device-side ordering; it exists so that a
 * host worker thread can work on the images of one frame while the main thread works on another context (FramePipeline). */
int vdo_frame_images_set_ctx(vdo_frame_images* f, vdo_ctx* ctx);
/* K9: static keypoint filter of Frame::Frame (:100-128) + depth gather (:178-194); outputs in input order. */
int vdo_frame_static_filter(vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth,
                            int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y,
                            float* depth_out, int* n_out);
/* The same for UseSampleFeature = 1 (src/Frame.cc:132-166: the destination is tested on all four sides), and the sampler that
 * replaces ORB there: Frame::SampleKeyPoints (:672-737), 3000 random grid positions from cv::RNG(seed); host only. */
int vdo_frame_static_filter_sampled(vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth,
                                    int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y,
                                    float* depth_out, int* n_out);
int vdo_sample_keypoints(int rows, int cols, uint64_t seed, int capacity, float* x_out, float* y_out, int* n_out);
/* K10: semi-dense object sampling (:201-228), raster order.  Pass key_x == NULL to keep results on the device. */
int vdo_frame_object_sample(vdo_frame_images* f, float th_depth_obj, int step, int cap,
                            float* key_x, float* key_y, float* corr_x, float* corr_y,
                            float* flow_x, float* flow_y, float* depth_out, int32_t* label, int* n_out);
/* K10 on the stream of `on_ctx` (NULL: the image set's own) with a scratch set of its own: for a caller that samples the objects on
 * a second host thread while the owning thread runs K9 / RenewFrameInfo on the same image set.  Host outputs required. */
int vdo_frame_object_sample_on(vdo_ctx* on_ctx, vdo_frame_images* f, float th_depth_obj, int step, int cap,
                               float* key_x, float* key_y, float* corr_x, float* corr_y,
                               float* flow_x, float* flow_y, float* depth_out, int32_t* label, int* n_out);
/* K9 + K10 of one image in one call and ONE synchronisation (Frame::Frame runs them back to back: src/Frame.cc:104-131 / 132-166,
 * 168-199): the arguments of vdo_frame_static_filter[_sampled] (sampled != 0: the UseSampleFeature branch) followed by those of
 * vdo_frame_object_sample (host outputs required). */
int vdo_frame_filters(vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth, int sampled,
                      int32_t* keep_idx, float* s_corr_x, float* s_corr_y, float* s_flow_x, float* s_flow_y, float* s_depth, int* n_static,
                      float th_depth_obj, int step, int cap,
                      float* key_x, float* key_y, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out, int32_t* label, int* n_obj);
/* The same on the stream of `on_ctx` instead of the image set's own context (NULL: the latter): for a caller that runs K9 / K10 on a
 * second host thread while the owning thread keeps using the image set (the scratch of this call is used by nothing else). */
int vdo_frame_filters_on(vdo_ctx* on_ctx, vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth, int sampled,
                         int32_t* keep_idx, float* s_corr_x, float* s_corr_y, float* s_flow_x, float* s_flow_y, float* s_depth, int* n_static,
                         float th_depth_obj, int step, int cap,
                         float* key_x, float* key_y, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out, int32_t* label, int* n_obj);

/* ---- Tracking-side gathers over the resident images ------------
 * All coordinate arrays are host pointers (the reference keeps them in std::vector<cv::KeyPoint>). */

/* Tracking::GrabImageRGBD, static correspondences: depth at ((int)kx,(int)ky) if inside the
 * 1-px-inset image and > 0, else -1.  Replaces src/Tracking.cc:259-275. */
int vdo_propagate_static(vdo_frame_images* f, int n, const float* kx, const float* ky, float* depth_out);
/* Object correspondences: (depth,label) if inside and 0 < depth < th_depth_obj, else (0.1, 0).
 * Replaces src/Tracking.cc:278-305. */
int vdo_propagate_object(vdo_frame_images* f, int n, const float* kx, const float* ky, float th_depth_obj,
                         float* depth_out, int32_t* label_out);
/* Optimizer::Get3DinWorld over a vector of keypoints: Twc[:3,:3]*x3Dc + Twc[:3,3] with cv::gemm
 * rounding (double accumulate, one rounding).  Replaces src/Optimizer.cc:2974-2995 (called per point
 * from Tracking.cc:1086,1094 and Optimizer builders).  K4 = fx,fy,cx,cy ; Twc row-major 4x4. */
int vdo_get3d_world(vdo_ctx* ctx, int n, const float* kx, const float* ky, const float* depth,
                    const float K4[4], const float Twc[16], float* xyz_out);
/* Tracking::GetSceneFlowObj: flow3d = X_cur - X_last with both points back-projected through
 * Frame::UnprojectStereoObject / UnprojectStereoObjectLast (src/Frame.cc:517-555); entries whose
 * semantic label is <= 0 in either frame get obj_label = -1 and zero flow.  Replaces src/Tracking.cc:1278-1364 (per-point part). */
int vdo_scene_flow(vdo_ctx* ctx, int n, const float* cur_x, const float* cur_y, const float* cur_d, const int32_t* cur_label, const float Tcw_cur[16],
                   const float* last_x, const float* last_y, const float* last_d, const int32_t* last_label, const float Tcw_last[16],
                   const float K4[4], float* flow3d_out, int32_t* obj_label_inout);
/* Tracking::RenewFrameInfo, static part (src/Tracking.cc:2666-2790): carry the inliers TM_sta, top
 * up from the ORB keypoints (stride-20 interleave, skipping keypoints within 1 px of a carried one),
 * then depth.  Outputs need capacity max_num_sta + 1 ; *n_out = count. */
int vdo_renew_static(vdo_frame_images* f, int n_tm, const int32_t* tm_sta, const float* stat_x, const float* stat_y,
                     int n_orb, const float* orb_x, const float* orb_y, int max_num_sta,
                     float* key_x, float* key_y, float* corr_x, float* corr_y, float* flow_x, float* flow_y,
                     int32_t* inlier_id, float* depth_out, int* n_out);
/* The same + Optimizer::Get3DinWorld of the new set (mvStat3DPointTmp, src/Tracking.cc:2784-2790) in one pass over the device
 * (one synchronisation): xyz_out [3 * (max_num_sta + 1)], nullable (K4 / Twc unused then). */
int vdo_renew_static_world(vdo_frame_images* f, int n_tm, const int32_t* tm_sta, const float* stat_x, const float* stat_y,
                           int n_orb, const float* orb_x, const float* orb_y, int max_num_sta, const float K4[4], const float Twc[16],
                           float* key_x, float* key_y, float* corr_x, float* corr_y, float* flow_x, float* flow_y,
                           int32_t* inlier_id, float* depth_out, float* xyz_out, int* n_out);
/* Tracking::UpdateMask (src/Tracking.cc:3015-3065): labels of this frame's mask at n positions
 * (-1 outside), and the warp of label `label` from `last`'s mask into `cur`'s mask by `last`'s flow. */
int vdo_mask_at(vdo_frame_images* f, int n, const float* cx, const float* cy, int32_t* label_out);
int vdo_mask_warp(vdo_frame_images* cur, vdo_frame_images* last, int32_t label);
int vdo_frame_images_download_mask(vdo_frame_images* f, int32_t* mask_out);
/* The resident depth image after K1 (metres): GrabImageRGBD converts the caller's imD in place (src/Tracking.cc:180-204); a host
 * caller that uploaded the raw map gets the converted one back with this. */
int vdo_frame_images_download_depth(vdo_frame_images* f, float* depth_out);

/* ---- Tracking bookkeeping around the gathers ------------------------------------ */

/* Tracking::DynObjTracking (src/Tracking.cc:1366-1612): groups the object points of the current frame
 * by semantic label; drops labels with > 50 % of their points in the image border (obj label -1),
 * labels whose share of slow points (||flow3d_xz|| < sf_mg_thres) exceeds sf_ds_thres become static
 * (obj label 0), labels with mean depth > th_depth_obj or < 150 points are dropped (-1); the survivors
 * get the motion label of the last-frame object whose semantic label wins the vote of their points'
 * last-frame labels (else a fresh id from *max_id_inout).  flow3d comes from vdo_scene_flow.
 * Outputs: obj_label_inout updated; accepted objects as CSR obj_off[n_obj+1] / obj_idx (capacity n),
 * obj_sem / obj_mod (capacity: number of distinct labels). */
typedef struct vdo_dyn_obj_params {
  int32_t img_w, img_h;
  int32_t shrink_row, shrink_col; /* 25 / 50 on KITTI, 0 / 0 otherwise (:1412-1416)               */
  float sf_mg_thres, sf_ds_thres; /* Tracking::fSFMgThres / fSFDsThres (yaml SFMgThres, SFDsThres) */
  float th_depth_obj;             /* mThDepthObj                                                   */
  int32_t f_id;                   /* frame id: 1 resets the id counter (:1536-1537)                 */
} vdo_dyn_obj_params;
int vdo_dyn_obj_tracking(const vdo_dyn_obj_params* prm, int n, const int32_t* sem_label, int32_t* obj_label_inout,
                         const float* key_x, const float* key_y, const float* depth, const float* flow3d, const int32_t* last_sem_label,
                         int n_last_obj, const int32_t* last_sem_pos, const int32_t* last_mod_label, const uint8_t* last_obj_stat,
                         int32_t* max_id_inout, int32_t* obj_off, int32_t* obj_idx, int32_t* obj_sem, int32_t* obj_mod, int* n_obj_out);

/* Tracking::RenewFrameInfo, object part (src/Tracking.cc:2806-2995).  `f` holds the NEW frame's images.
 * Carries the inliers of every tracked object (int-truncated position must lie on a mask != 0 with
 * 0 < depth < 25 and flow inside the image), tops every tracked object up to max_num_obj from the
 * semi-dense sampling tmp_* of the new image (stride-15 interleave, skipping samples within 1 px of a
 * carried point), then appends all samples of labels that are not tracked yet (object label -2).
 * Outputs (capacity `cap` each): key, depth, semantic label, flow, correspondence, inlier id (-1 for
 * added points), object label. */
int vdo_renew_object(vdo_frame_images* f, int n_obj, const int32_t* inl_off, const int32_t* inl_idx, const uint8_t* obj_stat,
                     const int32_t* sem_pos, const int32_t* mod_label,
                     const float* cur_x, const float* cur_y, const int32_t* cur_obj_label,
                     int n_tmp, const float* tmp_x, const float* tmp_y, const float* tmp_depth, const int32_t* tmp_label,
                     const float* tmp_flow_x, const float* tmp_flow_y, const float* tmp_corr_x, const float* tmp_corr_y,
                     int max_num_obj, int cap,
                     float* key_x, float* key_y, float* depth_out, int32_t* sem_out, float* flow_x, float* flow_y,
                     float* corr_x, float* corr_y, int32_t* dyn_inlier_id, int32_t* obj_label_out, int* n_out);
/* The same + the 3-D points of the new set (mvObj3DPoint, src/Tracking.cc:2981-2990) in one pass over the device (one
 * synchronisation): xyz_out [3 * cap], nullable (K4 / Twc unused then). */
int vdo_renew_object_world(vdo_frame_images* f, int n_obj, const int32_t* inl_off, const int32_t* inl_idx, const uint8_t* obj_stat,
                           const int32_t* sem_pos, const int32_t* mod_label,
                           const float* cur_x, const float* cur_y, const int32_t* cur_obj_label,
                           int n_tmp, const float* tmp_x, const float* tmp_y, const float* tmp_depth, const int32_t* tmp_label,
                           const float* tmp_flow_x, const float* tmp_flow_y, const float* tmp_corr_x, const float* tmp_corr_y,
                           int max_num_obj, int cap, const float K4[4], const float Twc[16],
                           float* key_x, float* key_y, float* depth_out, int32_t* sem_out, float* flow_x, float* flow_y,
                           float* corr_x, float* corr_y, int32_t* dyn_inlier_id, int32_t* obj_label_out, float* xyz_out, int* n_out);

/* Tracking::UpdateMask (src/Tracking.cc:2997-3068) in one stream-ordered sequence without host round
 * trips: per last-frame semantic label (ascending) the labels of `cur`'s mask at the flowed positions
 * vote; with >= 100 votes and background winning, that label's pixels of `last`'s mask are warped by
 * `last`'s flow into `cur`'s mask (visible to the next label's vote, as in the reference). */
int vdo_update_mask(vdo_frame_images* cur, vdo_frame_images* last, int n, const int32_t* last_sem_label,
                    const float* last_corr_x, const float* last_corr_y, int* n_recovered);
/* UpdateMask (K15) -> object part of the propagation (K11) -> GetSceneFlowObj (K13) in one call (src/Tracking.cc:2997-3068,
 * :283-305, :1278-1364): what vdo_update_mask + vdo_propagate_object + vdo_scene_flow do in that order, with one upload, one
 * download and one synchronisation.  obj_label_out starts at -2 like vObjLabel (Tracking.cc:1289). */
int vdo_object_chain(vdo_frame_images* cur, vdo_frame_images* last, int n, const int32_t* last_sem_label, const float* last_corr_x, const float* last_corr_y,
                     float th_depth_obj, const float Tcw_cur[16], const float* last_x, const float* last_y, const float* last_d, const float Tcw_last[16],
                     const float K4[4], int* n_recovered, float* depth_out, int32_t* sem_out, float* flow3d_out, int32_t* obj_label_out);
/* Optional, new (no counterpart in the reference): the inputs of vdo_object_chain that belong to the LAST frame (mLastFrame's object set: vSemObjLabel, mvObjCorres,
 * mvObjKeys, mvObjDepth - src/Tracking.cc:1040-1063 leaves them final at the end of Track()) sent to the device a frame ahead, asynchronously on `ctx`'s stream.  The
 * next vdo_object_chain on a frame of that context uses them if - compared value by value - they are what it is called with, and otherwise stages its inputs itself. */
int vdo_object_chain_prestage(vdo_ctx* ctx, int n, const int32_t* last_sem_label, const float* last_corr_x, const float* last_corr_y,
                              const float* last_x, const float* last_y, const float* last_d);

/* Tracklets: Tracking::GetStaticTrack / GetDynamicTrackNew (src/Tracking.cc:2201-2421) rebuild every
 * tracklet from frame 0 on every frame; this builder is incremental (one association vector per call)
 * and yields the same tracklets in the same order.  Host only. */
typedef struct vdo_tracks vdo_tracks;
int vdo_tracks_create(int with_object_label, vdo_tracks** out);
int vdo_tracks_destroy(vdo_tracks* t);
int vdo_tracks_add_frame(vdo_tracks* t, int n, const int32_t* asso /* index in the previous frame or -1 */, const int32_t* feat_label);
int vdo_tracks_size(vdo_tracks* t, int* n_tracks, int64_t* n_pairs);
int vdo_tracks_get(vdo_tracks* t, int32_t* track_off, int32_t* pair_frame, int32_t* pair_feat, int32_t* obj_id);
/* Only the tracks still observed in frame >= first_frame, whole and in creation order (buffers sized as for vdo_tracks_get; *n_tracks / *n_pairs = what was written):
 * what the windowed optimisation reads (Optimizer::PartialBatchOptimization, reference src/Optimizer.cc:42-1230, walks mpMap->TrackletSta of the window's frames). */
int vdo_tracks_get_since(vdo_tracks* t, int first_frame, int* n_tracks, int64_t* n_pairs, int32_t* track_off, int32_t* pair_frame, int32_t* pair_feat, int32_t* obj_id);

#ifdef __cplusplus
}
#endif
#endif /* VDO_SLAM_HIP_H_ */
