/* TEST INFRASTRUCTURE — C API of the CPU oracle (a restatement of the reference's
 * algorithms for the hot path).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (libvdo_hip.so)
 * never links or calls it.
 *
 * PINNED WHERE THE REFERENCE'S OWN SOURCES REACH, UNPINNED BELOW: the reference ships no
 * tests / golden vectors and its third-party dependencies (OpenCV 3.4, Eigen3, CSparse)
 * are absent.  oracle/ref/ compiles the reference's own first-party
 * sources AND its vendored g2o verbatim against shims (oracle/_ref/libref_*.so); the
 * oracle equals them - tests/test_ref_orb.py, test_ref_track.py, test_ref_g2o.py,
 * test_ref_full.py: integer / index work and the g2o edges bit for bit, the per-frame
 * optimisers float for float, the batch optimisers to 2e-6.  PARITY UNPINNED for what
 * sits under that: OpenCV's primitives (FAST, resize, blur, cv::gemm rounding,
 * solvePnPRansac incl. AP3P / EPnP) and the inside of Eigen / CSparse, which are
 * restatements of published algorithms on both sides of every comparison, checked only
 * by self-consistency KATs (tests/test_oracle_*.py).
 *
 * Struct layouts below deliberately equal the ones in include/vdo_slam_hip.h so the
 * Python tests can hand the same ctypes structures to both libraries.
 */
#ifndef VDO_ORACLE_H_
#define VDO_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Batch dynamic-SLAM factor graph in SoA form (the graph that
 * Optimizer::FullBatchOptimization / PartialBatchOptimization build,
 * src/Optimizer.cc:1232-1768 / :42-637).  Poses are g2o::VertexSE3 estimates
 * (Isometry3) stored as 12 doubles: R row-major (9) then t (3). */
typedef struct vdo_ba_graph {
  int32_t n_pose, n_point, n_eb, n_et, n_ep, n_prior;
  const double* pose;      /* [n_pose][12]  cameras (T_wc) and object motions (H) */
  const double* point;     /* [n_point][3]  VertexPointXYZ (world)               */
  /* EdgeSE3PointXYZ (edge_se3_pointxyz.cpp:99-140): z = point in camera frame, info = w*I */
  const int32_t* eb_pose;  /* [n_eb] */
  const int32_t* eb_point; /* [n_eb] */
  const double* eb_z;      /* [3][n_eb] SoA */
  const double* eb_w;      /* [n_eb] */
  /* LandmarkMotionTernaryEdge (types_dyn_slam3d.cpp:53-85): e = p1 - H^-1 p2 - z, info = w*I */
  const int32_t* et_p1;    /* [n_et] */
  const int32_t* et_p2;    /* [n_et] */
  const int32_t* et_pose;  /* [n_et] */
  const double* et_z;      /* [3][n_et] SoA */
  const double* et_w;      /* [n_et] */
  /* EdgeSE3 (edge_se3.cpp:77-104): measurement Z (12 doubles), info 6x6 row-major */
  const int32_t* ep_i;     /* [n_ep] */
  const int32_t* ep_j;     /* [n_ep] */
  const double* ep_z;      /* [n_ep][12] */
  const double* ep_info;   /* [n_ep][36] */
  /* EdgeSE3Prior (edge_se3_prior.cpp:89-102), offset parameter = identity */
  const int32_t* pr_pose;  /* [n_prior] */
  const double* pr_z;      /* [n_prior][12] */
  const double* pr_info;   /* [n_prior][36] */
  /* Huber deltas per edge class; <= 0 means no robust kernel.  The prior never has one
   * (src/Optimizer.cc:1364-1373). */
  double huber_eb, huber_et, huber_ep;
} vdo_ba_graph;

/* One linearisation (BlockSolver::buildSystem, block_solver.hpp:502-560) in
 * block form.  All arrays caller-allocated; any pointer may be NULL to skip. */
typedef struct vdo_ba_system {
  double* Hpp;     /* [n_pose][36]  diagonal 6x6 blocks, row-major              */
  double* bp;      /* [n_pose][6]                                              */
  double* Hll;     /* [n_point][9]  diagonal 3x3 blocks                         */
  double* bl;      /* [n_point][3]                                             */
  double* Hpl_eb;  /* [18][n_eb] SoA: 6x3 block pose x point of each binary edge (row-major index r*3+c) */
  double* Hll_et;  /* [9][n_et]  SoA: 3x3 block p1 x p2                          */
  double* Hlp1_et; /* [18][n_et] SoA: 3x6 block p1 x pose                        */
  double* Hlp2_et; /* [18][n_et] SoA: 3x6 block p2 x pose                        */
  double* Hpp_ep;  /* [n_ep][36] 6x6 block pose_i x pose_j                       */
  double chi2;        /* sum of e^T Omega e (activeChi2)                         */
  double robust_chi2; /* sum of rho(e) (activeRobustChi2, sparse_optimizer.cpp:102-114) */
} vdo_ba_system;

typedef struct vdo_lm_options {
  int32_t max_iterations;     /* optimize(n): 300 full batch, 100 partial            */
  double gain_threshold;      /* SparseOptimizerTerminateAction; <0 = not installed  */
  int32_t verbose;
  int32_t solver;             /* product only: 0 auto, 1 dense Cholesky, 2 PCG       */
  double pcg_tolerance;       /* product only                                        */
  int32_t pcg_max_iterations; /* product only                                        */
} vdo_lm_options;

#define VDO_LM_MAX_TRACE 512
typedef struct vdo_lm_stats {
  int32_t iterations;          /* outer iterations executed (optimize() return)      */
  int32_t total_trials;        /* sum of levenberg trials                            */
  int32_t stop_reason;         /* 0 max_iter,1 LM terminate,2 chi2 increase,3 gain action,4 fail */
  double initial_chi2, final_chi2, final_lambda;
  double chi2_trace[VDO_LM_MAX_TRACE];   /* robust chi2 after each outer iteration   */
  int32_t trials_trace[VDO_LM_MAX_TRACE];
  double ms_total, ms_linearize, ms_solve;
} vdo_lm_stats;

int vdo_oracle_ba_linearize(const vdo_ba_graph* g, vdo_ba_system* out);
/* pose_out [n_pose][12], point_out [n_point][3] */
int vdo_oracle_ba_optimize(const vdo_ba_graph* g, const vdo_lm_options* opt,
                           double* pose_out, double* point_out, vdo_lm_stats* stats);
/* Assemble the full normal equations as COO (upper triangle incl. diagonal) for the
 * scipy cross-check.  Order of unknowns: points (3 each) first, then poses (6 each).
 * Returns nnz written (or needed when rows==NULL). */
int64_t vdo_oracle_ba_normal_equations(const vdo_ba_graph* g, int32_t* rows, int32_t* cols,
                                       double* vals, int64_t cap, double* rhs);
/* Solve the normal equations (H + lambda I) x = b with the oracle's sparse Cholesky. */
int vdo_oracle_ba_solve(const vdo_ba_graph* g, double lambda, double* x);

/* Per-frame joint pose + optical-flow problem (Optimizer::PoseOptimizationFlow2Cam /
 * PoseOptimizationFlow2, src/Optimizer.cc:2333-2542 / 2755-2972).  Same layout as
 * include/vdo_slam_hip.h. */
typedef struct vdo_flow2_problem {
  int32_t n;              /* correspondences                                             */
  const double* obs;      /* [n][2] last-frame pixel (kpUn.pt)                           */
  const double* flow;     /* [n][2] measured optical flow (initial estimate + prior)     */
  const double* depth;    /* [n]    depth of the last-frame pixel                        */
  double K[4];            /* fx, fy, cx, cy                                              */
  double Twl[16];         /* 4x4 row-major, last-frame camera-to-world                   */
  double T0[16];          /* 4x4 row-major initial estimate (goes through toSE3Quat)     */
  double info_flow;       /* 0.1                                                         */
  double info_prior;      /* 0.3 camera / 0.5 object                                     */
  double huber_delta;     /* (double)sqrtf(0.04f)                                        */
  double chi2_gate;       /* 0.04f                                                       */
  int32_t max_iterations; /* 100 camera / 200 object                                     */
  int32_t ref_quirks;     /* 1: reproduce the BlockSolver_6_3 / 2-DoF mismatch (F3)      */
} vdo_flow2_problem;

/* returns the number of inliers (>=0) ; T_out 4x4 row-major ; flow_out [n][2] ; inlier_out [n] */
int vdo_oracle_flow2_optimize(const vdo_flow2_problem* p, double T_out[16], double* flow_out,
                              uint8_t* inlier_out, vdo_lm_stats* stats);

/* Non-joint per-frame pose refinement (Optimizer::PoseOptimizationNew src/Optimizer.cc:2177-2331,
 * Optimizer::PoseOptimizationObjMot :2544-2753): one VertexSE3Expmap, n unary reprojection edges,
 * information I2.  Same layout as include/vdo_slam_hip.h. */
typedef struct vdo_pose_problem {
  int32_t n;
  int32_t kind;           /* 0 EdgeSE3ProjectXYZOnlyPose (K) ; 1 EdgeSE3ProjectXYZOnlyObjMotion (P) */
  const double* obs;      /* [n][2] current-frame pixel                                  */
  const double* Xw;       /* [n][3] 3-D point                                            */
  double K[4];            /* fx, fy, cx, cy (kind 0)                                     */
  double P[12];           /* 3x4 row-major K*Tcw (kind 1)                                */
  double T0[16];          /* initial estimate                                            */
  double huber_delta;     /* (double)sqrtf(0.01f) kind 0 ; <= 0: no kernel (kind 1)      */
  double chi2_gate;       /* 0.01f                                                       */
  int32_t max_iterations; /* 100 / 200                                                   */
  int32_t pad;
} vdo_pose_problem;
int vdo_oracle_pose_optimize(const vdo_pose_problem* p, double T_out[16], uint8_t* inlier_out, vdo_lm_stats* stats);

/* ---- Tracking bookkeeping (tracking_logic_oracle.cpp) --------------------------------------*/
int vdo_oracle_dyn_obj_tracking(int n, const int32_t* sem_label, int32_t* obj_label_inout, const float* kx, const float* ky,
                                const float* depth, const float* flow3d, const int32_t* last_sem_label,
                                int n_last_obj, const int32_t* last_sem_pos, const int32_t* last_mod_label, const uint8_t* last_obj_stat,
                                int img_w, int img_h, int shrink_row, int shrink_col, float sf_mg_thres, float sf_ds_thres, float th_depth_obj,
                                int f_id, int32_t* max_id_inout,
                                int32_t* obj_off, int32_t* obj_idx, int32_t* obj_sem, int32_t* obj_mod);
int vdo_oracle_renew_object(int n_obj, const int32_t* inl_off, const int32_t* inl_idx, const uint8_t* obj_stat,
                            const int32_t* sem_pos, const int32_t* mod_label,
                            const float* cur_x, const float* cur_y, const int32_t* cur_obj_label,
                            int n_tmp, const float* tmp_x, const float* tmp_y, const float* tmp_depth, const int32_t* tmp_label,
                            const float* tmp_flow_x, const float* tmp_flow_y, const float* tmp_corr_x, const float* tmp_corr_y,
                            const int32_t* mask, const float* depth, const float* flow, int w, int h, int max_num_obj, int cap,
                            float* key_x, float* key_y, float* depth_out, int32_t* sem_out, float* flow_x, float* flow_y,
                            float* corr_x, float* corr_y, int32_t* dyn_inlier_id, int32_t* obj_label_out);
int vdo_oracle_update_mask(int n, const int32_t* last_sem_label, const float* last_corr_x, const float* last_corr_y,
                           const int32_t* mask_last, const float* flow_last, int w, int h, int32_t* mask_cur);
int vdo_oracle_build_tracks(int n_frames, const int32_t* asso_off, const int32_t* asso, const int32_t* feat_label,
                            int cap_tracks, int cap_pairs, int32_t* track_off, int32_t* pair_frame, int32_t* pair_feat, int32_t* obj_id);

/* ---- RANSAC initialiser (p3p_oracle.cpp): cv::solvePnPRansac(AP3P) restated up to the final refit --------*/
int vdo_oracle_quartic(double a4, double a3, double a2, double a1, double a0, double* roots);
int vdo_oracle_p3p(const double* f9, const double* P9, double* R_out, double* t_out);
void vdo_oracle_ransac_subsets(int n, int max_iters, int32_t* idx);
int vdo_oracle_p3p_ransac(int n, const double* X, const double* uv, const double* K4, int max_iters, double thr, double confidence,
                          double* T_out, uint8_t* inlier_out, int32_t* iters_run, int32_t* best_iter);
/* EPnP (4 control points) on all given points: T_out 4x4 camera-from-world, returns the mean reprojection error [px] */
double vdo_oracle_epnp(int n, const double* X, const double* uv, const double* K4, double* T_out);
/* the RANSAC above + (refit != 0) OpenCV 3.4's final EPnP re-estimation of the winning model on its inliers */
int vdo_oracle_pnp_ransac_refit(int n, const double* X, const double* uv, const double* K4, int max_iters, double thr, double confidence, int refit,
                                double* T_out, uint8_t* inlier_out, int32_t* iters_run, int32_t* best_iter);

/* ---- AP3P (ap3p_oracle.cpp): Ke & Roumeliotis' solver as OpenCV 3.4's ap3p.cpp lays it out - the solver the reference's calls name.
   The product runs Grunert (p3p_oracle.cpp restates it): these entries exist to compare the two.  coeffs5: a4 .. a0. */
int vdo_oracle_ap3p(const double* f9, const double* P9, double* R_out, double* t_out);
int vdo_oracle_ap3p_quartic(const double* coeffs5, double* roots4);
int vdo_oracle_ap3p_ransac(int n, const double* X, const double* uv, const double* K4, int max_iters, double thr, double confidence,
                           double* T_out, uint8_t* inlier_out, int32_t* iters_run, int32_t* best_iter);

/* ---- front-end (frontend_oracle.cpp) ------------------------------------------------------*/
typedef struct vdo_orb_params {   /* ORBextractor ctor arguments (include/ORBextractor.h:39-40) */
  int32_t n_features;     /* 2500 */
  float scale_factor;     /* 1.2  */
  int32_t n_levels;       /* 8    */
  int32_t ini_th;         /* 20   */
  int32_t min_th;         /* 7    */
} vdo_orb_params;
void vdo_oracle_depth_preprocess(float* depth, int64_t n, float bf, float factor);
void vdo_oracle_rgb2gray(const uint8_t* rgb, int64_t n_pixels, int channels, int rgb_order, uint8_t* gray);
int vdo_oracle_orb_level_sizes(const vdo_orb_params* p, int w, int h, int32_t* ws, int32_t* hs, int32_t* nfeat);
int vdo_oracle_orb_pyramid(const uint8_t* gray, int w, int h, const vdo_orb_params* p, uint8_t* levels_out);
int vdo_oracle_orb_fast_level(const uint8_t* gray, int w, int h, const vdo_orb_params* p, int level,
                              float* x, float* y, float* resp, int cap);
int vdo_oracle_orb_extract(const uint8_t* gray, int w, int h, const vdo_orb_params* p,
                           float* kx, float* ky, float* kresp, float* kangle, int32_t* koct, float* ksize, int cap);
/* the same + rotated BRIEF (computeOrbDescriptor, src/ORBextractor.cc:97-136; commented-out call site :1083-1091): desc [cap][32], nullable */
int vdo_oracle_orb_extract_desc(const uint8_t* gray, int w, int h, const vdo_orb_params* p,
                                float* kx, float* ky, float* kresp, float* kangle, int32_t* koct, float* ksize, int cap, uint8_t* desc);
void vdo_oracle_sincos_exact(float angle, float* s, float* c);
void vdo_oracle_orb_descriptor(const uint8_t* blurred, int w, float px, float py, float angle_deg, uint8_t* desc32);
const signed char* vdo_oracle_orb_pattern(void);
void vdo_oracle_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst);
int vdo_oracle_frame_static_filter_sampled(int n, const float* kx, const float* ky, const int32_t* mask, const float* depth, const float* flow,
                                           int w, int h, float th_depth,
                                           int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out);
int vdo_oracle_sample_keypoints(int rows, int cols, unsigned long long seed, float* x_out, float* y_out);   /* capacity 3000 */
int vdo_oracle_frame_static_filter(int n, const float* kx, const float* ky, const int32_t* koct,
                                   const int32_t* mask, const float* depth, const float* flow,
                                   int w, int h, float th_depth,
                                   int32_t* keep_idx, float* corr_x, float* corr_y, float* flow_x, float* flow_y, float* depth_out);
int vdo_oracle_frame_object_sample(const int32_t* mask, const float* depth, const float* flow, int w, int h,
                                   float th_depth_obj, int step, int cap,
                                   float* key_x, float* key_y, float* corr_x, float* corr_y,
                                   float* flow_x, float* flow_y, float* depth_out, int32_t* label);

/* ---- tracking-side stages (tracking_oracle.cpp) ------------------------------------------*/
void vdo_oracle_propagate_static(int n, const float* kx, const float* ky, const float* depth, int w, int h, float* depth_out);
void vdo_oracle_propagate_object(int n, const float* kx, const float* ky, const float* depth, const int32_t* mask, int w, int h,
                                 float th_obj, float* depth_out, int32_t* label_out);
void vdo_oracle_scene_flow(int n, const float* cur_x, const float* cur_y, const float* cur_d, const int32_t* cur_lab, const float* Tcw_cur,
                           const float* last_x, const float* last_y, const float* last_d, const int32_t* last_lab, const float* Tcw_last,
                           const float* K4, float* flow3d, int32_t* obj_label_inout);
void vdo_oracle_get3d_world(int n, const float* kx, const float* ky, const float* d, const float* K4, const float* Twc, float* xyz);
int vdo_oracle_renew_static(int n_tm, const int32_t* tm_sta, const float* stat_x, const float* stat_y,
                            int n_orb, const float* orb_x, const float* orb_y,
                            const int32_t* mask, const float* depth, const float* flow, int w, int h, int max_num_sta,
                            float* key_x, float* key_y, float* corr_x, float* corr_y, float* flow_x, float* flow_y,
                            int32_t* inlier_id, float* depth_out);
void vdo_oracle_mask_at(int n, const float* cx, const float* cy, const int32_t* mask, int w, int h, int32_t* out);
void vdo_oracle_mask_warp(const int32_t* mask_last, const float* flow_last, int w, int h, int32_t lab, int32_t* mask_cur);

/* ---- SE(3) helpers exposed for KATs --------------------------------------------*/
void vdo_oracle_se3_exp(const double u[6], double T16[16]);            /* SE3Quat::exp */
void vdo_oracle_iso_oplus(const double T12[12], const double d[6], double out12[12]); /* VertexSE3::oplusImpl */
void vdo_oracle_iso_to_mqt(const double T12[12], double e[6]);         /* toVectorMQT   */
void vdo_oracle_edge_se3_jac(const double Z12[12], const double Xi12[12], const double Xj12[12],
                             double e[6], double Ji[36], double Jj[36]);
void vdo_oracle_edge_prior_jac(const double Z12[12], const double X12[12], double e[6], double J[36]);
void vdo_oracle_edge_eb_jac(const double X12[12], const double p[3], const double z[3],
                            double e[3], double Jpose[18], double Jpoint[9]);
void vdo_oracle_edge_et_jac(const double H12[12], const double p1[3], const double p2[3], const double z[3],
                            double e[3], double Jp1[9], double Jp2[9], double Jh[18]);

/* more KAT exports for tests/test_ref_g2o.py (flow_oracle.cpp) */
void vdo_oracle_edge_flow2_jac(const double K4[4], const double Twl16[16], double depth, const double obs[2], const double flow_est[2], const double flow_meas[2],
                               const double T16[16], double err2[2], double Jpose12[12], double errp2[2]);
void vdo_oracle_huber(double delta, double e2, double rho2[2]);
void vdo_oracle_se3quat_oplus(const double T16[16], const double* u, double out16[16]);

#ifdef __cplusplus
}
#endif
#endif
