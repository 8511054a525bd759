/* dgs_reg.h -- C ABI of libdgs_reg.so: MI355X (gfx950) scan registration for delta_graph_slam.
 *
 * This is the drop-in boundary for ONE hot path of KennyRotella/delta_graph_slam: the
 * pcl::Registration<pcl::PointXYZ,pcl::PointXYZ> object that
 *   /root/reference/src/hdl_graph_slam/registrations.cpp:22-124  (select_registration_method) builds, and that
 *   /root/reference/apps/scan_matching_odometry_nodelet.cpp:173-270,309-346 and
 *   /root/reference/include/hdl_graph_slam/loop_detector.hpp:119-173 drive.
 * Each entry point names the reference interface it replaces.  The C++ adapter that re-exposes this ABI as a
 * pcl::Registration subclass is include/dgs/hip_registration.hpp; INTEGRATION.md shows the factory patch.
 *
 * Conventions
 *   - extern "C", opaque handle, POD structs, plain pointers + counts, int status (0 = DGS_OK), never throws.
 *   - Clouds are pcl::PointXYZ arrays: float[n][4] = x, y, z, pad (16-byte stride; the pad value is ignored).
 *   - Transforms are Eigen::Matrix4f memory: 16 floats, COLUMN-major.
 *   - `on_device` != 0 means the pointer is a device (HBM) pointer valid on the handle's device; the call then
 *     works on the handle's stream and does not touch host memory.  Ordering contract for device pointers: the DATA must be
 *     complete when the call is made (the handle's stream is not ordered against the stream that produced it: synchronise that
 *     stream, or make the handle share it with dgs_set_stream), and the library has finished READING the buffer when the call
 *     returns (set_input_* copy it, align_batch / fitness calls read it in place), so it may be freed or reused at once.
 *     With host pointers the library copies at the call and never retains the pointer.
 *   - A handle is used by one thread at a time; different handles are independent (two live handles per
 *     process is the reference's normal case: odometry + loop detector).
 *   - There is NO CPU fallback: every call fails with DGS_ERR_HIP when no gfx950 device is usable.
 */
#ifndef DGS_REG_H
#define DGS_REG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGS_ABI_VERSION 5

typedef struct dgs_handle dgs_handle;
typedef struct dgs_cloud dgs_cloud; /* a cloud resident in HBM together with its NN index / covariances (see below) */

enum dgs_status {
  DGS_OK = 0,
  DGS_ERR_INVALID_ARGUMENT = 1,
  DGS_ERR_HIP = 2,            /* a HIP runtime call failed; dgs_last_error() has the text */
  DGS_ERR_NO_TARGET = 3,      /* align()/fitness before setInputTarget (PCL: "no input target dataset") */
  DGS_ERR_NO_SOURCE = 4,
  DGS_ERR_GRID_TOO_LARGE = 5, /* voxel index would overflow (pcl::VoxelGridCovariance: "Leaf size is too small") */
  DGS_ERR_UNSUPPORTED = 6
};

/* registration_method strings of registrations.cpp:26-124 that this library serves */
enum dgs_method {
  DGS_METHOD_NDT = 0,  /* "NDT_OMP": pclomp::NormalDistributionsTransform, registrations.cpp:101-120 */
  DGS_METHOD_GICP = 1, /* "FAST_GICP": fast_gicp::FastGICP, registrations.cpp:27-36 */
  DGS_METHOD_VGICP = 2 /* "FAST_VGICP": fast_gicp::FastVGICP, registrations.cpp:48-56 */
};

/* fast_gicp::NeighborSearchMethod of FastVGICP (voxel offsets searched around the voxel of T * p) */
enum dgs_vgicp_search { DGS_VGICP_DIRECT1 = 0, DGS_VGICP_DIRECT7 = 1, DGS_VGICP_DIRECT27 = 2 };

/* pclomp::NeighborSearchMethod (registrations.cpp:113-119), same enumerator order as upstream */
enum dgs_ndt_search { DGS_NDT_KDTREE = 0, DGS_NDT_DIRECT26 = 1, DGS_NDT_DIRECT7 = 2, DGS_NDT_DIRECT1 = 3 };

/* NDT step-length control (computeStepLengthMT).
 * MORE_THUENTE (default): the More-Thuente line search as SURVEY.md App. A states it and PCL >= 1.8.1 executes it
 *   (interval check `(step_max - step_min) < 0`).
 * FIXED_STEP: the pre-1.8.1 PCL initialisation `(step_max - step_min) > 0`, which marks the interval converged at
 *   once, so the trial loop never runs and every iteration is one derivative evaluation at the clamped Newton step.
 *   Kept because un-pinned ndt_omp forks differ here; it oscillates on planar scenes (DESIGN.md "NDT sensitivity"). */
enum dgs_ndt_line_search { DGS_NDT_LS_FIXED_STEP = 0, DGS_NDT_LS_MORE_THUENTE = 1 };

/* Order of the per-point arithmetic of NDT computeDerivatives / updateDerivatives (DESIGN.md "Parity").
 * FAST: the <= 7 voxels of a point are folded in float and projected once through the point Jacobian (1/3 of the
 *   flops, FMA contraction allowed, the library expf, Gauss-Jordan Newton solve): the same algebra as upstream, re-associated.
 *   Opt-in: 2.4 x the throughput of UPSTREAM, a few ill-conditioned pairs per 32 land outside 1e-4 m of a CPU run.
 * UPSTREAM (default since ABI 5): every float operation of upstream's per-voxel update in upstream's order, individually rounded (no FMA), a
 *   platform-independent exp, each voxel's increments added to the point's double totals, the full (not exactly symmetric)
 *   6x6 Hessian, Jacobi-SVD Newton solve; the points' totals are summed in the GPU's own fixed order, so only the order of
 *   the double summation differs from a CPU run of upstream.
 * UPSTREAM_SEQUENTIAL: as UPSTREAM, and the per-point totals are written out and summed in point-index order like
 *   upstream's final loop: every evaluation is bit-identical to the CPU restatement (slow: one lane per sum). */
enum dgs_ndt_strict_order { DGS_NDT_ORDER_FAST = 0, DGS_NDT_ORDER_UPSTREAM = 1, DGS_NDT_ORDER_UPSTREAM_SEQUENTIAL = 2 };

/* fast_gicp::RegularizationMethod, same enumerator order as upstream */
enum dgs_gicp_regularization {
  DGS_GICP_REG_NONE = 0, DGS_GICP_REG_MIN_EIG = 1, DGS_GICP_REG_NORMALIZED_MIN_EIG = 2,
  DGS_GICP_REG_PLANE = 3, DGS_GICP_REG_FROBENIUS = 4
};
enum dgs_gicp_optimizer { DGS_GICP_OPT_GAUSS_NEWTON = 0, DGS_GICP_OPT_LEVENBERG_MARQUARDT = 1 };

typedef struct dgs_params {
  uint32_t struct_size; /* sizeof(dgs_params), set by dgs_params_init */
  int32_t method;       /* dgs_method */
  int32_t device;       /* HIP device ordinal; -1 = the calling thread's current device */
  int32_t num_threads;  /* reg_num_threads (registrations.cpp:30,102): accepted, ignored on the GPU */

  /* setTransformationEpsilon / setMaximumIterations (registrations.cpp:31-32,110-111) */
  double transformation_epsilon; /* default 0.01 */
  int32_t maximum_iterations;    /* default 64 */

  /* ---- NDT (pclomp::NormalDistributionsTransform) ---- */
  int32_t ndt_search_method;        /* setNeighborhoodSearchMethod; default DGS_NDT_DIRECT7 (registrations.cpp:103) */
  double ndt_resolution;            /* setResolution; factory default 0.5 (registrations.cpp:93) */
  double ndt_step_size;             /* upstream default 0.1 */
  double ndt_outlier_ratio;         /* upstream default 0.55 */
  double ndt_min_covar_eigvalue_mult; /* VoxelGridCovariance default 0.01 */
  int32_t ndt_min_points_per_voxel; /* VoxelGridCovariance default 6 */
  int32_t ndt_line_search;          /* dgs_ndt_line_search, default DGS_NDT_LS_MORE_THUENTE */
  int32_t ndt_mt_max_step_iterations; /* default 10 */
  int32_t ndt_fix_hessian_d1;       /* 0 = upstream h_ang_d1 table (z-term +sy); 1 = exact (-sy) */
  int32_t ndt_strict_order;         /* dgs_ndt_strict_order, default DGS_NDT_ORDER_UPSTREAM */

  /* ---- GICP (fast_gicp::FastGICP) ---- */
  double gicp_max_correspondence_distance; /* setMaxCorrespondenceDistance; factory default 2.5 (registrations.cpp:33) */
  double gicp_rotation_epsilon;            /* upstream default 2e-3 */
  double gicp_lm_init_lambda_factor;       /* upstream default 1e-9 */
  int32_t gicp_correspondence_randomness;  /* setCorrespondenceRandomness (k), default 20 (registrations.cpp:34) */
  int32_t gicp_regularization;             /* default DGS_GICP_REG_PLANE */
  int32_t gicp_optimizer;                  /* default DGS_GICP_OPT_LEVENBERG_MARQUARDT */
  int32_t gicp_lm_max_iterations;          /* default 10 */
  /* FAST_VGICP (uses the gicp_* fields above except gicp_max_correspondence_distance: VGICP has no distance gate) */
  int32_t vgicp_search_method;             /* default DGS_VGICP_DIRECT1 (FastVGICP constructor) */
  double vgicp_resolution;                 /* setResolution(reg_resolution), factory default 1.0 (registrations.cpp:52) */
  /* ---- six details of un-vendored upstream code behind named switches (ABI 5; [UPSTREAM-RECALL]).  Each
   * defaults to what the published upstream source does as far as it can be recalled; 0 restores the stand-in of ABI <= 4. ---- */
  int32_t ndt_newton_solver;            /* upstream evaluation orders (ndt_strict_order >= 1): 1 = Eigen::JacobiSVD's own two-sided Jacobi sequence
                                           (what computeTransformation's `sv.solve(-score_gradient)` runs); 0 = one-sided Hestenes Jacobi.  The FAST
                                           order keeps its Gauss-Jordan step either way. */
  int32_t ndt_hessian_recompute_double; /* computeStepLengthMT ends with computeHessian when the line search took extra trials: 1 = PCL's double-precision
                                           computeHessian / updateHessian, as ndt_omp kept it; 0 = the float computeDerivatives pass again.  Upstream
                                           orders only (the FAST order re-runs its own float pass). */
  int32_t ndt_guess_rotation_polar;     /* initial pose vector: Euler angles of Affine3f::rotation(), i.e. of the polar factor of the guess's 3x3
                                           (a float JacobiSVD) = 1, of the raw 3x3 = 0.  All orders. */
  int32_t ndt_exp_glibc;                /* upstream evaluation orders: updateDerivatives' `std::exp(float)`: 1 = glibc's expf (>= 2.27, the x86-64 FMA build)
                                           restated operation for operation -- the restatement is compared with the build image's libm on every
                                           float in [-104, 0] (tests/test_oracle_round4.py); 0 = the platform-independent polynomial of ABI <= 4
                                           (correctly rounded but for ~1e-9 of the arguments: NOT what a libm returns).  The FAST order uses the
                                           device library's expf either way.  (Was reserved0: same struct size.) */
  int32_t ndt_cov_eigensolver;          /* voxel covariances (VoxelGridCovariance: `eigensolver.compute(leaf.cov_)`, all orders): 1 = Eigen 3.3's
                                           SelfAdjointEigenSolver<Matrix3d>::compute restated -- scaling, the 3x3 Householder tridiagonalisation,
                                           implicit QR steps with Wilkinson's shift, selection sort; 0 = cyclic Jacobi (ABI <= 4).  The eigenvectors
                                           matter where a flat voxel's covariance is rebuilt from them (eigenvalue clamp). */
  int32_t gicp_cov_jacobi_svd;          /* FAST_GICP / FAST_VGICP covariance regularisation (calculate_covariances): 1 = Eigen::JacobiSVD<Matrix3d> restated (two-sided
                                           Jacobi; the routine fast_gicp calls), 0 (default) = the symmetric eigen-decomposition of ABI <= 4 -- the same factors up to
                                           rounding on regular neighbourhoods.  Not the default because JacobiSVD skips rotations below 2 eps maxDiag: on a rank-deficient
                                           neighbourhood (duplicated points, points on a line) the basis of the null space -- and with it U diag(1, 1, 1e-3) V^T -- then
                                           jumps with the last bit of the input covariance, which the device and a CPU sum in different orders.  (Was reserved1.) */
} dgs_params;

/* What the callers read back after align(): hasConverged(), getFinalTransformation(), and the
 * getFitnessScore() the loop detector takes per candidate (loop_detector.hpp:145-155). */
typedef struct dgs_result {
  float final_transformation[16]; /* getFinalTransformation(), column-major */
  int32_t converged;              /* hasConverged() */
  int32_t iterations;             /* nr_iterations_ */
  int32_t evaluations;            /* derivative / linearisation passes executed */
  int32_t status;                 /* dgs_status of this registration (batch entries fail independently) */
  double score;                   /* NDT: score (trans_probability * Ns); GICP: final sum of Mahalanobis errors */
  double fitness;                 /* getFitnessScore(max_range) when requested, else NaN */
} dgs_result;

/* Defaults = the reference factory's defaults for `method` (registrations.cpp:27-36 / 93-120). */
int dgs_params_init(dgs_params* params, int32_t method);

/* new pclomp::NormalDistributionsTransform / fast_gicp::FastGICP + setters (registrations.cpp:29-35,105-119) */
int dgs_create(const dgs_params* params, dgs_handle** out);
void dgs_destroy(dgs_handle* h);
const char* dgs_last_error(const dgs_handle* h); /* never NULL; "" when the last call succeeded */
int dgs_abi_version(void);

/* Run all of this handle's work on a caller-owned hipStream_t (NULL = a stream the handle owns). */
int dgs_set_stream(dgs_handle* h, void* hip_stream);
/* Block until everything this handle enqueued has finished. */
int dgs_synchronize(dgs_handle* h);

/* registration->setInputTarget(cloud): scan_matching_odometry_nodelet.cpp:180,254; loop_detector.hpp:124.
 * NDT: builds the voxel-Gaussian model (VoxelGridCovariance).  GICP: exact-NN index; covariances lazily. */
int dgs_set_input_target(dgs_handle* h, const float* xyz16, int64_t n, int32_t on_device);
/* registration->setInputSource(cloud): scan_matching_odometry_nodelet.cpp:185; loop_detector.hpp:138 */
int dgs_set_input_source(dgs_handle* h, const float* xyz16, int64_t n, int32_t on_device);

/* registration->align(*aligned, guess): scan_matching_odometry_nodelet.cpp:218; loop_detector.hpp:145.
 * `guess16` NULL = identity.  `aligned_xyz16` (nullable) receives final_transformation * source,
 * n_source points (host or device per `aligned_on_device`).  A registration that fails internally reports
 * converged = 0 and final_transformation = guess (the reference treats that as "skip this frame",
 * scan_matching_odometry_nodelet.cpp:222-226). */
int dgs_align(dgs_handle* h, const float* guess16, dgs_result* out, float* aligned_xyz16, int32_t aligned_on_device);

/* registration->getFitnessScore(max_range): loop_detector.hpp:148, scan_matching_odometry_nodelet.cpp:318.
 * Mean squared exact-1-NN distance of final_transformation * source to the target over points with
 * d^2 <= max_range (PCL compares the SQUARED distance with max_range; the reference's in-tree twin does the
 * same, information_matrix_calculator.cpp:97); DBL_MAX when no point qualifies. */
int dgs_get_fitness_score(dgs_handle* h, double max_range, double* score);

/* The inlier loop of publish_scan_matching_status (scan_matching_odometry_nodelet.cpp:321-332):
 * fraction of final_transformation * source points whose exact 1-NN squared distance to the target is
 * < max_sq_dist (the reference passes 0.5 * 0.5). */
int dgs_get_inlier_fraction(dgs_handle* h, double max_sq_dist, double* fraction);

/* registration->getSearchMethodTarget()->nearestKSearch(pt, 1, idx, sqdist) for m query points
 * (scan_matching_odometry_nodelet.cpp:327).  Exact; ties resolve to the lowest target index. */
int dgs_nearest_search_target(dgs_handle* h, const float* queries_xyz16, int64_t m, int32_t on_device,
                              int32_t* indices, float* sq_dists);

/* The candidate loop of LoopDetector::matching (loop_detector.hpp:137-156) as ONE batched call against the
 * current target: for c in [0, n): setInputSource(sources[c]); align(guess[c]); getFitnessScore(max_range).
 * `sources[c]` / `sizes[c]` may be ragged; `guesses16` is n*16 floats (NULL = identity); fitness is computed
 * when `compute_fitness` != 0.  results[c].status reports per-candidate failures.  The arg-min over
 * (converged, fitness) stays with the caller (loop_detector.hpp:149-155). */
int dgs_align_batch(dgs_handle* h, int32_t n, const float* const* sources, const int64_t* sizes, int32_t on_device,
                    const float* guesses16, int32_t compute_fitness, double fitness_max_range, dgs_result* results);

/* ---- device-resident clouds: KeyFrame::cloud kept in HBM -------------------------------------------------
 * The loop detector registers the same keyframe clouds again and again (every graph_update_interval tick,
 * /root/reference/apps/delta_graph_slam_nodelet.cpp:147-148,816; clouds live in KeyFrame::cloud, keyframe.hpp:51).  A dgs_cloud
 * is uploaded once; the exact-NN index and the GICP covariances derived from it are built on first use and kept, which is
 * what fast_gicp does per object when setInputSource sees the same pointer again.  A cloud belongs to the device of the
 * handle that created it, may be used by any handle on that device (one at a time) and must outlive the calls using it. */
int dgs_cloud_create(dgs_handle* h, const float* xyz16, int64_t n, int32_t on_device, dgs_cloud** out);
void dgs_cloud_destroy(dgs_cloud* cloud);
int64_t dgs_cloud_size(const dgs_cloud* cloud);
/* setInputTarget / setInputSource without a copy */
int dgs_set_input_target_cloud(dgs_handle* h, dgs_cloud* cloud);
int dgs_set_input_source_cloud(dgs_handle* h, dgs_cloud* cloud);
/* dgs_align_batch over resident clouds (loop_detector.hpp:137-156 with cached candidates) */
int dgs_align_batch_clouds(dgs_handle* h, int32_t n, dgs_cloud* const* sources, const float* guesses16, int32_t compute_fitness,
                           double fitness_max_range, dgs_result* results);

/* LoopDetector::find_candidates (/root/reference/include/hdl_graph_slam/loop_detector.hpp:83-111) over n keyframes on the device
 * (second half): keyframe i is a candidate iff
 *   new_accum_distance - accum_distance[i] >= accum_distance_thresh            (:93-96: "traveled distance ... too small" skips)
 *   and sqrt(dx * dx + dy * dy) <= distance_thresh, (dx, dy) = xy[i] - new_xy  (:98-105: Eigen's norm() of the 2-D difference, double)
 * `xy` holds n pairs (x, y) = node->estimate().translation().head<2>().  `indices` receives the candidates' positions in KEYFRAME ORDER
 * (the order matters: loop_detector.hpp:149 breaks score ties in favour of the later candidate); *n_out is always the full count,
 * DGS_ERR_INVALID_ARGUMENT when it exceeds `capacity`.  The "too close to the last loop edge" test (:85-87) stays with the caller.
 * Poses change at every graph optimisation, so the arrays travel with the call (host pointers, or device pointers with on_device). */
int dgs_find_loop_candidates(dgs_handle* h, const double* accum_distance, const double* xy, int64_t n, int32_t on_device, double new_accum_distance,
                             const double* new_xy, double accum_distance_thresh, double distance_thresh, int32_t* indices, int64_t capacity, int64_t* n_out);

/* InformationMatrixCalculator::calc_fitness_score(cloud1, cloud2, relpose, max_range)
 * (/root/reference/src/hdl_graph_slam/information_matrix_calculator.cpp:77-108; called per odometry edge and per loop
 * edge, apps/delta_graph_slam_nodelet.cpp:572,820): exact-NN index over cloud1, cloud2 transformed by the float cast of
 * relpose (column-major 16 floats, NULL = identity), mean squared NN distance over points with d^2 <= max_range, DBL_MAX
 * when none qualifies.  Uses buffers of its own: the handle's registration target / source / result are untouched. */
int dgs_calc_fitness_score(dgs_handle* h, const float* cloud1_xyz16, int64_t n1, const float* cloud2_xyz16, int64_t n2,
                           int32_t on_device, const float* relpose16, double max_range, double* score);

/* pcl::VoxelGrid<PointXYZ> centroid down-sampling, the step right before the path
 * (/root/reference/apps/scan_matching_odometry_nodelet.cpp:83-89,155-165; apps/prefiltering_nodelet.cpp:59-63):
 * cell = floor(p / leaf) as PCL indexes it, output = centroid of every occupied cell in cell-index order, pad lane = 1.
 * Sums are formed in float in point-index order (PCL's std::sort leaves the order inside a cell unspecified).
 * `*n_out` receives the number of cells; fails with DGS_ERR_INVALID_ARGUMENT when out_capacity is smaller. */
int dgs_voxel_grid_filter(dgs_handle* h, const float* in_xyz16, int64_t n, int32_t in_on_device, float leaf_size, float* out_xyz16,
                          int64_t out_capacity, int32_t out_on_device, int64_t* n_out);

/* pcl::ApproximateVoxelGrid<PointXYZ>, the reference's other down-sampling choice
 * (/root/reference/apps/scan_matching_odometry_nodelet.cpp:90-96; apps/prefiltering_nodelet.cpp:64-69): upstream's single pass
 * through a 512-entry history table hashed by the cell, reproduced exactly -- same output points (float sums in point order,
 * divided by the float count) in the same order: a cell is emitted when another cell evicts it from its slot, the rest in slot
 * order at the end.  Same argument conventions as dgs_voxel_grid_filter; the input must be finite (upstream does not check). */
int dgs_approx_voxel_grid_filter(dgs_handle* h, const float* in_xyz16, int64_t n, int32_t in_on_device, float leaf_size, float* out_xyz16,
                                 int64_t out_capacity, int32_t out_on_device, int64_t* n_out);

/* ---- several GPUs of one process: the candidate loop of LoopDetector::matching sharded across devices -------------------------
 * The reference runs loop detection inside the nodelet manager process under main_thread_mutex
 * (/root/reference/apps/delta_graph_slam_nodelet.cpp:797,816; candidate loop loop_detector.hpp:137-156), so the multi-GPU form a
 * nodelet can link is ONE process driving G devices.  A dgs_group owns one dgs_handle, one host thread and one stream per listed
 * device.  dgs_group_set_input_target = loop_detector.hpp:124 on every member (G parallel host->device copies).
 * dgs_group_align_batch deals candidate c to member c mod G, every member runs its share as one dgs_align_batch, the fixed-size
 * result records are exchanged with ncclAllGather (RCCL over xGMI; communicators from ncclCommInitAll, library loaded with
 * dlopen) and returned in ORIGINAL candidate order; best_index / best_score (nullable) receive the arg-min of
 * loop_detector.hpp:126-156 over (converged, fitness) with its tie rule (on an equal score the later candidate wins), -1 /
 * DBL_MAX when no candidate converged.  The fitness_score_thresh test (:162) stays with the caller.  A group whose device list
 * names one device twice (a one-GPU rehearsal), or that cannot load RCCL, gathers on the host instead: same results.
 * Every member writes the records of its share ON THE DEVICE (optimiser state + fitness sums -> record), so the all-gather sends what
 * the kernels left in HBM; the group's functions leave the caller's current HIP device unchanged.
 * Sources and the target are host arrays (KeyFrame::cloud) or, below, clouds resident on the group's devices; a group is used by
 * one thread at a time. */
typedef struct dgs_group dgs_group;
int dgs_group_create(const dgs_params* params, const int32_t* devices, int32_t n_devices, dgs_group** out); /* params->device is ignored */
void dgs_group_destroy(dgs_group* g);
const char* dgs_group_last_error(const dgs_group* g);
int32_t dgs_group_size(const dgs_group* g);
int32_t dgs_group_uses_rccl(const dgs_group* g);               /* 1: the group holds RCCL communicators */
int32_t dgs_group_rccl_ranks(const dgs_group* g);              /* ncclCommCount of the group's communicator (0: no RCCL) */
int32_t dgs_group_last_gather_used_rccl(const dgs_group* g);   /* 1: the last dgs_group_align_batch exchanged its records with ncclAllGather */
dgs_handle* dgs_group_member(dgs_group* g, int32_t k);         /* member k's handle (e.g. for dgs_profile_*); owned by the group */
int dgs_group_set_input_target(dgs_group* g, const float* xyz16, int64_t n);
int dgs_group_align_batch(dgs_group* g, int32_t n, const float* const* sources, const int64_t* sizes, const float* guesses16,
                          int32_t compute_fitness, double fitness_max_range, dgs_result* results, int32_t* best_index, double* best_score);

/* Keyframe clouds resident on the group's devices.  KeyFrame::cloud is set once and never written again
 * (/root/reference/include/hdl_graph_slam/keyframe.hpp:51) and the same keyframes are candidates tick after tick
 * (apps/delta_graph_slam_nodelet.cpp:816 -> loop_detector.hpp:59-70), so a nodelet uploads each keyframe ONCE:
 *   owner >= 0: one copy, on member owner mod G (a candidate keyframe: owner = its id keeps the shares even);
 *   owner = -1: a copy on every member (the new keyframe, which is every member's target).
 * dgs_group_set_input_target_cloud = loop_detector.hpp:124 on every member; members without a copy take one from a holder, device
 * to device over xGMI, and keep it (a keyframe is a target first and a candidate on later ticks).
 * dgs_group_align_batch_clouds = dgs_group_align_batch without any upload: candidate c runs on member c mod G when that member
 * holds its cloud, else on the cloud's owner; results, arg-min and tie rule as above.  A dgs_group_cloud belongs to the group it
 * was created with (may be destroyed before or after it). */
typedef struct dgs_group_cloud dgs_group_cloud;
int dgs_group_cloud_create(dgs_group* g, const float* xyz16, int64_t n, int32_t owner, dgs_group_cloud** out);
void dgs_group_cloud_destroy(dgs_group_cloud* cloud);
int64_t dgs_group_cloud_size(const dgs_group_cloud* cloud);
int32_t dgs_group_cloud_copies(const dgs_group_cloud* cloud);   /* members that hold a copy */
/* Drops every copy but ONE: the one on member `owner mod G` when that member holds one, else the first holder's (owner < 0: the first
 * holder's).  A new keyframe is every member's target for one tick (dgs_group_set_input_target_cloud leaves a copy on every member); as
 * a candidate of later ticks it needs the one copy on its owner only -- the caller trims it when the tick is over.  A copy still bound to
 * its member as target or source is detached like in dgs_cloud_destroy. */
int dgs_group_cloud_trim(dgs_group* g, dgs_group_cloud* cloud, int32_t owner);
int dgs_group_set_input_target_cloud(dgs_group* g, dgs_group_cloud* cloud);
int dgs_group_align_batch_clouds(dgs_group* g, int32_t n, dgs_group_cloud* const* sources, const float* guesses16, int32_t compute_fitness,
                                 double fitness_max_range, dgs_result* results, int32_t* best_index, double* best_score);

/* ---- measurement hooks (bench.py roofline leg; not part of the reference surface) ---------------------- */
enum dgs_kernel_id {
  DGS_K_NDT_DERIVATIVES = 0, DGS_K_NDT_SOLVE = 1, DGS_K_NDT_VOXEL_BUILD = 2, DGS_K_NN_SEARCH = 3,
  DGS_K_GICP_LINEARIZE = 4, DGS_K_GICP_COVARIANCE = 5, DGS_K_TRANSFORM = 6, DGS_K_COUNT = 7
};
/* When enabled, every launch of a tracked kernel is bracketed by hipEvents on the launch stream. */
int dgs_profile_enable(dgs_handle* h, int32_t enable);
/* Sum of event-measured durations and number of launches since the last dgs_profile_reset (synchronises). */
int dgs_profile_get(dgs_handle* h, int32_t kernel_id, double* total_ms, int64_t* launches);
int dgs_profile_reset(dgs_handle* h);
/* Counts describing the current problem, for algorithmic-byte accounting:
 * out[0] = target points, out[1] = source points, out[2] = valid voxels V, out[3] = occupied voxels,
 * out[4] = voxel grid cells, out[5] = derivative evaluations of the last align / batch (sum over pairs). */
int dgs_get_counts(dgs_handle* h, int64_t out[8]);

/* Test hooks: single evaluations on the device, so tests can compare kernels with the oracle directly. */
/* NDT computeDerivatives at pose p (6 doubles).  T16 NULL = build the float transform from p. */
int dgs_ndt_derivatives(dgs_handle* h, const double* p6, const float* T16, double* score, double* grad6, double* hess36);
/* NDT computeHessian in PCL's double-precision form at pose p (the pass computeStepLengthMT ends with when a line search took extra
 * trials; dgs_params.ndt_hessian_recompute_double).  Upstream evaluation orders only (DGS_ERR_UNSUPPORTED otherwise). */
int dgs_ndt_hessian_double(dgs_handle* h, const double* p6, double* hess36);
/* NDT pose (x, y, z, rx, ry, rz) after every outer iteration of pair `pair` of the last align / align_batch;
 * poses6 holds up to 72 x 6 doubles, *len receives the number written (entry 0 is the initial guess). */
int dgs_ndt_get_trajectory(dgs_handle* h, int32_t pair, double* poses6, int32_t* len);
/* NDT voxel table dump.  First call with NULL arrays returns the number of occupied voxels in *n. */
int dgs_ndt_get_voxels(dgs_handle* h, int64_t* n, int64_t* keys, int32_t* counts, int32_t* valid, double* mean3,
                       double* icov9);

/* Test hook: squared exact 1-NN distances of m query points to the target through the index the fitness pass uses (the
 * one-lane-per-query grid of nn_grid.hip with its tree fallback); must equal dgs_nearest_search_target's sq_dists bit for bit. */
int dgs_nn_fitness_distances(dgs_handle* h, const float* queries_xyz16, int64_t m, int32_t on_device, float* sq_dists);

/* GICP regularised k-NN covariances (FastGICP::calculate_covariances): which = 0 source, 1 target;
 * cov9 receives 9 doubles (row-major 3x3) per point. */
int dgs_gicp_get_covariances(dgs_handle* h, int32_t which, double* cov9);
/* GICP FastGICP::linearize (error_only = 0: new correspondences at the pose; returns sum of errors, H 6x6, b 6) or
 * FastGICP::compute_error (error_only = 1: correspondences / Mahalanobis matrices of the last linearisation).
 * The pose is a row-major double 4x4 (Eigen::Isometry3d).  With DGS_METHOD_VGICP the same calls are FastVGICP's. */
int dgs_gicp_linearize(dgs_handle* h, const double* T16_rowmajor, int32_t error_only, double* error, double* hess36, double* b6);
/* Test hook (FAST_VGICP): the target's Gaussian voxel map in ascending (z, y, x) voxel-coordinate order: coord3 int32[3], counts,
 * mean double[3], cov double[9] per voxel; *n_voxels is always set, arrays are filled when capacity suffices. */
int dgs_vgicp_get_voxels(dgs_handle* h, int64_t capacity, int32_t* coord3, int32_t* counts, double* mean3, double* cov9, int64_t* n_voxels);

#ifdef __cplusplus
}
#endif
#endif /* DGS_REG_H */
