/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/README.md).
 * C entry points of the CPU restatement, loaded with ctypes by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg.  Nothing under delta_graph_slam_amd/ may include or load this. */
#ifndef ORACLE_API_H
#define ORACLE_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ndt_params {
  double resolution, step_size, outlier_ratio, transformation_epsilon, min_covar_eigvalue_mult;
  int32_t max_iterations, search_method, min_points_per_voxel, line_search, mt_max_step_iterations, num_threads, fix_hessian_d1, exp_libm;
  int32_t newton_solver, hessian_recompute_double, guess_rotation_polar;   /* round 4: NdtParams in ndt_cpu.hpp */
  int32_t cov_eigensolver, pad0;
} orc_ndt_params;

typedef struct orc_result {
  float T[16];
  int32_t converged, iterations, evaluations, pad;
  double score;
} orc_result;

void orc_ndt_default_params(orc_ndt_params* p);
void* orc_ndt_create(const orc_ndt_params* p);
void orc_ndt_destroy(void* h);
void orc_ndt_set_target(void* h, const float* xyz16, int64_t n);
void orc_ndt_set_source(void* h, const float* xyz16, int64_t n);
void orc_ndt_align(void* h, const float* guess16, orc_result* out, double* trajectory, int32_t* traj_len);
double orc_ndt_derivatives(void* h, const double* p6, const float* T16_or_null, double* g6, double* H36, int32_t compute_hessian);
/* voxel table dump: returns number of leaves; arrays sized by a first call with null pointers */
int64_t orc_ndt_voxels(void* h, int64_t* keys, int32_t* counts, int32_t* valid, double* mean3, double* cov9, double* icov9);
void orc_ndt_grid(void* h, int32_t* min_b3, int32_t* max_b3, int32_t* div_b3);
void orc_euler_angles_012(const float* T16, float* out3);
void orc_pose_to_matrix_f32(const double* p6, float* T16);
void orc_svd_solve6(const double* A36, const double* b6, double* x6);
void orc_jsvd_solve6(const double* A36, const double* b6, double* x6, int32_t* sweeps_rotations2);
void orc_affine_rotation_f32(const float* T16, float* R9_rowmajor);
void orc_ndt_hessian_double(void* h, const double* p6, double* H36);
double orc_det_exp(double x);
double orc_glibc_exp(double x);   /* linalg.hpp glibc_exp: std::exp(double) as glibc >= 2.28 computes it (FMA build) */
long long orc_glibc_exp_mismatches(long long n, uint64_t seed, double* first_bad);
float orc_glibc_expf(float x);   /* linalg.hpp glibc_expf: std::exp(float) as glibc >= 2.27 computes it on an FMA-capable x86-64 */
long long orc_glibc_expf_mismatches(uint32_t first_bits, uint32_t last_bits, uint32_t* first_bad);   /* ... against the host libm, float by float */
void orc_ldlt_solve6(const double* A36, const double* b6, double* x6);
void orc_sym_eig3(const double* A9, double* evals3, double* V9);
int orc_eigen_selfadjoint3(const double* A9, double* evals3, double* V9);   /* Eigen's SelfAdjointEigenSolver<Matrix3d>::compute restated; returns the QR steps */
int32_t orc_max_threads(void);

typedef struct orc_gicp_params {
  double transformation_epsilon, rotation_epsilon, max_correspondence_distance, lm_init_lambda_factor;
  int32_t max_iterations, k_correspondences, regularization, optimizer, lm_max_iterations, num_threads;
  int32_t cov_svd, pad0;
} orc_gicp_params;
void orc_gicp_default_params(orc_gicp_params* p);
void* orc_gicp_create(const orc_gicp_params* p);
/* FAST_VGICP (search_method 0 DIRECT1, 1 DIRECT7, 2 DIRECT27); drive it through the orc_gicp_* entry points */
void* orc_vgicp_create(const orc_gicp_params* p, double resolution, int32_t search_method);
/* voxel map dump in ascending (z, y, x) coordinate order; returns the voxel count (call with nulls first) */
int64_t orc_vgicp_voxels(void* h, int32_t* coord3, int32_t* counts, double* mean3, double* cov9);
void orc_gicp_destroy(void* h);
void orc_gicp_set_target(void* h, const float* xyz16, int64_t n);
void orc_gicp_set_source(void* h, const float* xyz16, int64_t n);
void orc_gicp_align(void* h, const float* guess16, orc_result* out);
/* pose as a row-major double 4x4 (Eigen::Isometry3d) */
double orc_gicp_linearize(void* h, const double* T16_rowmajor, double* H36, double* b6);
double orc_gicp_compute_error(void* h, const double* T16_rowmajor);
/* covariances: which = 0 source, 1 target; out 9 doubles per point */
void orc_gicp_covariances(void* h, int32_t which, double* out9);
void orc_gicp_correspondences(void* h, int32_t* corr, float* sq_dist);
double orc_fitness_score(const float* target, int64_t nt, const float* source, int64_t ns, const float* T16, double max_range,
                         double inlier_sq, int64_t* n_used, int64_t* n_inliers);
void orc_knn(const float* cloud, int64_t n, const float* queries, int64_t m, int32_t k, int32_t* idx, float* d2);
void orc_se3_exp(const double* a6, double* T16_rowmajor);
/* pcl::VoxelGrid centroid filter; out must hold n points; returns the number of cells */
int64_t orc_voxel_grid(const float* xyz16, int64_t n, float leaf, float* out_xyz16);
int64_t orc_approx_voxel_grid(const float* xyz16, int64_t n, float leaf, float* out_xyz16);   // pcl::ApproximateVoxelGrid; finite input

#ifdef __cplusplus
}
#endif
#endif
