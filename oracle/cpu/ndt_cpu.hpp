// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/README.md).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call this.
//
// CPU restatement of pclomp::NormalDistributionsTransform (koide3/ndt_omp, un-vendored and un-pinned:
// /root/reference/README.md:21-22, docker/noetic/Dockerfile:14-15), the object the reference builds at
// /root/reference/src/hdl_graph_slam/registrations.cpp:101-120 and drives through
// setInputTarget/setInputSource/align at apps/scan_matching_odometry_nodelet.cpp:180,185,218 and
// include/hdl_graph_slam/loop_detector.hpp:124,138,145.  The upstream source is not on disk; the algorithm
// is restated from SURVEY.md Appendix A (published algorithm: Magnusson 2009, ch. 6; More & Thuente 1994).
//
// Arithmetic follows upstream: per-point math in float, per-point totals and the final sums in double,
// per-point results stored and summed in index order (OpenMP over points, schedule(guided, 8)).
#pragma once
#include <cstdint>
#include <vector>
#include <unordered_map>

namespace orc {

enum NdtSearch { NDT_KDTREE = 0, NDT_DIRECT26 = 1, NDT_DIRECT7 = 2, NDT_DIRECT1 = 3 };  // pclomp enum order
enum NdtLineSearch { NDT_LS_FIXED_STEP = 0, NDT_LS_MORE_THUENTE = 1 };

struct NdtParams {
  double resolution = 1.0;               // pclomp default; reference sets reg_resolution (registrations.cpp:93,112)
  double step_size = 0.1;                // upstream default, never changed by the reference
  double outlier_ratio = 0.55;           // upstream default
  double transformation_epsilon = 0.01;  // registrations.cpp:110
  int max_iterations = 64;               // registrations.cpp:111
  int search_method = NDT_DIRECT7;       // registrations.cpp:103,113-119
  int min_points_per_voxel = 6;          // VoxelGridCovariance default
  double min_covar_eigvalue_mult = 0.01; // VoxelGridCovariance default
  int line_search = NDT_LS_MORE_THUENTE; // see ndt_cpu.cpp: computeStepLengthMT
  int mt_max_step_iterations = 10;
  int fix_hessian_d1 = 0;                // 0 = upstream table (h_ang_d1 z-term +sy), 1 = exact second derivative (-sy)
  int num_threads = 0;                   // registrations.cpp:102,106-108 (0 = all cores)
  int exp_libm = 1;                      // std::exp(float) of updateDerivatives: 1 = glibc's expf restated (linalg.hpp glibc_expf: equal to the image's libm on every float in [-104, 0]), 0 = det_expf (rounds 1-3's stand-in), 2 = the host libm's expf itself
  // ---- round 4: three upstream details that rounds 1-3 replaced by stand-ins (all [UPSTREAM-RECALL]) ----
  int newton_solver = 1;                 // 1 = Eigen::JacobiSVD's own two-sided Jacobi sequence (linalg.hpp jsvd_solve6); 0 = one-sided Hestenes Jacobi (rounds 1-3)
  int hessian_recompute_double = 1;      // computeStepLengthMT's closing computeHessian: 1 = PCL's double-precision computeHessian / updateHessian
                                         // (double x, x' - mean, inverse covariance, double angle tables, ONE pass accumulating straight into the
                                         // Hessian in index order), as ndt_omp kept it; 0 = the float computeDerivatives pass run again (rounds 1-3)
  int guess_rotation_polar = 1;          // initial p: Euler angles of Affine3f::rotation() = the polar factor of the guess's 3x3 (a float JacobiSVD);
                                         // 0 = Euler angles of the raw 3x3 (rounds 1-3)
  int cov_eigensolver = 1;               // voxel covariances: 1 = Eigen::SelfAdjointEigenSolver<Matrix3d>::compute restated (tridiagonalisation + implicit QR,
                                         // linalg.hpp eigen_selfadjoint3), 0 = cyclic Jacobi (rounds 1-3's stand-in)
};

struct Leaf {
  int nr_points = 0;
  double sum[3] = {0, 0, 0};
  double sq[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  float centroid[3] = {0, 0, 0};  // float centroid (voxel "output" cloud, KDTREE mode)
  double mean[3];
  double cov[9];
  double icov[9];
  double evals[3];
  bool valid = false;
};

struct NdtResult {
  float T[16];  // column-major 4x4 (Eigen::Matrix4f layout)
  int converged;
  int iterations;   // nr_iterations_
  int evaluations;  // computeDerivatives calls (+ hessian recomputations)
  int hessian_recomputes;  // of which: computeHessian passes at the end of a line search that took extra trials
  double score;     // final score (trans_probability * Ns)
};

class NdtCpu {
 public:
  explicit NdtCpu(const NdtParams& p) : prm(p) {}
  void set_target(const float* xyz16, int64_t n);
  void set_source(const float* xyz16, int64_t n);
  NdtResult align(const float* guess_colmajor16, double* trajectory /* nullable, [max_iter+2][6] */, int* traj_len);
  // one computeDerivatives evaluation at pose p (cloud transformed by T(p) built the upstream way)
  double derivatives(const double p[6], double g[6], double H[36], bool compute_hessian = true);
  // same, but the cloud is transformed by an explicit float 4x4 (the first evaluation of align uses the guess)
  double derivatives_with(const float* T_colmajor16, const double p[6], double g[6], double H[36], bool compute_hessian);
  // computeHessian(hessian, trans_cloud, p) in PCL's double form (hessian_recompute_double); uses the angle tables of the LAST
  // compute_angle_derivatives call, as upstream does ("unnecessary because only used after regular derivative calculation")
  void hessian_double_with(const float* T_colmajor16, double H[36]);
  // test hook: angle tables at p, then hessian_double_with
  void hessian_double(const double p[6], double H[36]);

  NdtParams prm;
  std::vector<float> target, source;  // xyz16
  int64_t nt = 0, ns = 0;
  // voxel grid
  std::unordered_map<int64_t, Leaf> leaves;
  int min_b[3], max_b[3], div_b[3];
  int64_t divb_mul[3];
  float leaf_size[3], inv_leaf_size[3];
  double gauss_d1 = 0, gauss_d2 = 0;
  int evaluations = 0;

 private:
  void compute_angle_derivatives(const double p[6]);
  int neighbours(const float xt[3], const Leaf** out) const;
  float j_ang[8][3];
  float h_ang[15][3];
  double j_ang_d[8][3];   // the double vectors j_ang_a_ .. h_ang_f3_ upstream keeps beside the float matrices
  double h_ang_d[15][3];
};

// Eigen::Matrix3f::eulerAngles(0,1,2) (Eigen 3.3 algorithm) on the rotation block of a col-major float 4x4.
void euler_angles_012(const float* T_colmajor16, float out[3]);
// Translation(p0..2) * AngleAxis(p3,X) * AngleAxis(p4,Y) * AngleAxis(p5,Z) in float, col-major 4x4.
void pose_to_matrix_f32(const double p[6], float* T_colmajor16);

}  // namespace orc
