// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/oracle.py).
//
// CPU restatement of fast_gicp::FastGICP on fast_gicp::LsqRegistration (SMRT-AIST/fast_gicp, un-vendored and
// un-pinned: /root/reference/README.md:21-22, docker/noetic/Dockerfile:14-15), the object the reference builds at
// /root/reference/src/hdl_graph_slam/registrations.cpp:27-36 and drives through setInputTarget / setInputSource /
// align at apps/scan_matching_odometry_nodelet.cpp:180,185,218 and include/hdl_graph_slam/loop_detector.hpp:124,138,145.
// Restated from SURVEY.md Appendix B (published algorithm: Segal, Haehnel, Thrun, "Generalized-ICP", RSS 2009;
// Levenberg-Marquardt driver as in fast_gicp's lsq_registration).  Also holds the pcl::Registration::getFitnessScore
// restatement (in-tree twin: src/hdl_graph_slam/information_matrix_calculator.cpp:77-108).
#pragma once
#include <cstdint>
#include <vector>

#include "kdtree.hpp"

namespace orc {

enum GicpReg { GICP_REG_NONE = 0, GICP_REG_MIN_EIG = 1, GICP_REG_NORMALIZED_MIN_EIG = 2, GICP_REG_PLANE = 3, GICP_REG_FROBENIUS = 4 };
enum GicpOpt { GICP_OPT_GN = 0, GICP_OPT_LM = 1 };

struct GicpParams {
  double transformation_epsilon = 0.01;       // registrations.cpp:31 (upstream default 5e-4)
  double rotation_epsilon = 2e-3;             // upstream default
  double max_correspondence_distance = 2.5;   // registrations.cpp:33
  double lm_init_lambda_factor = 1e-9;
  int max_iterations = 64;                    // registrations.cpp:32
  int k_correspondences = 20;                 // registrations.cpp:34
  int regularization = GICP_REG_PLANE;        // fast_gicp constructor default
  int optimizer = GICP_OPT_LM;
  int lm_max_iterations = 10;
  int num_threads = 0;
  int cov_svd = 0;   // covariance regularisation: 1 = Eigen::JacobiSVD<Matrix3d> restated (two-sided Jacobi, linalg.hpp jacobi_svd_square<double, 3>), what
                     // fast_gicp's calculate_covariances calls; 0 (default) = symmetric eigen-decomposition: the same factors up to rounding on regular
                     // neighbourhoods; on rank-deficient ones JacobiSVD's rotation threshold makes the result jump with the last bit of the input
};

struct GicpResult {
  float T[16];  // column-major
  int converged, iterations, evaluations;
  double error;
};

class GicpCpu {
 public:
  explicit GicpCpu(const GicpParams& p) : prm(p) {}
  virtual ~GicpCpu() = default;
  virtual void set_target(const float* xyz16, int64_t n);
  void set_source(const float* xyz16, int64_t n);
  GicpResult align(const float* guess_colmajor16);
  // single linearisation / error evaluation at a double 4x4 (row-major) pose, for tests
  virtual double linearize(const double* T4x4, double* H36, double* b6);
  virtual double compute_error(const double* T4x4);
  void ensure_covariances();

  GicpParams prm;
  std::vector<float> target, source;
  int64_t nt = 0, ns = 0;
  KdTree tree_t, tree_s;
  std::vector<double> cov_t, cov_s;  // 9 doubles (row-major 3x3) per point
  std::vector<int> corr;
  std::vector<float> sq_dist;
  std::vector<double> mahal;  // 9 per source point
  int evaluations = 0;

 protected:
  void calc_covariances(const std::vector<float>& cloud, int64_t n, const KdTree& tree, std::vector<double>& covs);
  void update_correspondences(const double* T);
  int threads() const;
};

// pcl::Registration::getFitnessScore(max_range): mean squared 1-NN distance of T*source to target over d^2 <= max_range
double fitness_score(const float* target, int64_t nt, const float* source, int64_t ns, const float* T_colmajor16, double max_range,
                     double inlier_sq, int64_t* n_used, int64_t* n_inliers);

void se3_exp(const double* a6, double* T4x4);

}  // namespace orc
