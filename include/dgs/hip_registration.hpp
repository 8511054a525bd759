// dgs::HipRegistration -- pcl::Registration<PointSource, PointTarget> over libdgs_reg.so (include/dgs_reg.h).
//
// This is the object a patched hdl_graph_slam::select_registration_method returns for registration_method
// "NDT_HIP" / "FAST_GICP_HIP" (reference factory: src/hdl_graph_slam/registrations.cpp:22-124).  It is
// consumed unchanged through the base-class pointer by
//   apps/scan_matching_odometry_nodelet.cpp:180,185,218,222,228,318,327   and
//   include/hdl_graph_slam/loop_detector.hpp:124,138,145,148,149,155.
// Header-only; needs PCL (pcl/registration/registration.h) and Eigen at the USER's build, nothing else.
//
// Contract kept from the reference's registration objects:
//   * setInputTarget / setInputSource take the caller's shared cloud; the points are copied to HBM at the call
//     (pcl::PointXYZ is already the 16-byte x, y, z, pad record the ABI wants), the pointer is also kept in the base class;
//   * align(out, guess) -> pcl::Registration::align -> computeTransformation(out, guess) below; `out` receives
//     final_transformation * source; failure of any kind (no device, HIP error) never throws: converged_ = false and
//     final_transformation_ = guess, which the callers already treat as "skip" (scan_matching_odometry_nodelet.cpp:222-226,
//     loop_detector.hpp:149);
//   * getFitnessScore(max_range) shadows pcl::Registration's (non-virtual there) and is answered on the device; call it
//     through the derived type (INTEGRATION.md section 3/4 patch the two call sites with a downcast) or use
//     dgs_get_fitness_score -- through a base-class pointer PCL's own single-threaded CPU loop runs on the same
//     final_transformation_ and the base kd-tree.  After setKeepPclTree(false) that tree is no longer rebuilt; it is then
//     pointed at a one-point sentinel cloud at 1e30, so a base-pointer getFitnessScore() returns DBL_MAX ("no score")
//     instead of a value computed against a stale target, and nearestKSearch reports an infinite distance.
#pragma once

#include <cfloat>
#include <cstring>
#include <string>

#include <pcl/registration/registration.h>

#include "../dgs_reg.h"

namespace dgs {

template <typename PointSource, typename PointTarget>
class HipRegistration : public pcl::Registration<PointSource, PointTarget, float> {
 public:
  using Base = pcl::Registration<PointSource, PointTarget, float>;
  using PointCloudSource = typename Base::PointCloudSource;
  using PointCloudSourceConstPtr = typename Base::PointCloudSourceConstPtr;
  using PointCloudTargetConstPtr = typename Base::PointCloudTargetConstPtr;
  using Matrix4 = typename Base::Matrix4;

  static_assert(sizeof(PointSource) == 16 && sizeof(PointTarget) == 16, "clouds must be 16-byte x, y, z, pad points (pcl::PointXYZ)");

  explicit HipRegistration(dgs_method method) {
    dgs_params_init(&params_, method);
    this->reg_name_ = (method == DGS_METHOD_GICP) ? "dgs::HipRegistration<FAST_GICP>" : (method == DGS_METHOD_VGICP) ? "dgs::HipRegistration<FAST_VGICP>"
                                                                                        : "dgs::HipRegistration<NDT>";
    // the reference's setters below write into params_; PCL's own setters (epsilon, iterations, distance) are read at align()
    this->transformation_epsilon_ = params_.transformation_epsilon;
    this->max_iterations_ = params_.maximum_iterations;
    this->corr_dist_threshold_ = params_.gicp_max_correspondence_distance;
  }
  ~HipRegistration() override {
    if (handle_) dgs_destroy(handle_);
  }
  HipRegistration(const HipRegistration&) = delete;
  HipRegistration& operator=(const HipRegistration&) = delete;

  // ---- the setters registrations.cpp calls on ndt_omp / fast_gicp objects (:30-34, :106-118) ----
  void setNumThreads(int n) { params_.num_threads = n; dirty_ = true; }
  void setResolution(float r) { params_.ndt_resolution = r; params_.vgicp_resolution = r; dirty_ = true; }  // NDT leaf / VGICP voxel size
  void setNeighborSearchMethod(int method) { params_.vgicp_search_method = method; dirty_ = true; }        // FastVGICP (dgs_vgicp_search)
  void setNeighborhoodSearchMethod(int method) { params_.ndt_search_method = method; dirty_ = true; }  // dgs_ndt_search == pclomp order
  void setCorrespondenceRandomness(int k) { params_.gicp_correspondence_randomness = k; dirty_ = true; }
  void setStepSize(double s) { params_.ndt_step_size = s; dirty_ = true; }
  // dgs_ndt_strict_order: 0 fast (re-associated, opt-in), 1 upstream operation order (default), 2 + index-order sums (validation, dgs_reg.h)
  void setNdtStrictOrder(int order) { params_.ndt_strict_order = order; dirty_ = true; }
  // the pieces of the upstream order, one by one (all on by default; dgs_reg.h): Eigen's two-sided JacobiSVD sequence for the Newton
  // step, PCL's double computeHessian after a line search, the guess's rotation as Affine3f::rotation() takes it, std::exp(float) as glibc computes it
  void setNdtUpstreamFidelity(bool jacobi_svd, bool hessian_double, bool guess_polar, bool exp_glibc = true, bool cov_eigen_qr = true) {
    params_.ndt_newton_solver = jacobi_svd;
    params_.ndt_hessian_recompute_double = hessian_double;
    params_.ndt_guess_rotation_polar = guess_polar;
    params_.ndt_exp_glibc = exp_glibc;
    params_.ndt_cov_eigensolver = cov_eigen_qr;
    dirty_ = true;
  }
  void setOulierRatio(double r) { params_.ndt_outlier_ratio = r; dirty_ = true; }  // (sic) upstream spelling
  void setRotationEpsilon(double e) { params_.gicp_rotation_epsilon = e; dirty_ = true; }
  void setRegularizationMethod(int m) { params_.gicp_regularization = m; dirty_ = true; }
  void setDevice(int ordinal) { params_.device = ordinal; dirty_ = true; }
  // false: skip PCL's CPU kd-tree rebuild in initCompute() on every new target (then use getInlierFraction() instead of
  // getSearchMethodTarget()->nearestKSearch())
  void setKeepPclTree(bool keep) {
    keep_pcl_tree_ = keep;
    this->force_no_recompute_ = !keep;
    if (!keep) park_pcl_tree();
    else this->target_cloud_updated_ = true;   // PCL rebuilds its tree at the next align()
  }

  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {
    Base::setInputTarget(cloud);
    target_dirty_ = true;
    if (!keep_pcl_tree_) park_pcl_tree();
  }
  void setInputSource(const PointCloudSourceConstPtr& cloud) override {
    Base::setInputSource(cloud);
    source_dirty_ = true;
  }

  // registration->getFitnessScore(max_range) on the device (see the header comment about base-class pointers)
  double getFitnessScore(double max_range = DBL_MAX) {
    double s = DBL_MAX;
    if (!handle_ || dgs_get_fitness_score(handle_, max_range, &s) != DGS_OK) return DBL_MAX;
    return s;
  }
  // the inlier loop of publish_scan_matching_status (scan_matching_odometry_nodelet.cpp:321-332) in one call
  double getInlierFraction(double max_sq_dist) {
    double f = 0.0;
    if (!handle_ || dgs_get_inlier_fraction(handle_, max_sq_dist, &f) != DGS_OK) return 0.0;
    return f;
  }
  const dgs_result& lastResult() const { return last_; }
  std::string lastError() const { return handle_ ? dgs_last_error(handle_) : dgs_last_error(nullptr); }
  dgs_handle* handle() { return handle_; }

 protected:
  bool ensure_handle() {
    params_.transformation_epsilon = this->transformation_epsilon_;
    params_.maximum_iterations = this->max_iterations_;
    params_.gicp_max_correspondence_distance = this->corr_dist_threshold_;
    if (handle_ && !dirty_ && params_.transformation_epsilon == applied_.transformation_epsilon &&
        params_.maximum_iterations == applied_.maximum_iterations &&
        params_.gicp_max_correspondence_distance == applied_.gicp_max_correspondence_distance)
      return true;
    if (handle_) dgs_destroy(handle_);
    handle_ = nullptr;
    if (dgs_create(&params_, &handle_) != DGS_OK) return false;
    applied_ = params_;
    dirty_ = false;
    target_dirty_ = source_dirty_ = true;
    return true;
  }

  void computeTransformation(PointCloudSource& output, const Matrix4& guess) override {
    this->converged_ = false;
    this->nr_iterations_ = 0;
    this->final_transformation_ = guess;
    if (!ensure_handle()) return;
    if (target_dirty_ && this->target_) {
      if (dgs_set_input_target(handle_, reinterpret_cast<const float*>(this->target_->points.data()), (int64_t)this->target_->points.size(), 0) != DGS_OK) return;
      target_dirty_ = false;
    }
    if (source_dirty_ && this->input_) {
      if (dgs_set_input_source(handle_, reinterpret_cast<const float*>(this->input_->points.data()), (int64_t)this->input_->points.size(), 0) != DGS_OK) return;
      source_dirty_ = false;
    }
    output.points.resize(this->input_->points.size());
    if (dgs_align(handle_, guess.data(), &last_, reinterpret_cast<float*>(output.points.data()), 0) != DGS_OK) return;
    std::memcpy(this->final_transformation_.data(), last_.final_transformation, sizeof(float) * 16);
    this->transformation_ = this->final_transformation_;
    this->converged_ = last_.converged != 0;
    this->nr_iterations_ = last_.iterations;
  }

  // the base kd-tree is not maintained (setKeepPclTree(false)): make that visible instead of leaving the previous target in it
  void park_pcl_tree() {
    if (!sentinel_) {
      typename pcl::PointCloud<PointTarget>::Ptr c(new pcl::PointCloud<PointTarget>());
      c->points.resize(1);
      c->points[0].x = c->points[0].y = c->points[0].z = 1e30f;
      c->width = 1;
      sentinel_ = c;
    }
    if (this->tree_) this->tree_->setInputCloud(sentinel_);
  }

  typename pcl::PointCloud<PointTarget>::ConstPtr sentinel_;
  dgs_params params_{};
  dgs_params applied_{};
  dgs_handle* handle_ = nullptr;
  dgs_result last_{};
  bool dirty_ = true, target_dirty_ = true, source_dirty_ = true, keep_pcl_tree_ = true;
};

}  // namespace dgs
