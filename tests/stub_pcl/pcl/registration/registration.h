#pragma once
// Shape of pcl::Registration (PCL 1.8 - 1.10) as far as the adapter and the reference call sites use it.
#include <cfloat>
#include <memory>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/search/kdtree.h>
namespace pcl {
template <typename PointSource, typename PointTarget, typename Scalar = float>
class Registration {
 public:
  using Matrix4 = Eigen::Matrix<Scalar, 4, 4>;
  using Ptr = std::shared_ptr<Registration<PointSource, PointTarget, Scalar>>;
  using KdTree = pcl::search::KdTree<PointTarget>;
  using KdTreePtr = typename KdTree::Ptr;
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;

  Registration() : tree_(new KdTree) { final_transformation_.setIdentity(); transformation_.setIdentity(); previous_transformation_.setIdentity(); }
  virtual ~Registration() {}
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; source_cloud_updated_ = true; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; target_cloud_updated_ = true; }
  KdTreePtr getSearchMethodTarget() const { return tree_; }
  Matrix4 getFinalTransformation() { return final_transformation_; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  bool hasConverged() const { return converged_; }
  double getFitnessScore(double max_range = DBL_MAX) {
    double score = 0; int nr = 0;
    std::vector<int> idx(1); std::vector<float> d2(1);
    for (const auto& p : input_->points) {
      PointSource q;
      for (int r = 0; r < 3; r++) q.data[r] = final_transformation_(r, 0) * p.x + final_transformation_(r, 1) * p.y + final_transformation_(r, 2) * p.z + final_transformation_(r, 3);
      tree_->nearestKSearch(q, 1, idx, d2);
      if (d2[0] <= max_range) { score += d2[0]; nr++; }
    }
    return nr > 0 ? score / nr : DBL_MAX;
  }
  void align(PointCloudSource& output) { align(output, Matrix4::Identity()); }
  void align(PointCloudSource& output, const Matrix4& guess) {
    if (!initCompute()) return;
    output.points = input_->points;
    output.width = (std::uint32_t)output.points.size();
    final_transformation_ = transformation_ = previous_transformation_ = Matrix4::Identity();
    converged_ = false;
    for (auto& p : output.points) p.data[3] = 1.0f;
    computeTransformation(output, guess);
  }

 protected:
  bool initCompute() {
    if (!target_ || target_->points.empty() || !input_ || input_->points.empty()) return false;
    if (target_cloud_updated_ && !force_no_recompute_) { tree_->setInputCloud(target_); target_cloud_updated_ = false; }
    return true;
  }
  virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;

  std::string reg_name_;
  KdTreePtr tree_;
  int nr_iterations_ = 0, max_iterations_ = 10;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  Matrix4 final_transformation_, transformation_, previous_transformation_;
  double transformation_epsilon_ = 0.0, corr_dist_threshold_ = 0.0;
  bool converged_ = false, target_cloud_updated_ = true, source_cloud_updated_ = true, force_no_recompute_ = false;
};
}  // namespace pcl
