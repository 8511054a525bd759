#pragma once
#include <cfloat>
#include <memory>
#include <vector>
#include <pcl/point_cloud.h>
namespace pcl { namespace search {
template <typename PointT>
class KdTree {  // brute force: exact, like FLANN with eps = 0 (stub scale only)
 public:
  using Ptr = std::shared_ptr<KdTree<PointT>>;
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { cloud_ = c; }
  int nearestKSearch(const PointT& q, int /*k = 1*/, std::vector<int>& idx, std::vector<float>& d2) const {
    idx.assign(1, -1); d2.assign(1, FLT_MAX);
    if (!cloud_) return 0;
    for (std::size_t i = 0; i < cloud_->points.size(); i++) {
      const PointT& p = cloud_->points[i];
      const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z, d = (dx * dx + dy * dy) + dz * dz;
      if (d < d2[0]) { d2[0] = d; idx[0] = (int)i; }
    }
    return 1;
  }
 private:
  typename PointCloud<PointT>::ConstPtr cloud_;
};
}}  // namespace pcl::search
