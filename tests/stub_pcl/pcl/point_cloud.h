#pragma once
#include <cstdint>
#include <memory>
#include <vector>
namespace pcl {
template <typename PointT>
class PointCloud {
 public:
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  std::uint32_t width = 0, height = 1;
  bool is_dense = true;
  std::size_t size() const { return points.size(); }
  const PointT& at(std::size_t i) const { return points.at(i); }
  PointT& at(std::size_t i) { return points.at(i); }
  bool empty() const { return points.empty(); }
};
}  // namespace pcl
