#pragma once
namespace pcl {
struct alignas(16) PointXYZ {
  union { float data[4]; struct { float x, y, z; }; };
  PointXYZ() : data{0.f, 0.f, 0.f, 1.f} {}
};
}  // namespace pcl
