// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/oracle.py).
// CPU restatement of pcl::VoxelGrid<pcl::PointXYZ>::applyFilter (centroid down-sampling), the filter the reference runs
// right before the registration path (/root/reference/apps/scan_matching_odometry_nodelet.cpp:83-89,155-165;
// apps/prefiltering_nodelet.cpp:59-63).  PCL sorts (cell index, point index) pairs with std::sort on the cell index only, so
// the order of points inside a cell -- and with it the last bit of the float sums -- is unspecified upstream; this
// restatement fixes it to point-index order (stable sort).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "oracle_api.h"

extern "C" int64_t orc_voxel_grid(const float* xyz16, int64_t n, float leaf, float* out_xyz16) {
  const float inv = 1.0f / leaf;
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[1], -mn[2]};
  auto finite = [&](int64_t i) { return std::isfinite(xyz16[i * 4]) && std::isfinite(xyz16[i * 4 + 1]) && std::isfinite(xyz16[i * 4 + 2]); };
  for (int64_t i = 0; i < n; i++) {
    if (!finite(i)) continue;
    for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], xyz16[i * 4 + a]); mx[a] = std::max(mx[a], xyz16[i * 4 + a]); }
  }
  if (!(mn[0] <= mx[0])) return 0;
  int min_b[3], div_b[3];
  for (int a = 0; a < 3; a++) {
    min_b[a] = static_cast<int>(std::floor(mn[a] * inv));
    div_b[a] = static_cast<int>(std::floor(mx[a] * inv)) - min_b[a] + 1;
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  std::vector<std::pair<unsigned, int64_t>> iv;
  iv.reserve(n);
  for (int64_t i = 0; i < n; i++) {
    if (!finite(i)) continue;
    int idx = 0;
    for (int a = 0; a < 3; a++) idx += static_cast<int>(std::floor(xyz16[i * 4 + a] * inv) - static_cast<float>(min_b[a])) * mul[a];
    iv.emplace_back(static_cast<unsigned>(idx), i);
  }
  std::stable_sort(iv.begin(), iv.end(), [](const std::pair<unsigned, int64_t>& a, const std::pair<unsigned, int64_t>& b) { return a.first < b.first; });
  int64_t m = 0;
  size_t first = 0;
  while (first < iv.size()) {
    size_t last = first;
    float s[3] = {0.f, 0.f, 0.f};
    while (last < iv.size() && iv[last].first == iv[first].first) {
      for (int a = 0; a < 3; a++) s[a] += xyz16[iv[last].second * 4 + a];
      last++;
    }
    const float cnt = static_cast<float>(last - first);
    for (int a = 0; a < 3; a++) out_xyz16[m * 4 + a] = s[a] / cnt;
    out_xyz16[m * 4 + 3] = 1.0f;
    m++;
    first = last;
  }
  return m;
}

// CPU restatement of pcl::ApproximateVoxelGrid<pcl::PointXYZ>::applyFilter (PCL 1.8-1.12 filters/impl/approximate_voxel_grid.hpp),
// the reference's other down-sampling choice (/root/reference/apps/scan_matching_odometry_nodelet.cpp:90-96,
// apps/prefiltering_nodelet.cpp:64-69): one pass over the points in order through a 512-entry history table hashed by the
// cell coordinates; a point whose table slot holds ANOTHER cell flushes that cell's centroid (float sums, divided by the float
// count) to the output and takes the slot over; what is left in the table at the end is flushed in slot order.  PCL is absent
// from /root/reference and from this image: parity unpinned, like the rest of oracle/.
extern "C" int64_t orc_approx_voxel_grid(const float* xyz16, int64_t n, float leaf, float* out_xyz16) {
  constexpr int kHist = 512;
  struct He { int ix, iy, iz, count; float c[3]; };
  std::vector<He> hist(kHist);
  for (auto& e : hist) { e.count = 0; e.c[0] = e.c[1] = e.c[2] = 0.f; e.ix = e.iy = e.iz = 0; }
  const float inv = 1.0f / leaf;
  int64_t op = 0;
  auto flush = [&](He& e) {
    const float cnt = static_cast<float>(e.count);
    for (int a = 0; a < 3; a++) out_xyz16[op * 4 + a] = e.c[a] / cnt;
    out_xyz16[op * 4 + 3] = 1.0f;
    op++;
  };
  for (int64_t cp = 0; cp < n; cp++) {
    const float* p = xyz16 + cp * 4;
    const int ix = static_cast<int>(std::floor(p[0] * inv)), iy = static_cast<int>(std::floor(p[1] * inv)), iz = static_cast<int>(std::floor(p[2] * inv));
    const unsigned hash = (static_cast<unsigned>(ix) * 7171u + static_cast<unsigned>(iy) * 3079u + static_cast<unsigned>(iz) * 4231u) & (kHist - 1);
    He& e = hist[hash];
    if (e.count && (ix != e.ix || iy != e.iy || iz != e.iz)) {
      flush(e);
      e.count = 0;
      e.c[0] = e.c[1] = e.c[2] = 0.f;
    }
    e.ix = ix; e.iy = iy; e.iz = iz;
    e.count++;
    for (int a = 0; a < 3; a++) e.c[a] += p[a];
  }
  for (auto& e : hist)
    if (e.count) flush(e);
  return op;
}
