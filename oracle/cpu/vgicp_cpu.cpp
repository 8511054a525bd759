// ORACLE -- TEST INFRASTRUCTURE ONLY.  See vgicp_cpu.hpp.
#include "vgicp_cpu.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "linalg.hpp"
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

static inline uint64_t pack_coord(const int* c) {
  // 21 bits per axis, offset binary
  return (static_cast<uint64_t>(static_cast<uint32_t>(c[0] + (1 << 20)) & 0x1FFFFFu)) | (static_cast<uint64_t>(static_cast<uint32_t>(c[1] + (1 << 20)) & 0x1FFFFFu) << 21) |
         (static_cast<uint64_t>(static_cast<uint32_t>(c[2] + (1 << 20)) & 0x1FFFFFu) << 42);
}

// GaussianVoxelMap::voxel_coord: (x / resolution - 0.5).floor()
static inline void voxel_coord(const double* x, double resolution, int* c) {
  for (int a = 0; a < 3; a++) c[a] = static_cast<int>(std::floor(x[a] / resolution - 0.5));
}

void VgicpCpu::set_target(const float* xyz16, int64_t n) {
  GicpCpu::set_target(xyz16, n);
  map_valid = false;
}

void VgicpCpu::offset(int k, int* d) const {
  d[0] = d[1] = d[2] = 0;
  if (search_method == VGICP_DIRECT1) return;
  if (search_method == VGICP_DIRECT7) {
    // (0,0,0) (1,0,0) (-1,0,0) (0,1,0) (0,-1,0) (0,0,1) (0,0,-1)
    static const int o[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    d[0] = o[k][0]; d[1] = o[k][1]; d[2] = o[k][2];
    return;
  }
  d[0] = k / 9 - 1;
  d[1] = (k / 3) % 3 - 1;
  d[2] = k % 3 - 1;
}

// GaussianVoxelMap::create_voxelmap with AdditiveGaussianVoxel: points appended in index order, then finalize()
void VgicpCpu::build_voxelmap() {
  ensure_covariances();
  index.clear();
  voxels.clear();
  voxel_coords.clear();
  for (int64_t i = 0; i < nt; i++) {
    const float* p = target.data() + i * 4;
    const double x[3] = {static_cast<double>(p[0]), static_cast<double>(p[1]), static_cast<double>(p[2])};
    int c[3];
    voxel_coord(x, resolution, c);
    const uint64_t key = pack_coord(c);
    auto it = index.find(key);
    if (it == index.end()) {
      it = index.emplace(key, static_cast<int>(voxels.size())).first;
      voxels.emplace_back();
      voxel_coords.insert(voxel_coords.end(), c, c + 3);
    }
    GaussianVoxel& v = voxels[it->second];
    v.num_points++;
    for (int a = 0; a < 3; a++) v.mean[a] += x[a];
    const double* C = cov_t.data() + static_cast<size_t>(i) * 9;
    for (int a = 0; a < 9; a++) v.cov[a] += C[a];
  }
  for (auto& v : voxels) {
    for (int a = 0; a < 3; a++) v.mean[a] /= v.num_points;
    for (int a = 0; a < 9; a++) v.cov[a] /= v.num_points;
  }
  map_valid = true;
}

void VgicpCpu::dump_voxels(int32_t* coord3, int32_t* counts, double* mean3, double* cov9) const {
  std::vector<int> order(voxels.size());
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    const int32_t* ca = voxel_coords.data() + 3 * a;
    const int32_t* cb = voxel_coords.data() + 3 * b;
    if (ca[2] != cb[2]) return ca[2] < cb[2];
    if (ca[1] != cb[1]) return ca[1] < cb[1];
    return ca[0] < cb[0];
  });
  for (size_t k = 0; k < order.size(); k++) {
    const int v = order[k];
    std::memcpy(coord3 + 3 * k, voxel_coords.data() + 3 * v, 3 * sizeof(int32_t));
    counts[k] = voxels[v].num_points;
    std::memcpy(mean3 + 3 * k, voxels[v].mean, 3 * sizeof(double));
    std::memcpy(cov9 + 9 * k, voxels[v].cov, 9 * sizeof(double));
  }
}

// FastVGICP::update_correspondences: voxel look-ups at T * p (double) and the combined-covariance inverses
void VgicpCpu::update_voxel_correspondences(const double* T) {
  const int no = n_offsets();
  vcorr.assign(static_cast<size_t>(ns) * no, -1);
  vmahal.assign(static_cast<size_t>(ns) * no * 9, 0.0);
  double R[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[r * 3 + c] = T[r * 4 + c];
#pragma omp parallel for num_threads(threads()) schedule(guided, 8)
  for (int64_t i = 0; i < ns; i++) {
    const float* p = source.data() + i * 4;
    double ta[3];
    for (int r = 0; r < 3; r++) ta[r] = T[r * 4 + 0] * p[0] + T[r * 4 + 1] * p[1] + T[r * 4 + 2] * p[2] + T[r * 4 + 3];
    int c0[3];
    voxel_coord(ta, resolution, c0);
    const double* CA = cov_s.data() + static_cast<size_t>(i) * 9;
    double RC[9], RCRt[9];
    mat3_mul(R, CA, RC);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) RCRt[r * 3 + c] = RC[r * 3 + 0] * R[c * 3 + 0] + RC[r * 3 + 1] * R[c * 3 + 1] + RC[r * 3 + 2] * R[c * 3 + 2];
    for (int k = 0; k < no; k++) {
      int d[3];
      offset(k, d);
      const int c[3] = {c0[0] + d[0], c0[1] + d[1], c0[2] + d[2]};
      auto it = index.find(pack_coord(c));
      if (it == index.end()) continue;
      const size_t slot = static_cast<size_t>(i) * no + k;
      vcorr[slot] = it->second;
      const double* CB = voxels[it->second].cov;
      double RCR[9];
      for (int a = 0; a < 9; a++) RCR[a] = CB[a] + RCRt[a];
      inv3(RCR, vmahal.data() + slot * 9);
    }
  }
}

double VgicpCpu::linearize(const double* T, double* H, double* b) {
  evaluations++;
  ensure_covariances();
  if (!map_valid) build_voxelmap();
  update_voxel_correspondences(T);
  const int no = n_offsets();
  for (int k = 0; k < 36; k++) H[k] = 0;
  for (int k = 0; k < 6; k++) b[k] = 0;
  double sum_errors = 0.0;
  // sequential, in (point, offset) order: the reference sums per-thread partials in a run-dependent order
  for (int64_t i = 0; i < ns; i++) {
    const float* pa = source.data() + i * 4;
    double ta[3];
    for (int r = 0; r < 3; r++) ta[r] = T[r * 4 + 0] * pa[0] + T[r * 4 + 1] * pa[1] + T[r * 4 + 2] * pa[2] + T[r * 4 + 3];
    for (int k = 0; k < no; k++) {
      const size_t slot = static_cast<size_t>(i) * no + k;
      const int v = vcorr[slot];
      if (v < 0) continue;
      const GaussianVoxel& vx = voxels[v];
      const double w = std::sqrt(static_cast<double>(vx.num_points));
      double e[3];
      for (int r = 0; r < 3; r++) e[r] = vx.mean[r] - ta[r];
      const double* M = vmahal.data() + slot * 9;
      double Me[3];
      for (int r = 0; r < 3; r++) Me[r] = M[r * 3 + 0] * e[0] + M[r * 3 + 1] * e[1] + M[r * 3 + 2] * e[2];
      sum_errors += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
      const double J[3][6] = {{0, -ta[2], ta[1], -1, 0, 0}, {ta[2], 0, -ta[0], 0, -1, 0}, {-ta[1], ta[0], 0, 0, 0, -1}};
      double MJ[3][6];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 6; c++) MJ[r][c] = M[r * 3 + 0] * J[0][c] + M[r * 3 + 1] * J[1][c] + M[r * 3 + 2] * J[2][c];
      for (int r = 0; r < 6; r++) {
        for (int c = 0; c < 6; c++) H[r * 6 + c] += w * (J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c]);
        b[r] += w * (J[0][r] * Me[0] + J[1][r] * Me[1] + J[2][r] * Me[2]);
      }
    }
  }
  return sum_errors;
}

double VgicpCpu::compute_error(const double* T) {
  evaluations++;
  const int no = n_offsets();
  double sum_errors = 0.0;
  for (int64_t i = 0; i < ns; i++) {
    const float* pa = source.data() + i * 4;
    double ta[3];
    for (int r = 0; r < 3; r++) ta[r] = T[r * 4 + 0] * pa[0] + T[r * 4 + 1] * pa[1] + T[r * 4 + 2] * pa[2] + T[r * 4 + 3];
    for (int k = 0; k < no; k++) {
      const size_t slot = static_cast<size_t>(i) * no + k;
      const int v = vcorr[slot];
      if (v < 0) continue;
      const GaussianVoxel& vx = voxels[v];
      const double w = std::sqrt(static_cast<double>(vx.num_points));
      double e[3];
      for (int r = 0; r < 3; r++) e[r] = vx.mean[r] - ta[r];
      const double* M = vmahal.data() + slot * 9;
      double Me[3];
      for (int r = 0; r < 3; r++) Me[r] = M[r * 3 + 0] * e[0] + M[r * 3 + 1] * e[1] + M[r * 3 + 2] * e[2];
      sum_errors += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
    }
  }
  return sum_errors;
}

}  // namespace orc
