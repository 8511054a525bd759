// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/oracle.py).
//
// CPU restatement of fast_gicp::FastVGICP (voxelised GICP; Koide, Yokozuka, Oishi, Banno, "Voxelized GICP for fast and
// accurate 3D point cloud registration", ICRA 2021) on fast_gicp::LsqRegistration, the object the reference builds at
// /root/reference/src/hdl_graph_slam/registrations.cpp:48-56 (registration_method "FAST_VGICP": setNumThreads, setResolution
// (reg_resolution, 1.0), setTransformationEpsilon, setMaximumIterations, setCorrespondenceRandomness) -- SURVEY.md §8f-4.
// fast_gicp is un-vendored and un-pinned (README.md:21-22); restated from the published algorithm and the class layout:
//   * target: k-NN covariances as FastGICP, then a GaussianVoxelMap: voxel coordinate floor(x / resolution - 0.5) per axis,
//     ADDITIVE accumulation (mean = sum of points / n, cov = sum of point covariances / n, n = points in the voxel);
//   * correspondences: every source point x voxel offsets (DIRECT1: own voxel; DIRECT7: + 6 face neighbours; DIRECT27: 3x3x3)
//     around the voxel of T * p (all double); no distance gate; Mahalanobis = (cov_voxel + R cov_p R^T)^-1;
//   * cost: sum of sqrt(n_voxel) * e^T M e with e = mean_voxel - T p; J = [skew(T p) | -I] scaled the same way;
//   * optimiser: the LM / GN driver of LsqRegistration, shared with GicpCpu.
#pragma once
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "gicp_cpu.hpp"

namespace orc {

enum VgicpSearch { VGICP_DIRECT1 = 0, VGICP_DIRECT7 = 1, VGICP_DIRECT27 = 2 };

struct GaussianVoxel {
  int num_points = 0;
  double mean[3] = {0, 0, 0};
  double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};

class VgicpCpu : public GicpCpu {
 public:
  VgicpCpu(const GicpParams& p, double resolution, int search_method) : GicpCpu(p), resolution(resolution), search_method(search_method) {}
  void set_target(const float* xyz16, int64_t n) override;
  double linearize(const double* T4x4, double* H36, double* b6) override;
  double compute_error(const double* T4x4) override;
  void build_voxelmap();
  int64_t voxel_count() const { return static_cast<int64_t>(voxels.size()); }
  // dump in ascending (z, y, x) coordinate order
  void dump_voxels(int32_t* coord3, int32_t* counts, double* mean3, double* cov9) const;

  double resolution;
  int search_method;
  bool map_valid = false;
  std::unordered_map<uint64_t, int> index;  // packed coordinate -> voxels[]
  std::vector<GaussianVoxel> voxels;
  std::vector<int32_t> voxel_coords;        // 3 per voxel
  // per correspondence (source point, offset): voxel id or -1, Mahalanobis 3x3
  std::vector<int> vcorr;
  std::vector<double> vmahal;
  int n_offsets() const { return search_method == VGICP_DIRECT1 ? 1 : search_method == VGICP_DIRECT7 ? 7 : 27; }

 private:
  void update_voxel_correspondences(const double* T);
  void offset(int k, int* d) const;
};

}  // namespace orc
