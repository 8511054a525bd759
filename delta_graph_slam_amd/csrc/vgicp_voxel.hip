// FAST_VGICP target model: fast_gicp::GaussianVoxelMap::create_voxelmap with AdditiveGaussianVoxel, built on the device.
//
// Replaces what fast_gicp::FastVGICP::linearize does on its first call after setInputTarget (the object the reference builds at
// /root/reference/src/hdl_graph_slam/registrations.cpp:48-56): every target point goes to the voxel floor(x / resolution - 0.5)
// (double arithmetic), a voxel's mean is the mean of its points and its covariance the mean of its points' regularised
// k-NN covariances.  Same machinery as the NDT voxel build (ndt_voxel.hip): key per point, stable radix sort (so a voxel's
// points are summed in point-index order, as upstream appends them), run-length encode, one lane per voxel, dense
// cell -> voxel table over the target's AABB.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "handle.h"

namespace dgs {

__device__ __host__ inline int vgicp_coord(double x, double resolution) { return (int)floor(x / resolution - 0.5); }

__global__ __launch_bounds__(kBlock) void vgicp_key_kernel(const float4* __restrict__ pts, int64_t n, VgicpMap m, uint32_t* __restrict__ keys,
                                                           uint32_t* __restrict__ vals) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  uint32_t key = 0xFFFFFFFFu;
  if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    const int c0 = vgicp_coord((double)p.x, m.resolution) - m.min_c[0];
    const int c1 = vgicp_coord((double)p.y, m.resolution) - m.min_c[1];
    const int c2 = vgicp_coord((double)p.z, m.resolution) - m.min_c[2];
    key = (uint32_t)(c0 + c1 * m.mul1 + c2 * m.mul2);
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

// one lane per occupied voxel: AdditiveGaussianVoxel::append over the voxel's points in index order, then finalize()
__global__ __launch_bounds__(kBlock) void vgicp_finalize_kernel(const float4* __restrict__ pts, const double* __restrict__ cov6,
                                                                const uint32_t* __restrict__ order, const uint32_t* __restrict__ run_keys,
                                                                const int* __restrict__ run_counts, const int* __restrict__ run_offsets,
                                                                const int* __restrict__ scalars, VgicpMap m, int* __restrict__ cell2vox,
                                                                VgicpVoxel* __restrict__ vox) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= scalars[0]) return;
  const uint32_t key = run_keys[r];
  VgicpVoxel v;
  v.n = 0;
  v.w = 0.0;
  for (int a = 0; a < 3; a++) { v.mean[a] = 0.0; v.coord[a] = 0; }
  for (int a = 0; a < 6; a++) v.cov[a] = 0.0;
  if (key != 0xFFFFFFFFu) {
    const int cnt = run_counts[r], off = run_offsets[r];
    for (int j = 0; j < cnt; j++) {
      const uint32_t i = order[off + j];
      const float4 p = pts[i];
      v.mean[0] += (double)p.x; v.mean[1] += (double)p.y; v.mean[2] += (double)p.z;
      const double* c = cov6 + (size_t)i * 6;
#pragma unroll
      for (int a = 0; a < 6; a++) v.cov[a] += c[a];
    }
    const double np = (double)cnt;
    for (int a = 0; a < 3; a++) v.mean[a] /= np;
    for (int a = 0; a < 6; a++) v.cov[a] /= np;
    v.n = cnt;
    v.w = sqrt(np);
    v.coord[0] = (int)(key % (uint32_t)m.mul1) + m.min_c[0];
    v.coord[1] = (int)((key / (uint32_t)m.mul1) % (uint32_t)m.div[1]) + m.min_c[1];
    v.coord[2] = (int)(key / (uint32_t)m.mul2) + m.min_c[2];
    cell2vox[key] = r;
  }
  vox[r] = v;
}

int vgicp_build_map(dgs_handle* h) {
  const int64_t n = h->nt;
  hipStream_t st = h->stream;
  h->vmap = VgicpMap{};
  h->vmap.resolution = h->prm.vgicp_resolution;
  h->vmap.search = h->prm.vgicp_search_method;
  h->vmap.n_offsets = h->prm.vgicp_search_method == DGS_VGICP_DIRECT1 ? 1 : h->prm.vgicp_search_method == DGS_VGICP_DIRECT7 ? 7 : 27;
  h->vmap_valid = false;
  h->vmap_voxels = 0;
  if (n == 0) return DGS_OK;
  int rc = gicp_ensure_target_covariance(h);
  if (rc) return rc;
  int slot = prof_begin(h, DGS_K_NDT_VOXEL_BUILD);
  float hmm[6];
  rc = cloud_minmax(h, h->tgt->pts.ptr, n, hmm);
  if (rc) return rc;
  VgicpMap& m = h->vmap;
  if (!(hmm[0] <= hmm[3])) {  // no finite point: an empty map, every look-up misses
    prof_end(h, DGS_K_NDT_VOXEL_BUILD, slot);
    m.div[0] = m.div[1] = m.div[2] = 0;
    h->vmap_valid = true;
    return DGS_OK;
  }
  int64_t cells = 1;
  for (int a = 0; a < 3; a++) {
    m.min_c[a] = vgicp_coord((double)hmm[a], m.resolution);   // the coordinate is monotone in x
    const int max_c = vgicp_coord((double)hmm[3 + a], m.resolution);
    m.div[a] = max_c - m.min_c[a] + 1;
    cells *= m.div[a];
    if (cells > INT32_MAX) {
      prof_end(h, DGS_K_NDT_VOXEL_BUILD, slot);
      h->err = "voxel resolution is too small for the input dataset. Integer indices would overflow.";
      return DGS_ERR_GRID_TOO_LARGE;
    }
  }
  m.mul1 = m.div[0];
  m.mul2 = m.div[0] * m.div[1];
  DGS_HIP_TRY(h, h->vcell2vox.reserve((size_t)cells));
  DGS_HIP_TRY(h, h->key_in.reserve(n));
  DGS_HIP_TRY(h, h->key_out.reserve(n));
  DGS_HIP_TRY(h, h->val_in.reserve(n));
  DGS_HIP_TRY(h, h->val_out.reserve(n));
  DGS_HIP_TRY(h, h->run_keys.reserve(n));
  DGS_HIP_TRY(h, h->run_counts.reserve(n));
  DGS_HIP_TRY(h, h->run_offsets.reserve(n));
  DGS_HIP_TRY(h, h->dev_scalars.reserve(8));
  DGS_HIP_TRY(h, h->vvox.reserve(n));
  size_t t1 = 0, t2 = 0, t3 = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, h->key_in.ptr, h->key_out.ptr, h->val_in.ptr, h->val_out.ptr, (int)n, 0, 32, st);
  (void)hipcub::DeviceRunLengthEncode::Encode(nullptr, t2, h->key_out.ptr, h->run_keys.ptr, h->run_counts.ptr, h->dev_scalars.ptr, (int)n, st);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t3, h->run_counts.ptr, h->run_offsets.ptr, (int)n, st);
  DGS_HIP_TRY(h, h->cub_temp.reserve(std::max(t1, std::max(t2, t3)) + 256));

  DGS_HIP_TRY(h, hipMemsetAsync(h->vcell2vox.ptr, 0xFF, (size_t)cells * sizeof(int), st));
  DGS_HIP_TRY(h, hipMemsetAsync(h->dev_scalars.ptr, 0, 8 * sizeof(int), st));
  DGS_HIP_TRY(h, hipMemsetAsync(h->run_counts.ptr, 0, (size_t)n * sizeof(int), st));
  const int nb = (int)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(vgicp_key_kernel, dim3(nb), dim3(kBlock), 0, st, h->tgt->pts.ptr, n, m, h->key_in.ptr, h->val_in.ptr);
  size_t tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(h->cub_temp.ptr, tb, h->key_in.ptr, h->key_out.ptr, h->val_in.ptr, h->val_out.ptr, (int)n, 0, 32, st));
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRunLengthEncode::Encode(h->cub_temp.ptr, tb, h->key_out.ptr, h->run_keys.ptr, h->run_counts.ptr, h->dev_scalars.ptr,
                                                       (int)n, st));
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceScan::ExclusiveSum(h->cub_temp.ptr, tb, h->run_counts.ptr, h->run_offsets.ptr, (int)n, st));
  hipLaunchKernelGGL(vgicp_finalize_kernel, dim3(nb), dim3(kBlock), 0, st, h->tgt->pts.ptr, h->tgt->cov.ptr, h->val_out.ptr, h->run_keys.ptr,
                     h->run_counts.ptr, h->run_offsets.ptr, h->dev_scalars.ptr, m, h->vcell2vox.ptr, h->vvox.ptr);
  prof_end(h, DGS_K_NDT_VOXEL_BUILD, slot);
  DGS_HIP_TRY(h, hipGetLastError());
  // number of runs (the last one may be the run of non-finite points)
  if (ensure_pinned(h, 4096) != DGS_OK) return DGS_ERR_HIP;
  int* hs = reinterpret_cast<int*>(h->pinned);
  DGS_HIP_TRY(h, hipMemcpyAsync(hs, h->dev_scalars.ptr, sizeof(int), hipMemcpyDeviceToHost, st));
  DGS_HIP_TRY(h, hipStreamSynchronize(st));
  h->vmap_voxels = hs[0];
  m.cell2vox = h->vcell2vox.ptr;
  m.vox = h->vvox.ptr;
  h->vmap_valid = true;
  return DGS_OK;
}

// Test hook: the voxel map in ascending (z, y, x) coordinate order (= ascending key), runs of non-finite points dropped.
int vgicp_voxels(dgs_handle* h, int64_t capacity, int32_t* coord3, int32_t* counts, double* mean3, double* cov9, int64_t* n_voxels) {
  if (!h->vmap_valid) {
    int rc = vgicp_build_map(h);
    if (rc) return rc;
  }
  std::vector<VgicpVoxel> v((size_t)h->vmap_voxels);
  DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (!v.empty()) DGS_HIP_TRY(h, hipMemcpy(v.data(), h->vvox.ptr, v.size() * sizeof(VgicpVoxel), hipMemcpyDeviceToHost));
  int64_t k = 0;
  for (const VgicpVoxel& x : v) {
    if (x.n <= 0) continue;
    if (k < capacity && coord3) {
      for (int a = 0; a < 3; a++) { coord3[3 * k + a] = x.coord[a]; mean3[3 * k + a] = x.mean[a]; }
      counts[k] = x.n;
      const double* c = x.cov;
      const double c9[9] = {c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]};
      std::memcpy(cov9 + 9 * k, c9, sizeof(c9));
    }
    k++;
  }
  *n_voxels = k;
  return DGS_OK;
}

}  // namespace dgs
