// K1 ndt_voxel_build: target cloud -> voxel-Gaussian model (pclomp::VoxelGridCovariance semantics).
//
// Replaces the work registration->setInputTarget() triggers for NDT_OMP
// (/root/reference/apps/scan_matching_odometry_nodelet.cpp:180,254; include/hdl_graph_slam/loop_detector.hpp:124;
// configured at src/hdl_graph_slam/registrations.cpp:105-119).  Algorithm: SURVEY.md App. A "Target model".
//
// MI355X design: the per-voxel moments are accumulated in double IN POINT-INDEX ORDER (stable radix sort by
// voxel key, then one lane walks each voxel's run), so the table is bit-reproducible and follows the
// upstream accumulation order -- no float/double atomics.  The table is tiny (V ~ 10^4 x 48 B) and lives in
// L2; the dense cell->voxel index costs 4 B per grid cell of HBM (288 GB makes that free).
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cmath>

#include "handle.h"
#include "small_linalg.h"

namespace dgs {

// ---- AABB of the finite points ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void minmax_kernel(const float4* __restrict__ pts, int64_t n, float* __restrict__ partial) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = pts[i];
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
      mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
      mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
  }
  __shared__ float sm[kBlock / kWave][6];
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn[a] = fminf(mn[a], __shfl_down(mn[a], off, 64));
      mx[a] = fmaxf(mx[a], __shfl_down(mx[a], off, 64));
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    for (int a = 0; a < 3; a++) { sm[wave][a] = mn[a]; sm[wave][3 + a] = mx[a]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = sm[0][threadIdx.x];
    for (int w = 1; w < kBlock / kWave; w++) v = (threadIdx.x < 3) ? fminf(v, sm[w][threadIdx.x]) : fmaxf(v, sm[w][threadIdx.x]);
    partial[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

__global__ void minmax_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out6) {
  // one wave: lane l folds blocks l, l+64, ... then a shuffle tree; min for columns 0..2, max for 3..5
  const int lane = threadIdx.x;
  float v[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int b = lane; b < nblocks; b += 64)
    for (int a = 0; a < 6; a++) v[a] = (a < 3) ? fminf(v[a], partial[b * 6 + a]) : fmaxf(v[a], partial[b * 6 + a]);
#pragma unroll
  for (int a = 0; a < 6; a++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float o = __shfl_down(v[a], off, 64);
      v[a] = (a < 3) ? fminf(v[a], o) : fmaxf(v[a], o);
    }
  }
  if (lane == 0)
    for (int a = 0; a < 6; a++) out6[a] = v[a];
}

// AABB of the finite points of a device cloud, left on the device (6 floats: min xyz, max xyz); no synchronisation.
int cloud_minmax_device(dgs_handle* h, const float4* pts, int64_t n, float** d_out6, hipStream_t stream) {
  hipStream_t st = stream ? stream : h->stream;
  const int mm_blocks = (int)std::min<int64_t>((n + kBlock - 1) / kBlock, 256);
  DGS_HIP_TRY(h, h->minmax_partial.reserve((size_t)512 * 6 + 8));
  float* d_final = h->minmax_partial.ptr + (size_t)512 * 6;
  hipLaunchKernelGGL(minmax_kernel, dim3(mm_blocks), dim3(kBlock), 0, st, pts, n, h->minmax_partial.ptr);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, st, h->minmax_partial.ptr, mm_blocks, d_final);
  *d_out6 = d_final;
  return DGS_OK;
}

// AABB of the finite points of a device cloud -> host (one stream sync).
int cloud_minmax(dgs_handle* h, const float4* pts, int64_t n, float out6[6]) {
  hipStream_t st = h->stream;
  float* d_final = nullptr;
  int rc = cloud_minmax_device(h, pts, n, &d_final);
  if (rc) return rc;
  if (ensure_pinned(h, 4096) != DGS_OK) return DGS_ERR_HIP;
  float* hmm = reinterpret_cast<float*>(h->pinned);
  DGS_HIP_TRY(h, hipMemcpyAsync(hmm, d_final, 6 * sizeof(float), hipMemcpyDeviceToHost, st));
  DGS_HIP_TRY(h, hipStreamSynchronize(st));
  for (int k = 0; k < 6; k++) out6[k] = hmm[k];
  return DGS_OK;
}

// ---- voxel key per point (VoxelGridCovariance first pass, index arithmetic in float as upstream) ---------
// clear_cells / clear_scalars (may be null): the dense cell table is reset to -1 and the scalar block to 0 by this same launch (grid
// stride), instead of two fill commands in front of it
__global__ __launch_bounds__(kBlock) void voxel_key_kernel(const float4* __restrict__ pts, int64_t n, VoxelGrid g,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int* __restrict__ clear_cells,
                                                           int64_t n_cells, int* __restrict__ clear_scalars) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (clear_cells)
    for (int64_t c = i; c < n_cells; c += (int64_t)gridDim.x * blockDim.x) clear_cells[c] = -1;
  if (clear_scalars && i < 8) clear_scalars[i] = 0;
  if (i >= n) return;
  const float4 p = pts[i];
  uint32_t key = 0xFFFFFFFFu;
  if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    const int i0 = (int)(floorf(p.x * g.inv_leaf) - (float)g.min_b[0]);
    const int i1 = (int)(floorf(p.y * g.inv_leaf) - (float)g.min_b[1]);
    const int i2 = (int)(floorf(p.z * g.inv_leaf) - (float)g.min_b[2]);
    key = (uint32_t)(i0 + i1 * g.mul1 + i2 * g.mul2);
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void gather_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ order, int64_t n,
                                                        float4* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = pts[order[i]];
}

// ---- second pass: one lane per occupied voxel ----------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void voxel_finalize_kernel(const float4* __restrict__ sorted_pts, const uint32_t* __restrict__ run_keys,
                                                                const int* __restrict__ run_counts, const int* __restrict__ run_offsets,
                                                                int* __restrict__ scalars, int min_points, double eig_mult,
                                                                int* __restrict__ cell2vox, VoxelRec* __restrict__ vox,
                                                                float4* __restrict__ centroid, double* __restrict__ dbg,
                                                                int* __restrict__ vcount, int* __restrict__ vvalid, VoxelStrictRec* __restrict__ vstrict, const int eigen_qr) {
#pragma clang fp contract(off)  // the table is compared bit for bit with the CPU checker: every operation individually rounded
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int num_runs = scalars[0];
  if (r >= num_runs) return;
  const uint32_t key = run_keys[r];
  const int cnt = run_counts[r];
  vcount[r] = cnt;
  vvalid[r] = 0;
  if (key == 0xFFFFFFFFu) {  // non-finite points
    vcount[r] = 0;
    return;
  }
  const int off = run_offsets[r];
  double s[3] = {0, 0, 0}, q[6] = {0, 0, 0, 0, 0, 0};
  float cf[3] = {0, 0, 0};
  // upstream's order (point index, one after the other) is kept; the loads of kBatch points are issued together so that a voxel
  // with hundreds of points pays one memory latency per kBatch points instead of one per point (the kernel is a few dozen waves:
  // its duration is the latency chain of the fullest voxel)
  constexpr int kBatch = 24;
  for (int j0 = 0; j0 < cnt; j0 += kBatch) {
    float4 pb[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; u++) pb[u] = (j0 + u < cnt) ? sorted_pts[off + j0 + u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < kBatch; u++) {
      if (j0 + u < cnt) {
        const float4 p = pb[u];
        const double x = p.x, y = p.y, z = p.z;
        s[0] += x; s[1] += y; s[2] += z;
        q[0] += x * x; q[1] += x * y; q[2] += x * z; q[3] += y * y; q[4] += y * z; q[5] += z * z;
        cf[0] += p.x; cf[1] += p.y; cf[2] += p.z;
      }
    }
  }
  const double np = (double)cnt;
  const double mean[3] = {s[0] / np, s[1] / np, s[2] / np};
  const float fn = (float)cnt;
  double* d = dbg + (size_t)r * 12;
  d[0] = mean[0]; d[1] = mean[1]; d[2] = mean[2];
  for (int k = 0; k < 9; k++) d[3 + k] = 0.0;
  VoxelRec rec;
  rec.mean[0] = mean[0]; rec.mean[1] = mean[1]; rec.mean[2] = mean[2];
  for (int k = 0; k < 6; k++) rec.icov[k] = 0.f;
  bool valid = false;
  if (cnt >= min_points) {
    const double sq[9] = {q[0], q[1], q[2], q[1], q[3], q[4], q[2], q[4], q[5]};
    double cov[9];
    const double f = (np - 1.0) / np;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) cov[a * 3 + b] = ((sq[a * 3 + b] - 2.0 * (s[a] * mean[b])) / np + mean[a] * mean[b]) * f;
    double ev[3], V[9];
    if (eigen_qr) eigen_selfadjoint3_d(cov, ev, V);   // Eigen's SelfAdjointEigenSolver sequence (dgs_params.ndt_cov_eigensolver = 1, default)
    else sym_eig3_d(cov, ev, V);                       // cyclic Jacobi (rounds 1-3)
    if (!(ev[0] < 0 || ev[1] < 0 || ev[2] <= 0)) {
      const double min_ev = eig_mult * ev[2];
      if (ev[0] < min_ev) {
        ev[0] = min_ev;
        if (ev[1] < min_ev) ev[1] = min_ev;
        double Vi[9], VD[9];
        inv3_d(V, Vi);
        for (int rr = 0; rr < 3; rr++)
          for (int c = 0; c < 3; c++) VD[rr * 3 + c] = V[rr * 3 + c] * ev[c];
        for (int rr = 0; rr < 3; rr++)
          for (int c = 0; c < 3; c++) cov[rr * 3 + c] = VD[rr * 3 + 0] * Vi[0 * 3 + c] + VD[rr * 3 + 1] * Vi[1 * 3 + c] + VD[rr * 3 + 2] * Vi[2 * 3 + c];
      }
      double icov[9];
      inv3_d(cov, icov);
      valid = true;
      for (int k = 0; k < 9; k++)
        if (isinf(icov[k])) valid = false;
      if (valid) {
        rec.icov[0] = (float)icov[0]; rec.icov[1] = (float)icov[1]; rec.icov[2] = (float)icov[2];
        rec.icov[3] = (float)icov[4]; rec.icov[4] = (float)icov[5]; rec.icov[5] = (float)icov[8];
        for (int k = 0; k < 9; k++) d[3 + k] = icov[k];
      }
    }
  }
  vox[r] = rec;
  {
    VoxelStrictRec sr;
    sr.mean[0] = mean[0]; sr.mean[1] = mean[1]; sr.mean[2] = mean[2];
    for (int k = 0; k < 9; k++) sr.C[k] = (float)d[3 + k];   // float(icov), what upstream's updateDerivatives casts per visit
    sr.pad = 0.f;
    vstrict[r] = sr;
  }
  centroid[r] = make_float4(cf[0] / fn, cf[1] / fn, cf[2] / fn, valid ? 1.f : 0.f);
  vvalid[r] = valid ? 1 : 0;
  cell2vox[key] = valid ? r : -1;
  if (valid) atomicAdd(&scalars[1], 1);
}

// ---- pcl::VoxelGrid centroid filter: one lane per occupied cell, float sums in point-index order ------
__global__ __launch_bounds__(kBlock) void voxel_centroid_kernel(const float4* __restrict__ sorted_pts, const uint32_t* __restrict__ run_keys,
                                                                const int* __restrict__ run_counts, const int* __restrict__ run_offsets,
                                                                const int* __restrict__ scalars, float4* __restrict__ out, int out_capacity) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= scalars[0] || r >= out_capacity) return;
  if (run_keys[r] == 0xFFFFFFFFu) return;  // the run of non-finite points sorts last: it is simply not emitted
  const int off = run_offsets[r], cnt = run_counts[r];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int j = 0; j < cnt; j++) {
    const float4 p = sorted_pts[off + j];
    sx += p.x; sy += p.y; sz += p.z;
  }
  const float fn = (float)cnt;
  out[r] = make_float4(sx / fn, sy / fn, sz / fn, 1.f);
}

__global__ void voxel_count_kernel(const uint32_t* __restrict__ run_keys, int* __restrict__ scalars) {
  // number of emitted cells = runs, minus the trailing run of non-finite points if there is one
  const int nr = scalars[0];
  scalars[2] = (nr > 0 && run_keys[nr - 1] == 0xFFFFFFFFu) ? nr - 1 : nr;
}

// out receives the centroids ordered by cell index (as pcl::VoxelGrid emits them); *n_out the number of cells.
int voxel_grid_filter(dgs_handle* h, const float4* in, int64_t n, float leaf, float4* out, int64_t out_capacity, int64_t* n_out) {
  hipStream_t st = h->stream;
  *n_out = 0;
  if (n == 0) return DGS_OK;
  float hmm[6];
  int rc = cloud_minmax(h, in, n, hmm);
  if (rc) return rc;
  if (!(hmm[0] <= hmm[3])) return DGS_OK;
  VoxelGrid g{};
  g.leaf = leaf;
  g.inv_leaf = 1.0f / leaf;
  int64_t cells = 1;
  for (int a = 0; a < 3; a++) {
    g.min_b[a] = (int)std::floor(hmm[a] * g.inv_leaf);
    g.max_b[a] = (int)std::floor(hmm[3 + a] * g.inv_leaf);
    g.div_b[a] = g.max_b[a] - g.min_b[a] + 1;
    cells *= (int64_t)((hmm[3 + a] - hmm[a]) * g.inv_leaf) + 1;
  }
  if (cells > INT32_MAX || (int64_t)g.div_b[0] * g.div_b[1] * g.div_b[2] > INT32_MAX) {
    h->err = "Leaf size is too small for the input dataset. Integer indices would overflow.";
    return DGS_ERR_GRID_TOO_LARGE;
  }
  g.mul1 = g.div_b[0];
  g.mul2 = g.div_b[0] * g.div_b[1];
  DGS_HIP_TRY(h, h->key_in.reserve(n));
  DGS_HIP_TRY(h, h->key_out.reserve(n));
  DGS_HIP_TRY(h, h->val_in.reserve(n));
  DGS_HIP_TRY(h, h->val_out.reserve(n));
  DGS_HIP_TRY(h, h->vg_run_keys.reserve(n));
  DGS_HIP_TRY(h, h->run_counts.reserve(n));
  DGS_HIP_TRY(h, h->run_offsets.reserve(n));
  DGS_HIP_TRY(h, h->vg_scalars.reserve(8));
  DGS_HIP_TRY(h, h->scratch_cloud.reserve(n));
  size_t t1 = 0, t2 = 0, t3 = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, h->key_in.ptr, h->key_out.ptr, h->val_in.ptr, h->val_out.ptr, (int)n, 0, 32, st);
  (void)hipcub::DeviceRunLengthEncode::Encode(nullptr, t2, h->key_out.ptr, h->vg_run_keys.ptr, h->run_counts.ptr, h->vg_scalars.ptr, (int)n, st);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t3, h->run_counts.ptr, h->run_offsets.ptr, (int)n, st);
  DGS_HIP_TRY(h, h->cub_temp.reserve(std::max(t1, std::max(t2, t3)) + 256));
  const int nb = (int)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(voxel_key_kernel, dim3(nb), dim3(kBlock), 0, st, in, n, g, h->key_in.ptr, h->val_in.ptr, (int*)nullptr, (int64_t)0, h->vg_scalars.ptr);
  size_t tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(h->cub_temp.ptr, tb, h->key_in.ptr, h->key_out.ptr, h->val_in.ptr, h->val_out.ptr, (int)n, 0, 32, st));
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRunLengthEncode::Encode(h->cub_temp.ptr, tb, h->key_out.ptr, h->vg_run_keys.ptr, h->run_counts.ptr, h->vg_scalars.ptr, (int)n, st));
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceScan::ExclusiveSum(h->cub_temp.ptr, tb, h->run_counts.ptr, h->run_offsets.ptr, (int)n, st));
  hipLaunchKernelGGL(gather_kernel, dim3(nb), dim3(kBlock), 0, st, in, h->val_out.ptr, n, h->scratch_cloud.ptr);
  hipLaunchKernelGGL(voxel_centroid_kernel, dim3(nb), dim3(kBlock), 0, st, h->scratch_cloud.ptr, h->vg_run_keys.ptr, h->run_counts.ptr, h->run_offsets.ptr,
                     h->vg_scalars.ptr, out, (int)std::min<int64_t>(out_capacity, INT32_MAX));
  hipLaunchKernelGGL(voxel_count_kernel, dim3(1), dim3(1), 0, st, h->vg_run_keys.ptr, h->vg_scalars.ptr);
  int* hs = reinterpret_cast<int*>(h->pinned);
  DGS_HIP_TRY(h, hipMemcpyAsync(hs, h->vg_scalars.ptr, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  DGS_HIP_TRY(h, hipStreamSynchronize(st));
  DGS_HIP_TRY(h, hipGetLastError());
  *n_out = hs[2];
  return DGS_OK;
}

// ---- pcl::ApproximateVoxelGrid -------------------------------------------------------------------------------
// Upstream is ONE sequential pass through a 512-entry history table hashed by the cell coordinates: a point whose slot holds
// another cell flushes that cell's centroid to the output and takes the slot over; the rest is flushed in slot order at the end.
// The slots are independent of each other except for the ORDER of the output, so the pass parallelises exactly:
//   1. stable sort of the points by slot (9 bits) -> every slot's subsequence, still in point order;
//   2. a run = consecutive points of a subsequence with the same cell = one output point; head flags + scan number the runs;
//   3. a run is flushed when the next run of its slot begins (trigger = index of that run's first point) or, if it is the last of
//      its slot, in the final sweep (trigger = n + slot): sorting the runs by trigger gives upstream's output order;
//   4. one lane per output point sums its run in point order (float, as upstream) and divides by the float count.
constexpr unsigned kApproxHist = 512;

__device__ __forceinline__ void approx_cell(const float4 p, const float inv_leaf, int& ix, int& iy, int& iz, unsigned& slot) {
  ix = (int)floorf(p.x * inv_leaf);
  iy = (int)floorf(p.y * inv_leaf);
  iz = (int)floorf(p.z * inv_leaf);
  slot = ((unsigned)ix * 7171u + (unsigned)iy * 3079u + (unsigned)iz * 4231u) & (kApproxHist - 1u);
}

__global__ __launch_bounds__(kBlock) void approx_key_kernel(const float4* __restrict__ pts, int n, float inv_leaf, uint32_t* __restrict__ keys,
                                                            uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int ix, iy, iz;
  unsigned slot;
  approx_cell(pts[i], inv_leaf, ix, iy, iz, slot);
  keys[i] = slot;
  vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void approx_head_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ slots, const uint32_t* __restrict__ order,
                                                             int n, float inv_leaf, int* __restrict__ head) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  int h = 1;
  if (s > 0 && slots[s - 1] == slots[s]) {
    int ax, ay, az, bx, by, bz;
    unsigned t;
    approx_cell(pts[order[s]], inv_leaf, ax, ay, az, t);
    approx_cell(pts[order[s - 1]], inv_leaf, bx, by, bz, t);
    h = (ax != bx || ay != by || az != bz) ? 1 : 0;
  }
  head[s] = h;
}

__global__ __launch_bounds__(kBlock) void approx_runs_kernel(const int* __restrict__ head, const int* __restrict__ run_id, int n, uint32_t* __restrict__ run_start,
                                                             int* __restrict__ scalars) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  if (head[s]) run_start[run_id[s] - 1] = (uint32_t)s;
  if (s == n - 1) scalars[0] = run_id[s];
}

__global__ __launch_bounds__(kBlock) void approx_trigger_kernel(const uint32_t* __restrict__ slots, const uint32_t* __restrict__ order,
                                                                const uint32_t* __restrict__ run_start, const int* __restrict__ scalars, int n,
                                                                uint32_t* __restrict__ trigger, uint32_t* __restrict__ run) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int n_runs = scalars[0];
  uint32_t t = 0xFFFFFFFFu;   // padding sorts last
  if (r < n_runs) {
    const uint32_t s0 = run_start[r];
    const uint32_t slot = slots[s0];
    t = (uint32_t)n + slot;   // the last run of its slot: flushed by the final sweep over the table
    if (r + 1 < n_runs) {
      const uint32_t s1 = run_start[r + 1];
      if (slots[s1] == slot) t = order[s1];   // flushed when this point arrives
    }
  }
  trigger[r] = t;
  run[r] = (uint32_t)r;
}

__global__ __launch_bounds__(kBlock) void approx_centroid_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ order,
                                                                 const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ runs_in_output_order,
                                                                 const int* __restrict__ scalars, int n, float4* __restrict__ out, int out_capacity) {
#pragma clang fp contract(off)
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_runs = scalars[0];
  if (o >= n_runs || o >= out_capacity) return;
  const int r = (int)runs_in_output_order[o];
  const int s0 = (int)run_start[r], s1 = (r + 1 < n_runs) ? (int)run_start[r + 1] : n;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int s = s0; s < s1; s++) {
    const float4 p = pts[order[s]];
    sx += p.x; sy += p.y; sz += p.z;
  }
  const float fn = (float)(s1 - s0);
  out[o] = make_float4(sx / fn, sy / fn, sz / fn, 1.f);
}

// out receives the centroids in upstream's output order; *n_out their number.  The input must be finite (upstream does not look).
int approx_voxel_grid_filter(dgs_handle* h, const float4* in, int64_t n64, float leaf, float4* out, int64_t out_capacity, int64_t* n_out) {
  hipStream_t st = h->stream;
  *n_out = 0;
  if (n64 == 0) return DGS_OK;
  const int n = (int)n64;
  const float inv_leaf = 1.0f / leaf;
  DGS_HIP_TRY(h, h->key_in.reserve(n));
  DGS_HIP_TRY(h, h->key_out.reserve(n));
  DGS_HIP_TRY(h, h->val_in.reserve(n));
  DGS_HIP_TRY(h, h->val_out.reserve(n));
  DGS_HIP_TRY(h, h->vg_run_keys.reserve(n));
  DGS_HIP_TRY(h, h->run_counts.reserve(n));
  DGS_HIP_TRY(h, h->run_offsets.reserve(n));
  DGS_HIP_TRY(h, h->vg_scalars.reserve(8));
  DGS_HIP_TRY(h, h->scratch_cloud.reserve(n));   // 4 n words: the second sort's outputs live here
  uint32_t* trig_sorted = reinterpret_cast<uint32_t*>(h->scratch_cloud.ptr);
  uint32_t* runs_sorted = trig_sorted + n;
  size_t t1 = 0, t2 = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, h->key_in.ptr, h->key_out.ptr, h->val_in.ptr, h->val_out.ptr, n, 0, 32, st);
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, t2, h->run_counts.ptr, h->run_offsets.ptr, n, st);
  DGS_HIP_TRY(h, h->cub_temp.reserve(std::max(t1, t2) + 256));
  DGS_HIP_TRY(h, hipMemsetAsync(h->vg_scalars.ptr, 0, 8 * sizeof(int), st));
  const int nb = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(approx_key_kernel, dim3(nb), dim3(kBlock), 0, st, in, n, inv_leaf, h->key_in.ptr, h->val_in.ptr);
  size_t tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(h->cub_temp.ptr, tb, h->key_in.ptr, h->key_out.ptr, h->val_in.ptr, h->val_out.ptr, n, 0, 9, st));
  hipLaunchKernelGGL(approx_head_kernel, dim3(nb), dim3(kBlock), 0, st, in, h->key_out.ptr, h->val_out.ptr, n, inv_leaf, h->run_counts.ptr);
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceScan::InclusiveSum(h->cub_temp.ptr, tb, h->run_counts.ptr, h->run_offsets.ptr, n, st));
  hipLaunchKernelGGL(approx_runs_kernel, dim3(nb), dim3(kBlock), 0, st, h->run_counts.ptr, h->run_offsets.ptr, n, h->vg_run_keys.ptr, h->vg_scalars.ptr);
  hipLaunchKernelGGL(approx_trigger_kernel, dim3(nb), dim3(kBlock), 0, st, h->key_out.ptr, h->val_out.ptr, h->vg_run_keys.ptr, h->vg_scalars.ptr, n, h->key_in.ptr,
                     h->val_in.ptr);
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(h->cub_temp.ptr, tb, h->key_in.ptr, trig_sorted, h->val_in.ptr, runs_sorted, n, 0, 32, st));
  hipLaunchKernelGGL(approx_centroid_kernel, dim3(nb), dim3(kBlock), 0, st, in, h->val_out.ptr, h->vg_run_keys.ptr, runs_sorted, h->vg_scalars.ptr, n, out,
                     (int)std::min<int64_t>(out_capacity, INT32_MAX));
  int* hs = reinterpret_cast<int*>(h->pinned);
  if (ensure_pinned(h, 64) != DGS_OK) return DGS_ERR_HIP;
  hs = reinterpret_cast<int*>(h->pinned);
  DGS_HIP_TRY(h, hipMemcpyAsync(hs, h->vg_scalars.ptr, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  DGS_HIP_TRY(h, hipStreamSynchronize(st));
  DGS_HIP_TRY(h, hipGetLastError());
  *n_out = hs[0];
  return DGS_OK;
}

// ---- host driver -------------------------------------------------------------------------------------------
int ndt_build_target(dgs_handle* h) {
  const int64_t n = h->nt;
  hipStream_t st = h->stream;
  const float res = (float)h->prm.ndt_resolution;
  h->grid = VoxelGrid{};
  h->grid.leaf = res;
  h->grid.inv_leaf = 1.0f / res;
  h->grid_cells = 0;
  h->counts_stale = true;
  if (n == 0) return DGS_OK;

  // 1. AABB
  float hmm[6];
  int slot = prof_begin(h, DGS_K_NDT_VOXEL_BUILD);
  {
    const int rc = cloud_minmax(h, h->tgt->pts.ptr, n, hmm);
    if (rc != DGS_OK) return rc;
  }
  if (!(hmm[0] <= hmm[3])) {  // no finite point
    prof_end(h, DGS_K_NDT_VOXEL_BUILD, slot);
    return DGS_OK;
  }

  // 2. grid extents (VoxelGridCovariance::applyFilter)
  VoxelGrid& g = h->grid;
  int64_t cells = 1;
  for (int a = 0; a < 3; a++) {
    g.min_b[a] = (int)std::floor(hmm[a] * g.inv_leaf);
    g.max_b[a] = (int)std::floor(hmm[3 + a] * g.inv_leaf);
    g.div_b[a] = g.max_b[a] - g.min_b[a] + 1;
    const int64_t dx = (int64_t)((hmm[3 + a] - hmm[a]) * g.inv_leaf) + 1;
    cells *= dx;
  }
  const int64_t dense_cells = (int64_t)g.div_b[0] * g.div_b[1] * g.div_b[2];
  if (cells > INT32_MAX || dense_cells > INT32_MAX) {
    prof_end(h, DGS_K_NDT_VOXEL_BUILD, slot);
    h->err = "Leaf size is too small for the input dataset. Integer indices would overflow.";
    return DGS_ERR_GRID_TOO_LARGE;
  }
  g.mul1 = g.div_b[0];
  g.mul2 = g.div_b[0] * g.div_b[1];
  h->grid_cells = dense_cells;
  h->n_occupied_bound = n;

  // 3. buffers
  DGS_HIP_TRY(h, h->cell2vox.reserve((size_t)dense_cells));
  DGS_HIP_TRY(h, h->key_in.reserve(n));
  DGS_HIP_TRY(h, h->key_out.reserve(n));
  DGS_HIP_TRY(h, h->val_in.reserve(n));
  DGS_HIP_TRY(h, h->val_out.reserve(n));
  DGS_HIP_TRY(h, h->run_keys.reserve(n));
  DGS_HIP_TRY(h, h->run_counts.reserve(n));
  DGS_HIP_TRY(h, h->run_offsets.reserve(n));
  DGS_HIP_TRY(h, h->dev_scalars.reserve(8));
  DGS_HIP_TRY(h, h->vox.reserve(n));
  DGS_HIP_TRY(h, h->vox_centroid.reserve(n));
  DGS_HIP_TRY(h, h->vox_dbg.reserve((size_t)n * 12));
  DGS_HIP_TRY(h, h->vox_strict.reserve(n));
  DGS_HIP_TRY(h, h->vox_count.reserve(n));
  DGS_HIP_TRY(h, h->vox_valid.reserve(n));
  DGS_HIP_TRY(h, h->scratch_cloud.reserve(n));
  size_t t1 = 0, t2 = 0, t3 = 0;
  int end_bit = 32;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, h->key_in.ptr, h->key_out.ptr, h->val_in.ptr, h->val_out.ptr, (int)n, 0, end_bit, st);
  (void)hipcub::DeviceRunLengthEncode::Encode(nullptr, t2, h->key_out.ptr, h->run_keys.ptr, h->run_counts.ptr, h->dev_scalars.ptr, (int)n, st);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t3, h->run_counts.ptr, h->run_offsets.ptr, (int)n, st);
  const size_t tmax = std::max(t1, std::max(t2, t3));
  DGS_HIP_TRY(h, h->cub_temp.reserve(tmax + 256));

  // 4. keys -> stable sort -> runs -> offsets
  // (the cell table and the scalar block are cleared by the key kernel itself; run_counts needs no clearing: only its first
  // scalars[0] entries, which the run-length encoding writes, are ever read as counts)
  const int nb = (int)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(voxel_key_kernel, dim3(nb), dim3(kBlock), 0, st, h->tgt->pts.ptr, n, g, h->key_in.ptr, h->val_in.ptr, h->cell2vox.ptr, dense_cells,
                     h->dev_scalars.ptr);
  size_t tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(h->cub_temp.ptr, tb, h->key_in.ptr, h->key_out.ptr, h->val_in.ptr, h->val_out.ptr, (int)n, 0,
                                                    end_bit, st));
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRunLengthEncode::Encode(h->cub_temp.ptr, tb, h->key_out.ptr, h->run_keys.ptr, h->run_counts.ptr,
                                                       h->dev_scalars.ptr, (int)n, st));
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceScan::ExclusiveSum(h->cub_temp.ptr, tb, h->run_counts.ptr, h->run_offsets.ptr, (int)n, st));
  hipLaunchKernelGGL(gather_kernel, dim3(nb), dim3(kBlock), 0, st, h->tgt->pts.ptr, h->val_out.ptr, n, h->scratch_cloud.ptr);

  // 5. per-voxel statistics
  hipLaunchKernelGGL(voxel_finalize_kernel, dim3(nb), dim3(kBlock), 0, st, h->scratch_cloud.ptr, h->run_keys.ptr, h->run_counts.ptr,
                     h->run_offsets.ptr, h->dev_scalars.ptr, h->prm.ndt_min_points_per_voxel, h->prm.ndt_min_covar_eigvalue_mult,
                     h->cell2vox.ptr, h->vox.ptr, h->vox_centroid.ptr, h->vox_dbg.ptr, h->vox_count.ptr, h->vox_valid.ptr, h->vox_strict.ptr,
                     h->prm.ndt_cov_eigensolver);
  prof_end(h, DGS_K_NDT_VOXEL_BUILD, slot);
  DGS_HIP_TRY(h, hipGetLastError());
  g.cell2vox = h->cell2vox.ptr;
  g.vox = h->vox.ptr;
  g.centroid = h->vox_centroid.ptr;
  return DGS_OK;
}

}  // namespace dgs
