// K7 fitness_score / exact nearest-neighbour search over the target cloud.
//
// Replaces the pcl::search::KdTree (FLANN, exact, eps = 0) that pcl::Registration keeps over the target and that
// the reference queries through
//   registration->getFitnessScore(max_range)                       include/hdl_graph_slam/loop_detector.hpp:148,
//                                                                  apps/scan_matching_odometry_nodelet.cpp:318
//   registration->getSearchMethodTarget()->nearestKSearch(pt,1,..) apps/scan_matching_odometry_nodelet.cpp:327
// (in-tree statement of the same loop: src/hdl_graph_slam/information_matrix_calculator.cpp:77-108).
//
// MI355X design: no pointer-chasing kd-tree.  Points are sorted by 30-bit Morton code (one radix sort); an
// IMPLICIT complete binary tree of AABBs over 8-point leaves sits on top (heap order, no child pointers, built
// bottom-up by min/max of two children).  A query first takes an upper bound from the leaf at its own Morton
// position, then walks the tree stacklessly (heap-index arithmetic) pruning on exact float AABB distances, so
// the search is exact and unbounded (fitness_score_max_range defaults to DBL_MAX, loop_detector.hpp:46).
// Squared distances are formed with explicitly rounded mul/add in FLANN's L2_Simple order (dx^2 + dy^2 + dz^2),
// so they are bit-identical to a CPU float evaluation; ties resolve to the lowest original index.
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cmath>

#include "handle.h"

namespace dgs {

constexpr int kLeaf = 8;

struct BvhView {
  const float4* sorted;   // Morton order, w = original index (bit pattern)
  const float4* lo;
  const float4* hi;
  const uint32_t* keys;   // sorted Morton codes
  int n;
  int leaves;             // power of two
  float org[3];
  float scale;            // Morton quantisation: q = (x - org) * scale, clamped to [0, 1023]
};

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__device__ __forceinline__ uint32_t morton30(float x, float y, float z, const float* org, float scale) {
  const float fx = fminf(fmaxf((x - org[0]) * scale, 0.f), 1023.f);
  const float fy = fminf(fmaxf((y - org[1]) * scale, 0.f), 1023.f);
  const float fz = fminf(fmaxf((z - org[2]) * scale, 0.f), 1023.f);
  return (expand_bits10((uint32_t)fx) << 2) | (expand_bits10((uint32_t)fy) << 1) | expand_bits10((uint32_t)fz);
}

__device__ __forceinline__ float sqdist_rn(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = sub_rn(ax, bx), dy = sub_rn(ay, by), dz = sub_rn(az, bz);
  return add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
}

__device__ __forceinline__ float aabb_sqdist_rn(const float4 lo, const float4 hi, float x, float y, float z) {
  const float dx = fmaxf(fmaxf(sub_rn(lo.x, x), sub_rn(x, hi.x)), 0.f);
  const float dy = fmaxf(fmaxf(sub_rn(lo.y, y), sub_rn(y, hi.y)), 0.f);
  const float dz = fmaxf(fmaxf(sub_rn(lo.z, z), sub_rn(z, hi.z)), 0.f);
  return add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
}

// ---- build ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void morton_kernel(const float4* __restrict__ pts, int n, float o0, float o1, float o2, float scale,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  const float org[3] = {o0, o1, o2};
  // non-finite points sort last and never win a query (their distance compares false)
  keys[i] = (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) ? morton30(p.x, p.y, p.z, org, scale) : 0x3FFFFFFFu;
  vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void gather_index_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ order, int n,
                                                              float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t o = order[i];
  float4 p = pts[o];
  p.w = __uint_as_float(o);
  out[i] = p;
}

__global__ __launch_bounds__(kBlock) void bvh_leaf_kernel(const float4* __restrict__ sorted, int n, int leaves, float4* __restrict__ lo,
                                                          float4* __restrict__ hi) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= leaves) return;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int k = 0; k < kLeaf; k++) {
    const int i = j * kLeaf + k;
    if (i < n) {
      const float4 p = sorted[i];
      if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
      }
    }
  }
  const int node = leaves - 1 + j;
  lo[node] = make_float4(mn[0], mn[1], mn[2], 0.f);
  hi[node] = make_float4(mx[0], mx[1], mx[2], 0.f);
}

__global__ __launch_bounds__(kBlock) void bvh_level_kernel(int first, int count, float4* __restrict__ lo, float4* __restrict__ hi) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int node = first + t;
  const float4 l0 = lo[2 * node + 1], l1 = lo[2 * node + 2], h0 = hi[2 * node + 1], h1 = hi[2 * node + 2];
  lo[node] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.f);
  hi[node] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.f);
}

// ---- query ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void scan_leaf(const BvhView& b, int leaf, float x, float y, float z, float& best, int& best_idx) {
  const int base = leaf * kLeaf;
#pragma unroll
  for (int k = 0; k < kLeaf; k++) {
    const int i = base + k;
    if (i < b.n) {
      const float4 p = b.sorted[i];
      const float d = sqdist_rn(x, y, z, p.x, p.y, p.z);
      const int oi = (int)__float_as_uint(p.w);
      if (d < best || (d == best && oi < best_idx)) {
        best = d;
        best_idx = oi;
      }
    }
  }
}

__device__ void nn_query(const BvhView& b, float x, float y, float z, float& best, int& best_idx) {
  best = INFINITY;
  best_idx = 0x7FFFFFFF;
  // upper bound from the leaf at the query's own Morton position
  const uint32_t code = morton30(x, y, z, b.org, b.scale);
  int lo = 0, hi = b.n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (b.keys[mid] < code) lo = mid + 1; else hi = mid;
  }
  const int leaf0 = min(lo, b.n - 1) / kLeaf;
  scan_leaf(b, leaf0, x, y, z, best, best_idx);
  const int first_leaf = b.leaves - 1;
  int node = 0;
  for (;;) {
    const float d = aabb_sqdist_rn(b.lo[node], b.hi[node], x, y, z);
    if (d <= best) {
      if (node >= first_leaf) {
        if (node - first_leaf != leaf0) scan_leaf(b, node - first_leaf, x, y, z, best, best_idx);
      } else {
        node = 2 * node + 1;
        continue;
      }
    }
    while (node != 0 && (node & 1) == 0) node = (node - 1) >> 1;  // climb while we are a right child
    if (node == 0) break;
    node += 1;  // right sibling
  }
}

__global__ __launch_bounds__(kBlock) void nn_search_kernel(const BvhView b, const float4* __restrict__ q, int m, int* __restrict__ idx,
                                                           float* __restrict__ sq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const float4 p = q[i];
  float best;
  int bi;
  nn_query(b, p.x, p.y, p.z, best, bi);
  idx[i] = bi;
  sq[i] = best;
}

// fitness / inlier accumulation: per block one row {sum d2 (d2 <= max_range), count, inliers (d2 < inlier_sq)}
__global__ __launch_bounds__(kBlock) void nn_fitness_kernel(const BvhView b, const float4* const* __restrict__ src_ptrs, const int* __restrict__ sizes,
                                                            const float* __restrict__ Tbase, size_t T_stride, float max_range, float inlier_sq,
                                                            double* __restrict__ partial, int blocks_per_pair) {
  const int pair = blockIdx.y;
  const float4* __restrict__ src = src_ptrs[pair];
  const int n = sizes[pair];
  const float* T = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Tbase) + (size_t)pair * T_stride);  // column-major
  const float t00 = T[0], t10 = T[1], t20 = T[2], t01 = T[4], t11 = T[5], t21 = T[6], t02 = T[8], t12 = T[9], t22 = T[10], t03 = T[12],
              t13 = T[13], t23 = T[14];
  double s = 0.0, c = 0.0, inl = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += blocks_per_pair * kBlock) {
    const float4 p = src[i];
    // pcl::transformPointCloud: ((m0 x + m1 y) + m2 z) + m3 in float
    const float x = add_rn(add_rn(add_rn(mul_rn(t00, p.x), mul_rn(t01, p.y)), mul_rn(t02, p.z)), t03);
    const float y = add_rn(add_rn(add_rn(mul_rn(t10, p.x), mul_rn(t11, p.y)), mul_rn(t12, p.z)), t13);
    const float z = add_rn(add_rn(add_rn(mul_rn(t20, p.x), mul_rn(t21, p.y)), mul_rn(t22, p.z)), t23);
    float best;
    int bi;
    nn_query(b, x, y, z, best, bi);
    if (best <= max_range) {  // PCL compares the SQUARED distance with max_range
      s += (double)best;
      c += 1.0;
    }
    if (best < inlier_sq) inl += 1.0;
  }
  __shared__ double sm[kBlock / kWave][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s = wave_sum(s); c = wave_sum(c); inl = wave_sum(inl);
  if (lane == 0) { sm[wave][0] = s; sm[wave][1] = c; sm[wave][2] = inl; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double v = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    partial[((size_t)pair * blocks_per_pair + blockIdx.x) * 4 + threadIdx.x] = v;
  }
}

__global__ void nn_fitness_final_kernel(const double* __restrict__ partial, int blocks_per_pair, int n_pairs, double* __restrict__ out) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= n_pairs) return;
  double s = 0, c = 0, inl = 0;
  for (int b = 0; b < blocks_per_pair; b++) {
    const double* r = partial + ((size_t)pair * blocks_per_pair + b) * 4;
    s += r[0]; c += r[1]; inl += r[2];
  }
  out[pair * 4 + 0] = s;
  out[pair * 4 + 1] = c;
  out[pair * 4 + 2] = inl;
  out[pair * 4 + 3] = 0;
}

// ---- host drivers ----------------------------------------------------------------------------------------------
int bvh_build(dgs_handle* h, Bvh& bvh, const float4* pts, int64_t n64) {
  hipStream_t st = h->stream;
  const int n = (int)n64;
  bvh.valid = false;
  bvh.n = n;
  if (n == 0) return DGS_OK;
  float mm[6];
  int rc = cloud_minmax(h, pts, n, mm);
  if (rc) return rc;
  if (!(mm[0] <= mm[3])) { mm[0] = mm[1] = mm[2] = 0.f; mm[3] = mm[4] = mm[5] = 1.f; }
  const float ext = std::max(std::max(mm[3] - mm[0], mm[4] - mm[1]), std::max(mm[5] - mm[2], 1e-6f));
  int leaves = 1;
  while ((int64_t)leaves * kLeaf < n) leaves <<= 1;
  int levels = 1;
  while ((1 << (levels - 1)) < leaves) levels++;
  bvh.leaves = leaves;
  bvh.levels = levels;
  DGS_HIP_TRY(h, bvh.sorted.reserve(n));
  DGS_HIP_TRY(h, bvh.keys.reserve(n));
  DGS_HIP_TRY(h, bvh.keys_alt.reserve(n));
  DGS_HIP_TRY(h, bvh.vals.reserve(n));
  DGS_HIP_TRY(h, bvh.vals_alt.reserve(n));
  DGS_HIP_TRY(h, bvh.node_lo.reserve((size_t)2 * leaves));
  DGS_HIP_TRY(h, bvh.node_hi.reserve((size_t)2 * leaves));
  size_t tb = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, bvh.keys.ptr, bvh.keys_alt.ptr, bvh.vals.ptr, bvh.vals_alt.ptr, n, 0, 30, st);
  DGS_HIP_TRY(h, h->cub_temp.reserve(tb + 256));
  const int nb = (n + kBlock - 1) / kBlock;
  const float scale = 1023.0f / ext;
  hipLaunchKernelGGL(morton_kernel, dim3(nb), dim3(kBlock), 0, st, pts, n, mm[0], mm[1], mm[2], scale, bvh.keys.ptr, bvh.vals.ptr);
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(h->cub_temp.ptr, tb, bvh.keys.ptr, bvh.keys_alt.ptr, bvh.vals.ptr, bvh.vals_alt.ptr, n, 0, 30, st));
  hipLaunchKernelGGL(gather_index_kernel, dim3(nb), dim3(kBlock), 0, st, pts, bvh.vals_alt.ptr, n, bvh.sorted.ptr);
  hipLaunchKernelGGL(bvh_leaf_kernel, dim3((leaves + kBlock - 1) / kBlock), dim3(kBlock), 0, st, bvh.sorted.ptr, n, leaves, bvh.node_lo.ptr, bvh.node_hi.ptr);
  for (int l = levels - 2; l >= 0; l--) {
    const int count = 1 << l, first = count - 1;
    hipLaunchKernelGGL(bvh_level_kernel, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, st, first, count, bvh.node_lo.ptr, bvh.node_hi.ptr);
  }
  DGS_HIP_TRY(h, hipGetLastError());
  bvh.org[0] = mm[0]; bvh.org[1] = mm[1]; bvh.org[2] = mm[2];
  bvh.scale = scale;
  bvh.valid = true;
  return DGS_OK;
}

static BvhView make_view(const Bvh& b) {
  BvhView v;
  v.sorted = b.sorted.ptr;
  v.lo = b.node_lo.ptr;
  v.hi = b.node_hi.ptr;
  v.keys = b.keys_alt.ptr;
  v.n = (int)b.n;
  v.leaves = b.leaves;
  v.org[0] = b.org[0]; v.org[1] = b.org[1]; v.org[2] = b.org[2];
  v.scale = b.scale;
  return v;
}

static int ensure_target_bvh(dgs_handle* h) {
  if (h->target_bvh.valid) return DGS_OK;
  return bvh_build(h, h->target_bvh, h->target.ptr, h->nt);
}

int nn_search(dgs_handle* h, const float4* queries, int64_t m, int32_t* d_idx, float* d_sq) {
  int rc = ensure_target_bvh(h);
  if (rc) return rc;
  const BvhView v = make_view(h->target_bvh);
  int slot = prof_begin(h, DGS_K_NN_SEARCH);
  hipLaunchKernelGGL(nn_search_kernel, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, v, queries, (int)m, d_idx, d_sq);
  prof_end(h, DGS_K_NN_SEARCH, slot);
  DGS_HIP_TRY(h, hipGetLastError());
  return DGS_OK;
}

int nn_fitness_batch(dgs_handle* h, int n_pairs, const float4* const* d_src_ptrs, const int* d_sizes, int max_size, const float* d_T,
                     size_t T_stride_bytes, double max_range, double inlier_sq, double* sums, int64_t* counts, int64_t* inliers) {
  int rc = ensure_target_bvh(h);
  if (rc) return rc;
  hipStream_t st = h->stream;
  const BvhView v = make_view(h->target_bvh);
  const int full = std::max(1, (max_size + kBlock - 1) / kBlock);
  const int bpp = std::max(1, std::min(full, std::max(8, 2048 / std::max(1, n_pairs))));
  DGS_HIP_TRY(h, h->nn_partials.reserve((size_t)n_pairs * bpp * 4 + (size_t)n_pairs * 4));
  double* d_out = h->nn_partials.ptr + (size_t)n_pairs * bpp * 4;
  if (ensure_pinned(h, 4096 + sizeof(double) * 4 * n_pairs) != DGS_OK) return DGS_ERR_HIP;
  // PCL's comparison is float(sq_dist) <= double(max_range); clamp so DBL_MAX keeps every finite distance
  const float mr = (max_range >= (double)FLT_MAX) ? FLT_MAX : (float)max_range;
  const float iq = (inlier_sq >= (double)FLT_MAX) ? FLT_MAX : (float)inlier_sq;
  int slot = prof_begin(h, DGS_K_NN_SEARCH);
  hipLaunchKernelGGL(nn_fitness_kernel, dim3(bpp, n_pairs), dim3(kBlock), 0, st, v, d_src_ptrs, d_sizes, d_T, T_stride_bytes, mr, iq,
                     h->nn_partials.ptr, bpp);
  prof_end(h, DGS_K_NN_SEARCH, slot);
  hipLaunchKernelGGL(nn_fitness_final_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, st, h->nn_partials.ptr, bpp, n_pairs, d_out);
  double* hout = reinterpret_cast<double*>(reinterpret_cast<char*>(h->pinned) + 4096);
  DGS_HIP_TRY(h, hipMemcpyAsync(hout, d_out, sizeof(double) * 4 * n_pairs, hipMemcpyDeviceToHost, st));
  DGS_HIP_TRY(h, hipStreamSynchronize(st));
  DGS_HIP_TRY(h, hipGetLastError());
  for (int i = 0; i < n_pairs; i++) {
    sums[i] = hout[i * 4 + 0];
    counts[i] = (int64_t)hout[i * 4 + 1];
    inliers[i] = (int64_t)hout[i * 4 + 2];
  }
  return DGS_OK;
}

int nn_fitness(dgs_handle* h, const float4* src, int64_t n, const float* T16, double max_range, double inlier_sq, double* sum, int64_t* count,
               int64_t* inliers) {
  // single pair: stage pointer / size / transform in a small device block
  hipStream_t st = h->stream;
  DGS_HIP_TRY(h, h->src_ptrs.reserve(1));
  DGS_HIP_TRY(h, h->src_sizes.reserve(1));
  DGS_HIP_TRY(h, h->inits.reserve(1));
  if (ensure_pinned(h, 8192) != DGS_OK) return DGS_ERR_HIP;
  char* base = reinterpret_cast<char*>(h->pinned) + 2048;
  const int ni = (int)n;
  std::memcpy(base, &src, sizeof(void*));
  std::memcpy(base + 16, &ni, sizeof(int));
  std::memcpy(base + 64, T16, sizeof(float) * 16);
  DGS_HIP_TRY(h, hipMemcpyAsync(h->src_ptrs.ptr, base, sizeof(void*), hipMemcpyHostToDevice, st));
  DGS_HIP_TRY(h, hipMemcpyAsync(h->src_sizes.ptr, base + 16, sizeof(int), hipMemcpyHostToDevice, st));
  DGS_HIP_TRY(h, hipMemcpyAsync(h->inits.ptr, base + 64, sizeof(float) * 16, hipMemcpyHostToDevice, st));
  return nn_fitness_batch(h, 1, h->src_ptrs.ptr, h->src_sizes.ptr, ni, reinterpret_cast<const float*>(h->inits.ptr), 64, max_range, inlier_sq, sum,
                          count, inliers);
}

}  // namespace dgs
