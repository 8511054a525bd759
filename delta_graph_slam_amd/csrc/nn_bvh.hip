// K7 fitness_score / exact nearest-neighbour search over the target cloud.
//
// Replaces the pcl::search::KdTree (FLANN, exact, eps = 0) that pcl::Registration keeps over the target and that
// the reference queries through
//   registration->getFitnessScore(max_range)                       include/hdl_graph_slam/loop_detector.hpp:148,
//                                                                  apps/scan_matching_odometry_nodelet.cpp:318
//   registration->getSearchMethodTarget()->nearestKSearch(pt,1,..) apps/scan_matching_odometry_nodelet.cpp:327
// (in-tree statement of the same loop: src/hdl_graph_slam/information_matrix_calculator.cpp:77-108).
//
// MI355X design.  A per-lane kd-tree walk is bound by the texture-addresser: every lane of a wave touches its own
// cache line at every step (measured: 215 dependent steps x 3 uncoalesced 16-B loads per query, 0.68 ms per 64k
// queries).  So the index is an 8-ARY implicit tree over Hilbert-sorted points and EIGHT LANES share one query:
//   * points are sorted by a 30-bit Hilbert index (one radix sort); a leaf is 8 consecutive points = one 128-B line;
//   * an internal node stores its 8 children's AABBs contiguously (two 128-B lines), heap order, no pointers;
//   * at a node each lane of the group tests one child box (the group reads exactly two full lines), the group
//     ballots the children whose box is within the current best, descends nearest-first and keeps one pending-children
//     byte per level (5 levels for 64k points) -- no stack, no scratch; at the last level the boxes stay in registers
//     while the qualifying leaves are scanned, each leaf scan being one coalesced line with one point per lane.
// The search is exact and unbounded (fitness_score_max_range defaults to DBL_MAX, loop_detector.hpp:46).  Squared
// distances are formed with individually rounded mul/add in FLANN's L2_Simple order (dx^2 + dy^2 + dz^2), so they
// equal a CPU float evaluation bit for bit; pruning uses <= and ties resolve to the lowest original index.
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cmath>

#include "handle.h"
#include "nn_group.h"

namespace dgs {

// ---- build ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void hilbert_key_kernel(const float4* __restrict__ pts, int n, const float* __restrict__ mm6,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  // quantisation frame from the device-resident AABB (no host round trip): cubic cells over the largest extent
  float org[3] = {mm6[0], mm6[1], mm6[2]};
  float ext = fmaxf(fmaxf(mm6[3] - mm6[0], mm6[4] - mm6[1]), fmaxf(mm6[5] - mm6[2], 1e-6f));
  if (!(mm6[0] <= mm6[3])) { org[0] = org[1] = org[2] = 0.f; ext = 1.f; }
  const float scale = 1023.0f / ext;
  // non-finite points sort last and never win a query (their distance compares false)
  keys[i] = (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) ? hilbert30(p.x, p.y, p.z, org, scale) : 0x3FFFFFFFu;
  vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void gather_index_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ order, int n, int n_pad,
                                                              float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  if (i < n) {
    const uint32_t o = order[i];
    float4 p = pts[o];
    p.w = __uint_as_float(o);
    out[i] = p;
  } else {
    out[i] = make_float4(NAN, NAN, NAN, __uint_as_float(0xFFFFFFFFu));  // padding: distance is NaN, never selected
  }
}

// boxes of the nodes at one level (first .. first+count-1), written into their parents' child-box arrays
__global__ __launch_bounds__(kBlock) void bvh_boxes_kernel(const float4* __restrict__ sorted, int n, int first, int count, int is_leaf_level,
                                                           int first_leaf, float4* __restrict__ box_lo, float4* __restrict__ box_hi) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int node = first + t;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (is_leaf_level) {
    const int j = node - first_leaf;
#pragma unroll
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
  } else {
#pragma unroll
    for (int k = 0; k < kFan; k++) {
      const float4 l = box_lo[node * kFan + k], h = box_hi[node * kFan + k];
      mn[0] = fminf(mn[0], l.x); mn[1] = fminf(mn[1], l.y); mn[2] = fminf(mn[2], l.z);
      mx[0] = fmaxf(mx[0], h.x); mx[1] = fmaxf(mx[1], h.y); mx[2] = fmaxf(mx[2], h.z);
    }
  }
  const int parent = (node - 1) / kFan, slot = (node - 1) % kFan;
  box_lo[parent * kFan + slot] = make_float4(mn[0], mn[1], mn[2], 0.f);
  box_hi[parent * kFan + slot] = make_float4(mx[0], mx[1], mx[2], 0.f);
}


__global__ __launch_bounds__(kBlock) void nn_search_kernel(const BvhView b, const float4* __restrict__ q, int m, int* __restrict__ idx,
                                                           float* __restrict__ sq, const int rounds) {
  // a wave takes 64 * rounds consecutive queries, 8 adjacent ones per round; each round's results bound the next round's searches
  const int first = ((blockIdx.x * kBlock + threadIdx.x) >> 6) * (8 * rounds) + ((threadIdx.x & 63) >> 3);
  float px = 0.f, py = 0.f, pz = 0.f, prev_best = INFINITY;
  bool prev_found = false;
  for (int r = 0; r < rounds; r++) {
    const int qi = first + r * 8;
    const bool alive = qi < m;
    const float4 p = alive ? q[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
    float best;
    int bi;
    nn_query_group(b, p.x, p.y, p.z, alive, nn_warm_bound_round(prev_best, prev_found, p.x, p.y, p.z, px, py, pz), best, bi);
    prev_found = alive && bi != 0x7FFFFFFF;
    prev_best = best;
    px = p.x; py = p.y; pz = p.z;
    if (alive && (threadIdx.x & 7) == 0) {
      idx[qi] = bi;
      sq[qi] = (bi != 0x7FFFFFFF) ? best : INFINITY;
    }
  }
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
  constexpr int QPB = kBlock / 8;  // queries per block per sweep
  const int sub = threadIdx.x & 7;
  // every wave walks a contiguous stretch of the source, 8 adjacent points per round: consecutive points of a scan are
  // neighbours, so the previous round's results bound this round's searches (nn_warm_bound_round) and most of the tree is
  // pruned before it is touched
  const int run = (n + blocks_per_pair * QPB - 1) / (blocks_per_pair * QPB);
  const int first = (blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * (8 * run) + ((threadIdx.x & 63) >> 3);
  float px = 0.f, py = 0.f, pz = 0.f, prev_best = INFINITY;
  bool prev_found = false;
  for (int r = 0; r < run; r++) {
    const int i = first + r * 8;
    const bool alive = i < n;
    const float4 p = alive ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    // pcl::transformPointCloud: ((m0 x + m1 y) + m2 z) + m3 in float, every step rounded
    const float x = affine_row_rn(t00, t01, t02, t03, p.x, p.y, p.z);
    const float y = affine_row_rn(t10, t11, t12, t13, p.x, p.y, p.z);
    const float z = affine_row_rn(t20, t21, t22, t23, p.x, p.y, p.z);
    float best;
    int bi;
    nn_query_group<false>(b, x, y, z, alive, nn_warm_bound_round(prev_best, prev_found, x, y, z, px, py, pz), best, bi);
    prev_found = alive && bi != 0x7FFFFFFF;
    prev_best = best;
    px = x; py = y; pz = z;
    if (alive && sub == 0) {
      if (bi == 0x7FFFFFFF) best = INFINITY;  // nothing found (empty index / non-finite query): as the unbounded search reports it
      if (best <= max_range) {  // PCL compares the SQUARED distance with max_range
        s += (double)best;
        c += 1.0;
      }
      if (best < inlier_sq) inl += 1.0;
    }
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

// one wave per pair: lane l sums rows l, l + 64, ... in order, then the lanes are summed in a fixed butterfly order
__global__ __launch_bounds__(kWave) void nn_fitness_final_kernel(const double* __restrict__ partial, int blocks_per_pair, int n_pairs,
                                                                 double* __restrict__ out) {
  const int pair = blockIdx.x;
  if (pair >= n_pairs) return;
  double s = 0, c = 0, inl = 0;
  for (int b = threadIdx.x; b < blocks_per_pair; b += kWave) {
    const double* r = partial + ((size_t)pair * blocks_per_pair + b) * 4;
    s += r[0]; c += r[1]; inl += r[2];
  }
  s = wave_sum(s); c = wave_sum(c); inl = wave_sum(inl);
  if (threadIdx.x == 0) {
    out[pair * 4 + 0] = s;
    out[pair * 4 + 1] = c;
    out[pair * 4 + 2] = inl;
    out[pair * 4 + 3] = 0;
  }
}

// ---- host drivers ----------------------------------------------------------------------------------------------
int bvh_build(dgs_handle* h, Bvh& bvh, const float4* pts, int64_t n64, hipStream_t stream) {
  hipStream_t st = stream ? stream : h->stream;
  const int n = (int)n64;
  bvh.valid = false;
  bvh.n = n;
  if (n == 0) return DGS_OK;
  float* d_mm = nullptr;
  int rc = cloud_minmax_device(h, pts, n, &d_mm, st);
  if (rc) return rc;
  const int n_leaves = (n + kLeaf - 1) / kLeaf;
  int depth = 1;
  int64_t slots = kFan;
  while (slots < n_leaves) { slots *= kFan; depth++; }
  if (depth > 8) { h->err = "cloud too large for the nearest-neighbour index (more than 8 levels)"; return DGS_ERR_UNSUPPORTED; }
  const int first_leaf = (int)((slots - 1) / (kFan - 1));
  const int n_pad = n_leaves * kLeaf;
  bvh.leaves = (int)slots;
  bvh.levels = depth;
  DGS_HIP_TRY(h, bvh.sorted.reserve(n_pad));
  DGS_HIP_TRY(h, bvh.keys.reserve(n));
  DGS_HIP_TRY(h, bvh.keys_alt.reserve(n));
  DGS_HIP_TRY(h, bvh.vals.reserve(n));
  DGS_HIP_TRY(h, bvh.vals_alt.reserve(n));
  DGS_HIP_TRY(h, bvh.node_lo.reserve((size_t)first_leaf * kFan));
  DGS_HIP_TRY(h, bvh.node_hi.reserve((size_t)first_leaf * kFan));
  size_t tb = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, bvh.keys.ptr, bvh.keys_alt.ptr, bvh.vals.ptr, bvh.vals_alt.ptr, n, 0, 30, st);
  DGS_HIP_TRY(h, h->cub_temp.reserve(tb + 256));
  const int nb = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(hilbert_key_kernel, dim3(nb), dim3(kBlock), 0, st, pts, n, d_mm, bvh.keys.ptr, bvh.vals.ptr);
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(h->cub_temp.ptr, tb, bvh.keys.ptr, bvh.keys_alt.ptr, bvh.vals.ptr, bvh.vals_alt.ptr, n, 0, 30, st));
  hipLaunchKernelGGL(gather_index_kernel, dim3((n_pad + kBlock - 1) / kBlock), dim3(kBlock), 0, st, pts, bvh.vals_alt.ptr, n, n_pad, bvh.sorted.ptr);
  // leaf slots first, then every internal level bottom-up (the root's own box is never needed)
  hipLaunchKernelGGL(bvh_boxes_kernel, dim3((unsigned)((slots + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, bvh.sorted.ptr, n, first_leaf, (int)slots, 1,
                     first_leaf, bvh.node_lo.ptr, bvh.node_hi.ptr);
  int64_t count = slots / kFan;
  for (int l = depth - 1; l >= 1; l--) {
    const int first = (int)((count - 1) / (kFan - 1));
    hipLaunchKernelGGL(bvh_boxes_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, bvh.sorted.ptr, n, first, (int)count, 0,
                       first_leaf, bvh.node_lo.ptr, bvh.node_hi.ptr);
    count /= kFan;
  }
  DGS_HIP_TRY(h, hipGetLastError());
  bvh.valid = true;
  return DGS_OK;
}

BvhView make_bvh_view(const Bvh& b) {
  BvhView v;
  v.sorted = b.sorted.ptr;
  v.box_lo = b.node_lo.ptr;
  v.box_hi = b.node_hi.ptr;
  v.n = (int)b.n;
  v.depth = b.levels;
  v.first_leaf = (int)(((int64_t)b.leaves - 1) / (kFan - 1));
  return v;
}

static int ensure_target_bvh(dgs_handle* h) {
  if (side_join(h) != DGS_OK) return DGS_ERR_HIP;  // an index being built on the side stream
  if (h->tgt->bvh.valid) return DGS_OK;
  return bvh_build(h, h->tgt->bvh, h->tgt->pts.ptr, h->nt);
}

// tree + (fitness pass) grid over the current target, on `st` (default: the handle's stream)
int ensure_target_index(dgs_handle* h, hipStream_t st) {
  int rc = DGS_OK;
  if (!h->tgt->bvh.valid) rc = bvh_build(h, h->tgt->bvh, h->tgt->pts.ptr, h->nt, st);
  if (rc == DGS_OK && h->use_grid && !h->tgt_grid.valid) rc = nn_grid_build(h, h->tgt_grid, h->tgt->bvh, h->tgt->pts.ptr, h->nt, st);
  return rc;
}

int nn_search(dgs_handle* h, const float4* queries, int64_t m, int32_t* d_idx, float* d_sq) {
  int rc = ensure_target_bvh(h);
  if (rc) return rc;
  const BvhView v = make_bvh_view(h->tgt->bvh);
  int slot = prof_begin(h, DGS_K_NN_SEARCH);
  // one round while the queries alone fill the chip (8192 resident waves of 8 queries), more rounds -- and their warm bounds -- beyond that
  const int rounds = (int)std::max<int64_t>(1, std::min<int64_t>(m / 65536, 8));
  const int64_t waves = (m + 8 * rounds - 1) / (8 * rounds);
  hipLaunchKernelGGL(nn_search_kernel, dim3((unsigned)((waves * kWave + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, v, queries, (int)m, d_idx, d_sq, rounds);
  prof_end(h, DGS_K_NN_SEARCH, slot);
  DGS_HIP_TRY(h, hipGetLastError());
  return DGS_OK;
}

int nn_fitness_batch(dgs_handle* h, int n_pairs, const float4* const* d_src_ptrs, const int* d_sizes, int max_size, const float* d_T,
                     size_t T_stride_bytes, double max_range, double inlier_sq, double* sums, int64_t* counts, int64_t* inliers) {
  int rc = ensure_target_bvh(h);
  h->use_grid = grid_wanted(h, (int64_t)n_pairs * max_size);
  if (rc == DGS_OK) rc = ensure_target_index(h);
  if (rc) return rc;
  return nn_fitness_batch_on(h, h->tgt->bvh, (h->use_grid && h->tgt_grid.valid) ? &h->tgt_grid : nullptr, n_pairs, d_src_ptrs, d_sizes, max_size, d_T, T_stride_bytes, max_range,
                             inlier_sq, sums, counts, inliers);
}

int nn_fitness_batch_on(dgs_handle* h, const Bvh& index, NnGrid* grid, int n_pairs, const float4* const* d_src_ptrs, const int* d_sizes, int max_size,
                        const float* d_T, size_t T_stride_bytes, double max_range, double inlier_sq, double* sums, int64_t* counts, int64_t* inliers) {
  hipStream_t st = h->stream;
  const BvhView v = make_bvh_view(index);
  // tree walk: one query per 8 lanes; grid: one query per lane
  const int full = std::max(1, (int)(((int64_t)max_size * (grid ? 1 : 8) + kBlock - 1) / kBlock));
  const int bpp = std::max(1, std::min(full, std::max(64, 8192 / std::max(1, n_pairs))));
  DGS_HIP_TRY(h, h->nn_partials.reserve((size_t)n_pairs * bpp * 4 + (size_t)n_pairs * 4));
  double* d_out = h->nn_partials.ptr + (size_t)n_pairs * bpp * 4;
  if (ensure_pinned(h, 4096 + sizeof(double) * 4 * n_pairs) != DGS_OK) return DGS_ERR_HIP;
  // PCL's comparison is float(sq_dist) <= double(max_range); clamp so DBL_MAX keeps every finite distance
  const float mr = (max_range >= (double)FLT_MAX) ? FLT_MAX : (float)max_range;
  const float iq = (inlier_sq >= (double)FLT_MAX) ? FLT_MAX : (float)inlier_sq;
  int slot = prof_begin(h, DGS_K_NN_SEARCH);
  if (grid) {
    const int rg = nn_grid_launch_fitness(h, *grid, index, n_pairs, d_src_ptrs, d_sizes, max_size, d_T, T_stride_bytes, mr, iq, h->nn_partials.ptr, bpp);
    if (rg) return rg;
  } else
    hipLaunchKernelGGL(nn_fitness_kernel, dim3(bpp, n_pairs), dim3(kBlock), 0, st, v, d_src_ptrs, d_sizes, d_T, T_stride_bytes, mr, iq,
                       h->nn_partials.ptr, bpp);
  prof_end(h, DGS_K_NN_SEARCH, slot);
  hipLaunchKernelGGL(nn_fitness_final_kernel, dim3(n_pairs), dim3(kWave), 0, st, h->nn_partials.ptr, bpp, n_pairs, d_out);
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
