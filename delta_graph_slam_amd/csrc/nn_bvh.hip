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

#include <vector>
#include <cstdio>
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

// ---- k-d order: the same implicit tree over a point order made by median splits ------------------------------
// With the Hilbert order a node's 8 children are runs of the curve: compact, but their boxes overlap (a query within reach of
// 3.3 nodes per level on a 64-beam scan).  Splitting every range at its middle rank along its widest axis, three times per
// 8-ary level, gives boxes that do not overlap along the split axes: 1.2-2.7 nodes per level, and the fitness pass over 32 x
// 65,536 queries drops from 0.83 to 0.45 ms (same results: any order of the points gives a correct tree).  The price is the
// build -- a sort per binary level instead of one sort -- so this order is used where the build hides behind other work (the
// loop-closure batch builds the target's index on the side stream while the candidates iterate).
//
// The tree's layout fixes the ranks: a range of W = 8 * 2^m slots splits at slot W / 2; the cloud fills the slots from the
// left, so every range is full except the last.  Levels with W > kKdChunk (2,048 slots) are one global sort each on the 32-bit key
// (range number, coordinate along the range's widest axis); from W = kKdChunk down one workgroup per chunk sorts in LDS.
// chunk / workgroup size of the LDS levels, measured on the bench step (the build shares the chip with the iteration launches, so
// what counts is how little it disturbs them): 4096 / 1024 13.9 k registrations/s, 4096 / 512 13.8 k, 2048 / 512 14.2-14.4 k,
// 2048 / 256 14.0 k, 1024 / 256 14.3 k, 1024 / 512 14.2 k, 8192 / 1024 13.1 k
#ifndef DGS_KD_CHUNK
#define DGS_KD_CHUNK 2048
#endif
#ifndef DGS_KD_THREADS
#define DGS_KD_THREADS 512
#endif
constexpr int kKdChunk = DGS_KD_CHUNK, kKdThreads = DGS_KD_THREADS, kKdPer = kKdChunk / kKdThreads;

__device__ __forceinline__ unsigned orderable_f32(float f) {   // monotone float -> unsigned (finite values and infinities)
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ bool finite3(const float4 p) { return (p.x - p.x == 0.f) && (p.y - p.y == 0.f) && (p.z - p.z == 0.f); }
// widest axis of a box given as orderable minima [0..2] and maxima [3..5]; an empty box (no finite point) -> axis 0
__device__ __forceinline__ int widest_axis(const unsigned* bb) {
  if (bb[0] > bb[3]) return 0;
  // extents compared as floats recovered from the orderable form
  float e[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const unsigned lo = bb[a], hi = bb[3 + a];
    const float flo = __uint_as_float((lo & 0x80000000u) ? (lo & 0x7FFFFFFFu) : ~lo), fhi = __uint_as_float((hi & 0x80000000u) ? (hi & 0x7FFFFFFFu) : ~hi);
    e[a] = fhi - flo;
  }
  return (e[0] >= e[1] && e[0] >= e[2]) ? 0 : (e[1] >= e[2] ? 1 : 2);
}

__global__ __launch_bounds__(kBlock) void kd_iota_kernel(uint32_t* __restrict__ vals, int n, unsigned* __restrict__ bbox, int n_seg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) vals[i] = (uint32_t)i;
  if (i < n_seg * 6) bbox[i] = (i % 6 < 3) ? 0xFFFFFFFFu : 0u;
}

// boxes of the ranges of W slots (W >= kBlock: a workgroup's points belong to one range): wave reduction, one LDS atomic per wave
// and value, one global atomic per workgroup and value (minima / maxima: the result does not depend on the order)
__global__ __launch_bounds__(kBlock) void kd_bbox_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ vals, int n, int W,
                                                         unsigned* __restrict__ bbox) {
  __shared__ unsigned sm[6];
  if (threadIdx.x < 6) sm[threadIdx.x] = (threadIdx.x < 3) ? 0xFFFFFFFFu : 0u;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned v[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  if (i < n) {
    const float4 p = pts[vals[i]];
    if (finite3(p)) {
      v[0] = v[3] = orderable_f32(p.x);
      v[1] = v[4] = orderable_f32(p.y);
      v[2] = v[5] = orderable_f32(p.z);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) v[k] = min(v[k], (unsigned)__shfl_xor((int)v[k], o));
#pragma unroll
    for (int k = 3; k < 6; k++) v[k] = max(v[k], (unsigned)__shfl_xor((int)v[k], o));
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) atomicMin(&sm[k], v[k]);
#pragma unroll
    for (int k = 3; k < 6; k++) atomicMax(&sm[k], v[k]);
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    unsigned* dst = bbox + (size_t)((blockIdx.x * blockDim.x) / W) * 6 + threadIdx.x;
    if (threadIdx.x < 3) atomicMin(dst, sm[threadIdx.x]); else atomicMax(dst, sm[threadIdx.x]);
  }
}

// key = (range number, coordinate along the range's widest axis) in 32 bits: the range number takes the top seg_bits, the coordinate
// (monotone unsigned form) loses its lowest seg_bits.  The split is then at the middle rank of a slightly coarsened coordinate --
// any split gives a correct tree, and a 32-bit sort is half the work of a 64-bit one.  Non-finite points go to the end of their range.
__global__ __launch_bounds__(kBlock) void kd_key_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ vals, int n, int W, int seg_bits,
                                                        const unsigned* __restrict__ bbox, uint32_t* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int seg = i / W;
  const float4 p = pts[vals[i]];
  unsigned bb[6];
#pragma unroll
  for (int k = 0; k < 6; k++) bb[k] = bbox[(size_t)seg * 6 + k];
  const int axis = widest_axis(bb);
  const float c = axis == 0 ? p.x : (axis == 1 ? p.y : p.z);
  const unsigned ck = finite3(p) ? orderable_f32(c) : 0xFFFFFFFFu;
  keys[i] = seg_bits ? (((unsigned)seg << (32 - seg_bits)) | (ck >> seg_bits)) : ck;
}

__global__ __launch_bounds__(kBlock) void kd_clear_bbox_kernel(unsigned* __restrict__ bbox, int n_seg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_seg * 6) bbox[i] = (i % 6 < 3) ? 0xFFFFFFFFu : 0u;
}

// the levels from W = kKdChunk down to W = 16 (-> leaves of 8) for one chunk of kKdChunk slots: bitonic sorts in LDS of
// (coordinate along the range's widest axis, point index) -- the index makes the order total, hence deterministic
__global__ __launch_bounds__(kKdThreads) void kd_local_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ vals_in,
                                                              uint32_t* __restrict__ vals_out, int n) {
  __shared__ unsigned s_key[kKdChunk];
  __shared__ uint32_t s_idx[kKdChunk];
  __shared__ unsigned s_bb[kKdChunk / 16][6];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * kKdChunk;
  const int cnt = min(kKdChunk, n - base);
#pragma unroll
  for (int r = 0; r < kKdPer; r++) {
    const int e = tid + r * kKdThreads;
    s_idx[e] = (e < cnt) ? vals_in[base + e] : 0xFFFFFFFFu;
  }
  __syncthreads();
  for (int W = kKdChunk; W >= 16; W >>= 1) {
    const int n_seg = kKdChunk / W;
    for (int t = tid; t < n_seg * 6; t += kKdThreads) s_bb[t / 6][t % 6] = (t % 6 < 3) ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    float4 p[kKdPer];
    bool fin[kKdPer];
#pragma unroll
    for (int r = 0; r < kKdPer; r++) {
      const int e = tid + r * kKdThreads;
      const uint32_t gi = s_idx[e];
      p[r] = (gi != 0xFFFFFFFFu) ? pts[gi] : make_float4(NAN, NAN, NAN, 0.f);
      fin[r] = gi != 0xFFFFFFFFu && finite3(p[r]);
      unsigned v[6] = {fin[r] ? orderable_f32(p[r].x) : 0xFFFFFFFFu, fin[r] ? orderable_f32(p[r].y) : 0xFFFFFFFFu, fin[r] ? orderable_f32(p[r].z) : 0xFFFFFFFFu,
                       fin[r] ? orderable_f32(p[r].x) : 0u, fin[r] ? orderable_f32(p[r].y) : 0u, fin[r] ? orderable_f32(p[r].z) : 0u};
      if (W >= kWave) {   // the wave's 64 slots lie in one range: reduce in the wave, one atomic per value
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
          for (int k = 0; k < 3; k++) v[k] = min(v[k], (unsigned)__shfl_xor((int)v[k], o));
#pragma unroll
          for (int k = 3; k < 6; k++) v[k] = max(v[k], (unsigned)__shfl_xor((int)v[k], o));
        }
        if ((tid & 63) == 0) {
#pragma unroll
          for (int k = 0; k < 3; k++) atomicMin(&s_bb[e / W][k], v[k]);
#pragma unroll
          for (int k = 3; k < 6; k++) atomicMax(&s_bb[e / W][k], v[k]);
        }
      } else if (fin[r]) {
#pragma unroll
        for (int k = 0; k < 3; k++) atomicMin(&s_bb[e / W][k], v[k]);
#pragma unroll
        for (int k = 3; k < 6; k++) atomicMax(&s_bb[e / W][k], v[k]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kKdPer; r++) {
      const int e = tid + r * kKdThreads;
      const int axis = widest_axis(s_bb[e / W]);
      const float c = axis == 0 ? p[r].x : (axis == 1 ? p[r].y : p[r].z);
      s_key[e] = fin[r] ? orderable_f32(c) : 0xFFFFFFFFu;
    }
    __syncthreads();
    // bitonic sort of every range of W slots by (key, index).  Slot e = tid + r * kKdThreads lives in register r of thread tid:
    // partners at distance j >= kKdThreads are registers of the same thread, j < 64 lanes of the same wave (shuffles), only
    // 64 <= j < kKdThreads goes through LDS behind a barrier (52 barrier stages over all levels instead of 354).
    unsigned kr[kKdPer];
    uint32_t ir[kKdPer];
#pragma unroll
    for (int r = 0; r < kKdPer; r++) { kr[r] = s_key[tid + r * kKdThreads]; ir[r] = s_idx[tid + r * kKdThreads]; }
    for (int k = 2; k <= W; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        if (j >= kKdThreads) {
          const int dr = j / kKdThreads;
#pragma unroll
          for (int r = 0; r < kKdPer; r++) {
            if ((r & dr) == 0) {
              const int e = tid + r * kKdThreads;
              const bool up = (k == W) || ((e & k) == 0);
              const int q = r | dr;
              const bool a_gt_b = kr[r] > kr[q] || (kr[r] == kr[q] && ir[r] > ir[q]);
              if (a_gt_b == up) { const unsigned tk = kr[r]; kr[r] = kr[q]; kr[q] = tk; const uint32_t ti = ir[r]; ir[r] = ir[q]; ir[q] = ti; }
            }
          }
        } else {
          if (j >= kWave) {
            __syncthreads();   // everybody has read what it needed of the previous exchange
#pragma unroll
            for (int r = 0; r < kKdPer; r++) { s_key[tid + r * kKdThreads] = kr[r]; s_idx[tid + r * kKdThreads] = ir[r]; }
            __syncthreads();
          }
#pragma unroll
          for (int r = 0; r < kKdPer; r++) {
            const int e = tid + r * kKdThreads;
            unsigned pk;
            uint32_t pi;
            if (j >= kWave) { pk = s_key[e ^ j]; pi = s_idx[e ^ j]; }
            else { pk = (unsigned)__shfl_xor((int)kr[r], j); pi = (uint32_t)__shfl_xor((int)ir[r], j); }
            const bool up = (k == W) || ((e & k) == 0);
            const bool lower = (e & j) == 0;                 // this slot is the lower one of its pair
            const bool mine_gt = kr[r] > pk || (kr[r] == pk && ir[r] > pi);
            // the lower slot keeps the smaller key when ascending; the upper slot the larger one
            const bool take = (lower == up) ? mine_gt : !mine_gt;
            if (take) { kr[r] = pk; ir[r] = pi; }
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kKdPer; r++) { s_key[tid + r * kKdThreads] = kr[r]; s_idx[tid + r * kKdThreads] = ir[r]; }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < kKdPer; r++) {
    const int e = tid + r * kKdThreads;
    if (e < cnt) vals_out[base + e] = s_idx[e];
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

// which pairs of a batch a fitness launch walks: blockIdx.y -> id[blockIdx.y]; n = 0: every pair, blockIdx.y itself
struct NnPairList {
  int n;
  unsigned short id[62];
};

// fitness / inlier accumulation: per block one row {sum d2 (d2 <= max_range), count, inliers (d2 < inlier_sq)}
__global__ __launch_bounds__(kBlock) void nn_fitness_kernel(const BvhView b, const float4* const* __restrict__ src_ptrs, const int* __restrict__ sizes,
                                                            const float* __restrict__ Tbase, size_t T_stride, float max_range, float inlier_sq,
                                                            double* __restrict__ partial, int blocks_per_pair, const NnPairList list) {
  const int pair = list.n ? (int)list.id[blockIdx.y] : (int)blockIdx.y;   // a chosen subset of the batch (early fitness), or all of it
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
int bvh_build(dgs_handle* h, Bvh& bvh, const float4* pts, int64_t n64, hipStream_t stream, bool kd_order) {
  hipStream_t st = stream ? stream : h->stream;
  const int n = (int)n64;
  bvh.valid = false;
  bvh.n = n;
  if (n == 0) return DGS_OK;
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
  const int nb = (n + kBlock - 1) / kBlock;
  const uint32_t* order = nullptr;
  if (kd_order) {
    // ranges of W slots, W halving; the first W that needs a split is the smallest power of two >= n
    int64_t W = 16;
    while (W < n) W <<= 1;
    const int max_seg = (int)std::max<int64_t>(1, (n + kKdChunk - 1) / kKdChunk);
    DGS_HIP_TRY(h, bvh.kd_bbox.reserve((size_t)max_seg * 6 + 6));
    size_t tb = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, bvh.keys.ptr, bvh.keys_alt.ptr, bvh.vals.ptr, bvh.vals_alt.ptr, n, 0, 32, st);
    DGS_HIP_TRY(h, h->cub_temp.reserve(tb + 256));
    uint32_t* va = bvh.vals.ptr;
    uint32_t* vb = bvh.vals_alt.ptr;
    hipLaunchKernelGGL(kd_iota_kernel, dim3(nb), dim3(kBlock), 0, st, va, n, bvh.kd_bbox.ptr, 1);
    for (; W > kKdChunk; W >>= 1) {
      const int n_seg = (int)((n + W - 1) / W);
      int seg_bits = 0;
      while ((1 << seg_bits) < n_seg) seg_bits++;
      hipLaunchKernelGGL(kd_bbox_kernel, dim3(nb), dim3(kBlock), 0, st, pts, va, n, (int)W, bvh.kd_bbox.ptr);
      hipLaunchKernelGGL(kd_key_kernel, dim3(nb), dim3(kBlock), 0, st, pts, va, n, (int)W, seg_bits, bvh.kd_bbox.ptr, bvh.keys.ptr);
      size_t tbytes = h->cub_temp.cap;
      DGS_HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(h->cub_temp.ptr, tbytes, bvh.keys.ptr, bvh.keys_alt.ptr, va, vb, n, 0, 32, st));
      std::swap(va, vb);
      const int next_seg = (int)((n + W / 2 - 1) / (W / 2));
      hipLaunchKernelGGL(kd_clear_bbox_kernel, dim3((next_seg * 6 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, bvh.kd_bbox.ptr, next_seg);
    }
    hipLaunchKernelGGL(kd_local_kernel, dim3(max_seg), dim3(kKdThreads), 0, st, pts, va, vb, n);
    order = vb;
  } else {
    float* d_mm = nullptr;
    int rc = cloud_minmax_device(h, pts, n, &d_mm, st);
    if (rc) return rc;
    size_t tb = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, bvh.keys.ptr, bvh.keys_alt.ptr, bvh.vals.ptr, bvh.vals_alt.ptr, n, 0, 30, st);
    DGS_HIP_TRY(h, h->cub_temp.reserve(tb + 256));
    hipLaunchKernelGGL(hilbert_key_kernel, dim3(nb), dim3(kBlock), 0, st, pts, n, d_mm, bvh.keys.ptr, bvh.vals.ptr);
    tb = h->cub_temp.cap;
    DGS_HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(h->cub_temp.ptr, tb, bvh.keys.ptr, bvh.keys_alt.ptr, bvh.vals.ptr, bvh.vals_alt.ptr, n, 0, 30, st));
    order = bvh.vals_alt.ptr;
  }
  bvh.kd = kd_order;
  hipLaunchKernelGGL(gather_index_kernel, dim3((n_pad + kBlock - 1) / kBlock), dim3(kBlock), 0, st, pts, order, n, n_pad, bvh.sorted.ptr);
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
  return bvh_build(h, h->tgt->bvh, h->tgt->pts.ptr, h->nt, nullptr, (h->batch_kd || h->nn_kd_all) && !h->use_grid);
}

// tree + (fitness pass) grid over the current target, on `st` (default: the handle's stream)
int ensure_target_index(dgs_handle* h, hipStream_t st) {
  int rc = DGS_OK;
  if (!h->tgt->bvh.valid) rc = bvh_build(h, h->tgt->bvh, h->tgt->pts.ptr, h->nt, st, (h->batch_kd || h->nn_kd_all) && !h->use_grid);
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
  h->use_grid = grid_wanted(h, (int64_t)n_pairs * max_size);   // first: the grid is derived from a Hilbert ordered index
  int rc = ensure_target_bvh(h);
  if (rc == DGS_OK) rc = ensure_target_index(h);
  if (rc) return rc;
  return nn_fitness_batch_on(h, h->tgt->bvh, (h->use_grid && h->tgt_grid.valid) ? &h->tgt_grid : nullptr, n_pairs, d_src_ptrs, d_sizes, max_size, d_T, T_stride_bytes, max_range,
                             inlier_sq, sums, counts, inliers);
}

// The fitness pass in three steps, so that dgs_align_batch can walk the candidates that have finished while the others still iterate
// (ndt_align_pairs, `early_fit`): prepare (buffers, rows per pair -- a function of the WHOLE batch, so a pair's sum does not depend on
// which launch walked it), enqueue (the walk of the listed pairs, or of all, on any stream), totals (per pair, and their copy to the host).
int nn_fitness_prepare(dgs_handle* h, int n_pairs, int max_size, bool grid) {
  // tree walk: one query per 8 lanes; grid: one query per lane
  const int full = std::max(1, (int)(((int64_t)max_size * (grid ? 1 : 8) + kBlock - 1) / kBlock));
  static const int total_blocks = std::getenv("DGS_NN_BLOCKS") ? std::atoi(std::getenv("DGS_NN_BLOCKS")) : 8192;   // launch-shape sweeps
  const int bpp = std::max(1, std::min(full, std::max(64, total_blocks / std::max(1, n_pairs))));
  DGS_HIP_TRY(h, h->nn_partials.reserve((size_t)n_pairs * bpp * 4 + (size_t)n_pairs * 4));
  h->nn_out = h->nn_partials.ptr + (size_t)n_pairs * bpp * 4;
  h->nn_bpp = bpp;
  if (h->fit_host_cap < n_pairs) {
    if (h->fit_host) (void)hipHostFree(h->fit_host);
    h->fit_host = nullptr;
    h->fit_host_cap = 0;
    DGS_HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->fit_host), sizeof(double) * 4 * (size_t)(n_pairs + 64), hipHostMallocDefault));
    h->fit_host_cap = n_pairs + 64;
  }
  return DGS_OK;
}

void nn_fitness_enqueue(dgs_handle* h, hipStream_t st, const Bvh& index, const int* ids, int count, const float4* const* d_src_ptrs, const int* d_sizes,
                        const float* d_T, size_t T_stride_bytes, double max_range, double inlier_sq, int background_lds_kb) {
  const BvhView v = make_bvh_view(index);
  // background (beside the iteration launches of the same batch): a dynamic LDS request that nothing uses caps the walk at 160 / kb
  // workgroups per CU, so that an iteration launch finds room at once -- a walk's workgroup lives ~100 us, several launches long
  const unsigned lds = (unsigned)std::max(0, background_lds_kb) * 1024u;
  // PCL's comparison is float(sq_dist) <= double(max_range); clamp so DBL_MAX keeps every finite distance
  const float mr = (max_range >= (double)FLT_MAX) ? FLT_MAX : (float)max_range;
  const float iq = (inlier_sq >= (double)FLT_MAX) ? FLT_MAX : (float)inlier_sq;
  NnPairList list{};
  if (!ids) {   // pairs 0 .. count - 1
    hipLaunchKernelGGL(nn_fitness_kernel, dim3(h->nn_bpp, count), dim3(kBlock), 0, st, v, d_src_ptrs, d_sizes, d_T, T_stride_bytes, mr, iq, h->nn_partials.ptr,
                       h->nn_bpp, list);
    return;
  }
  constexpr int kMax = (int)(sizeof(list.id) / sizeof(list.id[0]));
  for (int o = 0; o < count; o += kMax) {
    list.n = std::min(kMax, count - o);
    for (int k = 0; k < list.n; k++) list.id[k] = (unsigned short)ids[o + k];
    hipLaunchKernelGGL(nn_fitness_kernel, dim3(h->nn_bpp, list.n), dim3(kBlock), lds, st, v, d_src_ptrs, d_sizes, d_T, T_stride_bytes, mr, iq,
                       h->nn_partials.ptr, h->nn_bpp, list);
  }
}

// per-pair totals and the copy to the host, on the handle's stream; no synchronisation (nn_fitness_read after one)
int nn_fitness_totals_enqueue(dgs_handle* h, int n_pairs) {
  hipStream_t st = h->stream;
  hipLaunchKernelGGL(nn_fitness_final_kernel, dim3(n_pairs), dim3(kWave), 0, st, h->nn_partials.ptr, h->nn_bpp, n_pairs, const_cast<double*>(h->nn_out));
  DGS_HIP_TRY(h, hipMemcpyAsync(h->fit_host, h->nn_out, sizeof(double) * 4 * n_pairs, hipMemcpyDeviceToHost, st));
  return DGS_OK;
}

void nn_fitness_read(const dgs_handle* h, int n_pairs, double* sums, int64_t* counts, int64_t* inliers) {
  for (int i = 0; i < n_pairs; i++) {
    sums[i] = h->fit_host[i * 4 + 0];
    counts[i] = (int64_t)h->fit_host[i * 4 + 1];
    inliers[i] = (int64_t)h->fit_host[i * 4 + 2];
  }
}

int nn_fitness_batch_on(dgs_handle* h, const Bvh& index, NnGrid* grid, int n_pairs, const float4* const* d_src_ptrs, const int* d_sizes, int max_size,
                        const float* d_T, size_t T_stride_bytes, double max_range, double inlier_sq, double* sums, int64_t* counts, int64_t* inliers) {
  hipStream_t st = h->stream;
  int rc = nn_fitness_prepare(h, n_pairs, max_size, grid != nullptr);
  if (rc) return rc;
  int slot = prof_begin(h, DGS_K_NN_SEARCH);
  if (grid) {
    const float mr = (max_range >= (double)FLT_MAX) ? FLT_MAX : (float)max_range;
    const float iq = (inlier_sq >= (double)FLT_MAX) ? FLT_MAX : (float)inlier_sq;
    const int rg = nn_grid_launch_fitness(h, *grid, index, n_pairs, d_src_ptrs, d_sizes, max_size, d_T, T_stride_bytes, mr, iq, h->nn_partials.ptr, h->nn_bpp);
    if (rg) return rg;
  } else {
    nn_fitness_enqueue(h, st, index, nullptr, n_pairs, d_src_ptrs, d_sizes, d_T, T_stride_bytes, max_range, inlier_sq, 0);
  }
  prof_end(h, DGS_K_NN_SEARCH, slot);
  rc = nn_fitness_totals_enqueue(h, n_pairs);
  if (rc) return rc;
  DGS_HIP_TRY(h, hipStreamSynchronize(st));
  DGS_HIP_TRY(h, hipGetLastError());
  nn_fitness_read(h, n_pairs, sums, counts, inliers);
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
