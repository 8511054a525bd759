// K4 knn_covariance, K5 gicp_correspond, K6 gicp_linearize / gicp_error and the Levenberg-Marquardt driver on the device.
//
// Replaces fast_gicp::FastGICP::computeTransformation (on fast_gicp::LsqRegistration), i.e. what
// registration->align(*aligned, guess) runs at /root/reference/apps/scan_matching_odometry_nodelet.cpp:218 and
// /root/reference/include/hdl_graph_slam/loop_detector.hpp:145 when registration_method is FAST_GICP (the value every
// shipped launch file sets; object configured at src/hdl_graph_slam/registrations.cpp:27-36).  Algorithm: SURVEY.md App. B.
//
// MI355X design
//   * No pointer kd-trees: both clouds get the implicit 8-ary AABB index of nn_bvh.hip (Hilbert ordered; the target of a large
//     batch k-d ordered).  k-NN covariances: one wave per index leaf (gicp_knn_leaf_kernel below); per-iteration 1-NN
//     correspondences: 8-lane groups (nn_group.h), exact and bounded by max_correspondence_distance.
//   * One linearisation = correspond (8 lanes / point) + linearize (1 lane / point, all double: RCR = C_B + R C_A R^T,
//     3x3 inverse, J = [skew(Tp) | -I], 21 + 6 + 1 sums) with the same wave-DPP -> LDS -> fixed-order partial rows as NDT,
//     so the sums are bit-reproducible (upstream's per-thread OpenMP partials are not).
//   * The optimiser step (LM / GN, se3_exp, the rho test, the convergence test) runs on one wave in the LAST workgroup of a pair's
//     linearize slice (gicp_close_round; gicp_solve_kernel is the same step as its own launch) and queues the next evaluation in
//     place: no host round trip per iteration.
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>

#include "handle.h"
#include "nn_group.h"
#include "small_linalg.h"
#include "solve6.h"

namespace dgs {

// ================================================================================================ K4 covariances
__device__ void regularize_cov(const double* cov9, int method_and_flag, double* out6) {
  const bool jacobi_svd = (method_and_flag & 0x100) != 0;   // dgs_params.gicp_cov_jacobi_svd rides on the method word
  const int method = method_and_flag & 0xff;
  double C[9];
  if (method == DGS_GICP_REG_NONE) {
    for (int a = 0; a < 9; a++) C[a] = cov9[a];
  } else if (method == DGS_GICP_REG_FROBENIUS) {
    const double lambda = 1e-3;
    double A[9], Ai[9];
    for (int a = 0; a < 9; a++) A[a] = cov9[a] + ((a % 4 == 0) ? lambda : 0.0);
    inv3_d(A, Ai);
    double nrm = 0;
    for (int a = 0; a < 9; a++) nrm += Ai[a] * Ai[a];
    nrm = sqrt(nrm);
    for (int a = 0; a < 9; a++) Ai[a] /= nrm;
    inv3_d(Ai, C);
  } else {
    // Eigen::JacobiSVD<Matrix3d> svd(cov, ComputeFullU | ComputeFullV); cov = U values.asDiagonal() V^T (singular values descending) -- or, rounds
    // 1-3's stand-in, the eigen-decomposition of the symmetric PSD matrix (the same factors up to rounding, U = V)
    double U[9], V[9], sv[3];
    if (jacobi_svd) {
      jacobi_svd3_d(cov9, U, V, sv);
    } else {
      double ev[3], E[9];
      sym_eig3_d(cov9, ev, E);
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) U[r * 3 + c] = V[r * 3 + c] = E[r * 3 + (2 - c)];   // descending order: column c <- eigenvector 2 - c
      sv[0] = fabs(ev[2]); sv[1] = fabs(ev[1]); sv[2] = fabs(ev[0]);
    }
    double vals[3];
    if (method == DGS_GICP_REG_PLANE) {
      vals[0] = 1; vals[1] = 1; vals[2] = 1e-3;
    } else if (method == DGS_GICP_REG_MIN_EIG) {
      for (int a = 0; a < 3; a++) vals[a] = fmax(sv[a], 1e-3);
    } else {
      for (int a = 0; a < 3; a++) vals[a] = fmax(sv[a] / sv[0], 1e-3);
    }
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) C[r * 3 + c] = U[r * 3 + 0] * vals[0] * V[c * 3 + 0] + U[r * 3 + 1] * vals[1] * V[c * 3 + 1] + U[r * 3 + 2] * vals[2] * V[c * 3 + 2];
  }
  out6[0] = C[0]; out6[1] = C[1]; out6[2] = C[2]; out6[3] = C[4]; out6[4] = C[5]; out6[5] = C[8];
}

// FastGICP::calculate_covariances: exact k-NN of every point in its own cloud, covariance of the neighbours, regularised.
// Two kernels.  (1) gicp_knn_kernel: the search alone, so its registers are the k-NN set and the walk, nothing of the covariance;
// every wave takes a contiguous stretch of the cloud in the index's own (Hilbert) order, 8 adjacent points per round (the 8
// groups walk nearly the same nodes), and each round's searches start from a bound on the k-th distance taken from the previous
// round: all k neighbours of q' lie within r_k(q') + |q - q'| of q (pruning only: same sets).  It writes the set of every
// point to `nbr` (slot r * 8 + sub of the 8-lane group -> nbr[pos * 32 + slot], -1 = nothing found).  (2)
// gicp_cov_from_knn_kernel: 8 lanes per point gather the neighbours and reduce mean and covariance in double; one lane per point regularises.
__global__ __launch_bounds__(kBlock) void gicp_knn_kernel(const BvhView b, const int n, const int k, const int run, int* __restrict__ nbr) {
  const int sub = threadIdx.x & 7;
  const int wave = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int first = wave * (8 * run) + ((threadIdx.x & 63) >> 3);
  float px = 0.f, py = 0.f, pz = 0.f, prev_kth = INFINITY;
  bool prev_found = false;
  for (int r = 0; r < run; r++) {
    const int pos = first + r * 8;
    if (__all(pos >= n)) break;   // wave-uniform: the stretch is past the end of the cloud
    const float4 q = (pos < n) ? b.sorted[pos] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int i = (pos < n) ? (int)__float_as_uint(q.w) : -1;
    const bool alive = pos < n && i >= 0 && i < n;
    const float bound = nn_warm_bound_round(prev_kth, prev_found, q.x, q.y, q.z, px, py, pz);
    KnnList L;
    knn_query_group(b, q.x, q.y, q.z, alive, k, L, bound);
    const unsigned long long worst = knn_largest(L);
    prev_kth = __uint_as_float((unsigned)(worst >> 32));
    prev_found = alive && prev_kth < INFINITY;
    px = q.x; py = q.y; pz = q.z;
    if (alive) {
#pragma unroll
      for (int s = 0; s < kKnnSlots; s++) {
        const int slot = s * 8 + sub;
        if (slot < k) nbr[(size_t)pos * kKnnMax + slot] = (L.dist(s) < INFINITY) ? L.index(s) : -1;
      }
    }
  }
}

// ---- k-NN sets, one WAVE per leaf of the index (8 Hilbert-adjacent query points), in three cooperative steps:
//  (a) a window of 8 leaves around the query leaf (64 points, one per lane) gives every query an upper bound T_j on its k-th
//      squared distance: the k-th smallest of its distances to the window (bit-descent over ballot counts);
//  (b) ONE walk of the tree for the 8 queries together: a frontier of nodes in LDS, 8 nodes x 8 child boxes per pass, a child
//      qualifies when it is within T_j of ANY query j (the very test of the per-query walk, so no neighbour can be missed), the
//      qualifying leaves end up in an LDS list (<= kLeafCap; typical 15-20);
//  (c) per query: distances to all candidate points (lane = point), the k smallest (distance, index) keys selected by the same
//      bit-descent, ties at the k-th distance broken towards the lower index, and written to nbr.
// The sets are the same as those of the per-query walk (knn_query_group): same float distance, same order relation.  Per wave this
// is ~10 k instructions instead of ~50 k (there, every one of ~40 insertions per query costs ~65 instructions of cross-lane
// minimum / replace-the-largest).
// Sparse corners of a cloud (0.7 % of the leaves of a 64-beam scan) have window bounds so loose that the lists overflow.  Those
// waves retry with the bounds scaled down (s T_j): whatever s, a query whose candidates hold >= k points within s T_j has its
// exact answer among them (every point within s T_j sits in a gathered leaf); s is bisected between "overflows" and "too few
// points" until every query of the leaf is answered.  Only what still fails after kKnnRetries -- or is irregular: fewer than 8
// leaves, a non-finite point in the window, fewer than k finite window points -- takes the per-query walk.
#ifndef DGS_KNN_LEAF_CAP
#define DGS_KNN_LEAF_CAP 64
#endif
constexpr int kLeafCap = DGS_KNN_LEAF_CAP, kFrontCap = 64, kLeafChunks = kLeafCap / 8, kKnnRetries = 12;
constexpr unsigned kInfBits = 0x7F800000u;

__device__ __forceinline__ float readlane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int lanes_below(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }

// k-th smallest (1-based) of the wave's values d[0 .. chunks) (bit patterns of non-negative floats, kInfBits = none), built from
// the top bit down: V keeps the largest prefix with fewer than k values below it.  If some prefix has EXACTLY k values below it
// the descent stops there and returns it with lt_only = true: then { d < V } is the answer set and ties cannot matter.
// `upper`: bit pattern of a value known to be >= the k-th smallest (0 = nothing known).  The descent is scalar work -- a compare,
// a population count and a branch per chunk and bit, ~18 rounds per call on scan data, two thirds of this kernel's instructions -- and
// its first eight rounds only find the binade of the answer: with a bound, the binades at and just below the bound's are tried
// directly (one round each) and the descent starts at the mantissa.
template <int NC>
__device__ __forceinline__ int count_below(const unsigned (&d)[NC], const int chunks, const unsigned trial) {
  int cnt = 0;
#pragma unroll
  for (int c = 0; c < NC; c++)
    if (c < chunks) cnt += __popcll(__ballot(d[c] < trial));
  return cnt;
}
template <int NC>
__device__ __forceinline__ unsigned kth_smallest_bits(const unsigned (&d)[NC], const int chunks, const int k, bool& lt_only, const unsigned upper = 0u) {
  unsigned V = 0;
  int bit = 30;
  lt_only = false;
  if (upper >= (4u << 23) && upper < 0x7F800000u) {
    const unsigned e = upper >> 23;   // the k-th smallest is below (e + 1) << 23
#pragma unroll 1
    for (unsigned t = 0; t < 3u; t++) {
      const unsigned base = (e - t) << 23;
      const int cnt = count_below<NC>(d, chunks, base);
      if (cnt == k) { lt_only = true; return base; }
      if (cnt < k) { V = base; bit = 22; break; }   // fewer than k below this binade, at least k below the next: the answer has this exponent
    }
  }
#pragma unroll 1
  for (; bit >= 0; bit--) {
    const unsigned trial = V | (1u << bit);
    const int cnt = count_below<NC>(d, chunks, trial);
    if (cnt == k) { lt_only = true; return trial; }
    if (cnt < k) V = trial;
  }
  return V;
}

// An UPPER BOUND of the k-th smallest of the wave's values: the descent above cut after the top `rounds` bits, the undecided low bits
// set (so that at least k values are <= the result).  For the window bound of step (a), which only prunes.
__device__ __forceinline__ unsigned kth_smallest_upper_bound(const unsigned d, const int k, const int rounds) {
  unsigned V = 0;
  int bit = 30;
#pragma unroll 1
  for (; bit > 30 - rounds; bit--) {
    const unsigned trial = V | (1u << bit);
    const int cnt = __popcll(__ballot(d < trial));
    if (cnt == k) return trial;        // exactly k values below trial: the k-th smallest is below it
    if (cnt < k) V = trial;
  }
  return V | ((2u << bit) - 1u);        // the k-th smallest has prefix V: it is at most V with every lower bit set
}

// Step (a) for NQ queries j0 .. j0 + NQ - 1 of the leaf at once: per query an UPPER BOUND of the k-th smallest distance to the 64 window
// points (the top 14 bits of the bit descent, the undecided low bits set: within 1.6 %, and it only prunes), written to lane j of Tl.
// Returns false when a query has fewer than k finite window points (the caller takes the careful path).
template <int NQ>
__device__ __forceinline__ bool knn_window_bounds(const float4 wp, const bool wfinite, const int w0, const int j0, const int k, float& Tl) {
  const int lane = threadIdx.x & 63;
  unsigned d[NQ], V[NQ];
  bool done[NQ];
  bool ok = true;
#pragma unroll
  for (int t = 0; t < NQ; t++) {
    const float qx = readlane_f32(wp.x, w0 + j0 + t), qy = readlane_f32(wp.y, w0 + j0 + t), qz = readlane_f32(wp.z, w0 + j0 + t);
    const float dp = sqdist_rn(qx, qy, qz, wp.x, wp.y, wp.z);
    d[t] = (wfinite && dp < INFINITY) ? __float_as_uint(dp) : kInfBits;
    ok = ok && (__popcll(__ballot(d[t] < kInfBits)) >= k);
    V[t] = 0;
    done[t] = false;
  }
  if (!ok) return false;
  constexpr int kRounds = 14;
#pragma unroll 1
  for (int bit = 30; bit > 30 - kRounds; bit--) {
#pragma unroll
    for (int t = 0; t < NQ; t++) {
      const unsigned trial = V[t] | (1u << bit);
      const int cnt = __popcll(__ballot(d[t] < trial));
      if (!done[t]) {
        if (cnt == k) { V[t] = trial; done[t] = true; }   // exactly k values below trial: the k-th smallest is below it
        else if (cnt < k) V[t] = trial;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NQ; t++) {
    const unsigned bound = done[t] ? V[t] : (V[t] | ((2u << (30 - kRounds)) - 1u));   // prefix V: at most V with every lower bit set
    if (lane == j0 + t) Tl = __uint_as_float(bound);
  }
  return true;
}

// step (b): the leaves within Tl[j] * scale of any query j of `open`; returns their number, or -1 when a list overflowed.
// Lane layout: 8 frontier nodes x 8 children per pass, the queries in a loop inside.  Measured and dropped (round 3, `profiles/r03`):
// (query x child) lanes with one node per pass (no query loop, but 4-8x the passes: 0.174-0.190 against 0.168-0.173 ms per 65,536-point
// cloud), and one test per node against the box of the open queries above the leaves (an eighth of the tests, but the sparse rings of a
// 16-beam scan then overflow the frontier and walk again: 0.081 -> 0.160 ms per 26,668-point cloud).
__device__ __forceinline__ int knn_gather_leaves(const BvhView& b, const float4 wp, const int w0, const int n_q, const unsigned open, const float Tl,
                                                 const float scale, int (*front)[kFrontCap], int* leaves) {
  const int lane = threadIdx.x & 63;
  int cur = 0, count = 1, level0 = 0;
  if (b.depth >= 3) {
    // The 64 nodes two levels below the root (heap ids 9..72, their boxes at 8..71) in ONE pass, a node per lane: it replaces the
    // three dependent passes root -> 8 -> 64 every wave would otherwise repeat.  Empty nodes carry inverted boxes and never hit.
    const float4 lo = load16_at(b.box_lo, 8u + lane), hi = load16_at(b.box_hi, 8u + lane);
    bool hit = false;
    // over the set bits of `open` (with 2 or 4 waves per leaf a wave has 4 or 2 of the 8 queries: a loop over all 8 spent most of its
    // scalar instructions skipping the others)
#pragma unroll 1
    for (unsigned rest = open; rest != 0u; rest &= rest - 1u) {
      const int j = __builtin_ctz(rest);
      const float qx = readlane_f32(wp.x, w0 + j), qy = readlane_f32(wp.y, w0 + j), qz = readlane_f32(wp.z, w0 + j);
      hit = hit || (aabb_sqdist_rn(lo, hi, qx, qy, qz) <= readlane_f32(Tl, j) * scale);
    }
    const unsigned long long m = __ballot(hit);
    if (hit) front[0][lanes_below(m)] = 9 + lane;
    count = __popcll(m);
    level0 = 2;
  } else if (lane == 0) {
    front[0][0] = 0;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  for (int level = level0; level < b.depth; level++) {
    const bool last = level == b.depth - 1;
    int next = 0;
    for (int base = 0; base < count; base += 8) {
      const int slot = base + (lane >> 3);
      const bool valid = slot < count;
      const int node = valid ? front[cur][slot] : 0;
      const unsigned ofs = (unsigned)node * kFan + (lane & 7);
      const float4 lo = load16_at(b.box_lo, ofs), hi = load16_at(b.box_hi, ofs);
      bool hit = false;
#pragma unroll 1
      for (unsigned rest = open; rest != 0u; rest &= rest - 1u) {
        const int j = __builtin_ctz(rest);
        const float qx = readlane_f32(wp.x, w0 + j), qy = readlane_f32(wp.y, w0 + j), qz = readlane_f32(wp.z, w0 + j);
        hit = hit || (aabb_sqdist_rn(lo, hi, qx, qy, qz) <= readlane_f32(Tl, j) * scale);
      }
      hit = hit && valid;
      const unsigned long long m = __ballot(hit);
      const int pos = next + lanes_below(m);
      const int child = node * kFan + 1 + (lane & 7);
      if (hit) {
        if (last) { if (pos < kLeafCap) leaves[pos] = child - b.first_leaf; }
        else if (pos < kFrontCap) front[cur ^ 1][pos] = child;
      }
      next += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (next > (last ? kLeafCap : kFrontCap)) return -1;
    cur ^= 1;
    count = next;
  }
  return count;
}


// step (c) for a candidate list of exactly NCH chunks (NCH * 8 leaves, the last chunk possibly partial): per open query the k-th smallest
// candidate distance, the selection (ties at the k-th distance to the lower index) and its neighbour list; answered queries leave `open`
template <int NCH>
__device__ __forceinline__ void knn_select(const BvhView& b, const float4 wp, const int w0, const int n_q, const int n_cand, const int k, const float Tl,
                                           const float scale, const int* leaves, int* __restrict__ nbr, const int l, unsigned& open) {
  const int lane = threadIdx.x & 63;
  float4 cp[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    cp[c] = make_float4(NAN, NAN, NAN, __uint_as_float(0xFFFFFFFFu));
    {
      const int li = c * 8 + (lane >> 3);
      if (li < n_cand) cp[c] = load16_at(b.sorted, (unsigned)(leaves[li] * kLeaf + (lane & 7)));
    }
  }
#pragma unroll 1
  for (unsigned rest = open; rest != 0u; rest &= rest - 1u) {
    const int j = __builtin_ctz(rest);
    const float qx = readlane_f32(wp.x, w0 + j), qy = readlane_f32(wp.y, w0 + j), qz = readlane_f32(wp.z, w0 + j);
    const unsigned Ttry = __float_as_uint(readlane_f32(Tl, j) * scale);
    unsigned d[NCH];
    int within = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      d[c] = kInfBits;
      {
        const float dp = sqdist_rn(qx, qy, qz, cp[c].x, cp[c].y, cp[c].z);   // NaN for padding / empty lanes
        if (dp < INFINITY && (int)__float_as_uint(cp[c].w) >= 0) d[c] = __float_as_uint(dp);
        within += __popcll(__ballot(d[c] <= Ttry));
      }
    }
    if (within < k) continue;   // the scaled bound was too small for this query: nothing can be concluded
    bool lt = true;
    // exactly k candidates within the bound (it sits at most 1.6 % above the k-th distance of the window): they are the answer, no descent
    const unsigned V = (within == k) ? Ttry + 1u : kth_smallest_bits<NCH>(d, NCH, k, lt, Ttry);   // at least k are within Ttry: the answer is <= Ttry
    // selection: everything below V; if V IS the k-th smallest, the k - (count below) lowest indices among the values equal to V
    bool sel[NCH];
    int below = 0, equal = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      sel[c] = d[c] < V;
      { below += __popcll(__ballot(d[c] < V)); equal += __popcll(__ballot(d[c] == V)); }
    }
    if (!lt) {
      const int need = k - below;
      if (equal == need) {
#pragma unroll
        for (int c = 0; c < NCH; c++) sel[c] = d[c] <= V;
      } else {
        // more values at the k-th distance than places left: lowest index first (rare: duplicate or symmetric points)
        for (int t = 0; t < need; t++) {
          int mine = 0x7FFFFFFF;
#pragma unroll
          for (int c = 0; c < NCH; c++)
            if (d[c] == V && !sel[c]) mine = min(mine, (int)__float_as_uint(cp[c].w));
          int best = mine;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
#pragma unroll
          for (int c = 0; c < NCH; c++)
            if (d[c] == V && (int)__float_as_uint(cp[c].w) == best) sel[c] = true;
        }
      }
    }
    int* __restrict__ out = nbr + (size_t)(l * kLeaf + j) * kKnnMax;
    int written = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      {
        const unsigned long long m = __ballot(sel[c]);
        if (sel[c]) out[written + lanes_below(m)] = (int)__float_as_uint(cp[c].w);
        written += __popcll(m);
      }
    }
    open &= ~(1u << j);
  }
}

template <int N>
__device__ __forceinline__ void knn_select_dispatch(const int chunks, const BvhView& b, const float4 wp, const int w0, const int n_q, const int n_cand, const int k,
                                                    const float Tl, const float scale, const int* leaves, int* __restrict__ nbr, const int l, unsigned& open) {
  if (chunks <= N || N == kLeafChunks) knn_select<N>(b, wp, w0, n_q, n_cand, k, Tl, scale, leaves, nbr, l, open);
  else if constexpr (N < kLeafChunks) knn_select_dispatch<N + 1>(chunks, b, wp, w0, n_q, n_cand, k, Tl, scale, leaves, nbr, l, open);
}

// `parts` (1, 2, 4 or 8) waves share a leaf, each answering 8 / parts of its queries: a small cloud has too few leaves to fill the
// chip, and a wave's 8 selections are one dependent chain -- shorter chains on more waves, at the price of one walk per part.
// -DDGS_KNN_WAVES: waves per SIMD the compiler must leave room for.  4 = what the kernel's 109 registers give anyway; 5 / 6 / 7 / 8 (with
// -DDGS_KNN_LEAF_CAP = 64 / 48 / 40 / 32) were measured and are no faster, spills or not: the kernel is bound by the scalar unit.
#ifndef DGS_KNN_WAVES
#define DGS_KNN_WAVES 4
#endif
__global__ __launch_bounds__(kBlock, DGS_KNN_WAVES) void gicp_knn_leaf_kernel(const BvhView b, const int n, const int k, const int parts, int* __restrict__ nbr, int* __restrict__ stats) {
  __shared__ int s_front[kBlock / kWave][2][kFrontCap];
  __shared__ int s_leaves[kBlock / kWave][kLeafCap];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n_leaves = (n + kLeaf - 1) / kLeaf;
  const int wid = blockIdx.x * (kBlock / kWave) + wv;
  const int l = wid / parts, part = wid % parts;      // this wave's query leaf and its share of the queries (wave-uniform)
  if (l >= n_leaves) return;
  bool fast = n_leaves >= 8;
  // ---- (a) window of 8 leaves, one point per lane; the queries are lanes w0 + j
  const int start = fast ? min(max(l - 3, 0), n_leaves - 8) : 0;
  const int w0 = fast ? (l - start) * 8 : 0;
  float4 wp = make_float4(NAN, NAN, NAN, __uint_as_float(0xFFFFFFFFu));
  if (fast) wp = load16_at(b.sorted, (unsigned)(start * 8 + lane));
  const bool wreal = fast && (start * 8 + lane) < n;
  const bool wfinite = wreal && (wp.x - wp.x == 0.f) && (wp.y - wp.y == 0.f) && (wp.z - wp.z == 0.f);
  if (__ballot(wreal && !wfinite) != 0ull) fast = false;   // a non-finite point nearby: the careful path
  const int n_q = min(kLeaf, n - l * kLeaf);                // queries of this leaf (the last leaf may be partial)
  const int per = kLeaf / parts, j_lo = part * per, j_hi = min(n_q, j_lo + per);
  if (j_lo >= j_hi) return;
#ifdef DGS_KNN_STATS
  const unsigned long long ts0 = wall_clock64();
  unsigned long long acc_b = 0, acc_c = 0;
#endif
  float Tl = 0.f;   // lane j holds T_j
  if (fast) {
    // the bounds of this wave's queries, their descents interleaved (one descent is a chain of dependent ballot -> count -> branch
    // rounds; 2 / 4 / 8 of them side by side hide each other's latency)
    switch (j_hi - j_lo) {
      case 8: fast = knn_window_bounds<8>(wp, wfinite, w0, j_lo, k, Tl); break;
      case 4: fast = knn_window_bounds<4>(wp, wfinite, w0, j_lo, k, Tl); break;
      case 2: fast = knn_window_bounds<2>(wp, wfinite, w0, j_lo, k, Tl); break;
      default:
        for (int j = j_lo; j < j_hi && fast; j++) fast = knn_window_bounds<1>(wp, wfinite, w0, j, k, Tl);
        break;
    }
  }
#ifdef DGS_KNN_STATS
  const unsigned long long ts1 = wall_clock64();
#endif
  unsigned open = ((1u << j_hi) - 1u) & ~((1u << j_lo) - 1u);
  int iters = 0;
  if (fast) {
    float scale = 1.f, s_small = 0.f, s_over = 0.f;   // s_small: answered nobody new; s_over: overflowed (0 = not seen yet)
    for (; open != 0u && iters < kKnnRetries; iters++) {
#ifdef DGS_KNN_STATS
      const unsigned long long tb0 = wall_clock64();
#endif
      const int n_cand = knn_gather_leaves(b, wp, w0, n_q, open, Tl, scale, s_front[wv], s_leaves[wv]);
#ifdef DGS_KNN_STATS
      const unsigned long long tb1 = wall_clock64();
      acc_b += tb1 - tb0;
#endif
      if (n_cand < 0) {
        s_over = scale;
        scale = (s_small > 0.f) ? sqrtf(s_small * s_over) : scale * 0.0625f;
        continue;
      }
      // ---- (c) candidates: chunk c = leaves c * 8 .. c * 8 + 7 of the list, lane -> (leaf lane / 8, point lane % 8).  The chunk count
      // is a template argument: with it as a run-time bound every one of the dozen per-chunk loops below carried a scalar compare and a
      // branch per possible chunk and round -- the kernel was bound by the CU's one scalar unit (2,600 scalar against 1,360 vector
      // instructions per wave, `profiles/r03/knn_*`), not by anything it computes.
      const int chunks = (n_cand + 7) >> 3;
      const unsigned was_open = open;
      knn_select_dispatch<1>(chunks, b, wp, w0, n_q, n_cand, k, Tl, scale, s_leaves[wv], nbr, l, open);
#ifdef DGS_KNN_STATS
      acc_c += wall_clock64() - tb1;
#endif
      if (open != 0u) {
        if (open != was_open) s_over = 0.f;   // fewer queries now: what overflowed before may fit
        s_small = scale;
        scale = (s_over > 0.f) ? sqrtf(s_small * s_over) : fminf(1.f, scale * 4.f);
      }
    }
  }
#ifdef DGS_KNN_STATS
  if (lane == 0) {
    atomicAdd(&stats[0], 1); atomicAdd(&stats[1], (fast && open == 0u) ? 1 : 0); atomicAdd(&stats[2], iters); atomicAdd(&stats[3], iters > 1 ? 1 : 0);
    atomicAdd(&stats[4], (int)(ts1 - ts0)); atomicAdd(&stats[5], (int)acc_b); atomicAdd(&stats[6], (int)acc_c); atomicAdd(&stats[7], (int)(wall_clock64() - ts0));
  }
#endif
  if (fast && open == 0u) return;
  // ---- the careful path: the per-query walk, 8 lanes per query
  {
    const int sub = lane & 7, jq = lane >> 3, pos = l * kLeaf + jq;
    const bool mine = jq >= j_lo && jq < j_hi;
    const float4 q = (mine && pos < n) ? b.sorted[pos] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int i = (mine && pos < n) ? (int)__float_as_uint(q.w) : -1;
    const bool alive = mine && pos < n && i >= 0 && i < n;
    KnnList L;
    knn_query_group(b, q.x, q.y, q.z, alive, k, L);
    if (alive) {
#pragma unroll
      for (int s = 0; s < kKnnSlots; s++) {
        const int slot = s * 8 + sub;
        if (slot < k) nbr[(size_t)pos * kKnnMax + slot] = (L.dist(s) < INFINITY) ? L.index(s) : -1;
      }
    }
  }
}

// One wave per 64 points, two phases.  (1) eight rounds of 8 points, 8 lanes per point: gather the neighbours, mean and covariance in
// double, reduced inside the 8-lane group; the six covariance entries of every point go to LDS.  (2) lane t regularises point t.
// (Regularising in phase 1's layout left 56 of 64 lanes idle through the 3x3 decomposition, which is most of this kernel's
// instructions: 1,964 VALU instructions per wave of 8 points.)  Same arithmetic per point as before, bit for bit.
#ifndef DGS_COV_ROUNDS
#define DGS_COV_ROUNDS 4
#endif
// A wave takes kCovRounds * 8 points: rounds of 8 points (8 lanes each) for the sums, then one lane per point for the regularisation.
// 8 rounds fill every lane of the second phase but leave a 65,536-point cloud with one wave per SIMD and nothing to hide its gathers
// behind; measured 8 / 4 / 2 rounds: 23.6 / 20.8 / 25.4 us per 65,536-point cloud, 21.5 / 14.6 / 15.2 us per 26,668 points.
constexpr int kCovRounds = DGS_COV_ROUNDS, kCovPerWave = kCovRounds * 8;
__global__ __launch_bounds__(kBlock) void gicp_cov_from_knn_kernel(const BvhView b, const float4* __restrict__ pts, const int n, const int k, const int method,
                                                                   const int* __restrict__ nbr, double* __restrict__ cov6) {
  __shared__ double s_c[kBlock / kWave][kCovPerWave][6];
  __shared__ int s_i[kBlock / kWave][kCovPerWave];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane & 7, grp = lane >> 3;
  const int wave_base = (blockIdx.x * (kBlock / kWave) + wv) * kCovPerWave;   // first point (position in index order) of this wave
  if (wave_base >= n) return;
  const double kk = (double)k;
#pragma unroll 2
  for (int r = 0; r < kCovRounds; r++) {
    const int pos = wave_base + r * 8 + grp;
    int i = -1;
    if (pos < n) i = (int)__float_as_uint(b.sorted[pos].w);
    const bool live = i >= 0 && i < n;
    // neighbours of this lane (slots q * 8 + sub < k); slots that found nothing are zero columns, as upstream's matrix
    double px[kKnnSlots], py[kKnnSlots], pz[kKnnSlots];
    bool use[kKnnSlots];
    double sx = 0, sy = 0, sz = 0;
#pragma unroll
    for (int q = 0; q < kKnnSlots; q++) {
      use[q] = (q * 8 + sub) < k;
      const int j = (live && use[q]) ? nbr[(size_t)pos * kKnnMax + q * 8 + sub] : -1;
      const float4 p = (j >= 0) ? pts[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      px[q] = p.x; py[q] = p.y; pz[q] = p.z;
      if (use[q]) { sx += px[q]; sy += py[q]; sz += pz[q]; }
    }
    const double mx = group8_sum_f64(sx) / kk, my = group8_sum_f64(sy) / kk, mz = group8_sum_f64(sz) / kk;
    double c[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < kKnnSlots; q++) {
      if (use[q]) {
        const double dx = px[q] - mx, dy = py[q] - my, dz = pz[q] - mz;
        c[0] += dx * dx; c[1] += dx * dy; c[2] += dx * dz; c[3] += dy * dy; c[4] += dy * dz; c[5] += dz * dz;
      }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) c[a] = group8_sum_f64(c[a]) / kk;
    if (sub == 0) {
#pragma unroll
      for (int a = 0; a < 6; a++) s_c[wv][r * 8 + grp][a] = c[a];
      s_i[wv][r * 8 + grp] = live ? i : -1;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's LDS stores have landed
  if (lane >= kCovPerWave) return;
  const int i = s_i[wv][lane];
  if (i < 0) return;
  const double* c = s_c[wv][lane];
  const double cov9[9] = {c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]};
  double out6[6];
  regularize_cov(cov9, method, out6);
#pragma unroll
  for (int a = 0; a < 6; a++) cov6[(size_t)i * 6 + a] = out6[a];
}

// ================================================================================================ K5 correspondences
// FastGICP::update_correspondences, search part: x' = float(T) * p in float, exact 1-NN in the target, accepted iff
// d^2 < corr_dist_threshold^2
__global__ __launch_bounds__(kBlock) void gicp_correspond_kernel(const BvhView b, const GicpItem* __restrict__ items,
                                                                 const GicpPair* __restrict__ pairs, const int n_pairs, const int cap_blocks,
                                                                 const float max_sq) {
  // workgroups go to the pairs whose queued evaluation is a linearisation (an LM trial re-uses the stored correspondences)
  int pair, slice, bpp;
  if (!deal_workgroup(n_pairs, cap_blocks, [&](int pi) { return pairs[pi].active != 0 && pairs[pi].eval_kind == 0; }, pair, slice, bpp)) return;
  const GicpPair& st = pairs[pair];
  const GicpItem it = items[pair];
  const int n = it.n;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = (float)st.Teval[k];
  const int sub = threadIdx.x & 7;
  constexpr int QPB = kBlock / 8;
  // every wave walks a contiguous stretch of the source in ITS index's (Hilbert) order (w = original index), 8 adjacent
  // points per round: the previous round's correspondences bound this round's searches (nn_warm_bound_round)
  const int run = (n + bpp * QPB - 1) / (bpp * QPB);
  const int first = (slice * (kBlock / kWave) + (threadIdx.x >> 6)) * (8 * run) + ((threadIdx.x & 63) >> 3);
  float px = 0.f, py = 0.f, pz = 0.f, prev_best = INFINITY;
  bool prev_found = false;
  for (int r = 0; r < run; r++) {
    const int pos = first + r * 8;
    const float4 p = (pos < n) ? it.src_sorted[pos] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int i = (pos < n) ? (int)__float_as_uint(p.w) : -1;
    const bool alive = pos < n && i >= 0 && i < n;
    const float x = affine_row_rn(T[0], T[1], T[2], T[3], p.x, p.y, p.z);
    const float y = affine_row_rn(T[4], T[5], T[6], T[7], p.x, p.y, p.z);
    const float z = affine_row_rn(T[8], T[9], T[10], T[11], p.x, p.y, p.z);
    float best;
    int bi;
    // nothing farther than the threshold can be a correspondence
    const float bound = fminf(max_sq, nn_warm_bound_round(prev_best, prev_found, x, y, z, px, py, pz));
    nn_query_group(b, x, y, z, alive, bound, best, bi);
    prev_found = alive && bi != 0x7FFFFFFF;
    prev_best = best;
    px = x; py = y; pz = z;
    if (alive && sub == 0) {
      const bool ok = (bi != 0x7FFFFFFF) && (best < max_sq);
      it.corr[i] = ok ? bi : -1;
      it.corr_sq[i] = (bi != 0x7FFFFFFF) ? best : max_sq;
    }
  }
}

// One correspondence's contribution to E = sum w e^T M e, b = sum w J^T M e, H = sum w J^T M J with J = [skew(t) | -I]
// (t = T p, e = mean_B - t; FastGICP: w = 1, FastVGICP: w = sqrt(points in the voxel)).  acc: E, b[6], upper triangle of H.
template <bool WEIGHTED>
__device__ __forceinline__ void gicp_accumulate(double* acc, const double* M, const double e0, const double e1, const double e2, const double t0,
                                                const double t1, const double t2, const double w, const bool full) {
  double m0 = M[0] * e0 + M[1] * e1 + M[2] * e2;
  double m1 = M[1] * e0 + M[3] * e1 + M[4] * e2;
  double m2 = M[2] * e0 + M[4] * e1 + M[5] * e2;
  double err = e0 * m0 + e1 * m1 + e2 * m2;
  if (WEIGHTED) err *= w;
  acc[0] += err;
  if (full) {
    double Mw[6];
#pragma unroll
    for (int k = 0; k < 6; k++) Mw[k] = WEIGHTED ? w * M[k] : M[k];
    if (WEIGHTED) { m0 *= w; m1 *= w; m2 *= w; }
    // b = J^T M e = [ (Me) x t ; -Me ]
    acc[1] += m1 * t2 - m2 * t1;
    acc[2] += m2 * t0 - m0 * t2;
    acc[3] += m0 * t1 - m1 * t0;
    acc[4] += -m0;
    acc[5] += -m1;
    acc[6] += -m2;
    // H = J^T M J = [[S^T M S, -S^T M], [-M S, M]] with S = skew(t):  G = M S (3x3), then S^T G and -G^T
    const double Mf[9] = {Mw[0], Mw[1], Mw[2], Mw[1], Mw[3], Mw[4], Mw[2], Mw[4], Mw[5]};
    const double Sk[9] = {0, -t2, t1, t2, 0, -t0, -t1, t0, 0};
    double G[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) G[r * 3 + c] = Mf[r * 3 + 0] * Sk[0 * 3 + c] + Mf[r * 3 + 1] * Sk[1 * 3 + c] + Mf[r * 3 + 2] * Sk[2 * 3 + c];
    double Hrr[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) Hrr[r * 3 + c] = Sk[0 * 3 + r] * G[0 * 3 + c] + Sk[1 * 3 + r] * G[1 * 3 + c] + Sk[2 * 3 + r] * G[2 * 3 + c];
    // upper triangle, row-major: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
    acc[7] += Hrr[0]; acc[8] += Hrr[1]; acc[9] += Hrr[2]; acc[10] += -G[0 * 3 + 0]; acc[11] += -G[1 * 3 + 0]; acc[12] += -G[2 * 3 + 0];
    acc[13] += Hrr[4]; acc[14] += Hrr[5]; acc[15] += -G[0 * 3 + 1]; acc[16] += -G[1 * 3 + 1]; acc[17] += -G[2 * 3 + 1];
    acc[18] += Hrr[8]; acc[19] += -G[0 * 3 + 2]; acc[20] += -G[1 * 3 + 2]; acc[21] += -G[2 * 3 + 2];
    acc[22] += Mw[0]; acc[23] += Mw[1]; acc[24] += Mw[2];
    acc[25] += Mw[3]; acc[26] += Mw[4];
    acc[27] += Mw[5];
  }
}

// Mahalanobis matrix of one correspondence: (C_B + R C_A R^T)^-1, symmetric 6-vector (3x3 block of upstream's 4x4)
__device__ __forceinline__ void gicp_mahalanobis(const double* T, const double* CA, const double* CB, double* M) {
  const double a0 = CA[0], a1 = CA[1], a2 = CA[2], a3 = CA[3], a4 = CA[4], a5 = CA[5];
  double RC[9];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const double r0 = T[r * 4 + 0], r1 = T[r * 4 + 1], r2 = T[r * 4 + 2];
    RC[r * 3 + 0] = r0 * a0 + r1 * a1 + r2 * a2;
    RC[r * 3 + 1] = r0 * a1 + r1 * a3 + r2 * a4;
    RC[r * 3 + 2] = r0 * a2 + r1 * a4 + r2 * a5;
  }
  double S[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) S[r * 3 + c] = RC[r * 3 + 0] * T[c * 4 + 0] + RC[r * 3 + 1] * T[c * 4 + 1] + RC[r * 3 + 2] * T[c * 4 + 2];
  S[0] += CB[0]; S[1] += CB[1]; S[2] += CB[2]; S[3] += CB[1]; S[4] += CB[3]; S[5] += CB[4]; S[6] += CB[2]; S[7] += CB[4]; S[8] += CB[5];
  double Mi[9];
  inv3_d(S, Mi);
  M[0] = Mi[0]; M[1] = Mi[1]; M[2] = Mi[2]; M[3] = Mi[4]; M[4] = Mi[5]; M[5] = Mi[8];
}

// wave DPP sums -> LDS -> one fixed-order row of kAccumPad doubles per workgroup
template <bool FUSED = false>
__device__ __forceinline__ void gicp_block_reduce(const double* acc, double* __restrict__ row) {
  __shared__ double sm[kBlock / kWave][kAccumPad];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kAccum; k++) {
    const double v = wave_sum_to_lane63(acc[k]);
    if (lane == 63) sm[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kAccumPad) {
    double v = 0.0;
    if (threadIdx.x < kAccum) v = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    if (FUSED) {
      handoff_store_row(row + threadIdx.x, v);   // write-through: see gicp_close_round and common.h, "in-launch hand-off"
      handoff_drain_stores();
    } else {
      row[threadIdx.x] = v;
    }
  }
}

// ---- the optimiser (forced inline: inside the fused linearize kernels the launch bound of the linearize loop must govern, the
// tail spills what does not fit)
// ================================================================================================ solver
// so3_exp / se3_exp of fast_gicp (quaternion form); out = rows 0..2 of the 4x4, row-major 3x4
__device__ __forceinline__ void se3_exp_dev(const double* a, double* T) {
  const double wx = a[0], wy = a[1], wz = a[2];
  const double theta_sq = wx * wx + wy * wy + wz * wz;
  double imag, real;
  if (theta_sq < 1e-10) {
    const double tq = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * tq;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * tq;
  } else {
    const double theta = sqrt(theta_sq), half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  const double qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  const double theta = sqrt(theta_sq);
  double V[9];
  if (theta < 1e-10) {
    for (int i = 0; i < 9; i++) V[i] = R[i];
  } else {
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) O2[r * 3 + c] = O[r * 3 + 0] * O[0 * 3 + c] + O[r * 3 + 1] * O[1 * 3 + c] + O[r * 3 + 2] * O[2 * 3 + c];
    const double c1 = (1.0 - cos(theta)) / theta_sq, c2 = (theta - sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
  }
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[r * 4 + c] = R[r * 3 + c];
    T[r * 4 + 3] = V[r * 3 + 0] * a[3] + V[r * 3 + 1] * a[4] + V[r * 3 + 2] * a[5];
  }
}

__device__ __forceinline__ void iso_mul(const double* A, const double* B, double* C) {  // 3x4 isometries, C = A * B
  double T[12];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 4; c++) T[r * 4 + c] = A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c] + A[r * 4 + 2] * B[2 * 4 + c];
    T[r * 4 + 3] += A[r * 4 + 3];
  }
  for (int k = 0; k < 12; k++) C[k] = T[k];
}

__device__ __forceinline__ bool gicp_is_converged(const double* delta, const GicpConsts& c) {
  double rmax = 0, tmax = 0;
  for (int r = 0; r < 3; r++) {
    for (int cc = 0; cc < 3; cc++) rmax = fmax(rmax, fabs(delta[r * 4 + cc] - (r == cc ? 1.0 : 0.0)) / c.rot_eps);
    tmax = fmax(tmax, fabs(delta[r * 4 + 3]) / c.trans_eps);
  }
  return fmax(rmax, tmax) < 1;
}

__device__ __forceinline__ void gicp_queue(GicpPair* st, const double* T, int kind, bool writer) {
  if (writer) {
    for (int k = 0; k < 12; k++) st->Teval[k] = T[k];
    st->eval_kind = kind;
  }
}

__device__ __forceinline__ void gicp_try_lm(GicpPair* st, GicpSolver& s, const GicpConsts& c, bool writer) {
  solve6_step(s.H, s.b, s.lambda, s.d);
  se3_exp_dev(s.d, s.delta);
  iso_mul(s.delta, s.x0, s.xi);
  gicp_queue(st, s.xi, 1, writer);
  s.phase = GP_ERROR_WAIT;
}

// after a successful optimisation step: convergence test of LsqRegistration::computeTransformation
__device__ __forceinline__ void gicp_after_step(GicpPair* st, GicpSolver& s, const GicpConsts& c, bool writer) {
  const bool conv = gicp_is_converged(s.delta, c);
  s.converged = conv ? 1 : 0;
  if (!conv && s.iteration + 1 < c.max_iterations) {
    s.iteration++;
    gicp_queue(st, s.x0, 0, writer);
    s.phase = GP_LINEARIZE_WAIT;
  } else {
    s.phase = GP_DONE;
  }
}

__device__ __forceinline__ void gicp_advance(GicpPair* st, GicpSolver& s, const GicpConsts& c, bool writer) {
  s.evaluations++;
  if (s.phase == GP_PROBE) {
    s.phase = GP_DONE;
    return;
  }
  if (s.phase == GP_LINEARIZE_WAIT) {
    if (c.optimizer == DGS_GICP_OPT_GAUSS_NEWTON) {
      solve6_step(s.H, s.b, 0.0, s.d);
      se3_exp_dev(s.d, s.delta);
      iso_mul(s.delta, s.x0, s.x0);
      gicp_after_step(st, s, c, writer);
      return;
    }
    if (s.lambda < 0.0) {
      double m = 0;
      for (int k = 0; k < 6; k++) m = fmax(m, fabs(s.H[k * 6 + k]));
      s.lambda = c.lm_init_lambda_factor * m;
    }
    s.nu = 2.0;
    s.lm_try = 0;
    gicp_try_lm(st, s, c, writer);
    return;
  }
  if (s.phase == GP_ERROR_WAIT) {
    double denom = 0;
    for (int k = 0; k < 6; k++) denom += s.d[k] * (s.lambda * s.d[k] - s.b[k]);
    const double rho = (s.y0 - s.yi) / denom;
    if (rho < 0) {
      if (gicp_is_converged(s.delta, c)) {  // step_lm returns true without moving x0
        gicp_after_step(st, s, c, writer);
        return;
      }
      s.lambda = s.nu * s.lambda;
      s.nu = 2 * s.nu;
      s.lm_try++;
      if (s.lm_try < c.lm_max_iterations) {
        gicp_try_lm(st, s, c, writer);
      } else {  // "lm not converged!!": the outer loop breaks, converged_ stays false
        s.converged = 0;
        s.phase = GP_DONE;
      }
      return;
    }
    for (int k = 0; k < 12; k++) s.x0[k] = s.xi[k];
    const double t = 2 * rho - 1;
    s.lambda = s.lambda * fmax(1.0 / 3.0, 1 - t * t * t);
    s.y0 = s.yi;
    gicp_after_step(st, s, c, writer);
  }
}

// The optimiser step of a registration from its partial rows: gicp_solve_kernel's body, also run by the LAST workgroup of a
// pair inside the linearize launch (fused rounds).  Rows are summed in the same order either way (bit-identical results).
template <bool FUSED>
__device__ __forceinline__ void gicp_step_from_rows(GicpPair* st, const double* __restrict__ partials, const int nblocks, const GicpConsts& c,
                                                    int* __restrict__ done_counter, const int launch) {
  __shared__ double sm[kBlock / kAccumPad][kAccumPad];
  __shared__ double tot[kAccumPad];
  __shared__ GicpSolver s_lds;   // the state lives in LDS: in registers it would cost the linearize loop half its occupancy
  const int col = threadIdx.x % kAccumPad, grp = threadIdx.x / kAccumPad;
  constexpr int G = kBlock / kAccumPad;
  double v = 0.0;
  for (int b = grp; b < nblocks; b += G) {
    const double* r = partials + (size_t)b * kAccumPad + col;
    v += FUSED ? handoff_load_row(r) : *r;
  }
  sm[grp][col] = v;
  {  // state -> LDS, word by word, by the whole workgroup
    const int* src = reinterpret_cast<const int*>(&st->s);
    int* dst = reinterpret_cast<int*>(&s_lds);
    for (int w = threadIdx.x; w < (int)(sizeof(GicpSolver) / sizeof(int)); w += kBlock) dst[w] = src[w];
  }
  __syncthreads();
  if (threadIdx.x < kAccumPad) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < G; k++) t += sm[k][threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x >= kWave) return;
  const bool writer = threadIdx.x == 0;
  GicpSolver& s = s_lds;
  if (st->eval_kind == 0) {
    if (writer) {
      s.y0 = tot[0];
      for (int k = 0; k < 6; k++) s.b[k] = tot[1 + k];
      int q = 7;
      for (int i = 0; i < 6; i++)
        for (int j = i; j < 6; j++) {
          s.H[i * 6 + j] = tot[q];
          s.H[j * 6 + i] = tot[q];
          q++;
        }
    }
  } else if (writer) {
    s.yi = tot[0];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  gicp_advance(st, s, c, writer);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  {
    const int* src = reinterpret_cast<const int*>(&s_lds);
    int* dst = reinterpret_cast<int*>(&st->s);
    for (int w = threadIdx.x; w < (int)(sizeof(GicpSolver) / sizeof(int)); w += kWave) dst[w] = src[w];
  }
  if (writer) {
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < 4; cc++) st->final_T[cc * 4 + r] = (float)s.x0[r * 4 + cc];
    st->final_T[3] = st->final_T[7] = st->final_T[11] = 0.f;
    st->final_T[15] = 1.f;
    if (s.phase == GP_DONE) {
      st->active = 0;
      if (FUSED) st->last_launch = launch;
      atomicAdd(done_counter, 1);
    }
  }
}

// Fused rounds: after its row is published (write-through stores, drained) a workgroup takes a ticket of its pair; the one that
// takes the last ticket runs the pair's optimiser step -- the scheme of ndt_derivatives_kernel<.., fused> (ndt_align.hip).
__device__ __forceinline__ void gicp_close_round(GicpPair* pairs, const int pair, const int nblocks, const double* __restrict__ partials_of_pair,
                                                 const GicpConsts& c, int* __restrict__ done_counter, const int launch) {
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    s_last = handoff_take_ticket(&pairs[pair].ticket, nblocks) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  gicp_step_from_rows<true>(pairs + pair, partials_of_pair, nblocks, c, done_counter, launch);
}

// ================================================================================================ K6 linearize / error
// fused: the linearize loop keeps its 4 waves per SIMD; the optimiser tail (one workgroup per pair and launch) spills what does not fit
template <bool FUSED>
__global__ __launch_bounds__(kBlock, FUSED ? 4 : 1) void gicp_linearize_kernel(const GicpItem* __restrict__ items, const float4* __restrict__ tgt,
                                                                const double* __restrict__ cov_t, GicpPair* __restrict__ pairs,
                                                                double* __restrict__ partials, const int n_pairs, const int cap_blocks,
                                                                int* __restrict__ pair_blocks, const GicpConsts consts, int* __restrict__ done_counter,
                                                                const int launch) {
  int pair, slice, nblocks;
  if (!deal_workgroup(n_pairs, cap_blocks, [&](int pi) { return FUSED ? (launch <= pairs[pi].last_launch) : (pairs[pi].active != 0); }, pair, slice, nblocks)) return;
  if (slice == 0 && threadIdx.x == 0) pair_blocks[pair] = nblocks;
  const GicpPair& st = pairs[pair];
  const GicpItem it = items[pair];
  const int n = it.n;
  const float4* __restrict__ src = it.src;
  const double* __restrict__ cov_s = it.cov_s;
  const int* __restrict__ corr = it.corr;
  double* __restrict__ mahal = it.mahal;
  const bool full = st.eval_kind == 0;
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = st.Teval[k];
  double acc[kAccum];
#pragma unroll
  for (int k = 0; k < kAccum; k++) acc[k] = 0.0;

  for (int i = slice * kBlock + threadIdx.x; i < n; i += nblocks * kBlock) {
    const int j = corr[i];
    if (j < 0) continue;
    const float4 pa = src[i], pb = tgt[j];
    double M[6];
    if (full) {
      gicp_mahalanobis(T, cov_s + (size_t)i * 6, cov_t + (size_t)j * 6, M);
      double* mo = mahal + (size_t)i * 6;
#pragma unroll
      for (int k = 0; k < 6; k++) mo[k] = M[k];
    } else {
      const double* mo = mahal + (size_t)i * 6;
#pragma unroll
      for (int k = 0; k < 6; k++) M[k] = mo[k];
    }
    const double ax = pa.x, ay = pa.y, az = pa.z;
    const double t0 = T[0] * ax + T[1] * ay + T[2] * az + T[3];
    const double t1 = T[4] * ax + T[5] * ay + T[6] * az + T[7];
    const double t2 = T[8] * ax + T[9] * ay + T[10] * az + T[11];
    gicp_accumulate<false>(acc, M, (double)pb.x - t0, (double)pb.y - t1, (double)pb.z - t2, t0, t1, t2, 1.0, full);
  }
  gicp_block_reduce<FUSED>(acc, partials + ((size_t)pair * cap_blocks + slice) * kAccumPad);
  if (FUSED) gicp_close_round(pairs, pair, nblocks, partials + (size_t)pair * cap_blocks * kAccumPad, consts, done_counter, launch);
}

// ================================================================================================ FAST_VGICP linearize / error
// FastVGICP::update_correspondences + linearize (eval_kind 0) or compute_error (eval_kind 1) in one pass: the voxel of T p
// (double) and its DIRECT1 / 7 / 27 neighbours are looked up in the dense cell table -- no tree, no distance gate --, every hit is
// a correspondence with Mahalanobis (cov_voxel + R cov_p R^T)^-1 and weight sqrt(points in the voxel).  An error-only
// evaluation re-uses the voxel ids and Mahalanobis matrices stored by the last linearisation, as upstream does.
template <bool FUSED>
__global__ __launch_bounds__(kBlock, FUSED ? 3 : 1) void vgicp_linearize_kernel(const GicpItem* __restrict__ items, const VgicpMap m,
                                                                 GicpPair* __restrict__ pairs, double* __restrict__ partials,
                                                                 const int n_pairs, const int cap_blocks, int* __restrict__ pair_blocks,
                                                                 const GicpConsts consts, int* __restrict__ done_counter, const int launch) {
  int pair, slice, nblocks;
  if (!deal_workgroup(n_pairs, cap_blocks, [&](int pi) { return FUSED ? (launch <= pairs[pi].last_launch) : (pairs[pi].active != 0); }, pair, slice, nblocks)) return;
  if (slice == 0 && threadIdx.x == 0) pair_blocks[pair] = nblocks;
  const GicpPair& st = pairs[pair];
  const GicpItem it = items[pair];
  const int n = it.n, no = m.n_offsets;
  const bool full = st.eval_kind == 0;
  double T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = st.Teval[k];
  double acc[kAccum];
#pragma unroll
  for (int k = 0; k < kAccum; k++) acc[k] = 0.0;

  for (int i = slice * kBlock + threadIdx.x; i < n; i += nblocks * kBlock) {
    const float4 pa = it.src[i];
    const double ax = pa.x, ay = pa.y, az = pa.z;
    const double t0 = T[0] * ax + T[1] * ay + T[2] * az + T[3];
    const double t1 = T[4] * ax + T[5] * ay + T[6] * az + T[7];
    const double t2 = T[8] * ax + T[9] * ay + T[10] * az + T[11];
    // GaussianVoxelMap::voxel_coord: floor(x / resolution - 0.5), relative to the table's first cell
    const int c0 = (int)floor(t0 / m.resolution - 0.5) - m.min_c[0];
    const int c1 = (int)floor(t1 / m.resolution - 0.5) - m.min_c[1];
    const int c2 = (int)floor(t2 / m.resolution - 0.5) - m.min_c[2];
    for (int k = 0; k < no; k++) {
      const size_t slot = (size_t)i * no + k;
      int v;
      if (full) {
        int dx = 0, dy = 0, dz = 0;
        if (m.search == DGS_VGICP_DIRECT7) {  // (0,0,0) (1,0,0) (-1,0,0) (0,1,0) (0,-1,0) (0,0,1) (0,0,-1)
          dx = (k == 1) - (k == 2);
          dy = (k == 3) - (k == 4);
          dz = (k == 5) - (k == 6);
        } else if (m.search == DGS_VGICP_DIRECT27) {
          dx = k / 9 - 1;
          dy = (k / 3) % 3 - 1;
          dz = k % 3 - 1;
        }
        const int x = c0 + dx, y = c1 + dy, z = c2 + dz;
        v = -1;
        if (x >= 0 && x < m.div[0] && y >= 0 && y < m.div[1] && z >= 0 && z < m.div[2]) v = m.cell2vox[x + y * m.mul1 + z * m.mul2];
        it.corr[slot] = v;
      } else {
        v = it.corr[slot];
      }
      if (v < 0) continue;
      const VgicpVoxel* vx = m.vox + v;
      double M[6];
      if (full) {
        gicp_mahalanobis(T, it.cov_s + (size_t)i * 6, vx->cov, M);
#pragma unroll
        for (int a = 0; a < 6; a++) it.mahal[slot * 6 + a] = M[a];
      } else {
#pragma unroll
        for (int a = 0; a < 6; a++) M[a] = it.mahal[slot * 6 + a];
      }
      gicp_accumulate<true>(acc, M, vx->mean[0] - t0, vx->mean[1] - t1, vx->mean[2] - t2, t0, t1, t2, vx->w, full);
    }
  }
  gicp_block_reduce<FUSED>(acc, partials + ((size_t)pair * cap_blocks + slice) * kAccumPad);
  if (FUSED) gicp_close_round(pairs, pair, nblocks, partials + (size_t)pair * cap_blocks * kAccumPad, consts, done_counter, launch);
}


__global__ __launch_bounds__(kBlock) void gicp_solve_kernel(GicpPair* __restrict__ pairs, const double* __restrict__ all_partials,
                                                            const int* __restrict__ pair_blocks, const int cap_blocks, const GicpConsts c,
                                                            int* __restrict__ done_counter) {
  GicpPair* st = pairs + blockIdx.x;  // one workgroup per registration of the batch
  if (!st->active) return;
  gicp_step_from_rows<false>(st, all_partials + (size_t)blockIdx.x * cap_blocks * kAccumPad, pair_blocks[blockIdx.x], c, done_counter, -1);
}

struct GicpInit {
  double x0[12];
  int probe_kind;  // -1: normal align; 0 / 1: probe a linearisation / an error evaluation
  int n;           // source size: an empty source never becomes active (PCL's initCompute refuses it)
};

__global__ void gicp_init_kernel(GicpPair* __restrict__ pairs, const GicpInit* __restrict__ inits, const int n_pairs) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= n_pairs) return;
  GicpPair* st = pairs + pi;
  const GicpInit* init = inits + pi;
  GicpSolver s;
  s.phase = (init->n <= 0) ? GP_DONE : (init->probe_kind >= 0) ? GP_PROBE : GP_LINEARIZE_WAIT;
  s.iteration = 0;
  s.evaluations = 0;
  s.converged = 0;
  s.lm_try = 0;
  s.pad = 0;
  for (int k = 0; k < 12; k++) { s.x0[k] = init->x0[k]; s.xi[k] = init->x0[k]; s.delta[k] = (k % 5 == 0) ? 1.0 : 0.0; st->Teval[k] = init->x0[k]; }
  for (int k = 0; k < 36; k++) s.H[k] = 0;
  for (int k = 0; k < 6; k++) { s.b[k] = 0; s.d[k] = 0; }
  s.y0 = s.yi = 0;
  s.lambda = -1.0;
  s.nu = 2.0;
  st->s = s;
  st->eval_kind = (init->probe_kind > 0) ? 1 : 0;
  st->active = (init->n > 0) ? 1 : 0;
  st->last_launch = (init->n > 0) ? 0x7FFFFFFF : -1;
  st->ticket = 0;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) st->final_T[c * 4 + r] = (float)init->x0[r * 4 + c];
  st->final_T[3] = st->final_T[7] = st->final_T[11] = 0.f;
  st->final_T[15] = 1.f;
}

// ================================================================================================ host side
static void fill_gconsts(dgs_handle* h) {
  GicpConsts& c = h->gconsts;
  const dgs_params& p = h->prm;
  c.trans_eps = p.transformation_epsilon;
  c.rot_eps = p.gicp_rotation_epsilon;
  c.lm_init_lambda_factor = p.gicp_lm_init_lambda_factor;
  const float dmax = (float)p.gicp_max_correspondence_distance;
  c.max_corr_sq = dmax * dmax;
  c.max_iterations = p.maximum_iterations;
  c.optimizer = p.gicp_optimizer;
  c.lm_max_iterations = p.gicp_lm_max_iterations;
  c.k = p.gicp_correspondence_randomness;
  c.regularization = p.gicp_regularization;
}

static void fill_gconsts(dgs_handle* h);
static int ensure_covariance(dgs_handle* h, CloudState& c);

int gicp_ensure_target_covariance(dgs_handle* h) {
  fill_gconsts(h);
  return ensure_covariance(h, *h->tgt);
}

static int ensure_covariance(dgs_handle* h, CloudState& c) {
  if (c.cov_valid && c.cov_k == h->gconsts.k && c.cov_reg == h->gconsts.regularization) return DGS_OK;
  if (h->gconsts.k > kKnnMax) {
    h->err = "reg_correspondence_randomness > 32 is not supported by the HIP k-NN";
    return DGS_ERR_UNSUPPORTED;
  }
  if (!c.bvh.valid) {
    // the target of a batch is searched by every candidate at every linearisation: worth the slower k-d ordered build (nn_bvh.hip)
    int rc = bvh_build(h, c.bvh, c.pts.ptr, c.n, nullptr, h->batch_kd && &c == h->tgt);
    if (rc) return rc;
  }
  DGS_HIP_TRY(h, c.cov.reserve((size_t)c.n * 6));
  const BvhView v = make_bvh_view(c.bvh);
  DGS_HIP_TRY(h, h->knn_nbr.reserve((size_t)c.n * kKnnMax));
  DGS_HIP_TRY(h, h->knn_stats.reserve(8));
  // rounds per wave: enough waves to fill the chip several times over, and stretches long enough for the warm bounds to pay
  const int run = (int)std::max<int64_t>(1, std::min<int64_t>(h->knn_rounds, (c.n + 8 * (int64_t)h->knn_min_waves - 1) / (8 * (int64_t)h->knn_min_waves)));
  const int64_t waves = (c.n + 8 * (int64_t)run - 1) / (8 * (int64_t)run);
  int slot = prof_begin(h, DGS_K_GICP_COVARIANCE);
  if (h->knn_leaf) {
    const int64_t n_leaves = (c.n + kLeaf - 1) / kLeaf;
    // waves per leaf (each answers 8 / parts of the leaf's queries after its own walk, whose loops run over ITS queries only): measured
    // 1 / 2 / 4 / 8 waves per leaf: 0.168 / 0.148 / 0.143 / 0.160 ms per 65,536-point cloud, 0.099 / 0.076 / 0.065 / 0.069 ms per 26,668 points
    int parts = 1;
    while (parts < 4 && n_leaves * parts < 32768) parts *= 2;
    if (h->knn_parts > 0) parts = h->knn_parts;
#ifdef DGS_KNN_STATS
    (void)hipMemsetAsync(h->knn_stats.ptr, 0, 8 * sizeof(int), h->stream);
#endif
    hipLaunchKernelGGL(gicp_knn_leaf_kernel, dim3((unsigned)((n_leaves * parts + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock), 0, h->stream, v, (int)c.n,
                       h->gconsts.k, parts, h->knn_nbr.ptr, h->knn_stats.ptr);
  } else {
    hipLaunchKernelGGL(gicp_knn_kernel, dim3((unsigned)((waves + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock), 0, h->stream, v, (int)c.n, h->gconsts.k, run,
                       h->knn_nbr.ptr);
  }
  constexpr int kCovPerBlock = kCovPerWave * (kBlock / kWave);
  hipLaunchKernelGGL(gicp_cov_from_knn_kernel, dim3((unsigned)(((int64_t)c.n + kCovPerBlock - 1) / kCovPerBlock)), dim3(kBlock), 0, h->stream, v, c.pts.ptr, (int)c.n,
                     h->gconsts.k, h->gconsts.regularization | (h->prm.gicp_cov_jacobi_svd ? 0x100 : 0), h->knn_nbr.ptr, c.cov.ptr);
  prof_end(h, DGS_K_GICP_COVARIANCE, slot);
  DGS_HIP_TRY(h, hipGetLastError());
#ifdef DGS_KNN_STATS
  if (h->knn_leaf) {
    int st[8];
    (void)hipMemcpyAsync(st, h->knn_stats.ptr, sizeof(st), hipMemcpyDeviceToHost, h->stream);
    (void)hipStreamSynchronize(h->stream);
    const double w = std::max(st[0], 1);
    fprintf(stderr, "[knn] n %lld waves %d answered cooperatively %d (%.2f %%) gather rounds per wave %.3f waves that retried %d; per wave (us): window bounds %.2f, "
            "leaf walk %.2f, selection %.2f, whole %.2f\n", (long long)c.n, st[0], st[1], 100.0 * st[1] / w, (double)st[2] / w, st[3], st[4] * 0.01 / w, st[5] * 0.01 / w,
            st[6] * 0.01 / w, st[7] * 0.01 / w);
  }
#endif
  c.cov_valid = true;
  c.cov_k = h->gconsts.k;
  c.cov_reg = h->gconsts.regularization;
  return DGS_OK;
}

// Launch shape of one batch: per-pair caps (what a lone registration gets) and grids dealt over the active pairs.
struct GicpLaunch {
  int n_pairs, cap_c, cap_l, grid_c, grid_l;
};

static GicpLaunch gicp_choose_launch(int n_pairs, int64_t max_n) {
  GicpLaunch L;
  L.n_pairs = n_pairs;
  L.cap_c = (int)std::max<int64_t>(1, std::min<int64_t>((max_n * 8 + kBlock - 1) / kBlock, 4096));
  L.cap_l = (int)std::max<int64_t>(1, std::min<int64_t>((max_n + kBlock - 1) / kBlock, 512));
  L.grid_c = (int)std::max<int64_t>(n_pairs, std::min<int64_t>((int64_t)n_pairs * L.cap_c, 8192));
  L.grid_l = (int)std::max<int64_t>(n_pairs, std::min<int64_t>((int64_t)n_pairs * L.cap_l, 2048));
  return L;
}

static void gicp_launch_round(dgs_handle* h, const GicpLaunch& L, const int launch) {
  // fused rounds (default; DGS_GICP_FUSED=0 restores the separate solve launch): the optimiser step of a pair runs in the last
  // workgroup of its linearize slice -- two dependent launches per evaluation instead of three (VGICP: one instead of two)
  const bool fused = h->gicp_fused;
  if (h->prm.method == DGS_METHOD_VGICP) {
    int slot = prof_begin(h, DGS_K_GICP_LINEARIZE);
    if (fused)
      hipLaunchKernelGGL(vgicp_linearize_kernel<true>, dim3(L.grid_l), dim3(kBlock), 0, h->stream, h->gitems.ptr, h->vmap, h->gpairs.ptr, h->partials.ptr,
                         L.n_pairs, L.cap_l, h->pair_blocks.ptr, h->gconsts, h->done_counter.ptr, launch);
    else
      hipLaunchKernelGGL(vgicp_linearize_kernel<false>, dim3(L.grid_l), dim3(kBlock), 0, h->stream, h->gitems.ptr, h->vmap, h->gpairs.ptr, h->partials.ptr,
                         L.n_pairs, L.cap_l, h->pair_blocks.ptr, h->gconsts, h->done_counter.ptr, launch);
    prof_end(h, DGS_K_GICP_LINEARIZE, slot);
    if (!fused)
      hipLaunchKernelGGL(gicp_solve_kernel, dim3(L.n_pairs), dim3(kBlock), 0, h->stream, h->gpairs.ptr, h->partials.ptr, h->pair_blocks.ptr, L.cap_l,
                         h->gconsts, h->done_counter.ptr);
    return;
  }
  const BvhView v = make_bvh_view(h->tgt->bvh);
  int slot = prof_begin(h, DGS_K_NN_SEARCH);
  hipLaunchKernelGGL(gicp_correspond_kernel, dim3(L.grid_c), dim3(kBlock), 0, h->stream, v, h->gitems.ptr, h->gpairs.ptr, L.n_pairs, L.cap_c,
                     h->gconsts.max_corr_sq);
  prof_end(h, DGS_K_NN_SEARCH, slot);
  slot = prof_begin(h, DGS_K_GICP_LINEARIZE);
  if (fused)
    hipLaunchKernelGGL(gicp_linearize_kernel<true>, dim3(L.grid_l), dim3(kBlock), 0, h->stream, h->gitems.ptr, h->tgt->pts.ptr, h->tgt->cov.ptr, h->gpairs.ptr,
                       h->partials.ptr, L.n_pairs, L.cap_l, h->pair_blocks.ptr, h->gconsts, h->done_counter.ptr, launch);
  else
    hipLaunchKernelGGL(gicp_linearize_kernel<false>, dim3(L.grid_l), dim3(kBlock), 0, h->stream, h->gitems.ptr, h->tgt->pts.ptr, h->tgt->cov.ptr, h->gpairs.ptr,
                       h->partials.ptr, L.n_pairs, L.cap_l, h->pair_blocks.ptr, h->gconsts, h->done_counter.ptr, launch);
  prof_end(h, DGS_K_GICP_LINEARIZE, slot);
  if (!fused)
    hipLaunchKernelGGL(gicp_solve_kernel, dim3(L.n_pairs), dim3(kBlock), 0, h->stream, h->gpairs.ptr, h->partials.ptr, h->pair_blocks.ptr, L.cap_l,
                       h->gconsts, h->done_counter.ptr);
}

// pinned staging: [0,64) done flags | inits | items | pairs read back
static size_t gicp_pinned_layout(int n, size_t* off_init, size_t* off_items, size_t* off_pairs) {
  size_t o = 64;
  *off_init = o;
  o += (size_t)n * sizeof(GicpInit);
  o = (o + 63) & ~(size_t)63;
  *off_items = o;
  o += (size_t)n * sizeof(GicpItem);
  o = (o + 63) & ~(size_t)63;
  *off_pairs = o;
  o += (size_t)n * sizeof(GicpPair);
  return o;
}

// Covariances and indices of every cloud, work arrays, per-pair state: everything a batch needs before its first round.
static int gicp_start(dgs_handle* h, int n, CloudState* const* srcs, const double* x0_rows, int probe_kind, GicpLaunch* L_out, int* n_live) {
  hipStream_t st = h->stream;
  fill_gconsts(h);
  int rc = ensure_covariance(h, *h->tgt);
  if (rc) return rc;
  const bool vgicp = h->prm.method == DGS_METHOD_VGICP;
  if (vgicp && !h->vmap_valid) {
    rc = vgicp_build_map(h);
    if (rc) return rc;
  }
  const int64_t per_point = vgicp ? h->vmap.n_offsets : 1;  // correspondences per source point
  int64_t total = 0, max_n = 1;
  *n_live = 0;
  for (int i = 0; i < n; i++) {
    if (srcs[i]->n <= 0) continue;
    rc = ensure_covariance(h, *srcs[i]);
    if (rc) return rc;
    total += srcs[i]->n;
    max_n = std::max<int64_t>(max_n, srcs[i]->n);
    (*n_live)++;
  }
  const GicpLaunch L = gicp_choose_launch(n, max_n);
  *L_out = L;
  DGS_HIP_TRY(h, h->corr.reserve((size_t)std::max<int64_t>(total, 1) * per_point));
  DGS_HIP_TRY(h, h->corr_sq.reserve((size_t)std::max<int64_t>(total, 1)));
  DGS_HIP_TRY(h, h->mahal.reserve((size_t)std::max<int64_t>(total, 1) * per_point * 6));
  DGS_HIP_TRY(h, h->gpairs.reserve(n));
  DGS_HIP_TRY(h, h->gitems.reserve(n));
  DGS_HIP_TRY(h, h->inits.reserve(n));  // sizeof(NdtInit) >= sizeof(GicpInit)
  static_assert(sizeof(NdtInit) >= sizeof(GicpInit), "init staging buffer is shared with NDT");
  DGS_HIP_TRY(h, h->pair_blocks.reserve(n));
  DGS_HIP_TRY(h, h->done_counter.reserve(16));
  DGS_HIP_TRY(h, h->partials.reserve((size_t)n * L.cap_l * kAccumPad + 64));
  size_t oi, ot, op;
  const size_t bytes = gicp_pinned_layout(n, &oi, &ot, &op);
  if (ensure_pinned(h, bytes) != DGS_OK) return DGS_ERR_HIP;
  char* base = reinterpret_cast<char*>(h->pinned);
  GicpInit* hin = reinterpret_cast<GicpInit*>(base + oi);
  GicpItem* hit = reinterpret_cast<GicpItem*>(base + ot);
  int64_t off = 0;
  for (int i = 0; i < n; i++) {
    const CloudState& c = *srcs[i];
    for (int k = 0; k < 12; k++) hin[i].x0[k] = x0_rows[12 * i + k];
    hin[i].probe_kind = probe_kind;
    hin[i].n = (int)c.n;
    hit[i].src = c.pts.ptr;
    hit[i].src_sorted = c.bvh.sorted.ptr;
    hit[i].cov_s = c.cov.ptr;
    hit[i].corr = h->corr.ptr + off * per_point;
    hit[i].corr_sq = h->corr_sq.ptr + off;
    hit[i].mahal = h->mahal.ptr + off * per_point * 6;
    hit[i].n = (int)c.n;
    hit[i].pad = 0;
    off += c.n;
  }
  DGS_HIP_TRY(h, hipMemcpyAsync(h->inits.ptr, hin, (size_t)n * sizeof(GicpInit), hipMemcpyHostToDevice, st));
  DGS_HIP_TRY(h, hipMemcpyAsync(h->gitems.ptr, hit, (size_t)n * sizeof(GicpItem), hipMemcpyHostToDevice, st));
  DGS_HIP_TRY(h, hipMemsetAsync(h->done_counter.ptr, 0, 16 * sizeof(int), st));
  DGS_HIP_TRY(h, hipMemsetAsync(h->pair_blocks.ptr, 0, (size_t)n * sizeof(int), st));
  hipLaunchKernelGGL(gicp_init_kernel, dim3((n + 63) / 64), dim3(64), 0, st, h->gpairs.ptr, reinterpret_cast<const GicpInit*>(h->inits.ptr), n);
  return DGS_OK;
}

static GicpPair* gicp_read_back(dgs_handle* h, int n, int* rc) {
  size_t oi, ot, op;
  (void)gicp_pinned_layout(n, &oi, &ot, &op);
  GicpPair* hp = reinterpret_cast<GicpPair*>(reinterpret_cast<char*>(h->pinned) + op);
  *rc = DGS_OK;
  if (hipMemcpyAsync(hp, h->gpairs.ptr, (size_t)n * sizeof(GicpPair), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
    h->err = "reading the GICP optimiser state back failed";
    *rc = DGS_ERR_HIP;
  }
  return hp;
}

// FastGICP::align for every source of a batch against the handle's target; the LM loops of all pairs advance together,
// one (correspond, linearize, solve) launch triple per round, and pairs that finish hand their workgroups to the rest.
int gicp_align_batch(dgs_handle* h, int n, CloudState* const* srcs, const float* guesses16, dgs_result* out) {
  hipStream_t st = h->stream;
  const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::vector<double> x0((size_t)n * 12);
  for (int i = 0; i < n; i++) {
    const float* G = guesses16 ? guesses16 + 16 * i : ident;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) x0[(size_t)i * 12 + r * 4 + c] = (double)G[c * 4 + r];  // Eigen::Isometry3d(guess.cast<double>())
  }
  GicpLaunch L;
  int n_live = 0;
  int rc = gicp_start(h, n, srcs, x0.data(), -1, &L, &n_live);
  if (rc) return rc;

  if (n_live > 0) {
    volatile int* flags = reinterpret_cast<volatile int*>(h->pinned);
    flags[0] = flags[1] = 0;
    if (ensure_poll_events(h) != DGS_OK) return DGS_ERR_HIP;
    hipEvent_t* ev = h->ev_poll;
    const long max_rounds = (long)h->prm.maximum_iterations * (h->prm.gicp_lm_max_iterations + 1) + 4;
    const int chunk = 4;
    long queued = 0;
    int launch_no = 0;
    auto enqueue_chunk = [&](int slot) -> int {
      for (int e = 0; e < chunk; e++) gicp_launch_round(h, L, launch_no++);
      queued += chunk;
      DGS_HIP_TRY(h, hipMemcpyAsync(const_cast<int*>(&flags[slot]), h->done_counter.ptr, sizeof(int), hipMemcpyDeviceToHost, st));
      DGS_HIP_TRY(h, hipEventRecord(ev[slot], st));
      return DGS_OK;
    };
    int cur = 0;
    rc = enqueue_chunk(0);
    while (rc == DGS_OK) {
      const bool more = queued < max_rounds;
      if (more) rc = enqueue_chunk(cur ^ 1);
      if (rc != DGS_OK) break;
      hipError_t e = hipEventSynchronize(ev[cur]);
      if (e != hipSuccess) { h->err = std::string("hipEventSynchronize: ") + hipGetErrorString(e); rc = DGS_ERR_HIP; break; }
      if (flags[cur] >= n_live) break;
      if (!more) break;
      cur ^= 1;
    }
    if (rc != DGS_OK) return rc;
  }
  GicpPair* hp = gicp_read_back(h, n, &rc);
  if (rc) return rc;
  long evals = 0;
  for (int i = 0; i < n; i++) {
    std::memcpy(out[i].final_transformation, hp[i].final_T, sizeof(float) * 16);
    const bool empty = srcs[i]->n <= 0;
    out[i].converged = (!empty && hp[i].s.phase == GP_DONE) ? hp[i].s.converged : 0;
    out[i].iterations = hp[i].s.iteration;
    out[i].evaluations = hp[i].s.evaluations;
    out[i].status = empty ? DGS_ERR_NO_SOURCE : DGS_OK;
    out[i].score = hp[i].s.y0;
    out[i].fitness = NAN;
    evals += hp[i].s.evaluations;
  }
  h->last_evaluations = evals;
  return DGS_OK;
}

int gicp_align(dgs_handle* h, const float* guess16, dgs_result* out) {
  CloudState* one[1] = {h->src};
  return gicp_align_batch(h, 1, one, guess16, out);
}

// device pointer and stride of the batch's final transforms (column-major float[16] per pair) for the fitness kernel
const float* gicp_final_transforms(dgs_handle* h, size_t* stride_bytes) {
  *stride_bytes = sizeof(GicpPair);
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(h->gpairs.ptr) + offsetof(GicpPair, final_T));
}

// Test hook: regularised covariances (9 doubles per point, row-major) of the source (which = 0) or target (1) cloud.
int gicp_covariances(dgs_handle* h, int which, double* host_out9, int64_t n) {
  fill_gconsts(h);
  int rc = which ? ensure_covariance(h, *h->tgt) : ensure_covariance(h, *h->src);
  if (rc) return rc;
  std::vector<double> c6((size_t)n * 6);
  DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));
  DGS_HIP_TRY(h, hipMemcpy(c6.data(), which ? h->tgt->cov.ptr : h->src->cov.ptr, c6.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; i++) {
    const double* c = c6.data() + i * 6;
    double* o = host_out9 + i * 9;
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[1]; o[4] = c[3]; o[5] = c[4]; o[6] = c[2]; o[7] = c[4]; o[8] = c[5];
  }
  return DGS_OK;
}

// Test hook: one linearize (error_only = 0: fresh correspondences at T) or compute_error (1: stored correspondences).
int gicp_probe(dgs_handle* h, const double* T16, int error_only, double* err, double* H36, double* b6) {
  GicpLaunch L;
  int n_live = 0;
  CloudState* one[1] = {h->src};
  int rc = gicp_start(h, 1, one, T16, error_only ? 1 : 0, &L, &n_live);
  if (rc) return rc;
  if (n_live == 0) return DGS_ERR_NO_SOURCE;
  gicp_launch_round(h, L, 0);
  GicpPair* hp = gicp_read_back(h, 1, &rc);
  if (rc) return rc;
  *err = error_only ? hp->s.yi : hp->s.y0;
  if (!error_only) {
    for (int k = 0; k < 36; k++) H36[k] = hp->s.H[k];
    for (int k = 0; k < 6; k++) b6[k] = hp->s.b[k];
  }
  return DGS_OK;
}

}  // namespace dgs
