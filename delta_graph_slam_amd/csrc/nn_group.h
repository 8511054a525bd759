// Cooperative exact nearest-neighbour traversal shared by nn_bvh.hip (fitness / nearestKSearch) and gicp.hip
// (correspondences, k-NN covariances).  See nn_bvh.hip for the data structure.
#pragma once
#include <cfloat>
#include <cmath>

#include "handle.h"

namespace dgs {

constexpr int kLeaf = 8;    // points per leaf = lanes per query group
constexpr int kFan = 8;     // children per node

struct BvhView {
  const float4* sorted;   // Hilbert order, padded to a multiple of 8; w = original index (bit pattern), -1 for padding
  const float4* box_lo;   // [node * 8 + child]
  const float4* box_hi;
  int n;
  int depth;              // internal levels D; leaf slots = 8^D
  int first_leaf;         // heap index of leaf slot 0 = (8^D - 1) / 7
};

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

// 30-bit 3-D HILBERT index of a point (10 bits per axis; Skilling's axes-to-transpose, then bit interleave).  Points are
// sorted by this key: unlike the Z-curve the Hilbert curve has no jumps, so runs of 8 / 64 / 512 ... consecutive points
// stay spatially compact and the AABBs of the implicit 8-ary tree overlap far less (55 -> ~15 node visits per query).
__device__ __forceinline__ uint32_t hilbert30(float x, float y, float z, const float* org, float scale) {
  uint32_t X0 = (uint32_t)fminf(fmaxf((x - org[0]) * scale, 0.f), 1023.f);
  uint32_t X1 = (uint32_t)fminf(fmaxf((y - org[1]) * scale, 0.f), 1023.f);
  uint32_t X2 = (uint32_t)fminf(fmaxf((z - org[2]) * scale, 0.f), 1023.f);
  const uint32_t M = 1u << 9;
#pragma unroll
  for (uint32_t Q = M; Q > 1; Q >>= 1) {  // inverse undo
    const uint32_t P = Q - 1;
    if (X0 & Q) X0 ^= P;  // i = 0: invert (the exchange branch is a no-op on X0 with itself)
    if (X1 & Q) X0 ^= P; else { const uint32_t t = (X0 ^ X1) & P; X0 ^= t; X1 ^= t; }
    if (X2 & Q) X0 ^= P; else { const uint32_t t = (X0 ^ X2) & P; X0 ^= t; X2 ^= t; }
  }
  X1 ^= X0;  // Gray encode
  X2 ^= X1;
  uint32_t t = 0;
#pragma unroll
  for (uint32_t Q = M; Q > 1; Q >>= 1)
    if (X2 & Q) t ^= Q - 1;
  X0 ^= t; X1 ^= t; X2 ^= t;
  return (expand_bits10(X0) << 2) | (expand_bits10(X1) << 1) | expand_bits10(X2);
}

// 16-byte record at a 32-bit BYTE offset from a wave-uniform base: lets the load take the base from SGPRs and the offset from one
// VGPR (global_load ... v_off, s[base]) instead of building a 64-bit address per lane (3 VALU instructions per load in a loop
// that is VALU-issue bound).  The index structures are far below 4 GB.
__device__ __forceinline__ float4 load16_at(const float4* __restrict__ base, unsigned index) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (index << 4));
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

// ---- cross-lane helpers inside an 8-lane group (DPP: no LDS traffic) -------------------------------------------
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned group8_min_u32(unsigned v) {
  v = min(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_u32<0x141>(v));  // row_half_mirror: lane i <-> 7 - i within each 8 lanes
  return v;
}
__device__ __forceinline__ float group8_min_f32(float v) {
  v = fminf(v, __uint_as_float(dpp_u32<0xB1>(__float_as_uint(v))));
  v = fminf(v, __uint_as_float(dpp_u32<0x4E>(__float_as_uint(v))));
  v = fminf(v, __uint_as_float(dpp_u32<0x141>(__float_as_uint(v))));
  return v;
}

// heap index of the first node of a level of the implicit 8-ary tree: (8^l - 1) / 7 = 0b...001001001 below bit 3 l
__device__ __forceinline__ int heap_first(int level) { return (int)(0x09249249u & ((1u << (3 * level)) - 1u)); }

// Pending children of the levels ABOVE the node being visited, one byte per level (bit c: child c still qualifies).  Invariant:
// the bytes of the current level and below are zero, so "anything left?" is pend != 0 and the deepest pending level is the top
// set byte.  32 bits serve trees of up to 5 levels (262,144 points: the last level scans its leaves in place and needs no byte),
// 64 bits the rest.
template <class PT>
__device__ __forceinline__ int pend_top_shift(PT pend);
template <>
__device__ __forceinline__ int pend_top_shift<unsigned>(unsigned pend) { return (31 - __clz((int)pend)) & ~7; }
template <>
__device__ __forceinline__ int pend_top_shift<unsigned long long>(unsigned long long pend) { return (63 - __clzll((long long)pend)) & ~7; }

// Leaves `node` (level sh / 8) for the deepest shallower level that still has pending children; levels with nothing pending
// are skipped without touching their boxes.  Returns false when nothing is pending anywhere (the query is finished).
template <class PT>
__device__ __forceinline__ bool nn_pop(const PT pend, int& node, int& sh) {
  if (pend == 0) return false;
  const int up_sh = pend_top_shift<PT>(pend);
  const int level = sh >> 3, up = up_sh >> 3;
  node = heap_first(up) + ((node - heap_first(level)) >> (3 * (level - up)));
  sh = up_sh;
  return true;
}

// Upper bound on the squared NN distance of q given the previous query q' of the same group and its squared NN distance:
// |q - NN(q')| <= |q' - NN(q')| + |q - q'| (triangle inequality), inflated against rounding.  Pruning only: the result
// of the bounded search is the exact NN (and the same index on ties) as the unbounded one.
__device__ __forceinline__ float nn_warm_bound(float prev_best, bool prev_found, float x, float y, float z, float px, float py, float pz) {
  if (!prev_found) return INFINITY;
  // hardware square roots (1 ulp): the bound is inflated by 1e-4 anyway, and a correctly rounded sqrtf costs ~20 instructions
  const float step = __builtin_amdgcn_sqrtf(sqdist_rn(x, y, z, px, py, pz));
  const float r = (__builtin_amdgcn_sqrtf(prev_best) + step) * 1.0001f + 1e-30f;
  const float b2 = r * r;
  return (b2 == b2) ? b2 : INFINITY;
}

// The 8 groups of a wave search 8 ADJACENT queries at a time (they then walk the same nodes together: no divergence, one
// L1 line serves all) and take the next 8 in the next round.  Bound for this group's query from the 8 results of the wave's
// previous round: lane (group, sub) looks at the previous query of group `sub`, the group keeps the tightest bound.
__device__ __forceinline__ float nn_warm_bound_round(float prev_best, bool prev_found, float x, float y, float z, float px, float py, float pz) {
  const int from = (threadIdx.x & 7) << 3;  // any lane of group `sub` holds that group's previous query
  const float ob = __shfl(prev_best, from), ox = __shfl(px, from), oy = __shfl(py, from), oz = __shfl(pz, from);
  const int of = __shfl(prev_found ? 1 : 0, from);
  return group8_min_f32(nn_warm_bound(ob, of != 0, x, y, z, ox, oy, oz));
}

// Exact 1-NN of (x, y, z) for the 8-lane group this lane belongs to; x, y, z must be equal across the group.
// The traversal is VALU-issue bound (PMC: the fitness kernel keeps the VALUs 100 % busy at a 96 % L2 hit rate), so the loop is
// written for instruction count: 32-bit pending word where the tree allows, no "which levels are stale" masking (see the
// invariant above), unsigned-integer minima on the bit patterns of the non-negative distances, 32-bit load offsets.
// (Measured and rejected: keeping the child-box distances of every level in registers to skip the reload on the way back up;
// visiting a leaf like a node so that every step is uniform -- the re-visits of the last level cost more than the divergence.)
// All 8 lanes return the same (best, best_idx); best_idx == 0x7FFFFFFF: nothing within `bound` (then best is meaningless).
// Lanes of a wave whose group is idle must still call this with `alive` = false (they follow the control flow only).
// WANT_INDEX = false (fitness: only the distance matters): best_idx is just 0 / 0x7FFFFFFF for found / not found, and ties need no
// second reduction.
template <class PT, bool WANT_INDEX>
__device__ __forceinline__ void nn_query_group_t(const BvhView& b, float x, float y, float z, bool alive, float bound, float& best, int& best_idx) {
  const int lane = threadIdx.x & 63;
  const unsigned sub = lane & 7, gshift = lane & ~7, bit = 1u << sub;
  // FLT_MAX instead of +inf: empty slots carry inverted boxes (distance +inf) and must never be entered, and "d <= best" then
  // needs no second test; no finite cloud has squared distances beyond FLT_MAX
  best = fminf(bound, FLT_MAX);
  best_idx = 0x7FFFFFFF;
  int node = 0, sh = 0;
  const int last_sh = 8 * (b.depth - 1);
  PT pend = 0;
  bool fresh = true;
  // a NaN coordinate makes every box distance compare as 0 (max(NaN, 0) = 0), the inverted boxes of empty slots included, and
  // the walk would run off the end of the point array: such a query has no neighbour, it only follows the control flow
  bool done = !alive || !(x == x && y == y && z == z);
#ifdef DGS_NN_STEPS
  int n_nodes = 0, n_leaves = 0, n_wasted = 0;
#endif
  while (__any(!done)) {
    if (!done) {
#ifdef DGS_NN_STEPS
      n_nodes++;
#endif
      const unsigned ofs = (unsigned)node * kFan + sub;
      const float4 lo = load16_at(b.box_lo, ofs), hi = load16_at(b.box_hi, ofs);
      const float d = aabb_sqdist_rn(lo, hi, x, y, z);
      unsigned mask = (unsigned)(__ballot(d <= best) >> gshift) & 0xFFu;
      if (!fresh) {  // back at a node: only the children noted as pending, and the note is consumed
        mask &= (unsigned)(pend >> sh) & 0xFFu;
        pend &= ~((PT)0xFFu << sh);
      }
#ifdef DGS_NN_STEPS
      if (!fresh && mask == 0) n_wasted++;
#endif
      if (sh == last_sh) {
        // children are leaves: scan every qualifying one nearest-first; the boxes stay in registers
        while (mask) {
          const unsigned key = (mask & bit) ? ((__float_as_uint(d) & ~7u) | sub) : 0xFFFFFFFFu;
          const unsigned c = group8_min_u32(key) & 7u;
          mask &= ~(1u << c);
#ifdef DGS_NN_STEPS
          n_leaves++;
#endif
          const unsigned leaf = ((unsigned)node * kFan + 1u + c) - (unsigned)b.first_leaf;
          const float4 p = load16_at(b.sorted, leaf * kLeaf + sub);
          const float dp = sqdist_rn(x, y, z, p.x, p.y, p.z);
          // non-negative floats order like their bit patterns; NaN (padding / non-finite points) sorts above +inf
          const unsigned dbits = __float_as_uint(dp);
          const unsigned dmin = group8_min_u32(dbits);
          const unsigned bbits = __float_as_uint(best);
          if (WANT_INDEX) {
            const unsigned imin = group8_min_u32((dbits == dmin) ? __float_as_uint(p.w) : 0xFFFFFFFFu);
            if (dmin < bbits || (dmin == bbits && (int)imin < best_idx)) {
              best = __uint_as_float(dmin);
              best_idx = (int)imin;
            }
          } else if (dmin <= bbits && dmin < 0x7F800000u) {  // a finite distance within the bound: found
            best = __uint_as_float(dmin);
            best_idx = 0;
          }
          mask &= (unsigned)(__ballot(d <= best) >> gshift) & 0xFFu;
        }
        fresh = false;
        if (!nn_pop<PT>(pend, node, sh)) done = true;
      } else if (mask) {
        const unsigned key = (mask & bit) ? ((__float_as_uint(d) & ~7u) | sub) : 0xFFFFFFFFu;
        const unsigned c = group8_min_u32(key) & 7u;
        pend |= (PT)(mask & ~(1u << c)) << sh;
        node = node * kFan + 1 + (int)c;
        sh += 8;
        fresh = true;
      } else {
        fresh = false;
        if (!nn_pop<PT>(pend, node, sh)) done = true;
      }
    }
  }
#ifdef DGS_NN_STEPS
  best_idx = n_wasted * 1000000 + n_nodes * 1000 + n_leaves;
#endif
}

template <bool WANT_INDEX = true>
__device__ __forceinline__ void nn_query_group(const BvhView& b, float x, float y, float z, bool alive, float bound, float& best, int& best_idx) {
  if (b.depth <= 5)
    nn_query_group_t<unsigned, WANT_INDEX>(b, x, y, z, alive, bound, best, best_idx);
  else
    nn_query_group_t<unsigned long long, WANT_INDEX>(b, x, y, z, alive, bound, best, best_idx);
}

// ---- exact k-NN for the 8-lane group (k <= 32): the k best (distance, index) pairs live in registers, 4 slots per lane, as an
// UNSORTED set of 64-bit keys (distance bits << 32 | index): squared distances are non-negative floats, so unsigned key order is
// the lexicographic (distance, index) order and the set is deterministic.  The set tracks its largest key (the k-th best);
// a better candidate replaces it.  Every cross-lane step is a DPP min / max inside the group: no LDS traffic, no sorted
// insertion (measured: the sorted list with bpermute shuffles spent more time inserting than traversing).
constexpr int kKnnSlots = 4;
constexpr int kKnnMax = kKnnSlots * 8;
struct KnnList {
  unsigned long long key[kKnnSlots];  // slot r * 8 + sub; slots >= k hold 0 (never the largest), empty slots hold (inf, ~slot)
  __device__ __forceinline__ float dist(int r) const { return __uint_as_float((unsigned)(key[r] >> 32)); }
  __device__ __forceinline__ int index(int r) const { return (int)(unsigned)key[r]; }
};
constexpr unsigned long long kKnnInvalid = ~0ull;

__device__ __forceinline__ double group8_sum_f64(double v) {
  {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, false);
    v += __hiloint2double(hi, lo);
  }
  {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4E, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4E, 0xf, 0xf, false);
    v += __hiloint2double(hi, lo);
  }
  {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x141, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x141, 0xf, 0xf, false);
    v += __hiloint2double(hi, lo);
  }
  return v;
}

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long group8_min_u64(unsigned long long v) {
  v = min(v, dpp_u64<0xB1>(v));
  v = min(v, dpp_u64<0x4E>(v));
  v = min(v, dpp_u64<0x141>(v));
  return v;
}
__device__ __forceinline__ unsigned long long group8_max_u64(unsigned long long v) {
  v = max(v, dpp_u64<0xB1>(v));
  v = max(v, dpp_u64<0x4E>(v));
  v = max(v, dpp_u64<0x141>(v));
  return v;
}

// largest key of the set = the current k-th best (group-uniform)
__device__ __forceinline__ unsigned long long knn_largest(const KnnList& L) {
  return group8_max_u64(max(max(L.key[0], L.key[1]), max(L.key[2], L.key[3])));
}

// All 8 lanes of the group end with the same distributed set.  `alive` = false lanes follow the control flow only.
// Same instruction-lean loop as nn_query_group_t (32-bit pending word for trees of up to 5 levels, FLT_MAX instead of +inf as
// the open bound so that one compare rejects the inverted boxes of empty slots).
// `bound`: an upper bound on the squared distance of the k-th neighbour (INFINITY when nothing is known) -- pruning only.
template <class PT>
__device__ __forceinline__ void knn_query_group_t(const BvhView& b, float x, float y, float z, bool alive, int k, float bound, KnnList& L) {
  const int lane = threadIdx.x & 63;
  const unsigned sub = lane & 7, gshift = lane & ~7, bit = 1u << sub;
#pragma unroll
  for (int r = 0; r < kKnnSlots; r++) {
    const int slot = r * 8 + (int)sub;
    L.key[r] = (slot < k) ? (((unsigned long long)__float_as_uint(INFINITY) << 32) | (unsigned)(0x7FFFFFFF - slot)) : 0ull;
  }
  unsigned long long worst = knn_largest(L);
  const float open_td = fminf(bound, FLT_MAX);
  float td = open_td;  // min(distance part of `worst`, bound, FLT_MAX): boxes farther than this cannot hold a better point
  int node = 0, sh = 0;
  const int last_sh = 8 * (b.depth - 1);
  PT pend = 0;
  bool fresh = true;
  bool done = !alive || !(x == x && y == y && z == z);   // NaN query: see nn_query_group_t
  while (__any(!done)) {
    if (!done) {
      const unsigned ofs = (unsigned)node * kFan + sub;
      const float4 lo = load16_at(b.box_lo, ofs), hi = load16_at(b.box_hi, ofs);
      const float d = aabb_sqdist_rn(lo, hi, x, y, z);
      unsigned mask = (unsigned)(__ballot(d <= td) >> gshift) & 0xFFu;
      if (!fresh) {
        mask &= (unsigned)(pend >> sh) & 0xFFu;
        pend &= ~((PT)0xFFu << sh);
      }
      if (sh == last_sh) {
        while (mask) {
          const unsigned bkey = (mask & bit) ? ((__float_as_uint(d) & ~7u) | sub) : 0xFFFFFFFFu;
          const unsigned c = group8_min_u32(bkey) & 7u;
          mask &= ~(1u << c);
          const unsigned leaf = ((unsigned)node * kFan + 1u + c) - (unsigned)b.first_leaf;
          const float4 p = load16_at(b.sorted, leaf * kLeaf + sub);
          const float dp = sqdist_rn(x, y, z, p.x, p.y, p.z);
          // this lane's point as a candidate key; padding / non-finite points never qualify
          unsigned long long cand = (dp < INFINITY) ? (((unsigned long long)__float_as_uint(dp) << 32) | __float_as_uint(p.w)) : kKnnInvalid;
          if (!(cand < worst)) cand = kKnnInvalid;
          // take the leaf's candidates smallest first: each one that gets in lowers the bar for the rest
          while ((unsigned)(__ballot(cand != kKnnInvalid) >> gshift) & 0xFFu) {
            const unsigned long long best = group8_min_u64(cand);
#pragma unroll
            for (int r = 0; r < kKnnSlots; r++)
              if (L.key[r] == worst) L.key[r] = best;  // keys are unique: exactly one slot of one lane holds the largest
            worst = knn_largest(L);
            if (cand == best || !(cand < worst)) cand = kKnnInvalid;
          }
          td = fminf(__uint_as_float((unsigned)(worst >> 32)), open_td);
          mask &= (unsigned)(__ballot(d <= td) >> gshift) & 0xFFu;
        }
        fresh = false;
        if (!nn_pop<PT>(pend, node, sh)) done = true;
      } else if (mask) {
        const unsigned bkey = (mask & bit) ? ((__float_as_uint(d) & ~7u) | sub) : 0xFFFFFFFFu;
        const unsigned c = group8_min_u32(bkey) & 7u;
        pend |= (PT)(mask & ~(1u << c)) << sh;
        node = node * kFan + 1 + (int)c;
        sh += 8;
        fresh = true;
      } else {
        fresh = false;
        if (!nn_pop<PT>(pend, node, sh)) done = true;
      }
    }
  }
}

__device__ __forceinline__ void knn_query_group(const BvhView& b, float x, float y, float z, bool alive, int k, KnnList& L, float bound = INFINITY) {
  if (b.depth <= 5)
    knn_query_group_t<unsigned>(b, x, y, z, alive, k, bound, L);
  else
    knn_query_group_t<unsigned long long>(b, x, y, z, alive, k, bound, L);
}

// host: device view of a built index (nn_bvh.hip)
BvhView make_bvh_view(const Bvh& b);

}  // namespace dgs
