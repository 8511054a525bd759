// K7 (fitness pass): exact unbounded 1-NN distances through a sparse 64-ary voxel hierarchy, ONE LANE PER QUERY.
//
// Replaces, for registration->getFitnessScore(max_range) (include/hdl_graph_slam/loop_detector.hpp:148,
// apps/scan_matching_odometry_nodelet.cpp:318; in-tree twin src/hdl_graph_slam/information_matrix_calculator.cpp:77-108), the walk
// of the 8-ary tree in nn_bvh.hip, which serves 8 queries per wave and spends 7/8 of every SIMD instruction repeating the same
// traversal logic (round 1: 15.9 k VALU instructions per wave of 64 lanes = 8 queries at a time, 0.5 % of the HBM roofline).
//
// Structure (built next to the tree at setInputTarget, on the same stream; everything but the points is a few MB and L2-resident):
//   * fine cells of size c, 4 x 4 x 4 of them form a coarse cell, 4 x 4 x 4 coarse cells an L2 cell; the target points are
//     sorted by (L2 cell, coarse sub-cell, fine sub-cell), so every cell of every level is a contiguous range of ONE array;
//   * per coarse cell a 64-bit occupancy mask of its fine cells + the rank of its first occupied fine cell; per L2 cell a 64-bit
//     mask of its coarse cells; a compact `cstart` array with one entry per OCCUPIED fine cell (its first point; the next entry
//     is its end).  No table has an entry per fine cell, so the boxes are implicit (computed from the indices) and c can be small;
//   * c is chosen ON THE DEVICE from the data: ~4 x the median distance between consecutive points of the Hilbert order (a
//     robust local spacing: dense regions dominate the median), enlarged until the L2 table fits its budget.
// Query, pass 1 (one lane per query): the 3 x 3 x 3 fine block; proven exact when best <= (distance to the block's border)^2
// (~80 % of the queries of aligned scans).  Open queries are queued (compaction: later passes run on full waves) and go to the
// 8-lane tree walk of nn_bvh.hip bounded by the best distance found; an optional middle pass (DGS_NN_GRID=2) searches the
// 3 x 3 x 3 coarse block through the masks first.  Distances land in a per-query array and are summed in a fixed order.
// Squared distances are FLANN's float sequence (sqdist_rn); box bounds are shrunk by 0.2 % against rounding: pruning only, the
// minimum over the target is exact.
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cmath>
#include <cstring>

#include "handle.h"
#include "nn_group.h"

namespace dgs {

constexpr unsigned kGridSentinel = 1u << 25;   // key of non-finite points (sorted last, never in a cell)
constexpr int kGridKeyBits = 26;
constexpr float kGridShrink = 0.998f;          // lower bounds are multiplied by this before they prune

// ---- build ---------------------------------------------------------------------------------------------------
// histogram of float exponents of the distance between consecutive points of the Hilbert order
__global__ __launch_bounds__(kBlock) void grid_spacing_kernel(const float4* __restrict__ hsorted, int n, unsigned* __restrict__ hist) {
  __shared__ unsigned sh[256];
  sh[threadIdx.x] = 0;
  __syncthreads();
  for (int i = blockIdx.x * kBlock + threadIdx.x; i + 1 < n; i += gridDim.x * kBlock) {
    const float4 a = hsorted[i], b = hsorted[i + 1];
    const float d = sqrtf(sqdist_rn(a.x, a.y, a.z, b.x, b.y, b.z));
    if (d > 0.f && d < INFINITY) atomicAdd(&sh[(__float_as_uint(d) >> 23) & 0xFF], 1u);
  }
  __syncthreads();
  if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

__global__ void grid_params_kernel(const float* __restrict__ mm6, const unsigned* __restrict__ hist, int n, float spacing_factor, NnGridParams* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  NnGridParams g;
  float lo[3] = {mm6[0], mm6[1], mm6[2]}, hi[3] = {mm6[3], mm6[4], mm6[5]};
  if (!(lo[0] <= hi[0])) { lo[0] = lo[1] = lo[2] = 0.f; hi[0] = hi[1] = hi[2] = 0.f; }   // no finite point
  // median of the consecutive-point distances, to a power of two
  unsigned long long total = 0;
  for (int e = 0; e < 256; e++) total += hist[e];
  unsigned long long acc = 0;
  int med = -1;
  for (int e = 0; e < 256 && med < 0; e++) {
    acc += hist[e];
    if (total && 2 * acc >= total) med = e;
  }
  const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), fmaxf(hi[2] - lo[2], 1e-3f));
  float c = (med >= 0) ? spacing_factor * __uint_as_float((unsigned)med << 23) : ext / 64.f;   // 2^e <= median < 2^(e+1)
  c = fmaxf(c, ext / 4096.f);
  for (int it = 0; it < 96; it++) {
    long long cells = 1;
    for (int a = 0; a < 3; a++) cells *= (long long)((hi[a] - lo[a]) / (16.f * c)) + 2;
    if (cells <= (long long)kNnGridL2Cells) break;
    c *= 1.125f;
  }
  g.c = c;
  g.inv_c = 1.0f / c;
  for (int a = 0; a < 3; a++) {
    g.org[a] = lo[a] - 0.5f * c;
    g.n2[a] = (int)((hi[a] - g.org[a]) * g.inv_c) / 16 + 1;
  }
  g.n = n;
  *out = g;
}

// fine cell coordinate of a float coordinate (the one formula both the build and the queries use)
__device__ __forceinline__ float grid_u(float p, float org, float inv_c) { return (p - org) * inv_c; }

__device__ __forceinline__ unsigned grid_key(int fx, int fy, int fz, const NnGridParams& g) {
  const unsigned l2 = (unsigned)(fx >> 4) + (unsigned)g.n2[0] * ((unsigned)(fy >> 4) + (unsigned)g.n2[1] * (unsigned)(fz >> 4));
  const unsigned cs = (unsigned)(((fx >> 2) & 3) | (((fy >> 2) & 3) << 2) | (((fz >> 2) & 3) << 4));
  const unsigned fs = (unsigned)((fx & 3) | ((fy & 3) << 2) | ((fz & 3) << 4));
  return (l2 << 12) | (cs << 6) | fs;
}

__global__ __launch_bounds__(kBlock) void grid_key_kernel(const float4* __restrict__ pts, int n, const NnGridParams* __restrict__ gp,
                                                          uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const NnGridParams g = *gp;
  const float4 p = pts[i];
  unsigned key = kGridSentinel;
  if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    const int fx = (int)floorf(grid_u(p.x, g.org[0], g.inv_c)), fy = (int)floorf(grid_u(p.y, g.org[1], g.inv_c)),
              fz = (int)floorf(grid_u(p.z, g.org[2], g.inv_c));
    if ((unsigned)fx < (unsigned)(16 * g.n2[0]) && (unsigned)fy < (unsigned)(16 * g.n2[1]) && (unsigned)fz < (unsigned)(16 * g.n2[2]))
      key = grid_key(fx, fy, fz, g);
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

// one lane per run of equal keys (= occupied fine cell, in key order): masks and bases of the two upper levels.
// clear != 0: undo what a previous build wrote through its own run keys (a dense memset would cost more than the build).
__global__ __launch_bounds__(kBlock) void grid_cells_kernel(const uint32_t* __restrict__ run_keys, const int* __restrict__ num_runs,
                                                            NnCoarse* __restrict__ coarse, unsigned long long* __restrict__ occ2, int clear) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= *num_runs) return;
  const unsigned k = run_keys[j];
  if (k >= kGridSentinel) return;
  if (clear) {
    coarse[k >> 6].mask = 0ull;
    occ2[k >> 12] = 0ull;
    return;
  }
  atomicOr(&coarse[k >> 6].mask, 1ull << (k & 63));
  atomicOr(&occ2[k >> 12], 1ull << ((k >> 6) & 63));
  if (j == 0 || (run_keys[j - 1] >> 6) != (k >> 6)) coarse[k >> 6].base = j;   // rank of the cell's first occupied fine cell
}

__global__ __launch_bounds__(kBlock) void grid_gather_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ order, int n,
                                                             float4* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = pts[order[i]];
}

int nn_grid_build(dgs_handle* h, NnGrid& G, const Bvh& bvh, const float4* pts, int64_t n64, hipStream_t stream) {
  hipStream_t st = stream ? stream : h->stream;
  const int n = (int)n64;
  G.valid = false;
  if (n == 0 || !bvh.valid) return DGS_OK;
  const bool fresh = G.coarse.ptr == nullptr;
  DGS_HIP_TRY(h, G.coarse.reserve((size_t)kNnGridL2Cells * 64 + 64));
  DGS_HIP_TRY(h, G.occ2.reserve((size_t)kNnGridL2Cells + 64));
  DGS_HIP_TRY(h, G.params.reserve(1));
  DGS_HIP_TRY(h, G.hist.reserve(256));
  DGS_HIP_TRY(h, G.scalars.reserve(8));
  if (fresh) {
    DGS_HIP_TRY(h, hipMemsetAsync(G.coarse.ptr, 0, G.coarse.cap * sizeof(NnCoarse), st));
    DGS_HIP_TRY(h, hipMemsetAsync(G.occ2.ptr, 0, G.occ2.cap * sizeof(unsigned long long), st));
    DGS_HIP_TRY(h, hipMemsetAsync(G.scalars.ptr, 0, 8 * sizeof(int), st));
    G.n_prev = 0;
  }
  const int nb = (n + kBlock - 1) / kBlock;
  // the previous build's cells are cleared through its own run keys (scalars[0] still holds its number of runs)
  if (G.n_prev > 0)
    hipLaunchKernelGGL(grid_cells_kernel, dim3((unsigned)((G.n_prev + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, G.run_keys.ptr, G.scalars.ptr, G.coarse.ptr,
                       G.occ2.ptr, 1);
  DGS_HIP_TRY(h, G.sorted.reserve(n));
  DGS_HIP_TRY(h, G.keys.reserve(n));
  DGS_HIP_TRY(h, G.keys_alt.reserve(n));
  DGS_HIP_TRY(h, G.vals.reserve(n));
  DGS_HIP_TRY(h, G.vals_alt.reserve(n));
  DGS_HIP_TRY(h, G.run_keys.reserve(n + 1));
  DGS_HIP_TRY(h, G.run_counts.reserve(n + 2));
  DGS_HIP_TRY(h, G.cstart.reserve(n + 2));
  size_t t1 = 0, t2 = 0, t3 = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, G.keys.ptr, G.keys_alt.ptr, G.vals.ptr, G.vals_alt.ptr, n, 0, kGridKeyBits, st);
  (void)hipcub::DeviceRunLengthEncode::Encode(nullptr, t2, G.keys_alt.ptr, G.run_keys.ptr, G.run_counts.ptr, G.scalars.ptr, n, st);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t3, G.run_counts.ptr, G.cstart.ptr, n + 1, st);
  DGS_HIP_TRY(h, h->cub_temp.reserve(std::max(t1, std::max(t2, t3)) + 256));
  float* d_mm = nullptr;
  int rc = cloud_minmax_device(h, pts, n, &d_mm, st);
  if (rc) return rc;
  DGS_HIP_TRY(h, hipMemsetAsync(G.hist.ptr, 0, 256 * sizeof(unsigned), st));
  DGS_HIP_TRY(h, hipMemsetAsync(G.run_counts.ptr, 0, (size_t)(n + 2) * sizeof(int), st));   // the scan runs over n + 1 entries: cstart[runs] = n
  hipLaunchKernelGGL(grid_spacing_kernel, dim3(std::min(nb, 256)), dim3(kBlock), 0, st, bvh.sorted.ptr, n, G.hist.ptr);
  hipLaunchKernelGGL(grid_params_kernel, dim3(1), dim3(1), 0, st, d_mm, G.hist.ptr, n, h->grid_spacing_factor, G.params.ptr);
  hipLaunchKernelGGL(grid_key_kernel, dim3(nb), dim3(kBlock), 0, st, pts, n, G.params.ptr, G.keys.ptr, G.vals.ptr);
  size_t tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRadixSort::SortPairs(h->cub_temp.ptr, tb, G.keys.ptr, G.keys_alt.ptr, G.vals.ptr, G.vals_alt.ptr, n, 0, kGridKeyBits, st));
  hipLaunchKernelGGL(grid_gather_kernel, dim3(nb), dim3(kBlock), 0, st, pts, G.vals_alt.ptr, n, G.sorted.ptr);
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceRunLengthEncode::Encode(h->cub_temp.ptr, tb, G.keys_alt.ptr, G.run_keys.ptr, G.run_counts.ptr, G.scalars.ptr, n, st));
  tb = h->cub_temp.cap;
  DGS_HIP_TRY(h, hipcub::DeviceScan::ExclusiveSum(h->cub_temp.ptr, tb, G.run_counts.ptr, G.cstart.ptr, n + 1, st));
  hipLaunchKernelGGL(grid_cells_kernel, dim3(nb), dim3(kBlock), 0, st, G.run_keys.ptr, G.scalars.ptr, G.coarse.ptr, G.occ2.ptr, 0);
  DGS_HIP_TRY(h, hipGetLastError());
  G.n_prev = n;
  G.n = n;
  G.valid = true;
  return DGS_OK;
}

// ---- query ---------------------------------------------------------------------------------------------------
struct NnGridView {
  const float4* sorted;
  const NnCoarse* coarse;
  const unsigned long long* occ2;
  const int* cstart;
  const NnGridParams* params;
};

#ifdef DGS_GRID_STATS
#define GRID_STAT(...) __VA_ARGS__
#else
#define GRID_STAT(...)
#endif

struct GridQuery {
  float x, y, z;        // the query
  float u[3];           // ... in fine-cell units
  float best;           // best squared distance so far
  float best_u;         // best in squared fine-cell units, inflated: a cell whose squared box distance exceeds it cannot win
  float inv_c2;         // 1 / (c^2 * shrink^2)
  GRID_STAT(int st_pts, st_cells;)
};

__device__ __forceinline__ void grid_take(GridQuery& q, float d) {
  if (d < q.best) {
    q.best = d;
    q.best_u = d * q.inv_c2;
  }
}

// scans sorted[b .. e), four loads in flight per step
__device__ __forceinline__ void grid_scan(const float4* __restrict__ sorted, int b, int e, GridQuery& q) {
  GRID_STAT(q.st_pts += e - b; q.st_cells += 1;)
  float best = q.best;
  for (int j = b; j < e; j += 4) {
    float4 p[4];
#pragma unroll
    for (int t = 0; t < 4; t++) p[t] = load16_at(sorted, (unsigned)min(j + t, e - 1));   // the tail repeats the last point
#pragma unroll
    for (int t = 0; t < 4; t++) best = fminf(best, sqdist_rn(q.x, q.y, q.z, p[t].x, p[t].y, p[t].z));
  }
  grid_take(q, best);
}

// squared distance (fine-cell units) from the query to the box [k * s, (k + 1) * s) per axis, s = 1 / 4 / 16 fine cells
__device__ __forceinline__ float grid_box_u(const GridQuery& q, int kx, int ky, int kz, float s) {
  const float lx = (float)kx * s, ly = (float)ky * s, lz = (float)kz * s;
  const float dx = fmaxf(fmaxf(lx - q.u[0], q.u[0] - (lx + s)), 0.f);
  const float dy = fmaxf(fmaxf(ly - q.u[1], q.u[1] - (ly + s)), 0.f);
  const float dz = fmaxf(fmaxf(lz - q.u[2], q.u[2] - (lz + s)), 0.f);
  return dx * dx + dy * dy + dz * dz;
}

// bits (x | y << 2 | z << 4) of a 4 x 4 x 4 block with lo[a] <= coordinate a <= hi[a]; empty when some lo > hi
__device__ __forceinline__ unsigned long long grid_box_mask(int lx, int hx, int ly, int hy, int lz, int hz) {
  if (lx > hx || ly > hy || lz > hz) return 0ull;
  const unsigned rx = ((2u << hx) - (1u << lx)) & 0xFu, ry = ((2u << hy) - (1u << ly)) & 0xFu, rz = ((2u << hz) - (1u << lz)) & 0xFu;
  const unsigned long long X = (unsigned long long)rx * 0x1111111111111111ull;
  const unsigned yp = ((ry & 1u) * 0xFu) | (((ry >> 1) & 1u) * 0xF0u) | (((ry >> 2) & 1u) * 0xF00u) | (((ry >> 3) & 1u) * 0xF000u);
  const unsigned long long Y = (unsigned long long)yp * 0x0001000100010001ull;
  const unsigned long long Z = ((rz & 1u) ? 0xFFFFull : 0ull) | (((rz >> 1) & 1u) ? 0xFFFF0000ull : 0ull) | (((rz >> 2) & 1u) ? 0xFFFF00000000ull : 0ull) |
                               (((rz >> 3) & 1u) ? 0xFFFF000000000000ull : 0ull);
  return X & Y & Z;
}

// the occupied fine cells `m` (subset of E.mask) of coarse cell (Kx, Ky, Kz): prune by the implicit box, scan what survives
__device__ __forceinline__ void grid_visit_fine(const NnGridView& v, const NnCoarse E, unsigned long long m, int Kx, int Ky, int Kz, GridQuery& q) {
  while (m) {
    const int bit = __ffsll((long long)m) - 1;
    m &= m - 1;
    const int fx = 4 * Kx + (bit & 3), fy = 4 * Ky + ((bit >> 2) & 3), fz = 4 * Kz + (bit >> 4);
    if (!(grid_box_u(q, fx, fy, fz, 1.f) <= q.best_u)) continue;
    const int idx = E.base + __popcll(E.mask & ((1ull << bit) - 1ull));
    grid_scan(v.sorted, v.cstart[idx], v.cstart[idx + 1], q);
  }
}

__device__ __forceinline__ NnCoarse grid_coarse(const NnGridView& v, const NnGridParams& g, int Kx, int Ky, int Kz) {
  NnCoarse E;
  E.mask = 0ull;
  E.base = 0;
  E.pad = 0;
  if ((unsigned)Kx < (unsigned)(4 * g.n2[0]) && (unsigned)Ky < (unsigned)(4 * g.n2[1]) && (unsigned)Kz < (unsigned)(4 * g.n2[2])) {
    const unsigned l2 = (unsigned)(Kx >> 2) + (unsigned)g.n2[0] * ((unsigned)(Ky >> 2) + (unsigned)g.n2[1] * (unsigned)(Kz >> 2));
    const unsigned cs = (unsigned)((Kx & 3) | ((Ky & 3) << 2) | ((Kz & 3) << 4));
    const uint4 raw = *reinterpret_cast<const uint4*>(v.coarse + ((l2 << 6) | cs));   // one 16-B load
    E.mask = ((unsigned long long)raw.y << 32) | raw.x;
    E.base = (int)raw.z;
  }
  return E;
}

// squared radius (true units) around the query inside which the 3 x 3 x 3 block of cells of `s` fine cells is complete
__device__ __forceinline__ float grid_proven_sq(const GridQuery& q, const int* k, float s, float c) {
  float m = s;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float lo = (float)k[a] * s;
    m = fminf(m, fminf(q.u[a] - lo, lo + s - q.u[a]));
  }
  const float r = c * (s + fmaxf(m, 0.f)) * kGridShrink;
  return r * r;
}

// ---- level 0: the own fine cell, then the other cells of the 3 x 3 x 3 fine block (they live in <= 2 x 2 x 2 coarse cells).
// Returns true when the result is proven exact.
__device__ __forceinline__ void grid_setup(GridQuery& q, const NnGridParams& g, float x, float y, float z, float best, int* f, int* K) {
  q.x = x; q.y = y; q.z = z;
  q.inv_c2 = (g.inv_c * g.inv_c) / (kGridShrink * kGridShrink);
  q.best = best;
  q.best_u = best * q.inv_c2;
  GRID_STAT(q.st_pts = 0; q.st_cells = 0;)
  q.u[0] = grid_u(x, g.org[0], g.inv_c); q.u[1] = grid_u(y, g.org[1], g.inv_c); q.u[2] = grid_u(z, g.org[2], g.inv_c);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    f[a] = (int)fminf(fmaxf(floorf(q.u[a]), -1024.f), (float)(16 * g.n2[a] + 1024));   // far outside the table: every range test rejects it
    K[a] = f[a] >> 2;    // arithmetic shift: cells below the table stay below it
  }
}

// Level 0, ONE LANE PER QUERY.  Phase A lists, per lane, the point ranges of the 3 x 3 x 3 fine block: per (y, z) row the
// x-neighbours inside one coarse cell are consecutive bits of its mask, hence ONE contiguous range of the sorted points; a row
// touches at most two coarse cells.  Phase B scans the concatenated ranges in ONE loop, so that a wave's trip count is the largest
// per-lane point total and not the sum of the largest cell of every step (nested per-cell loops cost 6 k instructions per wave).
// Measured alternatives: nested per-cell loops with box pruning (0.33 ms for this pass alone), larger cells (the
// scan is bound by the L1 addresser: 64 scattered 16-byte loads per instruction), and an 8-lanes-per-query cooperative search
// through the masks with nearest-first proposals (coalesced scans, but 2-4 x slower overall: the per-round control is heavy).
constexpr int kGridSegs = 18;   // 9 rows x <= 2 coarse cells
__device__ __forceinline__ bool grid_level0(const NnGridView& v, const NnGridParams& g, GridQuery& q, const int* f, const int* K, int2* __restrict__ segs) {
  int nseg = 0;
  const int Kx0 = (f[0] - 1) >> 2, Kx1 = (f[0] + 1) >> 2;
#pragma unroll
  for (int dz = -1; dz <= 1; dz++)
#pragma unroll
    for (int dy = -1; dy <= 1; dy++) {
      const int y = f[1] + dy, z = f[2] + dz;
      const int rowshift = ((y & 3) << 2) | ((z & 3) << 4);
#pragma unroll
      for (int sgm = 0; sgm < 2; sgm++) {
        const int Kx = sgm ? Kx1 : Kx0;
        if (sgm && Kx1 == Kx0) continue;
        const NnCoarse E = grid_coarse(v, g, Kx, y >> 2, z >> 2);
        const int xlo = max(f[0] - 1 - 4 * Kx, 0), xhi = min(f[0] + 1 - 4 * Kx, 3);
        const unsigned sel = (unsigned)(E.mask >> rowshift) & (((2u << xhi) - 1u) ^ ((1u << xlo) - 1u)) & 0xFu;
        if (sel) {
          const int first = __ffs((int)sel) - 1;
          const int idx = E.base + __popcll(E.mask & ((1ull << (rowshift + first)) - 1ull));
          segs[nseg * kBlock] = make_int2(v.cstart[idx], v.cstart[idx + __popc(sel)]);
          nseg++;
        }
      }
    }
  // phase B
  float best = q.best;
  int e = 0;
  int2 cur = nseg ? segs[0] : make_int2(0, 0);
  GRID_STAT(for (int t = 0; t < nseg; t++) { q.st_pts += segs[t * kBlock].y - segs[t * kBlock].x; q.st_cells += 1; })
  while (e < nseg) {
    float4 p[4];
#pragma unroll
    for (int t = 0; t < 4; t++) p[t] = load16_at(v.sorted, (unsigned)min(cur.x + t, cur.y - 1));   // the tail repeats the last point
#pragma unroll
    for (int t = 0; t < 4; t++) best = fminf(best, sqdist_rn(q.x, q.y, q.z, p[t].x, p[t].y, p[t].z));
    cur.x += 4;
    if (cur.x >= cur.y) {
      e++;
      if (e < nseg) cur = segs[e * kBlock];
    }
  }
  grid_take(q, best);
  return q.best <= grid_proven_sq(q, f, 1.f, g.c);
}

// ---- level 1 (optional pass, DGS_NN_GRID=2): the rest of the 3 x 3 x 3 coarse block, one lane per query, nested loops
__device__ __forceinline__ bool grid_level1(const NnGridView& v, const NnGridParams& g, GridQuery& q, const int* f, const int* K) {
  for (int dz = -1; dz <= 1; dz++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        const int Kx = K[0] + dx, Ky = K[1] + dy, Kz = K[2] + dz;
        if (!(grid_box_u(q, Kx, Ky, Kz, 4.f) <= q.best_u)) continue;
        const NnCoarse E = grid_coarse(v, g, Kx, Ky, Kz);
        const unsigned long long done = grid_box_mask(max(f[0] - 1 - 4 * Kx, 0), min(f[0] + 1 - 4 * Kx, 3), max(f[1] - 1 - 4 * Ky, 0), min(f[1] + 1 - 4 * Ky, 3),
                                                      max(f[2] - 1 - 4 * Kz, 0), min(f[2] + 1 - 4 * Kz, 3));
        grid_visit_fine(v, E, E.mask & ~done, Kx, Ky, Kz, q);
      }
  return q.best <= grid_proven_sq(q, K, 4.f, g.c);
}

// Queries of a batch are numbered pair * max_n + i; their squared NN distances go to dist[] (so that the fitness sums can be
// formed in a fixed order whatever order the open queries were queued in), open ones to a queue {query number, best so far}.
struct NnQueue {
  unsigned* items;
  float* best;
  int* count;
};

__device__ __forceinline__ void queue_push(const NnQueue& Q, bool open, unsigned item, float best) {
  const unsigned long long m = __ballot(open);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  const int first = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == first) base = atomicAdd(Q.count, __popcll(m));
  base = __shfl(base, first, 64);
  if (open) {
    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
    Q.items[slot] = item;
    Q.best[slot] = best;
  }
}

struct NnBatch {
  const float4* const* src_ptrs;
  const int* sizes;
  const float* Tbase;     // column-major 4 x 4 per pair, T_stride bytes apart
  size_t T_stride;
  int max_n;
};

// pcl::transformPointCloud of point i of pair `pair`: ((m0 x + m1 y) + m2 z) + m3 in float, every step rounded
__device__ __forceinline__ void batch_point(const NnBatch& B, int pair, int i, float& x, float& y, float& z) {
  const float* T = reinterpret_cast<const float*>(reinterpret_cast<const char*>(B.Tbase) + (size_t)pair * B.T_stride);
  const float4 p = B.src_ptrs[pair][i];
  x = affine_row_rn(T[0], T[4], T[8], T[12], p.x, p.y, p.z);
  y = affine_row_rn(T[1], T[5], T[9], T[13], p.x, p.y, p.z);
  z = affine_row_rn(T[2], T[6], T[10], T[14], p.x, p.y, p.z);
}

// pass 1: every query, one lane each, level 0 only.  ~80 % of aligned-scan queries end here; the rest is queued, so that the
// later (longer, more divergent) searches run on full waves of open queries instead of stalling 60 finished lanes behind 4.
__global__ __launch_bounds__(kBlock) void nn_grid_level0_kernel(const NnGridView v, const NnBatch B, float* __restrict__ dist, const NnQueue Q) {
  const int pair = blockIdx.y;
  const int n = B.sizes[pair];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (blockIdx.x * kBlock >= n) return;
  const NnGridParams g = *v.params;
  const bool alive = i < n;
  bool open = false;
  float best = INFINITY;
  __shared__ int2 segs[kGridSegs * kBlock];   // [segment][thread]: 36 KB per workgroup
  if (alive) {
    float x, y, z;
    batch_point(B, pair, i, x, y, z);
    GridQuery q;
    int f[3], K[3];
    grid_setup(q, g, x, y, z, INFINITY, f, K);
    // a query with a NaN coordinate has no neighbour (its box tests would all read 0): answered +inf here, never queued
    open = (x == x && y == y && z == z) && !grid_level0(v, g, q, f, K, segs + threadIdx.x);
    best = q.best;
    GRID_STAT(best = (float)(0 + 4 * min(q.st_cells, 63) + 256 * min(q.st_pts, 65535));)
    if (!open) dist[(size_t)pair * B.max_n + i] = best;
  }
  GRID_STAT(best = INFINITY;)
  queue_push(Q, open, (unsigned)(pair * B.max_n + i), best);
}

// pass 2 (optional, DGS_NN_GRID=2): the queued queries, one lane each, level 1 (3 x 3 x 3 coarse cells through the occupancy masks)
__global__ __launch_bounds__(kBlock) void nn_grid_level1_kernel(const NnGridView v, const NnBatch B, float* __restrict__ dist, const NnQueue Qin, const NnQueue Qout) {
  const int count = *Qin.count;
  const NnGridParams g = *v.params;
  for (int j0 = blockIdx.x * kBlock; j0 < count; j0 += gridDim.x * kBlock) {   // the queue's length is only known on the device
    const int j = j0 + threadIdx.x;
    bool open = false;
    float best = INFINITY;
    unsigned item = 0;
    if (j < count) {
      item = Qin.items[j];
      const int pair = (int)(item / (unsigned)B.max_n), i = (int)(item % (unsigned)B.max_n);
      float x, y, z;
      batch_point(B, pair, i, x, y, z);
      GridQuery q;
      int f[3], K[3];
      grid_setup(q, g, x, y, z, Qin.best[j], f, K);
      open = !grid_level1(v, g, q, f, K);
      best = q.best;
      GRID_STAT(best = (float)(1 + 4 * min(q.st_cells, 63) + 256 * min(q.st_pts, 65535));)
      if (!open) dist[item] = best;
    }
    GRID_STAT(best = INFINITY;)
    queue_push(Qout, open, item, best);
  }
}

// pass 3: what is still open (queries more than ~4 fine cells from every target point, or outside the table): the 8-lane tree
// walk of nn_bvh.hip, bounded by the best distance the grid passes found
__global__ __launch_bounds__(kBlock) void nn_tree_queue_kernel(const BvhView b, const NnBatch B, float* __restrict__ dist, const NnQueue Qin) {
  const int count = *Qin.count;
  for (int j0 = blockIdx.x * (kBlock / 8); j0 < count; j0 += gridDim.x * (kBlock / 8)) {
    const int j = j0 + (threadIdx.x >> 3);
    const bool alive = j < count;
    float x = 0.f, y = 0.f, z = 0.f, bound = INFINITY;
    unsigned item = 0;
    if (alive) {
      item = Qin.items[j];
      batch_point(B, (int)(item / (unsigned)B.max_n), (int)(item % (unsigned)B.max_n), x, y, z);
      bound = Qin.best[j];
    }
    float nb;
    int ni;
    nn_query_group<false>(b, x, y, z, alive, bound, nb, ni);
    if (alive && (threadIdx.x & 7) == 0) {
      dist[item] = (ni != 0x7FFFFFFF) ? nb : bound;
      GRID_STAT(dist[item] = 2.f;)
    }
  }
}

// fitness / inlier accumulation in a fixed order: per block one row {sum d2 (d2 <= max_range), count, inliers}
__global__ __launch_bounds__(kBlock) void nn_fitness_sum_kernel(const float* __restrict__ dist, const int* __restrict__ sizes, int max_n, float max_range,
                                                                float inlier_sq, double* __restrict__ partial, int blocks_per_pair) {
  const int pair = blockIdx.y;
  const int n = sizes[pair];
  double s = 0.0, c = 0.0, inl = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += blocks_per_pair * kBlock) {
    const float best = dist[(size_t)pair * max_n + i];
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
    const double val = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    partial[((size_t)pair * blocks_per_pair + blockIdx.x) * 4 + threadIdx.x] = val;
  }
}

static NnGridView make_grid_view(const NnGrid& G) {
  NnGridView v;
  v.sorted = G.sorted.ptr;
  v.coarse = G.coarse.ptr;
  v.occ2 = G.occ2.ptr;
  v.cstart = G.cstart.ptr;
  v.params = G.params.ptr;
  return v;
}

// squared NN distances of all queries of a batch -> G.dist[pair * max_n + i] (three passes, see above)
int nn_grid_distances(dgs_handle* h, NnGrid& G, const Bvh& index, int n_pairs, const float4* const* d_src_ptrs, const int* d_sizes, int max_n, const float* d_T,
                      size_t T_stride_bytes) {
  hipStream_t st = h->stream;
  const size_t total = (size_t)n_pairs * max_n;
  if (total == 0) return DGS_OK;
  if (total > 0x7FFFFFFFull) { h->err = "too many queries for one fitness batch"; return DGS_ERR_UNSUPPORTED; }
  DGS_HIP_TRY(h, G.dist.reserve(total));
  DGS_HIP_TRY(h, G.q_items.reserve(2 * total));
  DGS_HIP_TRY(h, G.q_best.reserve(2 * total));
  DGS_HIP_TRY(h, G.q_count.reserve(8));
  DGS_HIP_TRY(h, hipMemsetAsync(G.q_count.ptr, 0, 8 * sizeof(int), st));
  NnBatch B;
  B.src_ptrs = d_src_ptrs; B.sizes = d_sizes; B.Tbase = d_T; B.T_stride = T_stride_bytes; B.max_n = max_n;
  const NnQueue Q1{G.q_items.ptr, G.q_best.ptr, G.q_count.ptr}, Q2{G.q_items.ptr + total, G.q_best.ptr + total, G.q_count.ptr + 4};
  const NnGridView v = make_grid_view(G);
  const unsigned bx = (unsigned)((max_n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(nn_grid_level0_kernel, dim3(bx, n_pairs), dim3(kBlock), 0, st, v, B, G.dist.ptr, Q1);
  // the queues' lengths stay on the device: the next passes run grid-stride loops over them (a workgroup without work leaves at once)
  if (h->grid_levels >= 2)
    hipLaunchKernelGGL(nn_grid_level1_kernel, dim3((unsigned)std::min<size_t>((total + kBlock - 1) / kBlock, 2048)), dim3(kBlock), 0, st, v, B, G.dist.ptr, Q1, Q2);
  hipLaunchKernelGGL(nn_tree_queue_kernel, dim3((unsigned)std::min<size_t>((total + kBlock / 8 - 1) / (kBlock / 8), 2048)), dim3(kBlock), 0, st,
                     make_bvh_view(index), B, G.dist.ptr, h->grid_levels >= 2 ? Q2 : Q1);
  DGS_HIP_TRY(h, hipGetLastError());
  return DGS_OK;
}

int nn_grid_launch_fitness(dgs_handle* h, NnGrid& G, const Bvh& index, int n_pairs, const float4* const* d_src_ptrs, const int* d_sizes, int max_n, const float* d_T,
                           size_t T_stride_bytes, float max_range, float inlier_sq, double* partial, int bpp) {
  int rc = nn_grid_distances(h, G, index, n_pairs, d_src_ptrs, d_sizes, max_n, d_T, T_stride_bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(nn_fitness_sum_kernel, dim3(bpp, n_pairs), dim3(kBlock), 0, h->stream, G.dist.ptr, d_sizes, max_n, max_range, inlier_sq, partial, bpp);
  return DGS_OK;
}

// test hook: squared NN distance per query point (a one-"pair" batch under the identity transform, which is exact in float)
int nn_grid_search(dgs_handle* h, NnGrid& G, const Bvh& index, const float4* queries, int64_t m, float* d_sq) {
  hipStream_t st = h->stream;
  DGS_HIP_TRY(h, G.hook.reserve(64));
  struct { const float4* ptr; int n; int pad; float T[16]; } host;
  std::memset(&host, 0, sizeof(host));
  host.ptr = queries;
  host.n = (int)m;
  host.T[0] = host.T[5] = host.T[10] = host.T[15] = 1.f;
  DGS_HIP_TRY(h, hipMemcpyAsync(G.hook.ptr, &host, sizeof(host), hipMemcpyHostToDevice, st));
  DGS_HIP_TRY(h, hipStreamSynchronize(st));   // `host` is a stack object
  const char* base = reinterpret_cast<const char*>(G.hook.ptr);
  int rc = nn_grid_distances(h, G, index, 1, reinterpret_cast<const float4* const*>(base), reinterpret_cast<const int*>(base + 8), (int)m,
                             reinterpret_cast<const float*>(base + 16), 64);
  if (rc) return rc;
  DGS_HIP_TRY(h, hipMemcpyAsync(d_sq, G.dist.ptr, (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, st));
  return DGS_OK;
}

}  // namespace dgs
