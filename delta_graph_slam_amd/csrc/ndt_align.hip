// K2 ndt_derivatives + K3 ndt_solve: the NDT align() loop on the device.
//
// Replaces pclomp::NormalDistributionsTransform::computeTransformation, i.e. what
// registration->align(*aligned, guess) runs at /root/reference/apps/scan_matching_odometry_nodelet.cpp:218 and
// /root/reference/include/hdl_graph_slam/loop_detector.hpp:145 (object configured at
// src/hdl_graph_slam/registrations.cpp:105-119).  Algorithm: SURVEY.md App. A.
//
// MI355X design
//   * One launch of ndt_derivatives covers EVERY still-active pair of a batch and every source point (its ~1024
//     workgroups are re-dealt to the active pairs at each launch, on the device):
//     coalesced 16-B loads of XYZ1 points, 7 (1/27) dependent 4-B cell lookups, 48-B voxel records from L2,
//     float per-point math, double accumulation, wave shuffle -> LDS -> one 28-double partial row per block.
//     No atomics: the rows are summed in a fixed order by ndt_solve, so results are bit-reproducible.
//   * Per point the 7 voxels are folded first into b = sum w C q and N = sum w C - sum w d2 (Cq)(Cq)^T and then
//     projected once through the point Jacobian:  g = J^T b,  H = J^T N J + b . d2T/dp2  -- algebraically
//     the upstream per-voxel update (eq. 6.12/6.13) at ~1/3 of the flops.
//   * ndt_solve (one workgroup per pair) finishes the reduction and runs Newton + the More-Thuente state
//     machine on lane 0, then writes the next evaluation's float transform and angle tables in place, so an
//     iteration is two dependent launches and NO host round trip; finished pairs turn their blocks into
//     immediate returns.  The host only polls a "pairs done" counter once per chunk of iterations.
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "handle.h"
#include "solve6.h"

namespace dgs {

// ================================================================================================ derivatives
template <int SEARCH>
struct Offsets;
template <>
struct Offsets<DGS_NDT_DIRECT1> {
  static constexpr int N = 1;
};
template <>
struct Offsets<DGS_NDT_DIRECT7> {
  static constexpr int N = 7;
};
template <>
struct Offsets<DGS_NDT_DIRECT26> {
  static constexpr int N = 27;
};
template <>
struct Offsets<DGS_NDT_KDTREE> {
  static constexpr int N = 27;
};

template <int SEARCH>
__device__ __forceinline__ void neighbour_offset(int k, int& dx, int& dy, int& dz) {
  if (SEARCH == DGS_NDT_DIRECT1) {
    dx = dy = dz = 0;
  } else if (SEARCH == DGS_NDT_DIRECT7) {
    // (0,0,0) (+x) (-x) (+y) (-y) (+z) (-z): pclomp getNeighborhoodAtPoint7 order
    dx = (k == 1) - (k == 2);
    dy = (k == 3) - (k == 4);
    dz = (k == 5) - (k == 6);
  } else {
    dx = k / 9 - 1;
    dy = (k / 3) % 3 - 1;
    dz = k % 3 - 1;
  }
}

template <bool QUEUE = false, bool DONE_FLAG = false>
__device__ __forceinline__ bool ndt_close_evaluation(NdtPair* st, const double* partials_of_pair, int blocks_per_pair, const NdtConsts& c, int* done_counter, int launch,
                                                     NdtPair* hdr_next = nullptr, int need_h_in = -1);

// FUSED = false: derivatives only; ndt_solve_kernel (one workgroup per pair) follows as a second launch.
// FUSED = true: the workgroup of a pair that finishes LAST (a per-pair ticket) also sums the pair's partial rows in their fixed
// order and advances the optimiser, so an evaluation is ONE launch: no second kernel boundary, no second launch latency, and the
// optimiser steps of pairs that finish early overlap the derivative work of the others.  The hand-off is the write-through form
// of the agent-scope recipe: every byte of a row is stored sc1 (8-byte agent-scope stores), the storing wave drains, a
// workgroup barrier, ONE lane takes the ticket with an agent-scope atomic add; the workgroup whose add came last reads the rows
// with sc1 loads behind a barrier that lane joins.  Results do not depend on placement or timing; the rows are still added in
// slice order.  `launch` numbers the launches of one align: a pair takes part while launch <= its last_launch word,
// which its closing workgroup may write during a launch without changing what the other workgroups of that launch see.
// __launch_bounds__(kBlock, 4) holds the kernel at the derivative loop's 4 waves per SIMD; the optimiser tail (one workgroup per
// pair and launch) spills what does not fit.
// ---- the per-point work of computeDerivatives (fast order), shared by the launch-per-evaluation kernel and the queue kernel ----------
// The angle tables of the evaluation come through an accessor: the pair's record in HBM read with scalar loads (valid across a
// kernel boundary), or a copy in scalar registers made from coherent loads (inside the persistent queue kernel).
struct NdtHdrGlobal {
  const NdtPair& st;
  __device__ __forceinline__ float J(int k, int c) const { return st.jang[k][c]; }
  __device__ __forceinline__ float H(int k, int c) const { return st.hang[k][c]; }
};
struct NdtHdrRegs {
  float j[24], h[45];
  __device__ __forceinline__ float J(int k, int c) const { return j[k * 3 + c]; }
  __device__ __forceinline__ float H(int k, int c) const { return h[k * 3 + c]; }
};

// (A/B builds only since round 4, see ndt_point_loop.)  exp(x) for the default evaluation order: v_exp_f32 of x * log2(e) with the product's rounding error (and the low bits of log2 e)
// folded back in -- within ~1 ulp of the exact value for every x whose result is a normal float, 0 / inf beyond, NaN for NaN --
// in 6 instructions where the library expf takes 14 (its range reduction + ldexp buy correct subnormal results, which the NDT
// weights never need: such a term is < 1e-38 of the sum).  The validation orders use det_expf (common.h).
__device__ __forceinline__ float exp_hw(float x) {
  const float t = x * 1.44269502e+0f;                                         // float(log2 e) = 0x3FB8AA3B
  const float r = __builtin_fmaf(x, 1.92596299e-8f, __builtin_fmaf(x, 1.44269502e+0f, -t));   // x * log2 e - t, to ~2^-48 x
  const float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, r * 6.93147182e-1f, e);                            // 2^(t + r) = 2^t (1 + r ln 2 + ...)
}

template <int SEARCH, class HDR>
__device__ __forceinline__ void ndt_point_loop(const float (&T)[12], const HDR& hdr, const bool need_h, const float4* __restrict__ src, const int n,
                                               const int first, const int stride, const VoxelGrid& g, const double gd1, const float gd2,
                                               const int leaf_pow2, double (&acc)[kAccum]) {
  const float r2 = g.leaf * g.leaf;
  for (int i = first; i < n; i += stride) {
    const float4 x = src[i];
    // pcl::transformPointCloud in float, ((m0 x + m1 y) + m2 z) + m3 with every step rounded (no FMA contraction):
    // q = x' - mean is a cancellation, so one ulp of x' is ~1e-5 of a point's contribution -- keep x' exact.
    const float xt0 = affine_row_rn(T[0], T[1], T[2], T[3], x.x, x.y, x.z);
    const float xt1 = affine_row_rn(T[4], T[5], T[6], T[7], x.x, x.y, x.z);
    const float xt2 = affine_row_rn(T[8], T[9], T[10], T[11], x.x, x.y, x.z);
    // getNeighborhoodAtPoint: floor(x / leaf_size); x * (1 / leaf) is the same number when leaf is a power of two
    const int c0 = (int)floorf(leaf_pow2 ? xt0 * g.inv_leaf : xt0 / g.leaf);
    const int c1 = (int)floorf(leaf_pow2 ? xt1 * g.inv_leaf : xt1 / g.leaf);
    const int c2 = (int)floorf(leaf_pow2 ? xt2 * g.inv_leaf : xt2 / g.leaf);

    // ---- gather: voxel ids of the neighbourhood (independent loads, issued together)
    constexpr int NB = Offsets<SEARCH>::N;
    int vid[NB];
    // interior cells (every neighbour inside the grid) need no per-neighbour bounds test: base pointer + fixed offsets
    const bool interior = c0 > g.min_b[0] && c0 < g.max_b[0] && c1 > g.min_b[1] && c1 < g.max_b[1] && c2 > g.min_b[2] && c2 < g.max_b[2];
    if (interior) {
      const int* __restrict__ base = g.cell2vox + ((c0 - g.min_b[0]) + (c1 - g.min_b[1]) * g.mul1 + (c2 - g.min_b[2]) * g.mul2);
#pragma unroll
      for (int k = 0; k < NB; k++) {
        int dx, dy, dz;
        neighbour_offset<SEARCH>(k, dx, dy, dz);
        vid[k] = base[dx + dy * g.mul1 + dz * g.mul2];
      }
    } else {
#pragma unroll
      for (int k = 0; k < NB; k++) {
        int dx, dy, dz;
        neighbour_offset<SEARCH>(k, dx, dy, dz);
        const int a0 = c0 + dx, a1 = c1 + dy, a2 = c2 + dz;
        const bool inb = a0 >= g.min_b[0] && a0 <= g.max_b[0] && a1 >= g.min_b[1] && a1 <= g.max_b[1] && a2 >= g.min_b[2] && a2 <= g.max_b[2];
        vid[k] = inb ? g.cell2vox[(a0 - g.min_b[0]) + (a1 - g.min_b[1]) * g.mul1 + (a2 - g.min_b[2]) * g.mul2] : -1;
      }
    }

    // ---- fold the neighbourhood:  A = sum w C,  b = sum w C q,  M = sum w d2 (Cq)(Cq)^T,  score
    float N[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0}, sc = 0.f;   // N = A - M, accumulated directly
#if defined(DGS_FAST_AM) && DGS_FAST_AM == 1
    float Mm[6] = {0, 0, 0, 0, 0, 0};
#endif
    bool any = false;
#pragma unroll
    for (int k = 0; k < NB; k++) {
      if (vid[k] < 0) continue;
      if (SEARCH == DGS_NDT_KDTREE) {
        const float4 ce = g.centroid[vid[k]];
        const float ex = ce.x - xt0, ey = ce.y - xt1, ez = ce.z - xt2;
        if (!(ex * ex + ey * ey + ez * ez < r2)) continue;
      }
      const VoxelRec* __restrict__ rec = g.vox + vid[k];
      const float4* __restrict__ r4 = reinterpret_cast<const float4*>(rec);  // three aligned 16-B loads
      const double2 m01 = *reinterpret_cast<const double2*>(rec);
      const float4 rb = r4[1], rc = r4[2];
      const double mx = m01.x, my = m01.y;
      const double mz = __hiloint2double(__float_as_int(rb.y), __float_as_int(rb.x));
      const float q0 = (float)((double)xt0 - mx), q1 = (float)((double)xt1 - my), q2 = (float)((double)xt2 - mz);
      const float Cxx = rb.z, Cxy = rb.w, Cxz = rc.x, Cyy = rc.y, Cyz = rc.z, Czz = rc.w;
      const float u0 = q0 * Cxx + q1 * Cxy + q2 * Cxz;
      const float u1 = q0 * Cxy + q1 * Cyy + q2 * Cyz;
      const float u2 = q0 * Cxz + q1 * Cyz + q2 * Czz;
      // The library expf (<= 1 ulp).  Round 3 had put a 6-instruction hardware form here (exp_hw, <= 2 ulp; 2 % of the step): on the bench shard it
      // moved pair 20 of seed 40 to another optimum, 0.945 m from the reference's and outside the reference's own 34-twin band (0.872 m) --
      // isolated in round 4 with A/B builds (make variants, scripts/r4_fast_variants.py, profiles/r04/fast_order_variants.jsonl: the pair is
      // back inside the gate with expf or det_expf, whichever way N is accumulated) and reverted.  DGS_FAST_EXP=0 / 2 build the other two.
#if defined(DGS_FAST_EXP) && DGS_FAST_EXP == 0
      float e = exp_hw(-gd2 * (q0 * u0 + q1 * u1 + q2 * u2) * 0.5f);
#elif defined(DGS_FAST_EXP) && DGS_FAST_EXP == 2
      float e = det_expf(-gd2 * (q0 * u0 + q1 * u1 + q2 * u2) * 0.5f);
#else
      float e = expf(-gd2 * (q0 * u0 + q1 * u1 + q2 * u2) * 0.5f);
#endif
      // gauss_d1 is a double upstream: float(double(e) * d1), not e * float(d1) -- the float constant alone would scale score,
      // gradient and Hessian by (1 + 2.8e-8) at 1 m resolution, which was the whole per-evaluation difference to a CPU run
      const float score_inc = (float)(-gd1 * (double)e);
      e = gd2 * e;
      if (e > 1.f || e < 0.f || e != e) continue;  // upstream "error checking for invalid values"
      const float w = (float)((double)e * gd1);
      sc += score_inc;
      any = true;
      b[0] += w * u0; b[1] += w * u1; b[2] += w * u2;
      if (need_h) {   // a score + gradient evaluation (a More-Thuente trial) needs neither A nor M: wave-uniform, a scalar branch
        const float wd = w * gd2, t0 = wd * u0, t1 = wd * u1, t2 = wd * u2;
#if defined(DGS_FAST_AM) && DGS_FAST_AM == 1   // A/B build: A and M accumulated apart, N = A - M once per point (round 2's form)
        N[0] += w * Cxx; N[1] += w * Cxy; N[2] += w * Cxz; N[3] += w * Cyy; N[4] += w * Cyz; N[5] += w * Czz;
        Mm[0] += t0 * u0; Mm[1] += t0 * u1; Mm[2] += t0 * u2; Mm[3] += t1 * u1; Mm[4] += t1 * u2; Mm[5] += t2 * u2;
#else
        N[0] += w * Cxx; N[1] += w * Cxy; N[2] += w * Cxz; N[3] += w * Cyy; N[4] += w * Cyz; N[5] += w * Czz;
        N[0] -= t0 * u0; N[1] -= t0 * u1; N[2] -= t0 * u2; N[3] -= t1 * u1; N[4] -= t1 * u2; N[5] -= t2 * u2;
#endif
      }
    }
    if (!any) continue;

    // ---- project through the point Jacobian (eq. 6.18/6.19): J = [I | J3 J4 J5]
    // Rows 5..7 of the table have no z entry (computeAngleDerivatives writes exact zeros there), and the xy parts of rows 0 / 1 are kept:
    // the second-derivative rows f3 / f2 are exactly those (below).
    float xj[8];
    const float jxy0 = hdr.J(0, 0) * x.x + hdr.J(0, 1) * x.y, jxy1 = hdr.J(1, 0) * x.x + hdr.J(1, 1) * x.y;
    xj[0] = jxy0 + hdr.J(0, 2) * x.z;
    xj[1] = jxy1 + hdr.J(1, 2) * x.z;
#pragma unroll
    for (int k = 2; k < 5; k++) xj[k] = hdr.J(k, 0) * x.x + hdr.J(k, 1) * x.y + hdr.J(k, 2) * x.z;
#pragma unroll
    for (int k = 5; k < 8; k++) xj[k] = hdr.J(k, 0) * x.x + hdr.J(k, 1) * x.y;
    const float J3[3] = {0.f, xj[0], xj[1]}, J4[3] = {xj[2], xj[3], xj[4]}, J5[3] = {xj[5], xj[6], xj[7]};
    acc[0] += (double)sc;
    acc[1] += (double)b[0];
    acc[2] += (double)b[1];
    acc[3] += (double)b[2];
    acc[4] += (double)(b[1] * J3[1] + b[2] * J3[2]);
    acc[5] += (double)(b[0] * J4[0] + b[1] * J4[1] + b[2] * J4[2]);
    acc[6] += (double)(b[0] * J5[0] + b[1] * J5[1] + b[2] * J5[2]);
    if (need_h) {
#if defined(DGS_FAST_AM) && DGS_FAST_AM == 1
      const float N0 = N[0] - Mm[0], N1 = N[1] - Mm[1], N2 = N[2] - Mm[2], N3 = N[3] - Mm[3], N4 = N[4] - Mm[4], N5 = N[5] - Mm[5];
#else
      const float N0 = N[0], N1 = N[1], N2 = N[2], N3 = N[3], N4 = N[4], N5 = N[5];
#endif
      // N * J_k
      const float n3[3] = {N1 * J3[1] + N2 * J3[2], N3 * J3[1] + N4 * J3[2], N4 * J3[1] + N5 * J3[2]};
      const float n4[3] = {N0 * J4[0] + N1 * J4[1] + N2 * J4[2], N1 * J4[0] + N3 * J4[1] + N4 * J4[2], N2 * J4[0] + N4 * J4[1] + N5 * J4[2]};
      const float n5[3] = {N0 * J5[0] + N1 * J5[1] + N2 * J5[2], N1 * J5[0] + N3 * J5[1] + N4 * J5[2], N2 * J5[0] + N4 * J5[1] + N5 * J5[2]};
      // Of the fifteen second-derivative rows (eq. 6.21) nine are first-derivative rows (eq. 6.19) again, as computeAngleDerivatives
      // writes them -- the same double expressions or their exact negations, so the float entries are the same bits:
      //   a2 = -j1, a3 = j0, b2 = -j4, b3 = j3, c2 = -j7, c3 = j6;  f1 = xy part of d1, f2 = xy part of a2, f3 = xy part of a3;
      // e1..e3 (and c2, c3, f1..f3) have no z entry.  Same values as the full 15 x 3 products, 30 instructions fewer per point.
      float xh[15];
      xh[0] = -xj[1]; xh[1] = xj[0]; xh[2] = -xj[4]; xh[3] = xj[3]; xh[4] = -xj[7]; xh[5] = xj[6];
      const float hxy6 = hdr.H(6, 0) * x.x + hdr.H(6, 1) * x.y;
      xh[6] = hxy6 + hdr.H(6, 2) * x.z;
#pragma unroll
      for (int k = 7; k < 9; k++) xh[k] = hdr.H(k, 0) * x.x + hdr.H(k, 1) * x.y + hdr.H(k, 2) * x.z;
#pragma unroll
      for (int k = 9; k < 12; k++) xh[k] = hdr.H(k, 0) * x.x + hdr.H(k, 1) * x.y;
      xh[12] = hxy6; xh[13] = -jxy1; xh[14] = jxy0;
      // b . second derivatives: a=(0,xh0,xh1) b=(0,xh2,xh3) c=(0,xh4,xh5) d=(xh6..8) e=(xh9..11) f=(xh12..14)
      const float ba = b[1] * xh[0] + b[2] * xh[1];
      const float bb = b[1] * xh[2] + b[2] * xh[3];
      const float bc = b[1] * xh[4] + b[2] * xh[5];
      const float bd = b[0] * xh[6] + b[1] * xh[7] + b[2] * xh[8];
      const float be = b[0] * xh[9] + b[1] * xh[10] + b[2] * xh[11];
      const float bf = b[0] * xh[12] + b[1] * xh[13] + b[2] * xh[14];
      // upper triangle, row-major: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
      acc[7] += (double)N0;  acc[8] += (double)N1;  acc[9] += (double)N2;  acc[10] += (double)n3[0]; acc[11] += (double)n4[0]; acc[12] += (double)n5[0];
      acc[13] += (double)N3; acc[14] += (double)N4; acc[15] += (double)n3[1]; acc[16] += (double)n4[1]; acc[17] += (double)n5[1];
      acc[18] += (double)N5; acc[19] += (double)n3[2]; acc[20] += (double)n4[2]; acc[21] += (double)n5[2];
      acc[22] += (double)(J3[1] * n3[1] + J3[2] * n3[2] + ba);
      acc[23] += (double)(J3[1] * n4[1] + J3[2] * n4[2] + bb);
      acc[24] += (double)(J3[1] * n5[1] + J3[2] * n5[2] + bc);
      acc[25] += (double)(J4[0] * n4[0] + J4[1] * n4[1] + J4[2] * n4[2] + bd);
      acc[26] += (double)(J4[0] * n5[0] + J4[1] * n5[1] + J4[2] * n5[2] + be);
      acc[27] += (double)(J5[0] * n5[0] + J5[1] * n5[1] + J5[2] * n5[2] + bf);
    }
  }
}

// Block reduction of the 28 per-thread totals into one row.  A DPP butterfly over 28 doubles costs ~900 wave-instructions; instead
// every wave transposes through LDS, 14 values at a time: lane l stores value k at row k (stride 65 doubles: conflict-free both
// ways), then lane k adds the 64 entries of row k in lane order (fixed order -> reproducible).  ~290 wave-instructions.
// COHERENT: the row is handed over inside the launch (common.h, "in-launch hand-off"): write-through stores.
template <bool COHERENT>
__device__ __forceinline__ void ndt_block_row(const double (&acc)[kAccum], double* __restrict__ row_of_slice) {
  constexpr int HALF = kAccum / 2, RS = 65;
  __shared__ double tr[kBlock / kWave][HALF * RS];
  __shared__ double sm[kBlock / kWave][kAccumPad];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* my = tr[wave];
#pragma unroll
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int k = 0; k < HALF; k++) my[k * RS + lane] = acc[h * HALF + k];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
    if (lane < HALF) {
      double v = 0.0;
#pragma unroll 8
      for (int j = 0; j < 64; j++) v += my[lane * RS + j];
      sm[wave][h * HALF + lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
  __syncthreads();
  if (threadIdx.x < kAccumPad) {
    double v = 0.0;
    if (threadIdx.x < kAccum) v = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    double* row = row_of_slice + threadIdx.x;
    if (COHERENT) handoff_store_row(row, v);   // write-through (sc1): no release fence needed
    else *row = v;
  }
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f affine_row_rn2(float m0, float m1, float m2, float m3, v2f x, v2f y, v2f z) {
#pragma clang fp contract(off)
  return ((m0 * x + m1 * y) + m2 * z) + m3;   // v_pk_mul_f32 / v_pk_add_f32: every element individually rounded, as affine_row_rn
}

// PACK2 (instantiated in the EXPERIMENTS build only: measured 29-46 % slower): two source points per lane and step, the float fold and projection written on 2-vectors so that they compile to
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two points per instruction); look-ups, the double q = x' - mean, exp and the
// double accumulation stay per point.  The per-thread order of accumulation is unchanged (i, i + stride, i + 2 stride, ...).
template <int SEARCH, bool FUSED, bool PACK2>
__global__ __launch_bounds__(kBlock, PACK2 ? 2 : 4) void ndt_derivatives_kernel(const float4* const* __restrict__ src_ptrs, const int* __restrict__ src_sizes,
                                                                    NdtPair* __restrict__ pairs, const VoxelGrid g, const double gd1,
                                                                    const float gd2, const int leaf_pow2, double* __restrict__ partials,
                                                                    const int n_pairs, const int cap_blocks, int* __restrict__ pair_blocks,
                                                                    const NdtConsts consts, int* __restrict__ done_counter, const int launch) {
  // ---- map this workgroup to (still-active pair, slice).  The launch always has gridDim.x workgroups; they are dealt
  // evenly to the pairs that are still iterating, so a batch whose pairs converge at different iterations keeps the chip
  // busy on the stragglers instead of spinning up empty blocks.  Every wave derives the same mapping from the pairs'
  // `active` words (written by the previous solve launch): one strided load + ballot per 64 pairs, no inter-block traffic.
  const int lane_id = threadIdx.x & 63;
  int n_active = 0;
  for (int c0 = 0; c0 < n_pairs; c0 += 64) {
    const int pi = c0 + lane_id;
    const int a = (pi < n_pairs) ? (FUSED ? (int)(launch <= pairs[pi].last_launch) : pairs[pi].active) : 0;
    n_active += __popcll(__ballot(a != 0));
  }
  if (n_active == 0) return;
  const int blocks_per_pair = min((int)gridDim.x / n_active, cap_blocks);
  const int rank = blockIdx.x / blocks_per_pair, slice = blockIdx.x % blocks_per_pair;
  if (rank >= n_active) return;
  int pair = -1;
  {
    int seen = 0;
    for (int c0 = 0; c0 < n_pairs && pair < 0; c0 += 64) {
      const int pi = c0 + lane_id;
      const int a = (pi < n_pairs) ? (FUSED ? (int)(launch <= pairs[pi].last_launch) : pairs[pi].active) : 0;
      unsigned long long m = __ballot(a != 0);
      const int cnt = __popcll(m);
      if (rank < seen + cnt) {
        for (int k = rank - seen; k > 0; k--) m &= m - 1ull;  // drop the (rank - seen) lowest set bits
        pair = c0 + __ffsll((long long)m) - 1;
      }
      seen += cnt;
    }
  }
  pair = __builtin_amdgcn_readfirstlane(pair);
  if (slice == 0 && threadIdx.x == 0) pair_blocks[pair] = blocks_per_pair;
  const NdtPair& st = pairs[pair];
  const float4* __restrict__ src = src_ptrs[pair];
  const int n = src_sizes[pair];
  const bool need_h = st.need_hessian != 0;

  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = st.T[k];

  double acc[kAccum];
#pragma unroll
  for (int k = 0; k < kAccum; k++) acc[k] = 0.0;

  if constexpr (PACK2 && SEARCH == DGS_NDT_DIRECT7) {
    constexpr int NB = 7;
    const int stride = blocks_per_pair * kBlock;
    for (int i = slice * kBlock + threadIdx.x; i < n; i += 2 * stride) {
      const int ib = i + stride;
      const bool hb = ib < n;
      const float4 xa = src[i], xb = src[hb ? ib : i];
      v2f X = {xa.x, xb.x}, Y = {xa.y, xb.y}, Z = {xa.z, xb.z};
      const v2f xt0 = affine_row_rn2(T[0], T[1], T[2], T[3], X, Y, Z);
      const v2f xt1 = affine_row_rn2(T[4], T[5], T[6], T[7], X, Y, Z);
      const v2f xt2 = affine_row_rn2(T[8], T[9], T[10], T[11], X, Y, Z);
      int vid[2][NB];
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const float a0 = p ? xt0.y : xt0.x, a1 = p ? xt1.y : xt1.x, a2 = p ? xt2.y : xt2.x;
        const int c0 = (int)floorf(leaf_pow2 ? a0 * g.inv_leaf : a0 / g.leaf);
        const int c1 = (int)floorf(leaf_pow2 ? a1 * g.inv_leaf : a1 / g.leaf);
        const int c2 = (int)floorf(leaf_pow2 ? a2 * g.inv_leaf : a2 / g.leaf);
        const bool interior = c0 > g.min_b[0] && c0 < g.max_b[0] && c1 > g.min_b[1] && c1 < g.max_b[1] && c2 > g.min_b[2] && c2 < g.max_b[2];
        if (interior && (p == 0 || hb)) {
          const int* __restrict__ base = g.cell2vox + ((c0 - g.min_b[0]) + (c1 - g.min_b[1]) * g.mul1 + (c2 - g.min_b[2]) * g.mul2);
#pragma unroll
          for (int k = 0; k < NB; k++) {
            int dx, dy, dz;
            neighbour_offset<SEARCH>(k, dx, dy, dz);
            vid[p][k] = base[dx + dy * g.mul1 + dz * g.mul2];
          }
        } else {
#pragma unroll
          for (int k = 0; k < NB; k++) {
            int dx, dy, dz;
            neighbour_offset<SEARCH>(k, dx, dy, dz);
            const int b0 = c0 + dx, b1 = c1 + dy, b2 = c2 + dz;
            const bool inb = (p == 0 || hb) && b0 >= g.min_b[0] && b0 <= g.max_b[0] && b1 >= g.min_b[1] && b1 <= g.max_b[1] && b2 >= g.min_b[2] && b2 <= g.max_b[2];
            vid[p][k] = inb ? g.cell2vox[(b0 - g.min_b[0]) + (b1 - g.min_b[1]) * g.mul1 + (b2 - g.min_b[2]) * g.mul2] : -1;
          }
        }
      }
      v2f A[6], M[6], b[3], sc = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 6; k++) { A[k] = (v2f){0.f, 0.f}; M[k] = (v2f){0.f, 0.f}; }
#pragma unroll
      for (int k = 0; k < 3; k++) b[k] = (v2f){0.f, 0.f};
      bool any_a = false, any_b = false;
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const int va = vid[0][k], vb = vid[1][k];
        if (va < 0 && vb < 0) continue;
        const float4* __restrict__ ra4 = reinterpret_cast<const float4*>(g.vox + max(va, 0));
        const float4* __restrict__ rb4 = reinterpret_cast<const float4*>(g.vox + max(vb, 0));
        const float4 a0 = ra4[0], a1 = ra4[1], a2 = ra4[2], b0 = rb4[0], b1 = rb4[1], b2 = rb4[2];
        const double mxa = __hiloint2double(__float_as_int(a0.y), __float_as_int(a0.x)), mya = __hiloint2double(__float_as_int(a0.w), __float_as_int(a0.z)),
                     mza = __hiloint2double(__float_as_int(a1.y), __float_as_int(a1.x));
        const double mxb = __hiloint2double(__float_as_int(b0.y), __float_as_int(b0.x)), myb = __hiloint2double(__float_as_int(b0.w), __float_as_int(b0.z)),
                     mzb = __hiloint2double(__float_as_int(b1.y), __float_as_int(b1.x));
        // a voxel slot that is missing for one of the two points contributes exact zeros for it (q = 0 -> u = 0, w = 0)
        const v2f q0 = {va >= 0 ? (float)((double)xt0.x - mxa) : 0.f, vb >= 0 ? (float)((double)xt0.y - mxb) : 0.f};
        const v2f q1 = {va >= 0 ? (float)((double)xt1.x - mya) : 0.f, vb >= 0 ? (float)((double)xt1.y - myb) : 0.f};
        const v2f q2 = {va >= 0 ? (float)((double)xt2.x - mza) : 0.f, vb >= 0 ? (float)((double)xt2.y - mzb) : 0.f};
        const v2f Cxx = {a1.z, b1.z}, Cxy = {a1.w, b1.w}, Cxz = {a2.x, b2.x}, Cyy = {a2.y, b2.y}, Cyz = {a2.z, b2.z}, Czz = {a2.w, b2.w};
        const v2f u0 = q0 * Cxx + q1 * Cxy + q2 * Cxz;
        const v2f u1 = q0 * Cxy + q1 * Cyy + q2 * Cyz;
        const v2f u2 = q0 * Cxz + q1 * Cyz + q2 * Czz;
        const v2f arg = -gd2 * (q0 * u0 + q1 * u1 + q2 * u2) * 0.5f;
        v2f e = {expf(arg.x), expf(arg.y)};   // the library expf, as the default kernel (round 4)
        const float sia = (float)(-gd1 * (double)e.x), sib = (float)(-gd1 * (double)e.y);
        e = gd2 * e;
        const bool oka = va >= 0 && !(e.x > 1.f || e.x < 0.f || e.x != e.x), okb = vb >= 0 && !(e.y > 1.f || e.y < 0.f || e.y != e.y);
        const v2f w = {oka ? (float)((double)e.x * gd1) : 0.f, okb ? (float)((double)e.y * gd1) : 0.f};
        const v2f wd = w * gd2;
        sc += (v2f){oka ? sia : 0.f, okb ? sib : 0.f};
        any_a |= oka;
        any_b |= okb;
        b[0] += w * u0; b[1] += w * u1; b[2] += w * u2;
        A[0] += w * Cxx; A[1] += w * Cxy; A[2] += w * Cxz; A[3] += w * Cyy; A[4] += w * Cyz; A[5] += w * Czz;
        M[0] += wd * u0 * u0; M[1] += wd * u0 * u1; M[2] += wd * u0 * u2; M[3] += wd * u1 * u1; M[4] += wd * u1 * u2; M[5] += wd * u2 * u2;
      }
      if (!any_a && !any_b) continue;
      // a point without any contributing voxel projects exact zeros (also when its coordinates are not finite)
      if (!any_a) { X.x = 0.f; Y.x = 0.f; Z.x = 0.f; }
      if (!any_b) { X.y = 0.f; Y.y = 0.f; Z.y = 0.f; }
      v2f xj[8];
#pragma unroll
      for (int k = 0; k < 8; k++) xj[k] = st.jang[k][0] * X + st.jang[k][1] * Y + st.jang[k][2] * Z;
      const v2f g3 = b[1] * xj[0] + b[2] * xj[1];
      const v2f g4 = b[0] * xj[2] + b[1] * xj[3] + b[2] * xj[4];
      const v2f g5 = b[0] * xj[5] + b[1] * xj[6] + b[2] * xj[7];
#define DGS_ACC2(K, V) { const v2f v_ = (V); acc[K] += (double)v_.x; acc[K] += (double)v_.y; }
      DGS_ACC2(0, sc) DGS_ACC2(1, b[0]) DGS_ACC2(2, b[1]) DGS_ACC2(3, b[2]) DGS_ACC2(4, g3) DGS_ACC2(5, g4) DGS_ACC2(6, g5)
      if (need_h) {
        const v2f N0 = A[0] - M[0], N1 = A[1] - M[1], N2 = A[2] - M[2], N3 = A[3] - M[3], N4 = A[4] - M[4], N5 = A[5] - M[5];
        const v2f n30 = N1 * xj[0] + N2 * xj[1], n31 = N3 * xj[0] + N4 * xj[1], n32 = N4 * xj[0] + N5 * xj[1];
        const v2f n40 = N0 * xj[2] + N1 * xj[3] + N2 * xj[4], n41 = N1 * xj[2] + N3 * xj[3] + N4 * xj[4], n42 = N2 * xj[2] + N4 * xj[3] + N5 * xj[4];
        const v2f n50 = N0 * xj[5] + N1 * xj[6] + N2 * xj[7], n51 = N1 * xj[5] + N3 * xj[6] + N4 * xj[7], n52 = N2 * xj[5] + N4 * xj[6] + N5 * xj[7];
        v2f xh[15];
#pragma unroll
        for (int k = 0; k < 15; k++) xh[k] = st.hang[k][0] * X + st.hang[k][1] * Y + st.hang[k][2] * Z;
        const v2f ba = b[1] * xh[0] + b[2] * xh[1], bb = b[1] * xh[2] + b[2] * xh[3], bc = b[1] * xh[4] + b[2] * xh[5];
        const v2f bd = b[0] * xh[6] + b[1] * xh[7] + b[2] * xh[8], be = b[0] * xh[9] + b[1] * xh[10] + b[2] * xh[11];
        const v2f bf = b[0] * xh[12] + b[1] * xh[13] + b[2] * xh[14];
        DGS_ACC2(7, N0) DGS_ACC2(8, N1) DGS_ACC2(9, N2) DGS_ACC2(10, n30) DGS_ACC2(11, n40) DGS_ACC2(12, n50)
        DGS_ACC2(13, N3) DGS_ACC2(14, N4) DGS_ACC2(15, n31) DGS_ACC2(16, n41) DGS_ACC2(17, n51)
        DGS_ACC2(18, N5) DGS_ACC2(19, n32) DGS_ACC2(20, n42) DGS_ACC2(21, n52)
        DGS_ACC2(22, xj[0] * n31 + xj[1] * n32 + ba)
        DGS_ACC2(23, xj[0] * n41 + xj[1] * n42 + bb)
        DGS_ACC2(24, xj[0] * n51 + xj[1] * n52 + bc)
        DGS_ACC2(25, xj[2] * n40 + xj[3] * n41 + xj[4] * n42 + bd)
        DGS_ACC2(26, xj[2] * n50 + xj[3] * n51 + xj[4] * n52 + be)
        DGS_ACC2(27, xj[5] * n50 + xj[6] * n51 + xj[7] * n52 + bf)
      }
#undef DGS_ACC2
    }
  } else
  {
    ndt_point_loop<SEARCH>(T, NdtHdrGlobal{st}, need_h, src, n, slice * kBlock + (int)threadIdx.x, blocks_per_pair * kBlock, g, gd1, gd2, leaf_pow2, acc);
  }

  ndt_block_row<FUSED>(acc, partials + ((size_t)pair * cap_blocks + slice) * kAccumPad);
  if (!FUSED) return;
  // ---- publish this slice's row, take a ticket; the workgroup that takes the pair's last ticket closes the evaluation
  __shared__ int s_last;
  if (threadIdx.x < kAccumPad) handoff_drain_stores();   // the storing wave drains its stores
  __syncthreads();
  if (threadIdx.x == 0) s_last = handoff_take_ticket(&pairs[pair].ticket, blocks_per_pair) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
#ifdef DGS_CLOSE_STAMPS
  if (threadIdx.x == 0 && pairs[pair].s.nr_iterations == 1) pairs[pair].traj[kTrajCap - 1][5] = (double)wall_clock64();
#endif
  ndt_close_evaluation<false, true>(pairs + pair, partials + (size_t)pair * cap_blocks * kAccumPad, blocks_per_pair, consts, done_counter + pair, launch);
}

// Everything below this line -- the validation-mode evaluation, the optimiser (Newton step, More-Thuente state machine,
// transform / angle tables of the next evaluation) and the host code -- is compiled with floating-point contraction OFF: each
// operation is rounded on its own, as in a CPU build of upstream without FMA, so the optimiser's double arithmetic follows the
// CPU checker operation for operation.  Only the default derivative kernel above lets the compiler fuse multiply-adds.
#pragma clang fp contract(off)

// glibc's __exp_data.tab (N = 128): [2 i] = asuint64(tail_i), [2 i + 1] = asuint64(scale_i) - (i << 45), 2^(i/128) = scale_i (1 + tail_i); generated
// from 2^(i/128) at 120 decimal digits (glibc_exp_dev, ndt_strict.h)
__constant__ unsigned long long kGlibcExpTab[256] = {
    0x0000000000000000ull, 0x3ff0000000000000ull, 0x3c9b3b4f1a88bf6eull, 0x3feff63da9fb3335ull, 0xbc7160139cd8dc5dull, 0x3fefec9a3e778061ull,
    0xbc905e7a108766d1ull, 0x3fefe315e86e7f85ull, 0x3c8cd2523567f613ull, 0x3fefd9b0d3158574ull, 0xbc8bce8023f98efaull, 0x3fefd06b29ddf6deull,
    0x3c60f74e61e6c861ull, 0x3fefc74518759bc8ull, 0x3c90a3e45b33d399ull, 0x3fefbe3ecac6f383ull, 0x3c979aa65d837b6dull, 0x3fefb5586cf9890full,
    0x3c8eb51a92fdeffcull, 0x3fefac922b7247f7ull, 0x3c3ebe3d702f9cd1ull, 0x3fefa3ec32d3d1a2ull, 0xbc6a033489906e0bull, 0x3fef9b66affed31bull,
    0xbc9556522a2fbd0eull, 0x3fef9301d0125b51ull, 0xbc5080ef8c4eea55ull, 0x3fef8abdc06c31ccull, 0xbc91c923b9d5f416ull, 0x3fef829aaea92de0ull,
    0x3c80d3e3e95c55afull, 0x3fef7a98c8a58e51ull, 0xbc801b15eaa59348ull, 0x3fef72b83c7d517bull, 0xbc8f1ff055de323dull, 0x3fef6af9388c8deaull,
    0x3c8b898c3f1353bfull, 0x3fef635beb6fcb75ull, 0xbc96d99c7611eb26ull, 0x3fef5be084045cd4ull, 0x3c9aecf73e3a2f60ull, 0x3fef54873168b9aaull,
    0xbc8fe782cb86389dull, 0x3fef4d5022fcd91dull, 0x3c8a6f4144a6c38dull, 0x3fef463b88628cd6ull, 0x3c807a05b0e4047dull, 0x3fef3f49917ddc96ull,
    0x3c968efde3a8a894ull, 0x3fef387a6e756238ull, 0x3c875e18f274487dull, 0x3fef31ce4fb2a63full, 0x3c80472b981fe7f2ull, 0x3fef2b4565e27cddull,
    0xbc96b87b3f71085eull, 0x3fef24dfe1f56381ull, 0x3c82f7e16d09ab31ull, 0x3fef1e9df51fdee1ull, 0xbc3d219b1a6fbffaull, 0x3fef187fd0dad990ull,
    0x3c8b3782720c0ab4ull, 0x3fef1285a6e4030bull, 0x3c6e149289cecb8full, 0x3fef0cafa93e2f56ull, 0x3c834d754db0abb6ull, 0x3fef06fe0a31b715ull,
    0x3c864201e2ac744cull, 0x3fef0170fc4cd831ull, 0x3c8fdd395dd3f84aull, 0x3feefc08b26416ffull, 0xbc86a3803b8e5b04ull, 0x3feef6c55f929ff1ull,
    0xbc924aedcc4b5068ull, 0x3feef1a7373aa9cbull, 0xbc9907f81b512d8eull, 0x3feeecae6d05d866ull, 0xbc71d1e83e9436d2ull, 0x3feee7db34e59ff7ull,
    0xbc991919b3ce1b15ull, 0x3feee32dc313a8e5ull, 0x3c859f48a72a4c6dull, 0x3feedea64c123422ull, 0xbc9312607a28698aull, 0x3feeda4504ac801cull,
    0xbc58a78f4817895bull, 0x3feed60a21f72e2aull, 0xbc7c2c9b67499a1bull, 0x3feed1f5d950a897ull, 0x3c4363ed60c2ac11ull, 0x3feece086061892dull,
    0x3c9666093b0664efull, 0x3feeca41ed1d0057ull, 0x3c6ecce1daa10379ull, 0x3feec6a2b5c13cd0ull, 0x3c93ff8e3f0f1230ull, 0x3feec32af0d7d3deull,
    0x3c7690cebb7aafb0ull, 0x3feebfdad5362a27ull, 0x3c931dbdeb54e077ull, 0x3feebcb299fddd0dull, 0xbc8f94340071a38eull, 0x3feeb9b2769d2ca7ull,
    0xbc87deccdc93a349ull, 0x3feeb6daa2cf6642ull, 0xbc78dec6bd0f385full, 0x3feeb42b569d4f82ull, 0xbc861246ec7b5cf6ull, 0x3feeb1a4ca5d920full,
    0x3c93350518fdd78eull, 0x3feeaf4736b527daull, 0x3c7b98b72f8a9b05ull, 0x3feead12d497c7fdull, 0x3c9063e1e21c5409ull, 0x3feeab07dd485429ull,
    0x3c34c7855019c6eaull, 0x3feea9268a5946b7ull, 0x3c9432e62b64c035ull, 0x3feea76f15ad2148ull, 0xbc8ce44a6199769full, 0x3feea5e1b976dc09ull,
    0xbc8c33c53bef4da8ull, 0x3feea47eb03a5585ull, 0xbc845378892be9aeull, 0x3feea34634ccc320ull, 0xbc93cedd78565858ull, 0x3feea23882552225ull,
    0x3c5710aa807e1964ull, 0x3feea155d44ca973ull, 0xbc93b3efbf5e2228ull, 0x3feea09e667f3bcdull, 0xbc6a12ad8734b982ull, 0x3feea012750bdabfull,
    0xbc6367efb86da9eeull, 0x3fee9fb23c651a2full, 0xbc80dc3d54e08851ull, 0x3fee9f7df9519484ull, 0xbc781f647e5a3ecfull, 0x3fee9f75e8ec5f74ull,
    0xbc86ee4ac08b7db0ull, 0x3fee9f9a48a58174ull, 0xbc8619321e55e68aull, 0x3fee9feb564267c9ull, 0x3c909ccb5e09d4d3ull, 0x3feea0694fde5d3full,
    0xbc7b32dcb94da51dull, 0x3feea11473eb0187ull, 0x3c94ecfd5467c06bull, 0x3feea1ed0130c132ull, 0x3c65ebe1abd66c55ull, 0x3feea2f336cf4e62ull,
    0xbc88a1c52fb3cf42ull, 0x3feea427543e1a12ull, 0xbc9369b6f13b3734ull, 0x3feea589994cce13ull, 0xbc805e843a19ff1eull, 0x3feea71a4623c7adull,
    0xbc94d450d872576eull, 0x3feea8d99b4492edull, 0x3c90ad675b0e8a00ull, 0x3feeaac7d98a6699ull, 0x3c8db72fc1f0eab4ull, 0x3feeace5422aa0dbull,
    0xbc65b6609cc5e7ffull, 0x3feeaf3216b5448cull, 0x3c7bf68359f35f44ull, 0x3feeb1ae99157736ull, 0xbc93091fa71e3d83ull, 0x3feeb45b0b91ffc6ull,
    0xbc5da9b88b6c1e29ull, 0x3feeb737b0cdc5e5ull, 0xbc6c23f97c90b959ull, 0x3feeba44cbc8520full, 0xbc92434322f4f9aaull, 0x3feebd829fde4e50ull,
    0xbc85ca6cd7668e4bull, 0x3feec0f170ca07baull, 0x3c71affc2b91ce27ull, 0x3feec49182a3f090ull, 0x3c6dd235e10a73bbull, 0x3feec86319e32323ull,
    0xbc87c50422622263ull, 0x3feecc667b5de565ull, 0x3c8b1c86e3e231d5ull, 0x3feed09bec4a2d33ull, 0xbc91bbd1d3bcbb15ull, 0x3feed503b23e255dull,
    0x3c90cc319cee31d2ull, 0x3feed99e1330b358ull, 0x3c8469846e735ab3ull, 0x3feede6b5579fdbfull, 0xbc82dfcd978e9db4ull, 0x3feee36bbfd3f37aull,
    0x3c8c1a7792cb3387ull, 0x3feee89f995ad3adull, 0xbc907b8f4ad1d9faull, 0x3feeee07298db666ull, 0xbc55c3d956dcaebaull, 0x3feef3a2b84f15fbull,
    0xbc90a40e3da6f640ull, 0x3feef9728de5593aull, 0xbc68d6f438ad9334ull, 0x3feeff76f2fb5e47ull, 0xbc91eee26b588a35ull, 0x3fef05b030a1064aull,
    0x3c74ffd70a5fddcdull, 0x3fef0c1e904bc1d2ull, 0xbc91bdfbfa9298acull, 0x3fef12c25bd71e09ull, 0x3c736eae30af0cb3ull, 0x3fef199bdd85529cull,
    0x3c8ee3325c9ffd94ull, 0x3fef20ab5fffd07aull, 0x3c84e08fd10959acull, 0x3fef27f12e57d14bull, 0x3c63cdaf384e1a67ull, 0x3fef2f6d9406e7b5ull,
    0x3c676b2c6c921968ull, 0x3fef3720dcef9069ull, 0xbc808a1883ccb5d2ull, 0x3fef3f0b555dc3faull, 0xbc8fad5d3ffffa6full, 0x3fef472d4a07897cull,
    0xbc900dae3875a949ull, 0x3fef4f87080d89f2ull, 0x3c74a385a63d07a7ull, 0x3fef5818dcfba487ull, 0xbc82919e2040220full, 0x3fef60e316c98398ull,
    0x3c8e5a50d5c192acull, 0x3fef69e603db3285ull, 0x3c843a59ac016b4bull, 0x3fef7321f301b460ull, 0xbc82d52107b43e1full, 0x3fef7c97337b9b5full,
    0xbc892ab93b470dc9ull, 0x3fef864614f5a129ull, 0x3c74b604603a88d3ull, 0x3fef902ee78b3ff6ull, 0x3c83c5ec519d7271ull, 0x3fef9a51fbc74c83ull,
    0xbc8ff7128fd391f0ull, 0x3fefa4afa2a490daull, 0xbc8dae98e223747dull, 0x3fefaf482d8e67f1ull, 0x3c8ec3bc41aa2008ull, 0x3fefba1bee615a27ull,
    0x3c842b94c3a9eb32ull, 0x3fefc52b376bba97ull, 0x3c8a64a931d185eeull, 0x3fefd0765b6e4540ull, 0xbc8e37bae43be3edull, 0x3fefdbfdad9cbe14ull,
    0x3c77893b4d91cd9dull, 0x3fefe7c1819e90d8ull, 0x3c5305c14160cc89ull, 0x3feff3c22b8f71f1ull};

// glibc's __exp2f_data.tab: asuint64(2^(i/32)) - (i << 47), generated from 2^(i/32) at 80 decimal digits (glibc_expf_dev, common.h)
__constant__ unsigned long long kGlibcExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// ================================================================================================ validation modes
// dgs_params.ndt_strict_order >= 1: computeDerivatives / updateDerivatives in upstream's own operation order (SURVEY.md App. A
// "Per point"; the CPU checker states the same sequence).  Per point: float point gradient (3x6) and second-derivative
// vectors, then per neighbour voxel q = float(double(x') - mean), C = float(icov) (all 9 entries: after the eigenvalue clamp
// the covariance is rebuilt as V diag V^-1 and is not exactly symmetric), q^T C, exp, the float 3x6 product C * J, the float
// gradient / Hessian increments, each converted and added to the point's DOUBLE totals.  One point -> 43 doubles.
// LITERAL = false (default): the same values with upstream's structural zeros and ones not multiplied out -- the point gradient
// is [I | J3 J4 J5] with a zero in J3's first row, the point Hessian is zero outside its 3x3 rotational block: 1 * a, a + 0 and
// 0 * a are exact whenever a is finite, so C * J, x^T C H and J^T C J shrink from ~800 to ~450 float operations per voxel and the
// register copy of the 6x6x3 point Hessian to its 6 distinct vectors.  Every operation that remains is upstream's, in upstream's
// order.  Bit-identical to LITERAL = true (DGS_NDT_STRICT_LITERAL=1; test_strict_gpu.py::test_structural_zero_shortcuts_are_bit_identical)
// as long as the float products stay finite; where one overflows upstream turns 0 * inf into NaN and this path keeps inf -- both
// end in a non-finite Hessian and a failed registration.
template <int SEARCH, bool LITERAL>
__device__ __forceinline__ void ndt_point_strict(const float4 x, const float* T, const NdtPair& st, const VoxelGrid& g, const double* __restrict__ vtab,
                                                 const double gauss_d1, const float gd2, const int leaf_pow2, const bool need_h, double* out, const bool exp_libm) {
#pragma unroll
  for (int k = 0; k < kStrictAccum; k++) out[k] = 0.0;
  float xt[3];
  xt[0] = affine_row_rn(T[0], T[1], T[2], T[3], x.x, x.y, x.z);
  xt[1] = affine_row_rn(T[4], T[5], T[6], T[7], x.x, x.y, x.z);
  xt[2] = affine_row_rn(T[8], T[9], T[10], T[11], x.x, x.y, x.z);
  const int c0 = (int)floorf(leaf_pow2 ? xt[0] * g.inv_leaf : xt[0] / g.leaf);
  const int c1 = (int)floorf(leaf_pow2 ? xt[1] * g.inv_leaf : xt[1] / g.leaf);
  const int c2 = (int)floorf(leaf_pow2 ? xt[2] * g.inv_leaf : xt[2] / g.leaf);
  constexpr int NB = Offsets<SEARCH>::N;
  const float r2 = g.leaf * g.leaf;
  // computePointDerivatives
  const float xp[3] = {x.x, x.y, x.z};
  float pg[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
  float xj[8];
#pragma unroll
  for (int i = 0; i < 8; i++) xj[i] = st.jang[i][0] * xp[0] + st.jang[i][1] * xp[1] + st.jang[i][2] * xp[2];
  pg[1][3] = xj[0]; pg[2][3] = xj[1];
  pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4];
  pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
  // the 6 distinct vectors of the point Hessian's rotational block: (3,3) (3,4) (3,5) (4,4) (4,5) (5,5); zero without a Hessian
  float hv[6][3];
#pragma unroll
  for (int i = 0; i < 6; i++) hv[i][0] = hv[i][1] = hv[i][2] = 0.f;
  if (need_h) {
    float xh[15];
#pragma unroll
    for (int i = 0; i < 15; i++) xh[i] = st.hang[i][0] * xp[0] + st.hang[i][1] * xp[1] + st.hang[i][2] * xp[2];
    hv[0][1] = xh[0]; hv[0][2] = xh[1];     // a = (0, xh0, xh1)
    hv[1][1] = xh[2]; hv[1][2] = xh[3];     // b
    hv[2][1] = xh[4]; hv[2][2] = xh[5];     // c
    hv[3][0] = xh[6]; hv[3][1] = xh[7]; hv[3][2] = xh[8];       // d
    hv[4][0] = xh[9]; hv[4][1] = xh[10]; hv[4][2] = xh[11];     // e
    hv[5][0] = xh[12]; hv[5][1] = xh[13]; hv[5][2] = xh[14];    // f
  }
  // (i, j) of the rotational block -> its vector
  auto hvec = [&](int i, int j) -> const float* {
    const int lo = (i < j ? i : j) - 3, hi = (i < j ? j : i) - 3;
    return hv[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
  };
  double score_pt = 0.0, g_pt[6] = {0, 0, 0, 0, 0, 0}, h_pt[36];
#pragma unroll
  for (int k = 0; k < 36; k++) h_pt[k] = 0.0;
  // the neighbourhood's voxel ids first (independent loads, issued together): at 2 waves per SIMD (172 VGPRs are the double totals of
  // the point and of the thread) little else hides the table's latency.  (Loading the next voxel's record one iteration ahead was
  // tried: 24 more live registers, 1 wave per SIMD, 15 -> 19.7 ms per step.)
  int vids[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) {
    int dx, dy, dz;
    neighbour_offset<SEARCH>(k, dx, dy, dz);
    const int a0 = c0 + dx, a1 = c1 + dy, a2 = c2 + dz;
    const bool inb = a0 >= g.min_b[0] && a0 <= g.max_b[0] && a1 >= g.min_b[1] && a1 <= g.max_b[1] && a2 >= g.min_b[2] && a2 <= g.max_b[2];
    vids[k] = inb ? g.cell2vox[(a0 - g.min_b[0]) + (a1 - g.min_b[1]) * g.mul1 + (a2 - g.min_b[2]) * g.mul2] : -1;
  }
  if (SEARCH == DGS_NDT_KDTREE) {
#pragma unroll
    for (int k = 0; k < NB; k++) {
      if (vids[k] < 0) continue;
      const float4 ce = g.centroid[vids[k]];
      const float ex = ce.x - xt[0], ey = ce.y - xt[1], ez = ce.z - xt[2];
      if (!(ex * ex + ey * ey + ez * ez < r2)) vids[k] = -1;
    }
  }
#pragma unroll 1
  for (int k = 0; k < NB; k++) {
    int vid = vids[0];   // vids stays in registers: a dynamic subscript is a chain of selects, not scratch memory
#pragma unroll
    for (int j = 1; j < NB; j++) vid = (k == j) ? vids[j] : vid;
    if (vid < 0) continue;
    const double* __restrict__ rec = vtab + (size_t)vid * 12;  // mean[3], icov[9] (row-major), double
    float q[3], C[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) q[r] = (float)((double)xt[r] - rec[r]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) C[r][c] = (float)rec[3 + r * 3 + c];
    float qC[3];
#pragma unroll
    for (int c = 0; c < 3; c++) qC[c] = q[0] * C[0][c] + q[1] * C[1][c] + q[2] * C[2][c];
    const float e_arg = -gd2 * (q[0] * qC[0] + q[1] * qC[1] + q[2] * qC[2]) * 0.5f;
    float e_x_cov_x = exp_libm ? glibc_expf_dev(e_arg, kGlibcExp2fTab) : det_expf(e_arg);
    const float score_inc = (float)(-gauss_d1 * (double)e_x_cov_x);
    e_x_cov_x = gd2 * e_x_cov_x;
    if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;
    e_x_cov_x = (float)((double)e_x_cov_x * gauss_d1);
    float cPG[3][6];
    float g6[6];
    constexpr bool literal = LITERAL;
    if (!LITERAL) {
#pragma unroll
      for (int r = 0; r < 3; r++) {
        cPG[r][0] = C[r][0]; cPG[r][1] = C[r][1]; cPG[r][2] = C[r][2];            // C * (unit column): exact
        cPG[r][3] = C[r][1] * pg[1][3] + C[r][2] * pg[2][3];                      // (C0 * 0 + m1) + m2
#pragma unroll
        for (int c = 4; c < 6; c++) cPG[r][c] = C[r][0] * pg[0][c] + C[r][1] * pg[1][c] + C[r][2] * pg[2][c];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) cPG[r][c] = C[r][0] * pg[0][c] + C[r][1] * pg[1][c] + C[r][2] * pg[2][c];
    }
#pragma unroll
    for (int c = 0; c < 6; c++) g6[c] = q[0] * cPG[0][c] + q[1] * cPG[1][c] + q[2] * cPG[2][c];
#pragma unroll
    for (int c = 0; c < 6; c++) g_pt[c] += (double)(e_x_cov_x * g6[c]);
    if (need_h) {
      if (literal) {
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = 0; j < 6; j++) {
            float xCH = qC[0] * 0.f + qC[1] * 0.f + qC[2] * 0.f;
            if (i >= 3 && j >= 3) {
              const float* v = hvec(i, j);
              xCH = qC[0] * v[0] + qC[1] * v[1] + qC[2] * v[2];
            }
            const float pcp = pg[0][j] * cPG[0][i] + pg[1][j] * cPG[1][i] + pg[2][j] * cPG[2][i];
            h_pt[i * 6 + j] += (double)(e_x_cov_x * (-gd2 * g6[i] * g6[j] + xCH + pcp));
          }
      } else {
        float xch[6];   // x^T C H for the 6 distinct vectors
#pragma unroll
        for (int v = 0; v < 6; v++) xch[v] = qC[0] * hv[v][0] + qC[1] * hv[v][1] + qC[2] * hv[v][2];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = 0; j < 6; j++) {
            float t = -gd2 * g6[i] * g6[j];
            if (i >= 3 && j >= 3) {
              const int lo = (i < j ? i : j) - 3, hi = (i < j ? j : i) - 3;
              t = t + xch[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
            }
            // J^T C J: column j of J is a unit vector for j < 3, has a zero first entry for j == 3
            const float pcp = (j < 3) ? cPG[j][i] : (j == 3) ? (pg[1][3] * cPG[1][i] + pg[2][3] * cPG[2][i]) : (pg[0][j] * cPG[0][i] + pg[1][j] * cPG[1][i] + pg[2][j] * cPG[2][i]);
            h_pt[i * 6 + j] += (double)(e_x_cov_x * (t + pcp));
          }
      }
    }
    score_pt += (double)score_inc;
  }
  out[0] = score_pt;
#pragma unroll
  for (int k = 0; k < 6; k++) out[1 + k] = g_pt[k];
#pragma unroll
  for (int k = 0; k < 36; k++) out[7 + k] = h_pt[k];
}

// ================================================================================================ solver
// float transform + angle-derivative tables of pose x (computeAngleDerivatives: double trig, |angle| < 1e-4 snap),
// written to the pair's HBM record by lane 0 (`writer`); every lane computes the same values.
// double sin/cos is software on the GPU (~100s of instructions per call) and an evaluation needs twelve of them; when the
// whole wave runs this code with identical inputs (solve kernel), lane k evaluates angle k and the results are broadcast.
template <bool WAVE>
__device__ __forceinline__ void trig6(const double* ang, double* sn, double* cs) {
  if (WAVE) {
    const int lane = threadIdx.x & 63;
    double a = ang[0];
#pragma unroll
    for (int k = 1; k < 6; k++) a = (lane == k) ? ang[k] : a;
    double sv, cv;
    sincos(a, &sv, &cv);
#pragma unroll
    for (int k = 0; k < 6; k++) { sn[k] = readlane_f64(sv, k); cs[k] = readlane_f64(cv, k); }
  } else {
#pragma unroll
    for (int k = 0; k < 6; k++) sincos(ang[k], &sn[k], &cs[k]);
  }
}

template <bool COH>
__device__ __forceinline__ void hdr_put(float* p, float v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH>
__device__ __forceinline__ void hdr_put_int(int* p, int v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// `hdr` receives what the derivative pass of the evaluation reads (transform, angle tables, need_hessian): the pair's own record, or -- in
// the queue kernel -- the record slot of the pair's NEXT round; final_T always goes to the pair's record `st`.
// COH: read by other workgroups of the SAME launch (queue kernel): every word is written through (agent-scope stores)
template <bool WAVE, bool COH = false>
__device__ __forceinline__ void write_evaluation(NdtPair* st, NdtPair* hdr, NdtSolver& s, const NdtConsts& c, const double* x, int need_hessian, bool write_T, bool writer) {
  // angles 0..2: the FLOAT-rounded pose angles (transform entries), 3..5: the double pose angles (derivative tables)
  const double ang[6] = {(double)(float)x[3], (double)(float)x[4], (double)(float)x[5], x[3], x[4], x[5]};
  double sn[6], cs[6];
  trig6<WAVE>(ang, sn, cs);
  if (write_T) {
    // Eigen builds Translation * AngleAxis(x) * AngleAxis(y) * AngleAxis(z) in float.  One ulp of a rotation entry moves
    // a point at 50 m by 3 um, which the q = x' - mean cancellation turns into ~1e-4 of that point's contribution, so
    // the entries are formed reproducibly: trig of the FLOAT angle evaluated in double and rounded once (what a
    // correctly rounded cosf/sinf returns), products individually rounded in the source order of the expression.
    const float cx = (float)cs[0], sx = (float)sn[0], cy = (float)cs[1], sy = (float)sn[1], cz = (float)cs[2], sz = (float)sn[2];
    const float r00 = mul_rn(cy, cz), r01 = mul_rn(-cy, sz), r02 = sy;
    const float r10 = add_rn(mul_rn(cx, sz), mul_rn(mul_rn(sx, sy), cz)), r11 = sub_rn(mul_rn(cx, cz), mul_rn(mul_rn(sx, sy), sz)), r12 = mul_rn(-sx, cy);
    const float r20 = sub_rn(mul_rn(sx, sz), mul_rn(mul_rn(cx, sy), cz)), r21 = add_rn(mul_rn(sx, cz), mul_rn(mul_rn(cx, sy), sz)), r22 = mul_rn(cx, cy);
    const float t0 = (float)x[0], t1 = (float)x[1], t2 = (float)x[2];
    if (writer) {
      hdr_put<COH>(&hdr->T[0], r00); hdr_put<COH>(&hdr->T[1], r01); hdr_put<COH>(&hdr->T[2], r02); hdr_put<COH>(&hdr->T[3], t0);
      hdr_put<COH>(&hdr->T[4], r10); hdr_put<COH>(&hdr->T[5], r11); hdr_put<COH>(&hdr->T[6], r12); hdr_put<COH>(&hdr->T[7], t1);
      hdr_put<COH>(&hdr->T[8], r20); hdr_put<COH>(&hdr->T[9], r21); hdr_put<COH>(&hdr->T[10], r22); hdr_put<COH>(&hdr->T[11], t2);
      float* F = st->final_T;  // column-major
      hdr_put<COH>(&F[0], r00); hdr_put<COH>(&F[1], r10); hdr_put<COH>(&F[2], r20); hdr_put<COH>(&F[3], 0.f);
      hdr_put<COH>(&F[4], r01); hdr_put<COH>(&F[5], r11); hdr_put<COH>(&F[6], r21); hdr_put<COH>(&F[7], 0.f);
      hdr_put<COH>(&F[8], r02); hdr_put<COH>(&F[9], r12); hdr_put<COH>(&F[10], r22); hdr_put<COH>(&F[11], 0.f);
      hdr_put<COH>(&F[12], t0); hdr_put<COH>(&F[13], t1); hdr_put<COH>(&F[14], t2); hdr_put<COH>(&F[15], 1.f);
    }
  }
  double cx, cy, cz, sx, sy, sz;
  if (fabs(x[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cs[3]; sx = sn[3]; }
  if (fabs(x[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cs[4]; sy = sn[4]; }
  if (fabs(x[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cs[5]; sz = sn[5]; }
  if (writer) {
    float (*J)[3] = hdr->jang;
    hdr_put<COH>(&J[0][0], (float)(-sx * sz + cx * sy * cz)); hdr_put<COH>(&J[0][1], (float)(-sx * cz - cx * sy * sz)); hdr_put<COH>(&J[0][2], (float)(-cx * cy));
    hdr_put<COH>(&J[1][0], (float)(cx * sz + sx * sy * cz));  hdr_put<COH>(&J[1][1], (float)(cx * cz - sx * sy * sz));  hdr_put<COH>(&J[1][2], (float)(-sx * cy));
    hdr_put<COH>(&J[2][0], (float)(-sy * cz));                hdr_put<COH>(&J[2][1], (float)(sy * sz));                 hdr_put<COH>(&J[2][2], (float)(cy));
    hdr_put<COH>(&J[3][0], (float)(sx * cy * cz));            hdr_put<COH>(&J[3][1], (float)(-sx * cy * sz));           hdr_put<COH>(&J[3][2], (float)(sx * sy));
    hdr_put<COH>(&J[4][0], (float)(-cx * cy * cz));           hdr_put<COH>(&J[4][1], (float)(cx * cy * sz));            hdr_put<COH>(&J[4][2], (float)(-cx * sy));
    hdr_put<COH>(&J[5][0], (float)(-cy * sz));                hdr_put<COH>(&J[5][1], (float)(-cy * cz));                hdr_put<COH>(&J[5][2], 0.f);
    hdr_put<COH>(&J[6][0], (float)(cx * cz - sx * sy * sz));  hdr_put<COH>(&J[6][1], (float)(-cx * sz - sx * sy * cz)); hdr_put<COH>(&J[6][2], 0.f);
    hdr_put<COH>(&J[7][0], (float)(sx * cz + cx * sy * sz));  hdr_put<COH>(&J[7][1], (float)(cx * sy * cz - sx * sz));  hdr_put<COH>(&J[7][2], 0.f);
    float (*H)[3] = hdr->hang;
    if (need_hessian) {  // a score + gradient evaluation (More-Thuente trial) never reads the second-derivative tables
    hdr_put<COH>(&H[0][0], (float)(-cx * sz - sx * sy * cz)); hdr_put<COH>(&H[0][1], (float)(-cx * cz + sx * sy * sz)); hdr_put<COH>(&H[0][2], (float)(sx * cy));    // a2
    hdr_put<COH>(&H[1][0], (float)(-sx * sz + cx * sy * cz)); hdr_put<COH>(&H[1][1], (float)(-cx * sy * sz - sx * cz)); hdr_put<COH>(&H[1][2], (float)(-cx * cy));   // a3
    hdr_put<COH>(&H[2][0], (float)(cx * cy * cz));            hdr_put<COH>(&H[2][1], (float)(-cx * cy * sz));           hdr_put<COH>(&H[2][2], (float)(cx * sy));    // b2
    hdr_put<COH>(&H[3][0], (float)(sx * cy * cz));            hdr_put<COH>(&H[3][1], (float)(-sx * cy * sz));           hdr_put<COH>(&H[3][2], (float)(sx * sy));    // b3
    hdr_put<COH>(&H[4][0], (float)(-sx * cz - cx * sy * sz)); hdr_put<COH>(&H[4][1], (float)(sx * sz - cx * sy * cz));  hdr_put<COH>(&H[4][2], 0.f);                 // c2
    hdr_put<COH>(&H[5][0], (float)(cx * cz - sx * sy * sz));  hdr_put<COH>(&H[5][1], (float)(-sx * sy * cz - cx * sz)); hdr_put<COH>(&H[5][2], 0.f);                 // c3
    // d1: upstream PCL / ndt_omp carry +sy in the z slot; the exact second derivative is -sy (dgs_params.ndt_fix_hessian_d1)
    hdr_put<COH>(&H[6][0], (float)(-cy * cz));                hdr_put<COH>(&H[6][1], (float)(cy * sz));                 hdr_put<COH>(&H[6][2], (float)(c.fix_hessian_d1 ? -sy : sy));
    hdr_put<COH>(&H[7][0], (float)(-sx * sy * cz));           hdr_put<COH>(&H[7][1], (float)(sx * sy * sz));            hdr_put<COH>(&H[7][2], (float)(sx * cy));    // d2
    hdr_put<COH>(&H[8][0], (float)(cx * sy * cz));            hdr_put<COH>(&H[8][1], (float)(-cx * sy * sz));           hdr_put<COH>(&H[8][2], (float)(-cx * cy));   // d3
    hdr_put<COH>(&H[9][0], (float)(sy * sz));                 hdr_put<COH>(&H[9][1], (float)(sy * cz));                 hdr_put<COH>(&H[9][2], 0.f);                 // e1
    hdr_put<COH>(&H[10][0], (float)(-sx * cy * sz));          hdr_put<COH>(&H[10][1], (float)(-sx * cy * cz));          hdr_put<COH>(&H[10][2], 0.f);                // e2
    hdr_put<COH>(&H[11][0], (float)(cx * cy * sz));           hdr_put<COH>(&H[11][1], (float)(cx * cy * cz));           hdr_put<COH>(&H[11][2], 0.f);                // e3
    hdr_put<COH>(&H[12][0], (float)(-cy * cz));               hdr_put<COH>(&H[12][1], (float)(cy * sz));                hdr_put<COH>(&H[12][2], 0.f);                // f1
    hdr_put<COH>(&H[13][0], (float)(-cx * sz - sx * sy * cz)); hdr_put<COH>(&H[13][1], (float)(-cx * cz + sx * sy * sz)); hdr_put<COH>(&H[13][2], 0.f);              // f2
    hdr_put<COH>(&H[14][0], (float)(-sx * sz + cx * sy * cz)); hdr_put<COH>(&H[14][1], (float)(-cx * sy * sz - sx * cz)); hdr_put<COH>(&H[14][2], 0.f);              // f3
    }
    if (need_hessian == 2) {   // computeHessian in PCL's double form reads the double angle vectors (never inside the queue kernel: plain stores)
      double (*Jd)[3] = hdr->jang_d;
      Jd[0][0] = (-sx * sz + cx * sy * cz); Jd[0][1] = (-sx * cz - cx * sy * sz); Jd[0][2] = (-cx * cy);
      Jd[1][0] = (cx * sz + sx * sy * cz);  Jd[1][1] = (cx * cz - sx * sy * sz);  Jd[1][2] = (-sx * cy);
      Jd[2][0] = (-sy * cz);                Jd[2][1] = (sy * sz);                 Jd[2][2] = (cy);
      Jd[3][0] = (sx * cy * cz);            Jd[3][1] = (-sx * cy * sz);           Jd[3][2] = (sx * sy);
      Jd[4][0] = (-cx * cy * cz);           Jd[4][1] = (cx * cy * sz);            Jd[4][2] = (-cx * sy);
      Jd[5][0] = (-cy * sz);                Jd[5][1] = (-cy * cz);                Jd[5][2] = 0.0;
      Jd[6][0] = (cx * cz - sx * sy * sz);  Jd[6][1] = (-cx * sz - sx * sy * cz); Jd[6][2] = 0.0;
      Jd[7][0] = (sx * cz + cx * sy * sz);  Jd[7][1] = (cx * sy * cz - sx * sz);  Jd[7][2] = 0.0;
      double (*Hd)[3] = hdr->hang_d;
      Hd[0][0] = (-cx * sz - sx * sy * cz); Hd[0][1] = (-cx * cz + sx * sy * sz); Hd[0][2] = (sx * cy);
      Hd[1][0] = (-sx * sz + cx * sy * cz); Hd[1][1] = (-cx * sy * sz - sx * cz); Hd[1][2] = (-cx * cy);
      Hd[2][0] = (cx * cy * cz);            Hd[2][1] = (-cx * cy * sz);           Hd[2][2] = (cx * sy);
      Hd[3][0] = (sx * cy * cz);            Hd[3][1] = (-sx * cy * sz);           Hd[3][2] = (sx * sy);
      Hd[4][0] = (-sx * cz - cx * sy * sz); Hd[4][1] = (sx * sz - cx * sy * cz);  Hd[4][2] = 0.0;
      Hd[5][0] = (cx * cz - sx * sy * sz);  Hd[5][1] = (-sx * sy * cz - cx * sz); Hd[5][2] = 0.0;
      Hd[6][0] = (-cy * cz);                Hd[6][1] = (cy * sz);                 Hd[6][2] = (c.fix_hessian_d1 ? -sy : sy);
      Hd[7][0] = (-sx * sy * cz);           Hd[7][1] = (sx * sy * sz);            Hd[7][2] = (sx * cy);
      Hd[8][0] = (cx * sy * cz);            Hd[8][1] = (-cx * sy * sz);           Hd[8][2] = (-cx * cy);
      Hd[9][0] = (sy * sz);                 Hd[9][1] = (sy * cz);                 Hd[9][2] = 0.0;
      Hd[10][0] = (-sx * cy * sz);          Hd[10][1] = (-sx * cy * cz);          Hd[10][2] = 0.0;
      Hd[11][0] = (cx * cy * sz);           Hd[11][1] = (cx * cy * cz);           Hd[11][2] = 0.0;
      Hd[12][0] = (-cy * cz);               Hd[12][1] = (cy * sz);                Hd[12][2] = 0.0;
      Hd[13][0] = (-cx * sz - sx * sy * cz); Hd[13][1] = (-cx * cz + sx * sy * sz); Hd[13][2] = 0.0;
      Hd[14][0] = (-sx * sz + cx * sy * cz); Hd[14][1] = (-cx * sy * sz - sx * cz); Hd[14][2] = 0.0;
    }
    hdr_put_int<COH>(&hdr->need_hessian, need_hessian);
  }
#pragma unroll
  for (int k = 0; k < 6; k++) s.x_t[k] = x[k];
}

// ---- More-Thuente helpers (More & Thuente 1994; Sun & Yuan 2006 eq. 2.4.x) ---------------------------------
__device__ inline double mt_psi(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
__device__ inline double mt_dpsi(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

__device__ __forceinline__ double mt_trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return (fabs(a_c - a_l) < fabs(a_q - a_l)) ? a_c : 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return (fabs(a_c - a_t) >= fabs(a_s - a_t)) ? a_c : a_s;
  } else if (fabs(g_t) <= fabs(g_l)) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double a_n = (fabs(a_c - a_t) < fabs(a_s - a_t)) ? a_c : a_s;
    return (a_t > a_l) ? fmin(a_t + 0.66 * (a_u - a_t), a_n) : fmax(a_t + 0.66 * (a_u - a_t), a_n);
  }
  const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
  const double w = sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

__device__ __forceinline__ bool mt_update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t,
                                   double g_t) {
  if (f_t > f_l) {
    a_u = a_t; f_u = f_t; g_u = g_t;
    return false;
  } else if (g_t * (a_l - a_t) > 0) {
    a_l = a_t; f_l = f_t; g_l = g_t;
    return false;
  } else if (g_t * (a_l - a_t) < 0) {
    a_u = a_l; f_u = f_l; g_u = g_l;
    a_l = a_t; f_l = f_t; g_l = g_t;
    return false;
  }
  return true;
}

constexpr double kMu = 1.e-4, kNu = 0.9;

__device__ inline double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

// Starts one outer iteration from (score, grad, hess) at s.p.  Returns true when an evaluation was queued,
// false when the iteration finished without one (zero step) or the registration ended.
// STRICT: the upstream evaluation orders -- JacobiSVD(H).solve(-g) in the CPU checker's sequence of operations (Eigen's two-sided
// Jacobi across the wave, or rounds 2-3's one-sided Jacobi: dgs_params.ndt_newton_solver); otherwise the default order's Gauss-Jordan
// step.  A template argument, not a run-time test, so that the default order's fused kernel carries none of the SVD code.
// fast_solver (STRICT only): the direction from the Gauss-Jordan elimination instead -- a SPECULATED step (ndt_strict.h): the exact one follows
// beside the next launch and the closing behind it verifies; *fast_ok tells whether the elimination was well conditioned.
template <bool STRICT, bool SVD_REGS, bool COH = false>
__device__ __forceinline__ bool begin_iteration(NdtPair* st, NdtPair* hdr, NdtSolver& s, const NdtConsts& c, bool writer, const bool fast_solver = false, bool* fast_ok = nullptr) {
  double neg_g[6], delta[6], rc;
#pragma unroll
  for (int k = 0; k < 6; k++) neg_g[k] = -s.grad[k];
  if (STRICT && fast_solver) {
    gj_solve6_columns(s.hess, s.grad, delta, &rc);
    *fast_ok = rc > 1e-10;
    if (!*fast_ok) return false;
  } else if (STRICT) {
    if (c.newton_solver) jsvd_solve6_wave(s.hess, neg_g, delta);
    else if (SVD_REGS) svd_solve6_regs_dev(s.hess, neg_g, delta, 1e-17, 60);
    else svd_solve6_dev(s.hess, neg_g, delta, 1e-17, 60);
  } else {
    gj_solve6_columns(s.hess, s.grad, delta, &rc);
    if (!(rc > 1e-13)) svd_solve6_dev(s.hess, neg_g, delta);
  }
#ifdef DGS_CLOSE_STAMPS
  if (threadIdx.x == 0 && st->s.nr_iterations == 1) st->traj[kTrajCap - 2][0] = (double)wall_clock64();
#endif
  double norm = sqrt(dot6(delta, delta));
  if (norm == 0 || norm != norm) {
    s.converged = (norm == norm) ? 1 : 0;
    s.phase = PH_DONE;
    return false;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) s.dir[k] = delta[k] / norm;
  // computeStepLengthMT(p, dir, norm, step_size, eps / 2, ...)
  s.phi_0 = -s.score;
  s.d_phi_0 = -dot6(s.grad, s.dir);
  s.step_init = norm;
  if (s.d_phi_0 >= 0) {
    if (s.d_phi_0 == 0) {
      s.a_t = 0;  // "not a descent direction": zero step, no evaluation
      return false;
    }
    s.d_phi_0 = -s.d_phi_0;
#pragma unroll
    for (int k = 0; k < 6; k++) s.dir[k] = -s.dir[k];
  }
  const double step_max = c.step_size, step_min = c.trans_eps / 2;
  s.step_iterations = 0;
  s.trial_n = 0;
  s.trial_next = 0;
  s.a_l = 0; s.a_u = 0;
  s.f_l = mt_psi(0, s.phi_0, s.phi_0, s.d_phi_0, kMu);
  s.g_l = mt_dpsi(s.d_phi_0, s.d_phi_0, kMu);
  s.f_u = s.f_l;
  s.g_u = s.g_l;
  s.interval_converged = (c.line_search == DGS_NDT_LS_FIXED_STEP) ? ((step_max - step_min) > 0) : ((step_max - step_min) < 0);
  s.open_interval = 1;
  const double a_t = fmax(fmin(norm, step_max), step_min);
  s.a_t = a_t;
  double x[6];
#pragma unroll
  for (int k = 0; k < 6; k++) x[k] = s.p[k] + s.dir[k] * a_t;
#ifdef DGS_CLOSE_STAMPS
  if (threadIdx.x == 0 && st->s.nr_iterations == 1) st->traj[kTrajCap - 2][1] = (double)wall_clock64();
#endif
  write_evaluation<true, COH>(st, hdr, s, c, x, 1, true, writer);
#ifdef DGS_CLOSE_STAMPS
  if (threadIdx.x == 0 && st->s.nr_iterations == 1) st->traj[kTrajCap - 2][2] = (double)wall_clock64();
#endif
  s.phase = PH_MT_FIRST;
  return true;
}

// p += a_t * dir; convergence test of computeTransformation.  Returns true when the registration ended.
__device__ __forceinline__ bool end_iteration(NdtPair* st, NdtSolver& s, const NdtConsts& c, bool writer) {
  const double a = s.a_t;
#pragma unroll
  for (int k = 0; k < 6; k++) s.p[k] += s.dir[k] * a;
  if (writer && s.traj_len < kTrajCap) {
#pragma unroll
    for (int k = 0; k < 6; k++) st->traj[s.traj_len][k] = s.p[k];
  }
  s.traj_len++;
  bool conv = false;
  if (s.nr_iterations > c.max_iterations || (s.nr_iterations && (fabs(a) < c.trans_eps))) conv = true;
  s.nr_iterations++;
  if (conv) {
    s.converged = 1;
    s.phase = PH_DONE;
  }
  return conv;
}

__device__ inline bool mt_keep_going(const NdtSolver& s, const NdtConsts& c, double psi_t, double d_phi_t) {
  return !s.interval_converged && s.step_iterations < c.mt_max_step_iterations && !(psi_t <= 0 && d_phi_t <= -kNu * s.d_phi_0);
}

template <bool COH = false>
__device__ __forceinline__ void queue_trial(NdtPair* st, NdtPair* hdr, NdtSolver& s, const NdtConsts& c, double a_t, bool writer) {
  const double step_max = c.step_size, step_min = c.trans_eps / 2;
  a_t = fmax(fmin(a_t, step_max), step_min);
  s.a_t = a_t;
  double x[6];
#pragma unroll
  for (int k = 0; k < 6; k++) x[k] = s.p[k] + s.dir[k] * a_t;
  write_evaluation<true, COH>(st, hdr, s, c, x, 0, true, writer);
  s.phase = PH_MT_TRIAL;
}

// Consumes one evaluation result (already stored in s.score/grad/hess) and advances the state machine until
// the next evaluation is queued or the registration is finished.  Executed by all lanes of one wave in lock step.
// SVD_REGS: the stand-alone solve launch of the validation modes keeps the SVD workspace in registers (solve6.h)
// defer_solve (upstream order, ndt_strict.h): stop in front of the next iteration's Newton step (phase PH_SOLVE_PENDING); a later call
// with that phase -- from ndt_strict_solve_kernel -- continues there.
// speculate (upstream order, fused item-compacted kernel): take the next iteration's Newton step from the fast solver and tell the caller
// (*speculated) -- see NdtPair::spec_s.
template <bool SVD_REGS = false, bool COH = false, bool STRICT = false>
__device__ __forceinline__ void ndt_advance(NdtPair* st, NdtPair* hdr, NdtSolver& s, const NdtConsts& c, bool writer, bool defer_solve = false, bool speculate = false,
                                            bool* speculated = nullptr) {
  bool iteration_open = false;  // true: an iteration's line search has accepted its step, close it
  const bool resume = STRICT && s.phase == PH_SOLVE_PENDING;
  if (!resume) s.evaluations++;
#ifndef DGS_TRIAL_CACHE_ALL_ORDERS
#define DGS_TRIAL_CACHE_ALL_ORDERS 0
#endif
#ifndef DGS_AB_NO_TRIAL_CACHE
  if ((STRICT || DGS_TRIAL_CACHE_ALL_ORDERS) && (s.phase == PH_MT_FIRST || s.phase == PH_MT_TRIAL)) {
    // a trial point this line search has evaluated before takes the value it had then (NdtSolver::trial_x): same pose, same doubles, as on the CPU
    int hit = -1;
    for (int k = 0; k < s.trial_n; k++) {
      bool eq = true;
#pragma unroll
      for (int j = 0; j < 6; j++) eq = eq && (s.trial_x[k][j] == s.x_t[j]);
      if (eq && hit < 0) hit = k;
    }
    if (hit >= 0) {
      s.score = s.trial_score[hit];
#pragma unroll
      for (int j = 0; j < 6; j++) s.grad[j] = s.trial_grad[hit][j];
    } else {
      const int k = s.trial_next;
#pragma unroll
      for (int j = 0; j < 6; j++) { s.trial_x[k][j] = s.x_t[j]; s.trial_grad[k][j] = s.grad[j]; }
      s.trial_score[k] = s.score;
      s.trial_next = (k + 1) % NdtSolver::kTrialCache;
      if (s.trial_n < NdtSolver::kTrialCache) s.trial_n++;
    }
  }
#endif
  switch (resume ? PH_INIT_EVAL : s.phase) {
    case PH_PROBE:
      s.phase = PH_DONE;
      return;
    case PH_INIT_EVAL:
      break;
    case PH_MT_FIRST:
    case PH_MT_TRIAL: {
      const double phi_t = -s.score;
      const double d_phi_t = -dot6(s.grad, s.dir);
      const double psi_t = mt_psi(s.a_t, phi_t, s.phi_0, s.d_phi_0, kMu);
      const double d_psi_t = mt_dpsi(d_phi_t, s.d_phi_0, kMu);
      if (s.phase == PH_MT_TRIAL) {
        if (s.open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
          s.open_interval = 0;
          s.f_l = s.f_l + s.phi_0 - kMu * s.d_phi_0 * s.a_l;
          s.g_l = s.g_l + kMu * s.d_phi_0;
          s.f_u = s.f_u + s.phi_0 - kMu * s.d_phi_0 * s.a_u;
          s.g_u = s.g_u + kMu * s.d_phi_0;
        }
        if (s.open_interval)
          s.interval_converged = mt_update_interval(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, psi_t, d_psi_t);
        else
          s.interval_converged = mt_update_interval(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, phi_t, d_phi_t);
        s.step_iterations++;
      }
      if (mt_keep_going(s, c, psi_t, d_phi_t)) {
        const double a_n = s.open_interval ? mt_trial_value(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, psi_t, d_psi_t)
                                           : mt_trial_value(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, phi_t, d_phi_t);
        queue_trial<COH>(st, hdr, s, c, a_n, writer);
        return;
      }
      if (s.step_iterations) {  // computeHessian at the accepted point
        double x[6];
#pragma unroll
        for (int k = 0; k < 6; k++) x[k] = s.x_t[k];
        write_evaluation<true, COH>(st, hdr, s, c, x, (STRICT && c.hessian_double) ? 2 : 1, false, writer);
        s.phase = PH_MT_HESSIAN;
        return;
      }
      iteration_open = true;
    } break;
    case PH_MT_HESSIAN:
      iteration_open = true;
      break;
    default:
      return;
  }
  for (int guard = 0; guard < 4096; guard++) {
    if (iteration_open) {
      if (end_iteration(st, s, c, writer)) return;
    }
    if (STRICT && defer_solve) {   // the Newton step goes to the solve kernel
      s.phase = PH_SOLVE_PENDING;
      return;
    }
    if (STRICT && speculate) {
      bool ok = false;
      const int ph0 = s.phase, cv0 = s.converged;
      if (begin_iteration<STRICT, SVD_REGS, COH>(st, hdr, s, c, writer, true, &ok)) {   // evaluation queued from the speculated direction
        *speculated = true;
        return;
      }
      s.phase = ph0;
      s.converged = cv0;
      // ill-conditioned, or the fast step ends / skips the iteration: nothing was published that the exact step below does not overwrite
      // (its inputs -- p, score, gradient, Hessian -- are untouched)
    }
    if (begin_iteration<STRICT, SVD_REGS, COH>(st, hdr, s, c, writer)) return;  // evaluation queued
    if (s.phase == PH_DONE) return;
    iteration_open = true;                          // zero-step iteration: close it and try again
  }
  s.converged = 0;
  s.phase = PH_DONE;
}

#include "ndt_strict.h"   // ndt_strict_order 1: the upstream-order evaluation as a fused launch (uses ndt_advance above)

// Round 2's validation kernel, kept for ndt_strict_order 2 (ROWS = true; order 1 runs ndt_strict_kernel, ndt_strict.h).  ROWS = false:
// per-thread double totals over a strided set of points, block sums in a fixed order, one 48-double row per workgroup.  ROWS = true: the 43 per-point totals go to HBM, column-major per pair
// ([43][max_n]), for the sequential index-order sum of ndt_strict_seqsum_kernel.
template <int SEARCH, bool ROWS, bool LITERAL>
__global__ __launch_bounds__(kBlock, 2) void ndt_derivatives_strict_kernel(const float4* const* __restrict__ src_ptrs, const int* __restrict__ src_sizes,
                                                                        const NdtPair* __restrict__ pairs, const VoxelGrid g,
                                                                        const double* __restrict__ vtab, const double gauss_d1, const float gd2,
                                                                        const int leaf_pow2, double* __restrict__ partials, double* __restrict__ rows,
                                                                        const int max_n, const int n_pairs, const int cap_blocks,
                                                                        int* __restrict__ pair_blocks, const double gauss_d2, const size_t rows_pair_stride, const int exp_libm) {
  int pair, slice, blocks_per_pair;
  if (!deal_workgroup(n_pairs, cap_blocks, [&](int pi) { return pairs[pi].active != 0; }, pair, slice, blocks_per_pair)) return;
  if (slice == 0 && threadIdx.x == 0) pair_blocks[pair] = blocks_per_pair;
  const NdtPair& st = pairs[pair];
  const float4* __restrict__ src = src_ptrs[pair];
  const int n = src_sizes[pair];
  const bool need_h = st.need_hessian != 0;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = st.T[k];
  double acc[kStrictAccum];
#pragma unroll
  for (int k = 0; k < kStrictAccum; k++) acc[k] = 0.0;
  const int ncol = need_h ? kStrictAccum : 7;
  if (ROWS && st.need_hessian == 2) {
    // computeHessian in PCL's double form (evaluation kind 2): every (point, voxel) term to HBM, entry-major [36][n * NB], for the
    // sequential sum in upstream's order (ndt_strict_seqsum_kernel)
    constexpr int NB = Offsets<SEARCH>::N;
    const size_t row_stride = (size_t)max_n * NB;
    for (int i = slice * kBlock + threadIdx.x; i < n; i += blocks_per_pair * kBlock) {
      const float4 x = src[i];
      float xt[3];
      xt[0] = affine_row_rn(T[0], T[1], T[2], T[3], x.x, x.y, x.z);
      xt[1] = affine_row_rn(T[4], T[5], T[6], T[7], x.x, x.y, x.z);
      xt[2] = affine_row_rn(T[8], T[9], T[10], T[11], x.x, x.y, x.z);
      int vids[NB];
      const unsigned mask = strict_neighbourhood<SEARCH>(xt, g, leaf_pow2, vids);
      strict_point_hd<SEARCH, true>(x, xt, vids, mask, st, vtab, gauss_d1, gauss_d2, acc, rows + (size_t)pair * rows_pair_stride + (size_t)i * NB, row_stride, exp_libm ? kGlibcExpTab : nullptr);
    }
    return;
  }
  for (int i = slice * kBlock + threadIdx.x; i < n; i += blocks_per_pair * kBlock) {
    double o[kStrictAccum];
    ndt_point_strict<SEARCH, LITERAL>(src[i], T, st, g, vtab, gauss_d1, gd2, leaf_pow2, need_h, o, exp_libm != 0);
    if (ROWS) {
      double* __restrict__ col = rows + (size_t)pair * rows_pair_stride + i;
      for (int k = 0; k < ncol; k++) col[(size_t)k * max_n] = o[k];
    } else {
#pragma unroll
      for (int k = 0; k < kStrictAccum; k++) acc[k] += o[k];
    }
  }
  if (ROWS) return;
  __shared__ double sm[kBlock / kWave][kStrictPad];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kStrictAccum; k++) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) sm[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kStrictPad) {
    double v = 0.0;
    if (threadIdx.x < kStrictAccum) v = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    partials[((size_t)pair * cap_blocks + slice) * kStrictPad + threadIdx.x] = v;
  }
}

// ndt_strict_order 2: upstream's final loop -- score / gradient / Hessian entries summed over the points in index order, one
// lane per entry (a dependent chain of n double additions: this mode exists to prove bit-parity, not to be fast)
__global__ __launch_bounds__(kWave) void ndt_strict_seqsum_kernel(const NdtPair* __restrict__ pairs, const int* __restrict__ src_sizes,
                                                                  const double* __restrict__ rows, const int max_n, double* __restrict__ totals,
                                                                  const size_t rows_pair_stride, const int nb_slots) {
  const int pair = blockIdx.x;
  const NdtPair& st = pairs[pair];
  if (!st.active) return;
  const int c = threadIdx.x;
  if (c >= kStrictPad) return;
  if (st.need_hessian == 2) {
    // computeHessian (kind 2): the Hessian entries alone, every (point, voxel slot) term in upstream's order
    double v = 0.0;
    if (c >= 7 && c < kStrictAccum) {
      const size_t n = (size_t)src_sizes[pair] * nb_slots;
      const double* __restrict__ col = rows + (size_t)pair * rows_pair_stride + (size_t)(c - 7) * ((size_t)max_n * nb_slots);
      size_t i = 0;
      for (; i + 8 <= n; i += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = col[i + u];
#pragma unroll
        for (int u = 0; u < 8; u++) v += t[u];
      }
      for (; i < n; i++) v += col[i];
    }
    totals[(size_t)pair * kStrictPad + c] = v;
    return;
  }
  const int ncol = st.need_hessian ? kStrictAccum : 7;
  double v = 0.0;
  if (c < ncol) {
    const int n = src_sizes[pair];
    const double* __restrict__ col = rows + (size_t)pair * rows_pair_stride + (size_t)c * max_n;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = col[i + u];
#pragma unroll
      for (int u = 0; u < 8; u++) v += t[u];
    }
    for (; i < n; i++) v += col[i];
  }
  totals[(size_t)pair * kStrictPad + c] = v;
}

// Sums a pair's partial rows in slice order and advances its optimiser by one evaluation; executed by one whole workgroup.
// launch >= 0: fused launches (the pair leaves through last_launch); launch < 0: ndt_solve_kernel (the pair leaves through active).
#ifdef DGS_CLOSE_STAMPS   // diagnostic build only (make dbg): 100 MHz wall-clock stamps of the closing phases into the pair's last trajectory rows
#define CLOSE_STAMP(k) if (threadIdx.x == 0 && st->s.nr_iterations == 1) st->traj[kTrajCap - 1][k] = (double)wall_clock64();
#else
#define CLOSE_STAMP(k)
#endif
// QUEUE: called inside the persistent queue kernel -- the pair's record was written by another workgroup of the SAME launch and will be
// read by others: coherent (agent-scope) loads and write-through stores for every word of it.  Returns (to the closing wave) whether
// the registration has ended.
// DONE_FLAG: `done_counter` is this pair's own flag in HOST memory (pinned, device-visible): a finished pair stores launch + 1 into it and the host
// counts the flags at every chunk boundary -- no copy command between the chunks of launches (each cost the stream ~8 us: a blit kernel
// and two barriers).  Otherwise a device counter that the host copies back.
template <bool QUEUE, bool DONE_FLAG>
__device__ __forceinline__ bool ndt_close_evaluation(NdtPair* st, const double* partials_of_pair, int blocks_per_pair, const NdtConsts& c, int* done_counter, int launch,
                                                     NdtPair* hdr_next, int need_h_in) {
  CLOSE_STAMP(0)
  __shared__ NdtSolver s_lds;   // the optimiser state lives in LDS: a register copy costs ~150 VGPRs
  NdtSolver& s = s_lds;
  // Everything read here comes from memory (the state from the previous launch, the rows from this one): issue it all at once --
  // the state word by word across the workgroup (one 8-byte load per thread instead of 38 dependent 16-byte loads in every lane
  // of one wave, which took 5 us of the 7 us this function used to take), the rows as before -- and pay ONE memory latency.
  static_assert(sizeof(NdtSolver) % 8 == 0 && sizeof(NdtSolver) / 8 <= kBlock, "state words");
  constexpr int kWords = (int)(sizeof(NdtSolver) / 8);
  double word = 0.0;
  if (threadIdx.x < kWords) word = QUEUE ? __hip_atomic_load(reinterpret_cast<const double*>(&st->s) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                         : reinterpret_cast<const double*>(&st->s)[threadIdx.x];
  const int need_h = need_h_in >= 0 ? need_h_in : st->need_hessian;   // queue kernel: the flag of the round being closed comes from its record slot
  // ---- finish the reduction: 8 strided groups x 32 columns, fixed order
  __shared__ double tot[kAccumPad];
  __shared__ double sm[kBlock / kAccumPad][kAccumPad];
  const int col = threadIdx.x % kAccumPad, grp = threadIdx.x / kAccumPad;
  constexpr int G = kBlock / kAccumPad;
  double v = 0.0;
  // four rows in flight per thread, added in slice order (a missing row adds +0.0, which changes nothing)
  for (int b0 = grp; b0 < blocks_per_pair; b0 += 4 * G) {
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int b = b0 + k * G;
      const double* ptr = partials_of_pair + (size_t)min(b, blocks_per_pair - 1) * kAccumPad + col;
      // rows published inside this launch: agent-scope (sc1) loads, never a line this CU may hold from an earlier launch
      const double x = (launch >= 0) ? handoff_load_row(ptr) : *ptr;
      r[k] = (b < blocks_per_pair) ? x : 0.0;
    }
    v = (((v + r[0]) + r[1]) + r[2]) + r[3];
  }
  sm[grp][col] = v;
  if (threadIdx.x < kWords) reinterpret_cast<double*>(&s_lds)[threadIdx.x] = word;
  __syncthreads();
  if (threadIdx.x < kAccumPad) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < G; k++) t += sm[k][threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  CLOSE_STAMP(1)
  if (threadIdx.x >= kWave) return false;
  // ---- one wave advances the optimiser: every lane computes the same values, lane 0 writes the pair's record
  const bool writer = threadIdx.x == 0;
  // the totals into the optimiser state, one entry per lane: lanes 0..35 the symmetric Hessian (entry (i, j) <- upper-triangle slot of
  // (min, max)), 36..41 the gradient, 42 the score (all 64 lanes storing all 49 entries one after the other cost 0.7 us)
  {
    const int t = threadIdx.x;
    if (t < 36) {
      if (need_h) {
        const int i = t / 6, j = t % 6, lo = min(i, j), hi = max(i, j);
        s.hess[t] = tot[7 + lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
      }
    } else if (t < 42) {
      s.grad[t - 36] = tot[1 + t - 36];
    } else if (t == 42) {
      s.score = tot[0];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's LDS stores have landed
  }
  CLOSE_STAMP(2)
  ndt_advance<false, QUEUE>(st, hdr_next ? hdr_next : st, s, c, writer);
  CLOSE_STAMP(3)
  // write the state back word by word across the wave (lane 0 alone would issue 38 stores one after the other)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's LDS stores have landed
  for (int w = threadIdx.x; w < kWords; w += kWave) {
    const double v = reinterpret_cast<const double*>(&s_lds)[w];
    if (QUEUE) __hip_atomic_store(reinterpret_cast<double*>(&st->s) + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else reinterpret_cast<double*>(&st->s)[w] = v;
  }
  if (writer && s.phase == PH_DONE) {
    st->active = 0;
    if (launch >= 0) st->last_launch = launch;
    if (DONE_FLAG) __hip_atomic_store(done_counter, launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // which launch ended it (ndt_align_pairs: early fitness)
    else atomicAdd(done_counter, 1);
  }
  CLOSE_STAMP(4)
  return s.phase == PH_DONE;
}

__global__ __launch_bounds__(kBlock) void ndt_solve_kernel(NdtPair* __restrict__ pairs, const double* __restrict__ partials, const int cap_blocks,
                                                           int* __restrict__ pair_blocks, const NdtConsts c, int* __restrict__ done_counter,
                                                           const double* __restrict__ strict_totals, const int strict_from_rows) {
  const int pair = blockIdx.x;
  NdtPair* st = pairs + pair;
  if (!st->active) return;
  if (!strict_totals) {
    ndt_close_evaluation(st, partials + (size_t)pair * cap_blocks * kAccumPad, pair_blocks[pair], c, done_counter, -1);
    return;
  }
  // validation modes: the sums of this evaluation were formed by ndt_strict_kernel's rows / ndt_strict_seqsum.  A pair that the
  // derivative launch in front of this one did not evaluate (the launch served the other evaluation kind, ndt_strict.h) has no rows.
  if (strict_from_rows) {
    if (pair_blocks[pair] == 0) return;
  }
  __shared__ NdtSolver s_lds;
  NdtSolver& s = s_lds;
  __shared__ double tot[kStrictPad];
  int need_h = 0;
  if (threadIdx.x < kWave) {
    s_lds = st->s;
    need_h = st->need_hessian;
  }
  if (threadIdx.x < kStrictPad) {
    if (strict_from_rows) {
      // order 1: the workgroups' rows added in slice order (what ndt_strict_reduce_kernel did as a launch of its own), four loads in
      // flight, the additions in the same sequence
      const int nb = pair_blocks[pair];
      const double* base = partials + (size_t)pair * cap_blocks * kStrictPad + threadIdx.x;
      double v = 0.0;
      for (int b0 = 0; b0 < nb; b0 += 4) {
        double r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = base[(size_t)min(b0 + k, nb - 1) * kStrictPad];
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (b0 + k < nb) v += r[k];
      }
      tot[threadIdx.x] = v;
    } else {
      tot[threadIdx.x] = strict_totals[(size_t)pair * kStrictPad + threadIdx.x];
    }
  }
  __syncthreads();
  if (threadIdx.x >= kWave) return;
  const bool writer = threadIdx.x == 0;
  if (need_h != 2) {   // kind 2 (computeHessian alone) leaves score and gradient as the last trial left them
    s.score = tot[0];
#pragma unroll
    for (int k = 0; k < 6; k++) s.grad[k] = tot[1 + k];
  }
  if (need_h) {
#pragma unroll
    for (int k = 0; k < 36; k++) s.hess[k] = tot[7 + k];  // upstream's full 6x6 (not exactly symmetric in float)
  }
  ndt_advance<true, false, true>(st, st, s, c, writer);
  if (writer) {
    if (strict_from_rows) pair_blocks[pair] = 0;   // consumed
    st->s = s;
    if (s.phase == PH_DONE) {
      st->active = 0;
      atomicAdd(done_counter, 1);
    }
  }
}

// ================================================================================================ the queue kernel
// STATUS (round 3): correct -- bit for bit equal to the launch-per-evaluation path under the same slice schedule
// (tests/test_queue_gpu.py) -- and SLOWER: 3.9-4.9 ms per 32-candidate step against 1.5 ms (DESIGN.md has the phase breakdown).  Off by
// default and compiled into the experiments build only.
// ONE persistent launch per align instead of one launch per evaluation.  A launch per evaluation pays, at every kernel boundary, the
// whole serial tail of the slowest pair -- row hand-off, Newton step, More-Thuente state machine, trig of the next transform, the
// dependent loads of the next prologue: t = 10.6 us + 1.49 us x (pairs still iterating) per launch on the 32-candidate bench step,
// 38 launches, i.e. 0.4 ms of 1.44 ms spent with the chip waiting for 32 single waves (profiles/r03/tail_table_lockstep.json).  Here the pairs
// advance independently: a work item is (pair, round, slice); the workgroup that closes round r of a pair opens its round r + 1, and
// every other workgroup meanwhile works on the other pairs' slices -- the serial tail of one pair hides behind the derivative work of
// the rest.  Workgroups are workers that CLAIM items (no worker ever waits for a particular other worker, so the kernel cannot
// deadlock on workgroups that are not resident), pairs that finish stop offering items, the stragglers' rounds are cut into more
// slices and get the whole chip, and a worker leaves when no pair is iterating any more.
//   queue word of a pair (64 bits, own 64-byte line): [63:44] round | [43:32] slices of this round | [31:0] slices claimed.
//   claim = one agent-scope atomic add of 1; the returned word tells round, slice count and the claimed slice at once.
//   The slice count of a round is a fixed function of (batch shape, round number) -- never of timing -- so the partition of the
//   sums, and with it every result bit, is reproducible run to run (ndt_queue_slices; the launch-per-evaluation path can be run with
//   the same schedule for bit-for-bit comparison: DGS_NDT_QUEUE=0 DGS_NDT_SCHEDULE=1).
//   Coherence inside the launch is per access, as for the rows (common.h): the closing workgroup writes the pair's record through
//   (agent-scope stores), drains, and only then publishes the next round's queue word; workers read queue words, the record and the
//   rows with agent-scope loads.  A worker whose poll guard runs out raises `abort` and everybody leaves (the align reports an error)
//   instead of hanging the device.
__host__ __device__ inline int ndt_queue_slices(int round, int base, int cap) {
  const int f = round < 12 ? 1 : (round < 24 ? 2 : 4);
  return min(base * f, cap);
}
constexpr int kQueueHdrWords = 82;                    // NdtPair: T[12], jang[8][3], hang[15][3], need_hessian
static_assert(offsetof(NdtPair, need_hessian) == 81 * 4, "header layout");
// What a round's derivative pass reads of the pair's record (transform, angle tables, need_hessian: the first 82 words of NdtPair) has
// ONE SLOT PER ROUND, each in its own 128-byte lines: the closing workgroup of round r writes slot r + 1 through to memory before it
// publishes round r + 1, and no cache of any XCD can hold an older copy of an address that nobody has read in this launch yet -- so the
// workers read a slot with ordinary scalar loads, exactly like the launch-per-evaluation kernel reads the record after a kernel
// boundary (keeping the 81 table entries in scalar registers read back through v_readlane cost 58 more instructions per point).
constexpr size_t kQueueSlotBytes = 384;
__host__ __device__ inline NdtPair* queue_slot(char* ring, int ring_rounds, int pair, int round) {
  return reinterpret_cast<NdtPair*>(ring + ((size_t)pair * ring_rounds + round) * kQueueSlotBytes);
}
// Queue memory: 64-bit words, contiguous -- word 0 = control ([31:0] pairs still iterating, bit 32 abort), word 1 + p = pair p.  A worker
// looking for work reads control and up to 63 pairs with ONE wave-wide load of four 128-byte lines (one line per pair made every
// idle poll 34 line requests to the same memory channel: measured 110 polls per us by 768 workers, items three times slower).
__device__ __forceinline__ unsigned long long* queue_word(int* queue, int pair) { return reinterpret_cast<unsigned long long*>(queue) + 1 + pair; }
__device__ __forceinline__ unsigned long long* queue_ctl(int* queue) { return reinterpret_cast<unsigned long long*>(queue); }
constexpr int kQueueStatInts = 16;   // diagnostic build: counters behind the words

#ifdef DGS_EXPERIMENTS   // measured slower than one launch per evaluation (see the status note above): experiments build only
constexpr unsigned long long kQueueClosed = 0xFFFFFull << 44;   // round = all ones, no slices: the pair has finished
constexpr unsigned long long kQueueAbort = 1ull << 32;
template <int SEARCH>
__global__ __launch_bounds__(kBlock, 4) void ndt_queue_kernel(const float4* const* __restrict__ src_ptrs, const int* __restrict__ src_sizes, NdtPair* pairs,
                                                              const VoxelGrid g, const double gd1, const float gd2, const int leaf_pow2,
                                                              double* partials, const int n_pairs, const int cap_blocks, const int slices_base,
                                                              const NdtConsts consts, int* queue, int* __restrict__ done_counter, char* ring,
                                                              const int ring_rounds) {
  __shared__ unsigned long long s_item;
  __shared__ int s_pair;
  __shared__ int s_last;
  const int lane = threadIdx.x & 63;
  unsigned polls = 0;
#ifdef DGS_QUEUE_STATS
  unsigned acc_polls = 0, acc_failed = 0, acc_claims = 0, acc_closings = 0;
  unsigned long long acc_look = 0, acc_item = 0, acc_rec = 0, acc_loop = 0, acc_row = 0, acc_ticket = 0, acc_close = 0;
#define QSTAMP(var) const unsigned long long var = wall_clock64();
#else
#define QSTAMP(var)
#endif
  for (;;) {
    // ---- claim an item (wave 0; every lane holds the same values, lane 0 does the atomics)
    if (threadIdx.x < kWave) {
      int pair = -1;
      unsigned long long item = 0;
#ifdef DGS_QUEUE_STATS
      const unsigned long long t_claim0 = wall_clock64();
      unsigned st_polls = 0, st_failed = 0;
#endif
      unsigned idle = 0;
      for (;;) {
#ifdef DGS_QUEUE_STATS
        st_polls++;
#endif
        int left = 0, abort = 0;
        for (int c0 = 0; c0 <= n_pairs && pair < 0; c0 += 64) {
          const int idx = c0 + lane;   // queue word index: 0 = control, 1 + p = pair p
          unsigned long long w = 0;
          if (idx <= n_pairs) w = __hip_atomic_load(queue_ctl(queue) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (c0 == 0) {
            left = __builtin_amdgcn_readfirstlane((int)(unsigned)w);
            abort = __builtin_amdgcn_readfirstlane((int)(unsigned)(w >> 32)) & 1;
          }
          unsigned long long m = __ballot(idx >= 1 && idx <= n_pairs && (unsigned)w < (unsigned)((w >> 32) & 0xFFFull));   // pairs with unclaimed slices
          if (m == 0ull) continue;
          // ONE attempt per look, at a pair that depends on the worker (so that the workers spread over the pairs); a worker that loses
          // the race looks again instead of walking down a stale list (which is what turns a few late workers into a herd)
          const int rot = (int)((blockIdx.x * 11u + polls + idle) & 63u);
          m = (m >> rot) | (rot ? (m << (64 - rot)) : 0ull);
          const int cand = c0 + ((__ffsll((long long)m) - 1 + rot) & 63) - 1;
          unsigned lo = 0, hi = 0;
          if (lane == 0) {
            const unsigned long long old = __hip_atomic_fetch_add(queue_word(queue, cand), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo = (unsigned)old; hi = (unsigned)(old >> 32);
          }
          lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
          if (lo < (hi & 0xFFFu)) { pair = cand; item = ((unsigned long long)hi << 32) | lo; }
#ifdef DGS_QUEUE_STATS
          else st_failed++;
#endif
          c0 = n_pairs + 1;   // leave the scan: claimed, or look again
          idle = 0;
        }
        if (pair >= 0) break;
        if (left <= 0 || abort != 0) break;
        // a worker that can never be needed again leaves: at most cap_blocks slices per pair still iterating can ever be on offer
        if ((long long)blockIdx.x >= (long long)left * cap_blocks) break;
        if (++polls > (1u << 22)) {   // seconds of polling without finding work: something is wrong -- leave, all of us
          if (lane == 0) __hip_atomic_fetch_or(queue_ctl(queue), kQueueAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        // back off: the longer nothing was on offer, the longer the nap (0.5 us ... 8 us); whoever just lost a race looks again at once
        idle = min(idle + 1u, 5u);
        for (unsigned k = 0; k < (1u << (idle - 1u)); k++) __builtin_amdgcn_s_sleep(16);
      }
      if (lane == 0) { s_pair = pair; s_item = item; }
#ifdef DGS_QUEUE_STATS
      acc_polls += st_polls; acc_failed += st_failed; acc_claims += pair >= 0 ? 1 : 0; acc_look += wall_clock64() - t_claim0;
#endif
    }
    __syncthreads();
    const int pair = s_pair;
    if (pair < 0) {
#ifdef DGS_QUEUE_STATS
      if (threadIdx.x == 0) {   // diagnostic build: this worker's counters (100 MHz ticks), flushed once
        int* stat = queue + 2 * (n_pairs + 2);
        atomicAdd(&stat[2], (int)acc_polls); atomicAdd(&stat[3], (int)acc_failed); atomicAdd(&stat[4], (int)acc_claims); atomicAdd(&stat[7], (int)acc_closings);
        atomicAdd(&stat[5], (int)acc_look); atomicAdd(&stat[6], (int)acc_item);
        atomicAdd(&stat[8], (int)acc_rec); atomicAdd(&stat[9], (int)acc_loop); atomicAdd(&stat[10], (int)acc_row); atomicAdd(&stat[11], (int)acc_ticket); atomicAdd(&stat[12], (int)acc_close);
      }
#endif
      return;
    }
    QSTAMP(t_item0)
    const unsigned long long item = s_item;
    const int slice = (int)(unsigned)item, n_slices = (int)((item >> 32) & 0xFFFull), round = (int)(item >> 44);
    // ---- the round's record slot: scalar loads (see kQueueSlotBytes)
    const NdtPair& rec = *queue_slot(ring, ring_rounds, pair, round);
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = rec.T[k];
    const int need_h_word = rec.need_hessian;
    const bool need_h = need_h_word != 0;
    const float4* __restrict__ src = src_ptrs[pair];
    const int n = src_sizes[pair];
    double acc[kAccum];
#pragma unroll
    for (int k = 0; k < kAccum; k++) acc[k] = 0.0;
#ifdef DGS_QUEUE_STATS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    QSTAMP(t_rec)
    ndt_point_loop<SEARCH>(T, NdtHdrGlobal{rec}, need_h, src, n, slice * kBlock + (int)threadIdx.x, n_slices * kBlock, g, gd1, gd2, leaf_pow2, acc);
    QSTAMP(t_loop)
    ndt_block_row<true>(acc, partials + ((size_t)pair * cap_blocks + slice) * kAccumPad);
    // ---- publish the row, take a ticket; the workgroup that takes the round's last ticket closes it and opens the pair's next round
    if (threadIdx.x < kAccumPad) handoff_drain_stores();
    __syncthreads();
    QSTAMP(t_row)
    if (threadIdx.x == 0) s_last = handoff_take_ticket(&pairs[pair].ticket, n_slices) ? 1 : 0;
    __syncthreads();
    QSTAMP(t_ticket)
    if (s_last) {
      // the next round's slot starts as a copy of this round's transform (an evaluation that only adds the Hessian at the accepted point
      // keeps it); the optimiser then writes what changes.  Both through to memory, in this order.
      // The closing wave is ONE wave on a SIMD that it shares with the derivative loops of other workers: at equal priority its serial
      // chain (row sums, Newton step, line-search state machine, trig) runs at a third of its speed (measured 33 us against 8 us at the end
      // of a lockstep launch, where the SIMD is idle) -- and the pair's next round cannot open before it is through.  Raise it.
      __builtin_amdgcn_s_setprio(3);
      NdtPair* next = queue_slot(ring, ring_rounds, pair, min(round + 1, ring_rounds - 1));
      if (threadIdx.x < 12) {
        __hip_atomic_store(&next->T[threadIdx.x], T[threadIdx.x < 12 ? threadIdx.x : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      bool done = ndt_close_evaluation<true>(pairs + pair, partials + (size_t)pair * cap_blocks * kAccumPad, n_slices, consts, done_counter, 0x7FFFFFF0, next, need_h_word);
      if (round + 2 >= ring_rounds) done = true;   // cannot happen: the optimiser ends a registration long before its slots run out
      if (threadIdx.x < kWave) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the closing wave's write-through stores of the record have landed
        if (threadIdx.x == 0) {
          if (done) {
            __hip_atomic_store(queue_word(queue, pair), kQueueClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(queue_ctl(queue), ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one pair fewer (the count is > 0: no borrow into the abort bit)
          } else {
            const unsigned long long next = ((unsigned long long)(round + 1) << 44) | ((unsigned long long)ndt_queue_slices(round + 1, slices_base, cap_blocks) << 32);
            __hip_atomic_store(queue_word(queue, pair), next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
#ifdef DGS_QUEUE_STATS
    {
      const unsigned long long t_end = wall_clock64();
      acc_item += t_end - t_item0; acc_rec += t_rec - t_item0; acc_loop += t_loop - t_rec; acc_row += t_row - t_loop; acc_ticket += t_ticket - t_row;
      if (s_last) { acc_close += t_end - t_ticket; acc_closings++; }
    }
#endif
    __syncthreads();   // LDS (item, record, reduction buffers) is re-used by the next item
  }
}
#endif  // DGS_EXPERIMENTS

// ================================================================================================ init / export
__global__ void ndt_init_kernel(NdtPair* __restrict__ pairs, const NdtInit* __restrict__ inits, int n_pairs, const NdtConsts c, int probe,
                                int* __restrict__ done_counter, const float4* const* __restrict__ stage_ptrs, const int* __restrict__ stage_sizes,
                                const float4** __restrict__ src_ptrs, int* __restrict__ src_sizes, int* __restrict__ queue, const int queue_slices0,
                                char* __restrict__ ring, const int ring_rounds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 16) done_counter[i] = 0;   // the first block has 64 threads: the counter block is cleared here, not by a fill command
  if (queue && i == 0) *queue_ctl(queue) = (unsigned long long)n_pairs;   // queue kernel: pairs still iterating, abort bit clear
  if (queue && i < kQueueStatInts) queue[2 * (n_pairs + 2) + i] = 0;
  if (i >= n_pairs) return;
  if (queue) *queue_word(queue, i) = (unsigned long long)queue_slices0 << 32;   // round 0, nothing claimed
  src_ptrs[i] = stage_ptrs[i];
  src_sizes[i] = stage_sizes[i];
  NdtPair* st = pairs + i;
  const NdtInit& in = inits[i];
  NdtSolver s;
  s.phase = probe ? PH_PROBE : PH_INIT_EVAL;
  s.nr_iterations = 0;
  s.trial_n = 0;
  s.trial_next = 0;
  s.evaluations = 0;
  s.converged = 0;
  s.step_iterations = 0;
  s.interval_converged = 0;
  s.open_interval = 1;
  s.traj_len = 1;
  s.score = 0;
  s.phi_0 = s.d_phi_0 = s.a_t = s.a_l = s.f_l = s.g_l = s.a_u = s.f_u = s.g_u = s.step_init = 0;
  for (int k = 0; k < 6; k++) { s.p[k] = in.p0[k]; s.grad[k] = 0; s.dir[k] = 0; st->traj[0][k] = in.p0[k]; }
  for (int k = 0; k < 36; k++) s.hess[k] = 0;
  double x[6];
  for (int k = 0; k < 6; k++) x[k] = in.p0[k];
  write_evaluation<false>(st, st, s, c, x, probe == 2 ? 2 : 1, false, true);   // probe 2: the test hook of the double-precision computeHessian pass
  // the first evaluation transforms the cloud by the GUESS matrix itself (computeTransformation)
  const float* G = in.guess;
  st->T[0] = G[0]; st->T[1] = G[4]; st->T[2] = G[8];  st->T[3] = G[12];
  st->T[4] = G[1]; st->T[5] = G[5]; st->T[6] = G[9];  st->T[7] = G[13];
  st->T[8] = G[2]; st->T[9] = G[6]; st->T[10] = G[10]; st->T[11] = G[14];
  for (int k = 0; k < 16; k++) st->final_T[k] = G[k];
  st->s = s;
  st->active = 1;
  st->last_launch = 0x7FFFFFFF;
  st->ticket = 0;
  st->serve[0] = 0;    // upstream order, fused: the first evaluation (kind 1) is served by round 0's first kernel
  st->serve[1] = -1;
  st->serve[2] = -1;
  st->serve[3] = 0;
  st->spec_pending = 0;
  st->spec_result = 0;
  if (queue) {   // queue kernel: the record slot of round 0
    const int* from = reinterpret_cast<const int*>(st);
    int* to = reinterpret_cast<int*>(queue_slot(ring, ring_rounds, i, 0));
    for (int k = 0; k < kQueueHdrWords; k++) to[k] = from[k];
  }
}

struct NdtOut {
  float T[16];
  int converged, iterations, evaluations, pad;
  double score;
};

__global__ void ndt_export_kernel(const NdtPair* __restrict__ pairs, int n_pairs, NdtOut* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const NdtPair& st = pairs[i];
  NdtOut o;
  for (int k = 0; k < 16; k++) o.T[k] = st.final_T[k];
  o.converged = (st.s.phase == PH_DONE) ? st.s.converged : 0;
  o.iterations = st.s.nr_iterations;
  o.evaluations = st.s.evaluations;
  o.pad = 0;
  o.score = st.s.score;
  out[i] = o;
}

// ================================================================================================ host side
// Eigen 3.3 Matrix3f::eulerAngles(0,1,2) on the rotation block of a column-major float 4x4
// (computeTransformation: "convert initial guess matrix to 6 element transformation vector").
static void euler_angles_012(const float* T, float res[3]) {
  auto m = [&](int r, int c) { return T[c * 4 + r]; };
  const float kPi = 3.14159265358979323846f;
  res[0] = std::atan2(m(1, 2), m(2, 2));
  const float c2 = std::sqrt(m(0, 0) * m(0, 0) + m(0, 1) * m(0, 1));
  if (res[0] > 0.0f) {
    res[0] -= kPi;
    res[1] = std::atan2(-m(0, 2), -c2);
  } else {
    res[1] = std::atan2(-m(0, 2), c2);
  }
  const float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * m(2, 0) - c1 * m(1, 0), c1 * m(1, 1) - s1 * m(2, 1));
  res[0] = -res[0];
  res[1] = -res[1];
  res[2] = -res[2];
}

// Eigen::Transform<float, 3, Affine>::rotation() of the guess (dgs_params.ndt_guess_rotation_polar): computeRotationScaling, i.e. a
// 3 x 3 float JacobiSVD (Eigen 3.3's two-sided Jacobi: the sequence of solve6.h's jsvd, here in float on the host, once per align),
// x = det(U V^T), U.col(0) /= x, R = U V^T.  Every operation individually rounded (contraction is off in this part of the file).
// [UPSTREAM-RECALL: Eigen/src/Geometry/Transform.h, Eigen/src/SVD/JacobiSVD.h; the CPU checker carries its own statement.]  R: row-major.
static void affine_rotation_f32(const float* T_colmajor16, float* R) {
  constexpr int N = 3;
  const float precision = 2.f * FLT_EPSILON, tiny = FLT_MIN;
  float W[9], U[9], V[9];
  float scale = 0.f;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) { W[r * 3 + c] = T_colmajor16[c * 4 + r]; scale = std::fabs(W[r * 3 + c]) > scale ? std::fabs(W[r * 3 + c]) : scale; }
  if (scale == 0.f) scale = 1.f;
  for (int i = 0; i < 9; i++) { W[i] = W[i] / scale; U[i] = V[i] = (i % 4 == 0) ? 1.f : 0.f; }
  float max_diag = 0.f;
  for (int i = 0; i < N; i++) { const float a = std::fabs(W[i * 4]); if (a > max_diag) max_diag = a; }
  bool finished = false;
  for (int sweep = 0; sweep < 64 && !finished; sweep++) {
    finished = true;
    for (int p = 1; p < N; p++)
      for (int q = 0; q < p; q++) {
        const float pm = precision * max_diag;
        const float threshold = tiny > pm ? tiny : pm;
        if (!(std::fabs(W[p * N + q]) > threshold || std::fabs(W[q * N + p]) > threshold)) continue;
        finished = false;
        float m00 = W[p * N + p], m01 = W[p * N + q], m10 = W[q * N + p], m11 = W[q * N + q];
        const float t = m00 + m11, d = m10 - m01;
        float r1c, r1s;
        if (std::fabs(d) < tiny) { r1s = 0.f; r1c = 1.f; }
        else {
          const float u = t / d;
          const float tmp = std::sqrt(1.f + u * u);
          r1s = 1.f / tmp;
          r1c = u / tmp;
        }
        if (!(r1c == 1.f && r1s == 0.f)) {
          const float x0 = m00, y0 = m10, x1 = m01, y1 = m11;
          m00 = r1c * x0 + r1s * y0; m10 = -r1s * x0 + r1c * y0;
          m01 = r1c * x1 + r1s * y1; m11 = -r1s * x1 + r1c * y1;
        }
        float jrc, jrs;
        {
          const float deno = 2.f * std::fabs(m01);
          if (deno < tiny) { jrc = 1.f; jrs = 0.f; }
          else {
            const float tau = (m00 - m11) / deno;
            const float w = std::sqrt(tau * tau + 1.f);
            const float tt = (tau > 0.f) ? 1.f / (tau + w) : 1.f / (tau - w);
            const float sign_t = tt > 0.f ? 1.f : -1.f;
            const float nn = 1.f / std::sqrt(tt * tt + 1.f);
            jrs = -sign_t * (m01 / std::fabs(m01)) * std::fabs(tt) * nn;
            jrc = nn;
          }
        }
        const float jtc = jrc, jts = -jrs;
        const float jlc = r1c * jtc - r1s * jts;
        const float jls = r1c * jts + r1s * jtc;
        if (!(jlc == 1.f && jls == 0.f)) {
          for (int i = 0; i < N; i++) {
            const float xi = W[p * N + i], yi = W[q * N + i];
            W[p * N + i] = jlc * xi + jls * yi;
            W[q * N + i] = -jls * xi + jlc * yi;
          }
          for (int i = 0; i < N; i++) {
            const float xi = U[i * N + p], yi = U[i * N + q];
            U[i * N + p] = jlc * xi + jls * yi;
            U[i * N + q] = -jls * xi + jlc * yi;
          }
        }
        if (!(jrc == 1.f && -jrs == 0.f)) {
          const float c = jrc, s = -jrs;
          for (int i = 0; i < N; i++) {
            const float xi = W[i * N + p], yi = W[i * N + q];
            W[i * N + p] = c * xi + s * yi;
            W[i * N + q] = -s * xi + c * yi;
          }
          for (int i = 0; i < N; i++) {
            const float xi = V[i * N + p], yi = V[i * N + q];
            V[i * N + p] = c * xi + s * yi;
            V[i * N + q] = -s * xi + c * yi;
          }
        }
        const float app = std::fabs(W[p * N + p]), aqq = std::fabs(W[q * N + q]);
        const float mx = app < aqq ? aqq : app;
        if (max_diag < mx) max_diag = mx;
      }
  }
  float sv[3];
  for (int i = 0; i < N; i++) {
    const float a = W[i * 4];
    sv[i] = std::fabs(a) * scale;
    if (a < 0.f) for (int k = 0; k < N; k++) U[k * N + i] = -U[k * N + i];
  }
  for (int i = 0; i < N; i++) {   // descending order: first maximum of the tail, column swaps
    int pos = 0;
    float best = sv[i];
    for (int k = 1; k < N - i; k++) if (sv[i + k] > best) { best = sv[i + k]; pos = k; }
    if (best == 0.f) break;
    if (pos) {
      pos += i;
      std::swap(sv[i], sv[pos]);
      for (int k = 0; k < N; k++) { std::swap(U[k * N + i], U[k * N + pos]); std::swap(V[k * N + i], V[k * N + pos]); }
    }
  }
  auto prod = [&](const float* M, int i, int j) { return M[i * 3 + 0] * V[j * 3 + 0] + M[i * 3 + 1] * V[j * 3 + 1] + M[i * 3 + 2] * V[j * 3 + 2]; };
  float UVt[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) UVt[i * 3 + j] = prod(U, i, j);
  auto det3h = [&](int a, int b, int c) { return UVt[0 * 3 + a] * (UVt[1 * 3 + b] * UVt[2 * 3 + c] - UVt[1 * 3 + c] * UVt[2 * 3 + b]); };
  const float x = det3h(0, 1, 2) - det3h(1, 0, 2) + det3h(2, 0, 1);
  for (int k = 0; k < 3; k++) U[k * 3 + 0] /= x;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[i * 3 + j] = prod(U, i, j);
}

static void fill_consts(dgs_handle* h) {
  const dgs_params& p = h->prm;
  const double c1 = 10.0 * (1.0 - p.ndt_outlier_ratio);
  const double c2 = p.ndt_outlier_ratio / std::pow(p.ndt_resolution, 3);
  const double d3 = -std::log(c2);
  NdtConsts& c = h->consts;
  c.gauss_d1 = -std::log(c1 + c2) - d3;
  c.gauss_d2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / c.gauss_d1);
  c.step_size = p.ndt_step_size;
  c.trans_eps = p.transformation_epsilon;
  c.max_iterations = p.maximum_iterations;
  c.line_search = p.ndt_line_search;
  c.mt_max_step_iterations = p.ndt_mt_max_step_iterations;
  c.fix_hessian_d1 = p.ndt_fix_hessian_d1;
  c.search_method = p.ndt_search_method;
  c.strict_order = p.ndt_strict_order;
  c.newton_solver = p.ndt_newton_solver;
  c.hessian_double = (p.ndt_strict_order != DGS_NDT_ORDER_FAST && p.ndt_hessian_recompute_double) ? 1 : 0;
  c.exp_libm = p.ndt_exp_glibc ? 1 : 0;
}

struct NdtLaunch {
  int n_pairs;
  int cap_blocks;    // most slices one pair can get (= rows reserved per pair in `partials`)
  int total_blocks;  // workgroups per derivative launch
  int max_n;         // largest source of the batch (row length of the ndt_strict_order 2 per-point table)
  int queue_workers; // queue kernel: persistent workgroups; 0 = launch-per-evaluation path
  int queue_base;    // queue kernel: slices of a pair's first rounds (ndt_queue_slices)
};

// per-pair stride (doubles) of the ndt_strict_order 2 table: 43 per-point totals, or -- with the double-precision computeHessian pass --
// 36 entries x (point, voxel slot) terms
static size_t strict_rows_pair_stride(const dgs_handle* h, const NdtLaunch& L) {
  size_t s = (size_t)kStrictAccum * L.max_n;
  if (h->consts.hessian_double) {
    const int nb = (h->consts.search_method == DGS_NDT_DIRECT1) ? 1 : (h->consts.search_method == DGS_NDT_DIRECT7 ? 7 : 27);
    s = std::max(s, (size_t)36 * L.max_n * nb);
  }
  return s;
}

// ndt_strict_order 2: per-point totals / per-term table to HBM (round 2's kernel), then the sequential sums
template <int SEARCH>
static void launch_strict_rows(dgs_handle* h, const NdtLaunch& L, const dim3 grid, const int leaf_pow2) {
  const double gd1 = h->consts.gauss_d1;
  const float gd2 = (float)h->consts.gauss_d2;
  static const bool literal = std::getenv("DGS_NDT_STRICT_LITERAL") && std::atoi(std::getenv("DGS_NDT_STRICT_LITERAL")) != 0;
  const size_t stride = strict_rows_pair_stride(h, L);
#define DGS_LAUNCH_STRICT(LIT)                                                                                                                             \
  hipLaunchKernelGGL((ndt_derivatives_strict_kernel<SEARCH, true, LIT>), grid, dim3(kBlock), 0, h->stream, h->src_ptrs.ptr, h->src_sizes.ptr, h->pairs.ptr, h->grid, \
                     h->vox_dbg.ptr, gd1, gd2, leaf_pow2, h->partials.ptr, h->strict_rows.ptr, L.max_n, L.n_pairs, L.cap_blocks, h->pair_blocks.ptr,     \
                     h->consts.gauss_d2, stride, h->consts.exp_libm)
  if (literal) DGS_LAUNCH_STRICT(true); else DGS_LAUNCH_STRICT(false);
#undef DGS_LAUNCH_STRICT
  hipLaunchKernelGGL(ndt_strict_seqsum_kernel, dim3(L.n_pairs), dim3(kWave), 0, h->stream, h->pairs.ptr, h->src_sizes.ptr, h->strict_rows.ptr, L.max_n,
                     h->strict_totals.ptr, stride, Offsets<SEARCH>::N);
}

// ndt_strict_order 1: ndt_strict_kernel (ndt_strict.h); launch >= 0: fused (derivatives + closing workgroups), < 0: derivatives only.
// hd: the instantiation for the pairs waiting for the double-precision computeHessian pass (evaluation kind 2).
// dgs_handle::strict_kernel (DGS_NDT_STRICT_KERNEL at dgs_create): 3 (default) the item-compacted kernel, one launch per round; 2 the
// lane-per-point kernels, two launches per round.
// (Measured and dropped: the item-compacted kernel for the float kinds alone -- 64-point tiles, a third of the LDS, meant for three waves
// per SIMD -- with the lane-per-point computeHessian kernel as the round's second launch: the register allocator spilled the double
// accumulators, 47 ms per step.)
// (the item-compacted kernel carries ONE exponential -- glibc's, the default: with both compiled in it went from 4 to 27 spilled registers;
//  ndt_exp_glibc = 0, the rounds 1-3 polynomial, is served by the lane-per-point kernels)
static int strict_kernel_version(const dgs_handle* h) { return (h->strict_kernel == 2 || !h->consts.exp_libm) ? 2 : 3; }


// item-compacted kernel, fused launches: the Newton steps of the closings go to ndt_strict_solve_kernel on the third stream
static bool strict_solve_beside(const dgs_handle* h) {
  return strict_kernel_version(h) == 3 && h->n_occupied_bound < (1 << 25) && h->ndt_fused && h->solve_min_active > 0 && h->hd_stream != nullptr &&
         h->consts.strict_order == DGS_NDT_ORDER_UPSTREAM;
}

template <int SEARCH>
static void launch_strict_sums(dgs_handle* h, const NdtLaunch& L, const dim3 grid, const int leaf_pow2, const int launch, const bool hd) {
  if (strict_kernel_version(h) == 3 && h->n_occupied_bound < (1 << 25)) {
    if (hd && launch >= 0) return;   // one kernel serves every kind (launch < 0: the test hook asks for the kind it has set up)
    const int spec = (launch >= 0 && h->ndt_speculate && h->consts.newton_solver && !strict_solve_beside(h)) ? 1 : 0;
    const dim3 grid_s(grid.x + (spec ? L.n_pairs : 0));   // + one solver workgroup per pair, in front (ndt_strict.h)
    if (h->ndt_fixed_slices) {   // DGS_NDT_FIXED_SLICES=1: a pair's slices are a function of its own size (ndt_strict.h)
      if (launch >= 0)
        hipLaunchKernelGGL((ndt_strict3_kernel<SEARCH, true, true, true>), grid_s, dim3(kBlock), 0, h->stream, h->src_ptrs.ptr, h->src_sizes.ptr, h->pairs.ptr, h->grid, h->vox_strict.ptr,
                           h->vox_dbg.ptr, h->consts.gauss_d1, h->consts.gauss_d2, leaf_pow2, h->partials.ptr, L.n_pairs, L.cap_blocks, h->pair_blocks.ptr, h->consts,
                           h->done_flags, launch, strict_solve_beside(h) ? h->solve_min_active : 0, spec);
      else
        hipLaunchKernelGGL((ndt_strict3_kernel<SEARCH, false, true, true>), grid, dim3(kBlock), 0, h->stream, h->src_ptrs.ptr, h->src_sizes.ptr, h->pairs.ptr, h->grid, h->vox_strict.ptr,
                           h->vox_dbg.ptr, h->consts.gauss_d1, h->consts.gauss_d2, leaf_pow2, h->partials.ptr, L.n_pairs, L.cap_blocks, h->pair_blocks.ptr, h->consts,
                           h->done_counter.ptr, launch, 0, 0);
    return;
    }
    if (launch >= 0)
      hipLaunchKernelGGL((ndt_strict3_kernel<SEARCH, true, true>), grid_s, dim3(kBlock), 0, h->stream, h->src_ptrs.ptr, h->src_sizes.ptr, h->pairs.ptr, h->grid, h->vox_strict.ptr,
                         h->vox_dbg.ptr, h->consts.gauss_d1, h->consts.gauss_d2, leaf_pow2, h->partials.ptr, L.n_pairs, L.cap_blocks, h->pair_blocks.ptr, h->consts,
                         h->done_flags, launch, strict_solve_beside(h) ? h->solve_min_active : 0, spec);
    else
      hipLaunchKernelGGL((ndt_strict3_kernel<SEARCH, false, true>), grid, dim3(kBlock), 0, h->stream, h->src_ptrs.ptr, h->src_sizes.ptr, h->pairs.ptr, h->grid, h->vox_strict.ptr,
                         h->vox_dbg.ptr, h->consts.gauss_d1, h->consts.gauss_d2, leaf_pow2, h->partials.ptr, L.n_pairs, L.cap_blocks, h->pair_blocks.ptr, h->consts,
                         h->done_counter.ptr, launch, 0, 0);
    return;
  }
  const bool beside = hd && launch >= 0 && h->hd_overlap && h->hd_stream;   // the computeHessian launch on its own stream, beside the next round's first launch
  hipStream_t lst = beside ? h->hd_stream : h->stream;
  const int hd_lag = (launch >= 0 && h->hd_overlap && h->hd_stream) ? 2 : 1;
#define DGS_LAUNCH_SS(FUSED, HD, FLAGS)                                                                                                                          \
  hipLaunchKernelGGL((ndt_strict_kernel<SEARCH, FUSED, HD>), grid, dim3(kBlock), 0, lst, h->src_ptrs.ptr, h->src_sizes.ptr, h->pairs.ptr, h->grid, h->vox_strict.ptr, \
                     h->vox_dbg.ptr, h->consts.gauss_d1, h->consts.gauss_d2, leaf_pow2, h->partials.ptr, L.n_pairs, L.cap_blocks, h->pair_blocks.ptr, h->consts, FLAGS, launch, hd_lag)
  if (launch >= 0) {
    if (hd) DGS_LAUNCH_SS(true, true, h->done_flags); else DGS_LAUNCH_SS(true, false, h->done_flags);
  } else {
    if (hd) DGS_LAUNCH_SS(false, true, h->done_counter.ptr); else DGS_LAUNCH_SS(false, false, h->done_counter.ptr);
  }
#undef DGS_LAUNCH_SS
}

// launch >= 0: fused launch number `launch` of this align (derivatives + closing workgroups); < 0: derivatives only
static void launch_derivatives(dgs_handle* h, const NdtLaunch& L, int launch = -1, bool hd = false) {
  const dim3 grid(L.total_blocks), block(kBlock);
  const double gd1 = h->consts.gauss_d1;
  const float gd2 = (float)h->consts.gauss_d2;
  int fe = 0;
  const int leaf_pow2 = (std::frexp(h->grid.leaf, &fe) == 0.5f) ? 1 : 0;
  hipStream_t pst = (hd && launch >= 0 && h->hd_overlap && h->hd_stream && h->consts.strict_order == DGS_NDT_ORDER_UPSTREAM) ? h->hd_stream : h->stream;   // launch_strict_sums
  int slot = prof_begin(h, DGS_K_NDT_DERIVATIVES, pst);
  if (h->consts.strict_order == DGS_NDT_ORDER_UPSTREAM_SEQUENTIAL) {
    switch (h->consts.search_method) {
      case DGS_NDT_DIRECT1: launch_strict_rows<DGS_NDT_DIRECT1>(h, L, grid, leaf_pow2); break;
      case DGS_NDT_DIRECT26: launch_strict_rows<DGS_NDT_DIRECT26>(h, L, grid, leaf_pow2); break;
      case DGS_NDT_KDTREE: launch_strict_rows<DGS_NDT_KDTREE>(h, L, grid, leaf_pow2); break;
      default: launch_strict_rows<DGS_NDT_DIRECT7>(h, L, grid, leaf_pow2); break;
    }
    prof_end(h, DGS_K_NDT_DERIVATIVES, slot);
    return;
  }
  if (h->consts.strict_order == DGS_NDT_ORDER_UPSTREAM) {
    switch (h->consts.search_method) {
      case DGS_NDT_DIRECT1: launch_strict_sums<DGS_NDT_DIRECT1>(h, L, grid, leaf_pow2, launch, hd); break;
      case DGS_NDT_DIRECT26: launch_strict_sums<DGS_NDT_DIRECT26>(h, L, grid, leaf_pow2, launch, hd); break;
      case DGS_NDT_KDTREE: launch_strict_sums<DGS_NDT_KDTREE>(h, L, grid, leaf_pow2, launch, hd); break;
      default: launch_strict_sums<DGS_NDT_DIRECT7>(h, L, grid, leaf_pow2, launch, hd); break;
    }
    prof_end(h, DGS_K_NDT_DERIVATIVES, slot, pst);
    return;
  }
#define DGS_LAUNCH_DERIV(SEARCH, FUSED, PACK)                                                                                                            \
  hipLaunchKernelGGL((ndt_derivatives_kernel<SEARCH, FUSED, PACK>), grid, block, 0, h->stream, h->src_ptrs.ptr, h->src_sizes.ptr, h->pairs.ptr, h->grid, gd1, gd2, \
                     leaf_pow2, h->partials.ptr, L.n_pairs, L.cap_blocks, h->pair_blocks.ptr, h->consts, (launch >= 0 ? h->done_flags : h->done_counter.ptr), launch)
  if (launch >= 0) {
    switch (h->consts.search_method) {
      case DGS_NDT_DIRECT1: DGS_LAUNCH_DERIV(DGS_NDT_DIRECT1, true, false); break;
      case DGS_NDT_DIRECT26: DGS_LAUNCH_DERIV(DGS_NDT_DIRECT26, true, false); break;
      case DGS_NDT_KDTREE: DGS_LAUNCH_DERIV(DGS_NDT_KDTREE, true, false); break;
      default:
#ifdef DGS_EXPERIMENTS
        if (h->ndt_pack2) { DGS_LAUNCH_DERIV(DGS_NDT_DIRECT7, true, true); break; }
#endif
        DGS_LAUNCH_DERIV(DGS_NDT_DIRECT7, true, false);
        break;
    }
  } else {
    switch (h->consts.search_method) {
      case DGS_NDT_DIRECT1: DGS_LAUNCH_DERIV(DGS_NDT_DIRECT1, false, false); break;
      case DGS_NDT_DIRECT26: DGS_LAUNCH_DERIV(DGS_NDT_DIRECT26, false, false); break;
      case DGS_NDT_KDTREE: DGS_LAUNCH_DERIV(DGS_NDT_KDTREE, false, false); break;
      default:
#ifdef DGS_EXPERIMENTS
        if (h->ndt_pack2) { DGS_LAUNCH_DERIV(DGS_NDT_DIRECT7, false, true); break; }
#endif
        DGS_LAUNCH_DERIV(DGS_NDT_DIRECT7, false, false);
        break;
    }
  }
#undef DGS_LAUNCH_DERIV
  prof_end(h, DGS_K_NDT_DERIVATIVES, slot);
}

static void launch_solve(dgs_handle* h, const NdtLaunch& L) {
  int slot = prof_begin(h, DGS_K_NDT_SOLVE);
  hipLaunchKernelGGL(ndt_solve_kernel, dim3(L.n_pairs), dim3(kBlock), 0, h->stream, h->pairs.ptr, h->partials.ptr, L.cap_blocks, h->pair_blocks.ptr, h->consts,
                     h->done_counter.ptr, h->consts.strict_order != DGS_NDT_ORDER_FAST ? h->strict_totals.ptr : nullptr,
                     h->consts.strict_order == DGS_NDT_ORDER_UPSTREAM ? 1 : 0);
  prof_end(h, DGS_K_NDT_SOLVE, slot);
}

static NdtLaunch choose_launch(int n_pairs, int max_n) {
  NdtLaunch L;
  L.n_pairs = n_pairs;
  // tuning knobs (sweeps only; the defaults are the measured winners: profiles/r03/launch_shape_sweep.jsonl)
  static const int env_cap = std::getenv("DGS_NDT_CAP") ? std::atoi(std::getenv("DGS_NDT_CAP")) : 1024;
  static const int env_total = std::getenv("DGS_NDT_BLOCKS") ? std::atoi(std::getenv("DGS_NDT_BLOCKS")) : 1024;
  static const int env_ppt = std::getenv("DGS_NDT_PPT") ? std::max(1, std::atoi(std::getenv("DGS_NDT_PPT"))) : 2;
  // Slices (= partial rows) one pair can get: sized by the cloud, >= env_ppt points per thread -- 128 workgroups for a 65,536-point
  // scan as before, 391 for the 200,000-point indoor scan (which a fixed cap of 128 held on half of the 256 CUs), never more rows
  // than the closing workgroup's row sum was laid out for.
  static const int env_min = std::getenv("DGS_NDT_MIN_BLOCKS") ? std::max(1, std::atoi(std::getenv("DGS_NDT_MIN_BLOCKS"))) : 64;
  const int by_points = (max_n + kBlock * env_ppt - 1) / (kBlock * env_ppt);
  const int at_least = std::min(env_min, (max_n + kBlock - 1) / kBlock);   // small clouds: 64 workgroups while every thread still has a point (16,384 points: 13.3 against 15.1 us per evaluation)
  L.cap_blocks = std::max(1, std::min({std::max(by_points, at_least), env_cap, kMaxPartialBlocks}));
  L.total_blocks = (int)std::max<int64_t>(n_pairs, std::min<int64_t>((int64_t)n_pairs * L.cap_blocks, env_total));  // ~4 workgroups per CU
  L.max_n = std::max(max_n, 1);
  L.queue_workers = 0;
  L.queue_base = 0;
  return L;
}

// Shape of the persistent queue kernel for this batch: workers (3 workgroups per CU by default: the fourth slot of every CU stays free
// for the side stream's index build), and the slices of a pair's first rounds.
static void choose_queue(dgs_handle* h, NdtLaunch& L) {
  static const int env_workers = std::getenv("DGS_NDT_QUEUE_WORKERS") ? std::atoi(std::getenv("DGS_NDT_QUEUE_WORKERS")) : 768;
  static const int env_base = std::getenv("DGS_NDT_QUEUE_BASE") ? std::atoi(std::getenv("DGS_NDT_QUEUE_BASE")) : 0;
  const int64_t most = (int64_t)L.n_pairs * L.cap_blocks;
  L.queue_workers = (int)std::max<int64_t>(1, std::min<int64_t>(env_workers, most));
  L.queue_base = env_base > 0 ? std::min(env_base, L.cap_blocks) : std::max(1, std::min(L.cap_blocks, L.queue_workers / std::max(1, L.n_pairs)));
  (void)h;
}

// Uploads pointers / sizes / initial poses, runs init, returns the launch shape.
static int ndt_setup(dgs_handle* h, int n_pairs, const float4* const* src_ptrs_host, const int* sizes_host, const float* guesses16,
                     const double* probe_p6, NdtLaunch* launch_out, bool use_queue = false, int probe_kind = 1) {
  hipStream_t st = h->stream;
  fill_consts(h);
  int max_n = 0;
  for (int i = 0; i < n_pairs; i++) max_n = std::max(max_n, sizes_host[i]);
  NdtLaunch L = choose_launch(n_pairs, max_n);
  if (use_queue) {
    choose_queue(h, L);
    DGS_HIP_TRY(h, h->ndt_queue.reserve(2 * ((size_t)n_pairs + 2) + kQueueStatInts));
    // one record slot per pair and round; a registration ends within (max_iterations + 2) x (line-search trials + 2) evaluations
    const int per_iter_q = (h->prm.ndt_line_search == DGS_NDT_LS_FIXED_STEP && h->prm.ndt_step_size - h->prm.transformation_epsilon / 2 > 0) ? 1 : (h->prm.ndt_mt_max_step_iterations + 2);
    h->ndt_ring_rounds = (h->prm.maximum_iterations + 3) * per_iter_q + 8;
    DGS_HIP_TRY(h, h->ndt_ring.reserve((size_t)n_pairs * h->ndt_ring_rounds * kQueueSlotBytes));
  }
  *launch_out = L;
  DGS_HIP_TRY(h, h->pairs.reserve(n_pairs));
  DGS_HIP_TRY(h, h->inits.reserve(n_pairs));
  DGS_HIP_TRY(h, h->src_ptrs.reserve(n_pairs));
  DGS_HIP_TRY(h, h->src_sizes.reserve(n_pairs));
  DGS_HIP_TRY(h, h->partials.reserve((size_t)n_pairs * L.cap_blocks * (h->consts.strict_order ? kStrictPad : kAccumPad)));
  if (h->consts.strict_order) DGS_HIP_TRY(h, h->strict_totals.reserve((size_t)n_pairs * kStrictPad));
  if (h->consts.strict_order == DGS_NDT_ORDER_UPSTREAM_SEQUENTIAL) DGS_HIP_TRY(h, h->strict_rows.reserve((size_t)n_pairs * strict_rows_pair_stride(h, L)));
  DGS_HIP_TRY(h, h->pair_blocks.reserve(n_pairs));
  if (h->consts.strict_order == DGS_NDT_ORDER_UPSTREAM) DGS_HIP_TRY(h, hipMemsetAsync(h->pair_blocks.ptr, 0, sizeof(int) * n_pairs, st));   // "rows present" marks of the unfused upstream order
  DGS_HIP_TRY(h, h->done_counter.reserve(16));
  const size_t off_init = 256;
  const size_t off_ptr = off_init + sizeof(NdtInit) * n_pairs;
  const size_t off_size = off_ptr + sizeof(void*) * n_pairs;
  const size_t off_out = (off_size + sizeof(int) * n_pairs + 255) & ~(size_t)255;
  const size_t total = off_out + sizeof(NdtOut) * n_pairs + sizeof(NdtPair) + 256;
  if (ensure_pinned(h, total) != DGS_OK) return DGS_ERR_HIP;
  char* base = reinterpret_cast<char*>(h->pinned);
  NdtInit* hin = reinterpret_cast<NdtInit*>(base + off_init);
  const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int i = 0; i < n_pairs; i++) {
    const float* G = guesses16 ? guesses16 + 16 * i : ident;
    std::memcpy(hin[i].guess, G, sizeof(float) * 16);
    if (probe_p6) {
      for (int k = 0; k < 6; k++) hin[i].p0[k] = probe_p6[k];
    } else {
      float e[3];
      if (h->prm.ndt_guess_rotation_polar) {   // eig_transformation.rotation().eulerAngles(0, 1, 2): Affine3f::rotation() is the polar factor
        float R[9], Gp[16];
        affine_rotation_f32(G, R);
        std::memcpy(Gp, G, sizeof(Gp));
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) Gp[c * 4 + r] = R[r * 3 + c];
        euler_angles_012(Gp, e);
      } else {
        euler_angles_012(G, e);
      }
      hin[i].p0[0] = G[12]; hin[i].p0[1] = G[13]; hin[i].p0[2] = G[14];
      hin[i].p0[3] = e[0]; hin[i].p0[4] = e[1]; hin[i].p0[5] = e[2];
    }
  }
  std::memcpy(base + off_ptr, src_ptrs_host, sizeof(void*) * n_pairs);
  std::memcpy(base + off_size, sizes_host, sizeof(int) * n_pairs);
  // ONE copy of (initial poses | source pointers | sizes), contiguous in the pinned block as in the device staging buffer; the init
  // kernel hands the pointers and sizes on to the arrays the other kernels read
  const size_t stage_bytes = off_size + sizeof(int) * n_pairs - off_init;
  DGS_HIP_TRY(h, h->inits.reserve((stage_bytes + sizeof(NdtInit) - 1) / sizeof(NdtInit)));
  DGS_HIP_TRY(h, hipMemcpyAsync(h->inits.ptr, hin, stage_bytes, hipMemcpyHostToDevice, st));
  const char* dstage = reinterpret_cast<const char*>(h->inits.ptr);
  hipLaunchKernelGGL(ndt_init_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, st, h->pairs.ptr, h->inits.ptr, n_pairs, h->consts, probe_p6 ? probe_kind : 0,
                     h->done_counter.ptr, reinterpret_cast<const float4* const*>(dstage + (off_ptr - off_init)),
                     reinterpret_cast<const int*>(dstage + (off_size - off_init)), h->src_ptrs.ptr, h->src_sizes.ptr,
                     L.queue_workers > 0 ? h->ndt_queue.ptr : nullptr, L.queue_workers > 0 ? ndt_queue_slices(0, L.queue_base, L.cap_blocks) : 0,
                     reinterpret_cast<char*>(h->ndt_ring.ptr), h->ndt_ring_rounds);
  return DGS_OK;
}

#ifdef DGS_EXPERIMENTS
static void launch_queue(dgs_handle* h, const NdtLaunch& L) {
  const dim3 grid(L.queue_workers), block(kBlock);
  const double gd1 = h->consts.gauss_d1;
  const float gd2 = (float)h->consts.gauss_d2;
  int fe = 0;
  const int leaf_pow2 = (std::frexp(h->grid.leaf, &fe) == 0.5f) ? 1 : 0;
  int slot = prof_begin(h, DGS_K_NDT_DERIVATIVES);
#define DGS_LAUNCH_QUEUE(SEARCH)                                                                                                                       \
  hipLaunchKernelGGL((ndt_queue_kernel<SEARCH>), grid, block, 0, h->stream, h->src_ptrs.ptr, h->src_sizes.ptr, h->pairs.ptr, h->grid, gd1, gd2, leaf_pow2, \
                     h->partials.ptr, L.n_pairs, L.cap_blocks, L.queue_base, h->consts, h->ndt_queue.ptr, h->done_counter.ptr,                \
                     reinterpret_cast<char*>(h->ndt_ring.ptr), h->ndt_ring_rounds)
  switch (h->consts.search_method) {
    case DGS_NDT_DIRECT1: DGS_LAUNCH_QUEUE(DGS_NDT_DIRECT1); break;
    case DGS_NDT_DIRECT26: DGS_LAUNCH_QUEUE(DGS_NDT_DIRECT26); break;
    case DGS_NDT_KDTREE: DGS_LAUNCH_QUEUE(DGS_NDT_KDTREE); break;
    default: DGS_LAUNCH_QUEUE(DGS_NDT_DIRECT7); break;
  }
#undef DGS_LAUNCH_QUEUE
  prof_end(h, DGS_K_NDT_DERIVATIVES, slot);
}
#else
static void launch_queue(dgs_handle*, const NdtLaunch&) {}
#endif

static int ndt_export(dgs_handle* h, int n_pairs, dgs_result* results);

int ndt_align_pairs(dgs_handle* h, int n_pairs, const float4* const* src_ptrs_host, const int* sizes_host, const float* guesses16,
                    dgs_result* results) {
  hipStream_t st = h->stream;
  NdtLaunch L{};
  // the persistent queue kernel serves the default evaluation order (the validation orders keep their launch-per-evaluation kernels)
  const bool use_queue = kExperiments && h->ndt_queue_mode != 0 && h->ndt_fused && (h->consts.strict_order == DGS_NDT_ORDER_FAST) && n_pairs >= h->ndt_queue_min_pairs;
  int rc = ndt_setup(h, n_pairs, src_ptrs_host, sizes_host, guesses16, nullptr, &L, use_queue);
  if (rc != DGS_OK) return rc;
  if (use_queue) {
    launch_queue(h, L);
    // the queue kernel leaves one workgroup slot per CU free: the side stream's index build (dgs_align_batch) runs beside it
    if ((rc = side_build_now(h)) != DGS_OK) return rc;
    int* hq = reinterpret_cast<int*>(h->pinned);   // control word: [0] pairs unfinished, [1] abort bit
    DGS_HIP_TRY(h, hipMemcpyAsync(hq, h->ndt_queue.ptr, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
    rc = ndt_export(h, n_pairs, results);
    if (rc != DGS_OK) return rc;
#ifdef DGS_QUEUE_STATS
    {
      int q[8];
      (void)hipMemcpy(q, h->ndt_queue.ptr + 2 * (n_pairs + 2), sizeof(q), hipMemcpyDeviceToHost);
      int q2[8];
      (void)hipMemcpy(q2, h->ndt_queue.ptr + 2 * (n_pairs + 2) + 8, sizeof(q2), hipMemcpyDeviceToHost);
      const double it = std::max(1, q[4]);
      std::fprintf(stderr, "[queue] workers %d base %d: polls %d failed_claims %d claims %d closings %d; per worker: looking for work %.1f us, in items %.1f us; per item: %.2f us = record %.2f + "
                   "points %.2f + row %.2f + ticket %.2f; per closing %.2f us\n",
                   L.queue_workers, L.queue_base, q[2], q[3], q[4], q[7], q[5] * 0.01 / L.queue_workers, q[6] * 0.01 / L.queue_workers, q[6] * 0.01 / it, q2[0] * 0.01 / it,
                   q2[1] * 0.01 / it, q2[2] * 0.01 / it, q2[3] * 0.01 / it, q2[4] * 0.01 / std::max(1, q[7]));
    }
#endif
    if (hq[1] != 0 || hq[0] != 0) {
      h->err = "ndt_queue_kernel gave up (poll guard): " + std::to_string(hq[0]) + " registrations unfinished";
      return DGS_ERR_HIP;
    }
    return DGS_OK;
  }

  if (h->early_fit.on && (rc = nn_fitness_prepare(h, n_pairs, h->early_fit.max_n, false)) != DGS_OK) return rc;
  // ---- iterate: chunks of (derivatives, solve) launches; the host looks at the done counter one chunk behind
  volatile int* flags = reinterpret_cast<volatile int*>(h->pinned);  // [0], [1]: done counts of alternating chunks
  flags[0] = flags[1] = 0;
  if (ensure_poll_events(h) != DGS_OK) return DGS_ERR_HIP;
  const bool fused_flags = h->ndt_fused && h->consts.strict_order != DGS_NDT_ORDER_UPSTREAM_SEQUENTIAL;   // the default order and the upstream order close inside the launch
  if (fused_flags) {   // fused launches: every pair has a "finished" flag in pinned host memory that its closing workgroup sets
    if (h->done_flags_cap < n_pairs) {
      if (h->done_flags) (void)hipHostFree(h->done_flags);
      h->done_flags = nullptr;
      h->done_flags_cap = 0;
      DGS_HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->done_flags), sizeof(int) * (size_t)(n_pairs + 64), hipHostMallocDefault));
      h->done_flags_cap = n_pairs + 64;
    }
    for (int i = 0; i < n_pairs; i++) h->done_flags[i] = 0;   // nothing of this handle is in flight: the previous align has been synchronised
  }
  hipEvent_t* ev = h->ev_poll;
  // evaluations one iteration can take.  The fixed-step form initialises its interval as converged only while step_size > epsilon / 2; at or
  // below that (a nonsensical but legal parameter set: the soak drew step_size 0.05 with epsilon 0.1) it runs the trial loop like More-Thuente,
  // and a budget of one evaluation per iteration cut such pairs off unfinished
  const bool fixed_one = h->prm.ndt_line_search == DGS_NDT_LS_FIXED_STEP && h->prm.ndt_step_size - h->prm.transformation_epsilon / 2 > 0;
  const int per_iter = fixed_one ? 1 : (h->prm.ndt_mt_max_step_iterations + 2);
  // (upstream order with speculated Newton steps: an evaluation whose header the exact step refuses is made again -- at most twice the launches)
  const long max_evals = ((long)(h->prm.maximum_iterations + 3) * per_iter + 2) * ((h->consts.strict_order == DGS_NDT_ORDER_UPSTREAM && h->ndt_speculate) ? 2 : 1);
  const int chunk = 4;  // (derivatives, solve) launches between two looks at the done counter
  long queued = 0;
  const bool fused = fused_flags;
  int launch_no = 0;
  // DGS_NDT_SCHEDULE=1 (tests): every launch cuts the pairs into the slices the queue kernel would give that round, so that the two
  // paths sum the same partitions and can be compared bit for bit
  const bool schedule = h->ndt_schedule;
  NdtLaunch Lq = L;
  if (schedule) choose_queue(h, Lq);
  int round_no = 0;
  int launches_upto[2] = {0, 0};   // launches enqueued up to the end of the chunk of either slot
  int hd_rounds = 0;               // rounds whose computeHessian launch went to its own stream
  auto enqueue_chunk = [&](int slot, int launches) -> int {
    for (int e = 0; e < launches; e++) {
      NdtLaunch Lr = L;
      if (schedule) {
        Lr.cap_blocks = ndt_queue_slices(round_no, Lq.queue_base, L.cap_blocks);
        Lr.total_blocks = n_pairs * Lr.cap_blocks;
      }
      round_no++;
      const bool two_kinds = h->consts.strict_order == DGS_NDT_ORDER_UPSTREAM && h->consts.hessian_double && !(strict_kernel_version(h) == 3 && h->n_occupied_bound < (1 << 25));   // ndt_strict.h, lane-per-point kernels: kinds 0 / 1, then kind 2
      if (fused) {
        if (two_kinds && h->hd_overlap && h->hd_stream) {
          // round r: [main] first kernel, [computeHessian stream] second kernel beside the first kernel of round r + 1; the first kernel of
          // round r + 2 waits for it (NdtPair::serve, lag 2)
          constexpr int R = dgs_handle::kHdEvents;
          if (launch_no >= 2) DGS_HIP_TRY(h, hipStreamWaitEvent(st, h->ev_hd_b[(launch_no - 2) % R], 0));
          launch_derivatives(h, Lr, launch_no);
          DGS_HIP_TRY(h, hipEventRecord(h->ev_hd_a[launch_no % R], st));
          DGS_HIP_TRY(h, hipStreamWaitEvent(h->hd_stream, h->ev_hd_a[launch_no % R], 0));
          launch_derivatives(h, Lr, launch_no, true);
          DGS_HIP_TRY(h, hipEventRecord(h->ev_hd_b[launch_no % R], h->hd_stream));
          hd_rounds = launch_no + 1;
        } else if (h->consts.strict_order == DGS_NDT_ORDER_UPSTREAM && strict_solve_beside(h)) {
          // round r: [main] the derivative launch; [third stream] the Newton steps its closings left behind, beside the launch of round r + 1;
          // the launch of round r + 2 waits for them (NdtPair::serve, lag 2)
          constexpr int R = dgs_handle::kHdEvents;
          if (launch_no >= 2) DGS_HIP_TRY(h, hipStreamWaitEvent(st, h->ev_hd_b[(launch_no - 2) % R], 0));
          launch_derivatives(h, Lr, launch_no);
          DGS_HIP_TRY(h, hipEventRecord(h->ev_hd_a[launch_no % R], st));
          DGS_HIP_TRY(h, hipStreamWaitEvent(h->hd_stream, h->ev_hd_a[launch_no % R], 0));
          hipLaunchKernelGGL(ndt_strict_solve_kernel, dim3(n_pairs), dim3(kWave), 0, h->hd_stream, h->pairs.ptr, n_pairs, h->consts, h->done_flags, launch_no, 2);
          DGS_HIP_TRY(h, hipEventRecord(h->ev_hd_b[launch_no % R], h->hd_stream));
          hd_rounds = launch_no + 1;
        } else {
          launch_derivatives(h, Lr, launch_no);
          if (two_kinds) launch_derivatives(h, Lr, launch_no, true);   // same round number: NdtPair::serve
        }
        launch_no++;
      } else {
        launch_derivatives(h, Lr);
        launch_solve(h, Lr);
        if (two_kinds) {
          launch_derivatives(h, Lr, -1, true);
          launch_solve(h, Lr);
        }
      }
    }
    queued += launches;
    launches_upto[slot] = launch_no;
    if (!fused_flags) DGS_HIP_TRY(h, hipMemcpyAsync(const_cast<int*>(&flags[slot]), h->done_counter.ptr, sizeof(int), hipMemcpyDeviceToHost, st));
    DGS_HIP_TRY(h, hipEventRecord(ev[slot], st));
    return DGS_OK;
  };
  auto pairs_done = [&](int slot) -> int {
    if (!fused_flags) return flags[slot];
    int n = 0;
    for (int i = 0; i < n_pairs; i++) n += reinterpret_cast<volatile int*>(h->done_flags)[i] != 0;
    return n;
  };
  // ---- early fitness (dgs_align_batch, compute_fitness): the nearest-neighbour walk of a candidate needs only its final transform, and
  // the tail of a batch -- a few pairs still iterating, 11 us of latency per launch -- leaves most of the chip idle.  At every chunk
  // boundary the pairs whose closing launch is KNOWN to have completed (flag = launch + 1 <= the launches the synchronised event covers:
  // their state is in memory, not in some XCD's L2) get their walk on the low-priority side stream, behind the target's index build.
  // Whatever is left at the end goes on the main stream.  A pair's rows and their order do not depend on which launch walked it.
  const bool early = h->early_fit.on && fused_flags;
  std::vector<char> walked(early ? n_pairs : 0, 0);
  const float* fit_T = reinterpret_cast<const float*>(reinterpret_cast<const char*>(h->pairs.ptr) + offsetof(NdtPair, final_T));
  std::vector<int> ready;
  auto walk_finished = [&](int completed, bool rest) -> int {
    ready.clear();
    for (int i = 0; i < n_pairs; i++) {
      if (walked[i]) continue;
      const int f = reinterpret_cast<volatile int*>(h->done_flags)[i];
      if (rest || (f != 0 && f <= completed)) ready.push_back(i);
    }
    // a walk's workgroups live ~100 us whatever the number of pairs: few large launches, not one per finished pair
    if (ready.empty() || (!rest && ((int)ready.size() < h->early_fit.min_pairs || n_pairs - pairs_done(0) > h->early_fit.max_active))) return DGS_OK;
    for (int i : ready) walked[i] = 1;
    nn_fitness_enqueue(h, rest ? st : h->side_stream, h->tgt->bvh, ready.data(), (int)ready.size(), h->src_ptrs.ptr, h->src_sizes.ptr, fit_T, sizeof(NdtPair),
                       h->early_fit.max_range, 0.0, rest ? 0 : h->early_fit.lds_kb);
    if (!rest) {
      DGS_HIP_TRY(h, hipEventRecord(h->ev_join, h->side_stream));
      h->side_pending = true;
    }
    return DGS_OK;
  };
  int cur = 0;
  // with a build waiting for the side stream the first chunk is twice as long: the host needs ~0.1 ms to enqueue that build, and
  // the main stream must not run dry meanwhile.  (Also tried: the fused launches writing the count of finished pairs into pinned
  // host memory themselves instead of a copy command per chunk -- the one more kernel argument pushed the derivative loop over its
  // 128 VGPRs (4 spilled registers, 34 -> 40 us per launch): the copies stay.)
  rc = enqueue_chunk(0, h->side_build_deferred ? 2 * chunk : chunk);
  bool finished = false;
  bool first = true;
  while (rc == DGS_OK) {
    const bool more = queued < max_evals;
    if (more) rc = enqueue_chunk(cur ^ 1, chunk);
    if (rc != DGS_OK) break;
    // two chunks are in flight: now the host has time to enqueue what dgs_align_batch left for the side stream
    if (first && (rc = side_build_now(h)) != DGS_OK) break;
    first = false;
    hipError_t e = hipEventSynchronize(ev[cur]);
    if (e != hipSuccess) { h->err = std::string("hipEventSynchronize: ") + hipGetErrorString(e); rc = DGS_ERR_HIP; break; }
    if (early && (rc = walk_finished(launches_upto[cur], false)) != DGS_OK) break;
    if (pairs_done(cur) >= n_pairs) { finished = true; break; }
    if (!more) break;
    cur ^= 1;
  }
  if (hd_rounds > 0) {   // the main stream takes the computeHessian stream's launches in (they are in order: the last one covers them all)
    hipError_t e = hipStreamWaitEvent(st, h->ev_hd_b[(hd_rounds - 1) % dgs_handle::kHdEvents], 0);
    if (e != hipSuccess && rc == DGS_OK) { h->err = std::string("hipStreamWaitEvent: ") + hipGetErrorString(e); rc = DGS_ERR_HIP; }
  }
  if (rc != DGS_OK) return rc;
  (void)finished;  // pairs that did not finish inside max_evals export converged = 0
  if (early) {
    // the main stream waits for the side stream (index build, early walks), walks the rest, totals everything and copies it out: the
    // export's synchronisation below covers all of it
    if (side_join(h) != DGS_OK) return DGS_ERR_HIP;
    if ((rc = walk_finished(0, true)) != DGS_OK) return rc;
    if ((rc = nn_fitness_totals_enqueue(h, n_pairs)) != DGS_OK) return rc;
    h->early_fit.enqueued = true;
  }
  return ndt_export(h, n_pairs, results);
}

// ---- export: final transforms / flags / counts of every pair to the caller's result array
static int ndt_export(dgs_handle* h, int n_pairs, dgs_result* results) {
  hipStream_t st = h->stream;
  char* base = reinterpret_cast<char*>(h->pinned);
  const size_t off_out = ((256 + sizeof(NdtInit) * n_pairs + sizeof(void*) * n_pairs + sizeof(int) * n_pairs) + 255) & ~(size_t)255;
  NdtOut* hout = reinterpret_cast<NdtOut*>(base + off_out);
  NdtOut* dout = reinterpret_cast<NdtOut*>(h->partials.ptr);  // partial rows are dead now; reuse (>= 32 doubles per pair)
  hipLaunchKernelGGL(ndt_export_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, st, h->pairs.ptr, n_pairs, dout);
  DGS_HIP_TRY(h, hipMemcpyAsync(hout, dout, sizeof(NdtOut) * n_pairs, hipMemcpyDeviceToHost, st));
  DGS_HIP_TRY(h, hipStreamSynchronize(st));
  DGS_HIP_TRY(h, hipGetLastError());
  int64_t evals = 0;
  for (int i = 0; i < n_pairs; i++) {
    std::memcpy(results[i].final_transformation, hout[i].T, sizeof(float) * 16);
    results[i].converged = hout[i].converged;
    results[i].iterations = hout[i].iterations;
    results[i].evaluations = hout[i].evaluations;
    results[i].status = DGS_OK;
    results[i].score = hout[i].score;
    results[i].fitness = NAN;
    evals += hout[i].evaluations;
  }
  h->last_evaluations = evals;
  return DGS_OK;
}

// Test hook: poses after every outer iteration of pair `pair` of the last align / batch.
int ndt_trajectory(dgs_handle* h, int pair, double* out, int* len) {
  std::vector<char> buf(sizeof(NdtPair));
  DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));
  DGS_HIP_TRY(h, hipMemcpy(buf.data(), h->pairs.ptr + pair, sizeof(NdtPair), hipMemcpyDeviceToHost));
  const NdtPair* st = reinterpret_cast<const NdtPair*>(buf.data());
#ifdef DGS_CLOSE_STAMPS
  const int n = kTrajCap;
#else
  const int n = std::min(st->s.traj_len, kTrajCap);
#endif
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 6; k++) out[i * 6 + k] = st->traj[i][k];
  *len = n;
  return DGS_OK;
}

// Test hook: one computeDerivatives evaluation on the device.
int ndt_probe(dgs_handle* h, const double* p6, const float* T16, double* score, double* g6, double* H36, int kind) {
  hipStream_t st = h->stream;
  NdtLaunch L{};
  const float4* src = h->src->pts.ptr;
  const int n = (int)h->ns;
  float T[16];
  if (T16) {
    std::memcpy(T, T16, sizeof(T));
  } else {  // pose_to_matrix in float, as the solver builds it
    const float rx = (float)p6[3], ry = (float)p6[4], rz = (float)p6[5];
    const float cx = (float)std::cos((double)rx), sx = (float)std::sin((double)rx), cy = (float)std::cos((double)ry), sy = (float)std::sin((double)ry),
                cz = (float)std::cos((double)rz), sz = (float)std::sin((double)rz);
    auto mul = [](float a, float b) { volatile float r = a * b; return (float)r; };  // keep host products un-fused
    T[0] = mul(cy, cz); T[4] = mul(-cy, sz); T[8] = sy; T[12] = (float)p6[0];
    { volatile float a = mul(cx, sz), b = mul(mul(sx, sy), cz); T[1] = a + b; }
    { volatile float a = mul(cx, cz), b = mul(mul(sx, sy), sz); T[5] = a - b; }
    T[9] = mul(-sx, cy); T[13] = (float)p6[1];
    { volatile float a = mul(sx, sz), b = mul(mul(cx, sy), cz); T[2] = a - b; }
    { volatile float a = mul(sx, cz), b = mul(mul(cx, sy), sz); T[6] = a + b; }
    T[10] = mul(cx, cy); T[14] = (float)p6[2];
    T[3] = T[7] = T[11] = 0.f; T[15] = 1.f;
  }
  int rc = ndt_setup(h, 1, &src, &n, T, p6, &L, false, kind);
  if (rc != DGS_OK) return rc;
  launch_derivatives(h, L, -1, kind == 2);
  launch_solve(h, L);
  char* base = reinterpret_cast<char*>(h->pinned);
  NdtPair* hp = reinterpret_cast<NdtPair*>(base + ((h->pinned_bytes - sizeof(NdtPair) - 64) & ~(size_t)63));
  DGS_HIP_TRY(h, hipMemcpyAsync(hp, h->pairs.ptr, sizeof(NdtPair), hipMemcpyDeviceToHost, st));
  DGS_HIP_TRY(h, hipStreamSynchronize(st));
  DGS_HIP_TRY(h, hipGetLastError());
  *score = hp->s.score;
  for (int k = 0; k < 6; k++) g6[k] = hp->s.grad[k];
  for (int k = 0; k < 36; k++) H36[k] = hp->s.hess[k];
  return DGS_OK;
}

}  // namespace dgs
