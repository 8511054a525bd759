// 6x6 linear solves used by the optimiser kernels (NDT Newton step, GICP Levenberg-Marquardt step).
#pragma once
#include <cfloat>
#include <cmath>

#include "common.h"

namespace dgs {

// value of lane `src` (wave-uniform, known at compile time after unrolling): two v_readlane_b32 instead of the LDS crossbar round
// trip of a ds_bpermute -- the elimination below is one dependent chain, its latency is the optimiser's latency
__device__ __forceinline__ double readlane_f64(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// 6x6 solve by Gauss-Jordan elimination with row pivoting, one matrix element per lane (lanes 0..41 hold [A | b]);
// every value that steers control flow is wave-uniform.  *rcond receives min|pivot| / max|pivot|.
__device__ __forceinline__ void gj_solve6_wave(const double* A, const double* b, double* x, double* rcond) {
  const int lane = threadIdx.x & 63;
  const int i = (lane < 42) ? lane / 7 : 0, j = (lane < 42) ? lane % 7 : 0;
  double a = 0.0;
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c < 7; c++)
      if (lane == r * 7 + c) a = (c < 6) ? A[r * 6 + c] : b[r];
  double pmax = 0.0, pmin = DBL_MAX;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    int best_r = k;
    double best_v = -1.0;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const double v = fabs(readlane_f64(a, r * 7 + k));
      if (r >= k && v > best_v) { best_v = v; best_r = r; }
    }
    const int src = (i == k) ? best_r * 7 + j : ((i == best_r) ? k * 7 + j : lane);
    a = __shfl(a, src, 64);
    const double piv = readlane_f64(a, k * 7 + k);
    pmax = fmax(pmax, fabs(piv));
    pmin = fmin(pmin, fabs(piv));
    const double rowk = __shfl(a, k * 7 + j, 64);
    const double colk = __shfl(a, i * 7 + k, 64);
    if (piv != 0.0) a = (i == k) ? a / piv : a - colk * (rowk / piv);
  }
#pragma unroll
  for (int r = 0; r < 6; r++) x[r] = readlane_f64(a, r * 7 + 6);
  *rcond = (pmax > 0) ? pmin / pmax : 0.0;
}

// Gauss-Jordan with row pivoting, ONE COLUMN PER LANE: lanes 0..5 hold the columns of A, lane 6 the right-hand side -g, six doubles each,
// in registers.  The pivot search of column k is local to lane k (it owns the whole column), a row exchange is local to every lane,
// and what crosses lanes per pivot is the pivot row index, the pivot and the five multipliers -- v_readlane broadcasts from lane k,
// no LDS crossbar, no division per element (one reciprocal, computed by every lane).  ~50 instructions per pivot on a 12-register
// state, against the lane-per-element form above with its six ds_bpermute round trips and two divisions per pivot (3.5 us per Newton
// step at the end of every fused NDT launch, scripts/dbg_close_stamps.py) and against a whole-system-per-lane form, whose 84
// registers spill under the fused kernel's 128-VGPR budget (3.1 us).  Same pivoting rule as above; the pivot row is scaled by the
// reciprocal of the pivot, so the result differs from the form above in the last bits (both within cond * 1e-16 of the exact
// solution).  A: row-major 6 x 6 (any memory), g: the system solved is (A + diag_add I) x = -g; every lane receives x.
__device__ __forceinline__ void gj_solve6_columns(const double* __restrict__ A, const double* __restrict__ g, double* x, double* rcond, const double diag_add = 0.0) {
  const int lane = threadIdx.x & 63;
  const int col = lane < 7 ? lane : 6;   // lanes 7.. mirror lane 6 (idle copies)
  double a[6];
#pragma unroll
  for (int r = 0; r < 6; r++) a[r] = (col < 6) ? (r == col ? A[r * 6 + col] + diag_add : A[r * 6 + col]) : -g[r];
  double pmax = 0.0, pmin = DBL_MAX;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    // lane k: the row (>= k) with the largest |entry| of its column
    int best = k;
    double bv = fabs(a[k]);
#pragma unroll
    for (int r = k + 1; r < 6; r++) {
      const double v = fabs(a[r]);
      if (v > bv) { bv = v; best = r; }
    }
    best = __builtin_amdgcn_readlane(best, k);
#pragma unroll
    for (int r = k + 1; r < 6; r++)
      if (best == r) { const double t = a[k]; a[k] = a[r]; a[r] = t; }   // wave-uniform: every lane exchanges the same two rows
    const double piv = readlane_f64(a[k], k);
    pmax = fmax(pmax, fabs(piv));
    pmin = fmin(pmin, fabs(piv));
    if (piv != 0.0) {
      const double inv = 1.0 / piv;
      double f[6];
#pragma unroll
      for (int r = 0; r < 6; r++) f[r] = (r == k) ? 0.0 : readlane_f64(a[r], k);   // column k before the update = the multipliers * piv
      a[k] *= inv;
#pragma unroll
      for (int r = 0; r < 6; r++)
        if (r != k) a[r] -= f[r] * a[k];
    }
  }
#pragma unroll
  for (int r = 0; r < 6; r++) x[r] = readlane_f64(a[r], 6);
  *rcond = (pmax > 0) ? pmin / pmax : 0.0;
}

// Pseudo-inverse solve through a one-sided Jacobi SVD with Eigen::JacobiSVD's default rank threshold
// (6 * eps * s_max).  Slow path: only taken when the elimination above meets a (numerically) singular Hessian -- and the
// Newton solve of the NDT validation modes (ndt_strict_order), which run it with skip_tol 1e-17 / 60 sweeps: then every
// operation is the one the CPU checker executes, individually rounded, so the step comes out bit-identical.
// U and V live in LDS: the callers run this with every lane of ONE wave computing the same values, so one copy serves the
// wave (identical stores to one address are harmless), and the 144 VGPRs a register copy costs -- which would set the register
// allocation, hence the occupancy, of every kernel this function is linked into -- are not needed.
__device__ __forceinline__ void svd_solve6_dev(const double* A, const double* b, double* x, const double skip_tol = 4e-16, const int max_sweeps = 40) {
#pragma clang fp contract(off)
  __shared__ double svd_ws[72];
  double* U = svd_ws;
  double* V = svd_ws + 36;
  for (int i = 0; i < 36; i++) { U[i] = A[i]; V[i] = (i % 7 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < max_sweeps; sweep++) {
    bool rotated = false;
    for (int p = 0; p < 5; p++)
      for (int q = p + 1; q < 6; q++) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < 6; k++) { al += U[k * 6 + p] * U[k * 6 + p]; be += U[k * 6 + q] * U[k * 6 + q]; ga += U[k * 6 + p] * U[k * 6 + q]; }
        if (ga == 0.0 || fabs(ga) <= skip_tol * sqrt(al * be)) continue;
        rotated = true;
        const double ze = (be - al) / (2.0 * ga);
        const double t = (ze >= 0 ? 1.0 : -1.0) / (fabs(ze) + sqrt(1.0 + ze * ze));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 6; k++) {
          const double up = U[k * 6 + p], uq = U[k * 6 + q];
          U[k * 6 + p] = c * up - s * uq; U[k * 6 + q] = s * up + c * uq;
          const double vp = V[k * 6 + p], vq = V[k * 6 + q];
          V[k * 6 + p] = c * vp - s * vq; V[k * 6 + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sv[6], smax = 0;
  for (int j = 0; j < 6; j++) {
    double s2 = 0;
    for (int k = 0; k < 6; k++) s2 += U[k * 6 + j] * U[k * 6 + j];
    sv[j] = sqrt(s2);
    smax = fmax(smax, sv[j]);
  }
  const double thr = fmax(smax * 6.0 * DBL_EPSILON, DBL_MIN);
  for (int i = 0; i < 6; i++) x[i] = 0.0;
  for (int j = 0; j < 6; j++) {
    if (!(sv[j] > thr)) continue;
    double ub = 0;
    for (int k = 0; k < 6; k++) ub += U[k * 6 + j] * b[k];
    const double coef = ub / (sv[j] * sv[j]);
    for (int i = 0; i < 6; i++) x[i] += V[i * 6 + j] * coef;
  }
}


// The same solve with U and V in REGISTERS: every loop over rows / column pairs is unrolled so that all subscripts are static
// (144 VGPRs).  For the stand-alone solve launch of the NDT validation modes only: there the LDS copy above makes every rotation a
// chain of ~30 LDS round trips (190 us per Newton step) and nothing else shares the kernel's register budget.  Same operations in
// the same order as svd_solve6_dev.
__device__ __forceinline__ void svd_solve6_regs_dev(const double* A, const double* b, double* x, const double skip_tol, const int max_sweeps) {
#pragma clang fp contract(off)
  double U[36], V[36];
#pragma unroll
  for (int i = 0; i < 36; i++) { U[i] = A[i]; V[i] = (i % 7 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < max_sweeps; sweep++) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 5; p++)
#pragma unroll
      for (int q = p + 1; q < 6; q++) {
        double al = 0, be = 0, ga = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) { al += U[k * 6 + p] * U[k * 6 + p]; be += U[k * 6 + q] * U[k * 6 + q]; ga += U[k * 6 + p] * U[k * 6 + q]; }
        if (ga == 0.0 || fabs(ga) <= skip_tol * sqrt(al * be)) continue;
        rotated = true;
        const double ze = (be - al) / (2.0 * ga);
        const double t = (ze >= 0 ? 1.0 : -1.0) / (fabs(ze) + sqrt(1.0 + ze * ze));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const double up = U[k * 6 + p], uq = U[k * 6 + q];
          U[k * 6 + p] = c * up - s * uq; U[k * 6 + q] = s * up + c * uq;
          const double vp = V[k * 6 + p], vq = V[k * 6 + q];
          V[k * 6 + p] = c * vp - s * vq; V[k * 6 + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sv[6], smax = 0;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double s2 = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) s2 += U[k * 6 + j] * U[k * 6 + j];
    sv[j] = sqrt(s2);
    smax = fmax(smax, sv[j]);
  }
  const double thr = fmax(smax * 6.0 * DBL_EPSILON, DBL_MIN);
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] = 0.0;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    if (!(sv[j] > thr)) continue;
    double ub = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) ub += U[k * 6 + j] * b[k];
    const double coef = ub / (sv[j] * sv[j]);
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] += V[i * 6 + j] * coef;
  }
}

// x = A^-1 b with the pseudo-inverse fallback Eigen's JacobiSVD::solve would give on a singular A (wave-uniform)
__device__ __forceinline__ void solve6_wave(const double* A, const double* b, double* x) {
  double rc;
  gj_solve6_wave(A, b, x, &rc);
  if (!(rc > 1e-13)) svd_solve6_dev(A, b, x);
}

// x = -(H + lambda I)^-1 g: the Gauss-Newton / Levenberg-Marquardt step of the GICP optimisers, through the column-per-lane elimination
__device__ __forceinline__ void solve6_step(const double* H, const double* g, const double lambda, double* x) {
  double rc;
  gj_solve6_columns(H, g, x, &rc, lambda);
  if (!(rc > 1e-13)) {
    double Hl[36], nb[6];
    for (int k = 0; k < 36; k++) Hl[k] = H[k];
    for (int k = 0; k < 6; k++) { Hl[k * 6 + k] += lambda; nb[k] = -g[k]; }
    svd_solve6_dev(Hl, nb, x);
  }
}

}  // namespace dgs
