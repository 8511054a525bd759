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

// ---- Eigen::JacobiSVD<Matrix6d>(A, ComputeFullU | ComputeFullV).solve(b), lane-parallel (dgs_params.ndt_newton_solver = 1) --------
// The operation sequence of the CPU checker's restatement of Eigen's two-sided Jacobi (steps 1-4 of JacobiSVD::compute, SVDBase::rank,
// the solve), every operation individually rounded -- so the Newton step of the upstream evaluation orders comes out bit-identical --
// but laid out across ONE wave instead of run by every lane:
//   lanes 0..5   column j of the work matrix W,   lanes 8..13  column j of U,   lanes 16..21  column j of V     (six doubles each);
//   the 2x2 block of a pair (p, q) reaches every lane through v_readlane; the rotation angles (three square roots, six divisions: one
//   dependent chain) are computed by every lane alike; W <- J_left W is local to the W lanes (rows p, q of their columns); the three
//   column rotations W J_right, U J_left, V J_right are ONE exchange of columns p <-> q inside each 8-lane group (lane ^ (p ^ q)) and
//   three multiply-adds per entry, every group with its own (c, s).
// A one-wave solve executes ~45 rotations of ~230 instructions (5 sweeps on the NDT Hessians) where rounds 2-3's one-sided Jacobi ran 60
// sweeps of dot products on every lane (~85 k instructions, 95-190 us per Newton step).
// Must be called by all 64 lanes of a wave with identical arguments; A row-major 6 x 6 and b in LDS or any memory; every lane receives x.
template <int P, int Q>
__device__ __forceinline__ void jsvd_pair(double (&col)[6], double& max_diag, bool& finished, const int grp, const int j) {
#pragma clang fp contract(off)
  const double kMin = DBL_MIN;
  const double w_pq = readlane_f64(col[P], Q), w_qp = readlane_f64(col[Q], P);   // W(p, q) lives in column q, row p
  const double pm = (2.0 * DBL_EPSILON) * max_diag;
  const double threshold = kMin > pm ? kMin : pm;
  if (!(fabs(w_pq) > threshold || fabs(w_qp) > threshold)) return;
  finished = false;
  // ---- real_2x2_jacobi_svd
  double m00 = readlane_f64(col[P], P), m01 = w_pq, m10 = w_qp, m11 = readlane_f64(col[Q], Q);
  const double t = m00 + m11, d = m10 - m01;
  double r1c, r1s;
  if (fabs(d) < kMin) { r1s = 0.0; r1c = 1.0; }
  else {
    const double u = t / d;
    const double tmp = sqrt(1.0 + u * u);
    r1s = 1.0 / tmp;
    r1c = u / tmp;
  }
  if (!(r1c == 1.0 && r1s == 0.0)) {
    const double x0 = m00, y0 = m10, x1 = m01, y1 = m11;
    m00 = r1c * x0 + r1s * y0; m10 = -r1s * x0 + r1c * y0;
    m01 = r1c * x1 + r1s * y1; m11 = -r1s * x1 + r1c * y1;
  }
  double jrc, jrs;
  {
    const double deno = 2.0 * fabs(m01);
    if (deno < kMin) { jrc = 1.0; jrs = 0.0; }
    else {
      const double tau = (m00 - m11) / deno;
      const double w = sqrt(tau * tau + 1.0);
      const double tt = (tau > 0.0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
      const double sign_t = tt > 0.0 ? 1.0 : -1.0;
      const double nn = 1.0 / sqrt(tt * tt + 1.0);
      jrs = -sign_t * (m01 / fabs(m01)) * fabs(tt) * nn;
      jrc = nn;
    }
  }
  const double jtc = jrc, jts = -jrs;
  const double jlc = r1c * jtc - r1s * jts;
  const double jls = r1c * jts + r1s * jtc;
  const bool left_id = (jlc == 1.0 && jls == 0.0), right_id = (jrc == 1.0 && -jrs == 0.0);
  // ---- W.applyOnTheLeft(p, q, j_left): rows p, q of every W column
  if (grp == 0 && !left_id) {
    const double xi = col[P], yi = col[Q];
    col[P] = jlc * xi + jls * yi;
    col[Q] = -jls * xi + jlc * yi;
  }
  // ---- the three column rotations: x' = c x + s y (column p), y' = -s x + c y (column q)
  //      W, V: (c, s) = j_right.transpose() = (jrc, -jrs);   U: (c, s) = j_left
  const double c = (grp == 1) ? jlc : jrc, s = (grp == 1) ? jls : -jrs;
  const bool skip = (grp == 1) ? left_id : right_id;
  const bool is_p = (j == P), is_q = (j == Q);
  const double sg = is_p ? s : -s;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double other = __shfl_xor(col[k], P ^ Q, 64);
    const double v = c * col[k] + sg * other;    // column p: c x + s y;   column q: c y + (-s) x = -s x + c y
    col[k] = ((is_p || is_q) && !skip) ? v : col[k];
  }
  const double app = fabs(readlane_f64(col[P], P)), aqq = fabs(readlane_f64(col[Q], Q));
  const double mx = app < aqq ? aqq : app;
  if (max_diag < mx) max_diag = mx;
}

__device__ __forceinline__ void jsvd_solve6_wave(const double* A, const double* b, double* x, int* sweeps_out = nullptr) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int grp = min(lane >> 3, 2);           // 0: W, 1: U, 2: V (lanes 24.. mirror V lanes: idle copies)
  const int j = lane & 7;                      // column; 6 and 7 hold zeros
  const int jc = j < 6 ? j : 0;
  // ---- step 1: scale = max |A|, W = A / scale, U = V = I
  double col[6];
#pragma unroll
  for (int k = 0; k < 6; k++) col[k] = A[k * 6 + jc];
  double amax = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) { const double a = fabs(col[k]); if (a > amax) amax = a; }
  double scale = 0.0;
#pragma unroll
  for (int l = 0; l < 6; l++) { const double a = readlane_f64(amax, l); if (a > scale) scale = a; }
  if (scale == 0.0) scale = 1.0;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double w = col[k] / scale;
    col[k] = (j >= 6) ? 0.0 : (grp == 0 ? w : (k == j ? 1.0 : 0.0));
  }
  // ---- step 2: sweeps
  double dj = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) dj = (k == j) ? fabs(col[k]) : dj;
  double max_diag = 0.0;
#pragma unroll
  for (int l = 0; l < 6; l++) { const double a = readlane_f64(dj, l); if (a > max_diag) max_diag = a; }
  bool finished = false;
  int sweeps = 0;
  while (!finished && sweeps < 64) {
    finished = true;
    sweeps++;
    jsvd_pair<1, 0>(col, max_diag, finished, grp, j);
    jsvd_pair<2, 0>(col, max_diag, finished, grp, j);
    jsvd_pair<2, 1>(col, max_diag, finished, grp, j);
    jsvd_pair<3, 0>(col, max_diag, finished, grp, j);
    jsvd_pair<3, 1>(col, max_diag, finished, grp, j);
    jsvd_pair<3, 2>(col, max_diag, finished, grp, j);
    jsvd_pair<4, 0>(col, max_diag, finished, grp, j);
    jsvd_pair<4, 1>(col, max_diag, finished, grp, j);
    jsvd_pair<4, 2>(col, max_diag, finished, grp, j);
    jsvd_pair<4, 3>(col, max_diag, finished, grp, j);
    jsvd_pair<5, 0>(col, max_diag, finished, grp, j);
    jsvd_pair<5, 1>(col, max_diag, finished, grp, j);
    jsvd_pair<5, 2>(col, max_diag, finished, grp, j);
    jsvd_pair<5, 3>(col, max_diag, finished, grp, j);
    jsvd_pair<5, 4>(col, max_diag, finished, grp, j);
  }
  if (sweeps_out) *sweeps_out = sweeps;
  // ---- step 3: singular values |W_jj| * scale; U columns negated where W_jj < 0
  double wjj = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) wjj = (k == j) ? col[k] : wjj;
  wjj = __shfl(wjj, jc, 64);                   // every group: the diagonal entry of ITS column index, from the W lane
  const double svj = fabs(wjj) * scale;
  if (grp == 1 && wjj < 0.0) {
#pragma unroll
    for (int k = 0; k < 6; k++) col[k] = -col[k];
  }
  // ---- step 4: descending order (selection by the first maximum of the tail), kept as a permutation: sorted slot r <- column ord[r]
  double sv[6];
  int ord[6];
#pragma unroll
  for (int l = 0; l < 6; l++) { sv[l] = readlane_f64(svj, l); ord[l] = l; }
  int nonzero = 6;
  bool stop = false;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    int pos = i;
    double best = sv[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) if (sv[k] > best) { best = sv[k]; pos = k; }
    if (!stop && best == 0.0) { nonzero = i; stop = true; }
    if (!stop) {
#pragma unroll
      for (int k = i + 1; k < 6; k++)
        if (k == pos) { const double ts = sv[i]; sv[i] = sv[k]; sv[k] = ts; const int to = ord[i]; ord[i] = ord[k]; ord[k] = to; }
    }
  }
  // ---- rank and solve: tmp_r = (1 / s_r) (U(:, ord r) . b),  x = sum over r < rank, in sorted order, of V(:, ord r) tmp_r
  const double pt = sv[0] * (6.0 * DBL_EPSILON);
  const double premultiplied = pt > DBL_MIN ? pt : DBL_MIN;
  int rank = 0;
  {
    int r = nonzero - 1;
    bool going = true;
#pragma unroll
    for (int k = 5; k >= 0; k--) {
      if (going && k <= r) {
        if (sv[k] < premultiplied) r = k - 1; else going = false;
      }
    }
    rank = r + 1;
  }
  // U lane j: (1 / s_j) * (column . b), with s_j the singular value of column j (before sorting it is svj)
  double ub = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) ub = (k == 0) ? col[k] * b[k] : ub + col[k] * b[k];
  const double tj = (1.0 / svj) * ub;          // meaningful in U lanes
  const double tv = __shfl(tj, 8 + jc, 64);    // V lane j receives tmp of column j
  double prod[6];
#pragma unroll
  for (int k = 0; k < 6; k++) prod[k] = col[k] * tv;   // V lanes: V(k, j) * tmp_j
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] = 0.0;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    if (r < rank) {
      const int src = 16 + ord[r];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const double term = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(prod[i]), src), __builtin_amdgcn_readlane(__double2loint(prod[i]), src));
        x[i] = (r == 0) ? term : x[i] + term;
      }
    }
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
