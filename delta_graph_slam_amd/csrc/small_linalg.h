// Small double-precision linear algebra executed by one lane (3x3 inverse, symmetric 3x3 eigen-decomposition).
#pragma once
#include <hip/hip_runtime.h>

namespace dgs {

__device__ inline bool inv3_d(const double* A, double* Ai) {
#pragma clang fp contract(off)
  const double c00 = A[4] * A[8] - A[5] * A[7];
  const double c01 = A[5] * A[6] - A[3] * A[8];
  const double c02 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  const double id = 1.0 / det;
  Ai[0] = c00 * id;
  Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id;
  Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  Ai[3] = c01 * id;
  Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id;
  Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  Ai[6] = c02 * id;
  Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id;
  Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  return det != 0.0;
}

// symmetric 3x3 eigen-decomposition (cyclic Jacobi with full two-sided rotations, lower triangle authoritative), ascending
// eigenvalues.  The sequence of IEEE operations is fixed (no contraction) and is the one the CPU checker of this repository
// executes as well, so the voxel table can be compared with it bit for bit (dgs_params.ndt_strict_order).
__device__ inline void sym_eig3_d(const double* Ain, double* ev, double* V) {
#pragma clang fp contract(off)
  double A[9];
  A[0] = Ain[0]; A[4] = Ain[4]; A[8] = Ain[8];
  A[3] = A[1] = Ain[3]; A[6] = A[2] = Ain[6]; A[7] = A[5] = Ain[7];
  double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double dia = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-34 * dia || off == 0.0) break;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int q = p + 1; q < 3; q++) {
        const double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; k++) {  // A <- A * J
          const double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {  // A <- J^T * A
          const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double vkp = v[k * 3 + p], vkq = v[k * 3 + q];
          v[k * 3 + p] = c * vkp - s * vkq;
          v[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  const double e[3] = {A[0], A[4], A[8]};
  // stable ascending order of three values (ties keep their index order)
  int o0 = 0, o1 = 1, o2 = 2;
  if (e[o1] < e[o0]) { const int t = o0; o0 = o1; o1 = t; }
  if (e[o2] < e[o1]) { const int t = o1; o1 = o2; o2 = t; }
  if (e[o1] < e[o0]) { const int t = o0; o0 = o1; o1 = t; }
  const int o[3] = {o0, o1, o2};
  for (int k = 0; k < 3; k++) {
    ev[k] = e[o[k]];
    for (int r = 0; r < 3; r++) V[r * 3 + k] = v[r * 3 + o[k]];
  }
}

// Eigen::SelfAdjointEigenSolver<Matrix3d>::compute(A, ComputeEigenvectors) as Eigen 3.3.x executes it (what pclomp's VoxelGridCovariance runs on every
// voxel covariance): the lower triangle divided by its largest |entry|, the written-out 3 x 3 Householder tridiagonalisation, implicit symmetric QR
// steps with Wilkinson's shift (Givens rotations by makeGivens, the eigenvector matrix updated on the right) until every sub-diagonal entry is
// negligible, eigenvalues scaled back, selection sort ascending with the columns swapped along.  The same sequence of IEEE operations as the CPU
// checker's restatement (contraction off): the voxel table stays comparable bit for bit.  dgs_params.ndt_cov_eigensolver = 1 (default).
__device__ inline void eigen_selfadjoint3_d(const double* Ain, double* ev, double* V) {
#pragma clang fp contract(off)
  double m00 = Ain[0], m10 = Ain[3], m11 = Ain[4], m20 = Ain[6], m21 = Ain[7], m22 = Ain[8];
  double scale = fabs(m00);
  scale = fabs(m10) > scale ? fabs(m10) : scale;
  scale = fabs(m11) > scale ? fabs(m11) : scale;
  scale = fabs(m20) > scale ? fabs(m20) : scale;
  scale = fabs(m21) > scale ? fabs(m21) : scale;
  scale = fabs(m22) > scale ? fabs(m22) : scale;
  if (scale == 0.0) scale = 1.0;
  m00 /= scale; m10 /= scale; m11 /= scale; m20 /= scale; m21 /= scale; m22 /= scale;
  double diag[3], sub[2], Q[9];
  const double tiny = 2.2250738585072014e-308;   // DBL_MIN
  diag[0] = m00;
  const double v1norm2 = m20 * m20;
  if (v1norm2 <= tiny) {
    diag[1] = m11; diag[2] = m22; sub[0] = m10; sub[1] = m21;
    Q[0] = 1; Q[1] = 0; Q[2] = 0; Q[3] = 0; Q[4] = 1; Q[5] = 0; Q[6] = 0; Q[7] = 0; Q[8] = 1;
  } else {
    const double beta = sqrt(m10 * m10 + v1norm2);
    const double invBeta = 1.0 / beta;
    const double m01 = m10 * invBeta, m02 = m20 * invBeta;
    const double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
    diag[1] = m11 + m02 * q;
    diag[2] = m22 - m02 * q;
    sub[0] = beta;
    sub[1] = m21 - m01 * q;
    Q[0] = 1; Q[1] = 0; Q[2] = 0; Q[3] = 0; Q[4] = m01; Q[5] = m02; Q[6] = 0; Q[7] = m02; Q[8] = -m01;
  }
  const int n = 3, maxIterations = 30;
  int end = n - 1, start = 0, iter = 0;
  const double precision = 2.0 * 2.220446049250313e-16;
  while (end > 0) {
    for (int i = start; i < end; ++i)
      if (fabs(sub[i]) <= (fabs(diag[i]) + fabs(diag[i + 1])) * precision || fabs(sub[i]) <= tiny) sub[i] = 0.0;
    while (end > 0 && sub[end - 1] == 0.0) end--;
    if (end <= 0) break;
    iter++;
    if (iter > maxIterations * n) break;
    start = end - 1;
    while (start > 0 && sub[start - 1] != 0.0) start--;
    const double td = (diag[end - 1] - diag[end]) * 0.5;
    const double e = sub[end - 1];
    double mu = diag[end];
    if (td == 0.0) {
      mu -= fabs(e);
    } else {
      const double e2 = e * e;
      const double at = fabs(td), ae = fabs(e);
      const double p = at > ae ? at : ae;
      double hh = 0.0;
      if (p != 0.0) {
        const double qp = (at > ae ? ae : at) / p;
        hh = p * sqrt(1.0 + qp * qp);
      }
      if (e2 == 0.0) mu -= (e / (td + (td > 0.0 ? 1.0 : -1.0))) * (e / hh);
      else mu -= e2 / (td + (td > 0.0 ? hh : -hh));
    }
    double x = diag[start] - mu;
    double z = sub[start];
    for (int k = start; k < end; ++k) {
      double c, sn;
      if (z == 0.0) {
        c = x < 0.0 ? -1.0 : 1.0; sn = 0.0;
      } else if (x == 0.0) {
        c = 0.0; sn = z < 0.0 ? 1.0 : -1.0;
      } else if (fabs(x) > fabs(z)) {
        const double t = z / x;
        double u = sqrt(1.0 + t * t);
        if (x < 0.0) u = -u;
        c = 1.0 / u; sn = -t * c;
      } else {
        const double t = x / z;
        double u = sqrt(1.0 + t * t);
        if (z < 0.0) u = -u;
        sn = -1.0 / u; c = -t * sn;
      }
      const double sdk = sn * diag[k] + c * sub[k];
      const double dkp1 = sn * sub[k] + c * diag[k + 1];
      diag[k] = c * (c * diag[k] - sn * sub[k]) - sn * (c * sub[k] - sn * diag[k + 1]);
      diag[k + 1] = sn * sdk + c * dkp1;
      sub[k] = c * sdk - sn * dkp1;
      if (k > start) sub[k - 1] = c * sub[k - 1] - sn * z;
      x = sub[k];
      if (k < end - 1) {
        z = -sn * sub[k + 1];
        sub[k + 1] = c * sub[k + 1];
      }
      for (int i = 0; i < 3; i++) {
        const double xi = Q[i * 3 + k], yi = Q[i * 3 + k + 1];
        Q[i * 3 + k] = c * xi - sn * yi;
        Q[i * 3 + k + 1] = sn * xi + c * yi;
      }
    }
  }
  for (int i = 0; i < 3; i++) diag[i] *= scale;
  for (int i = 0; i < n - 1; ++i) {
    int k = 0;
    for (int j = 1; j < n - i; j++)
      if (diag[i + j] < diag[i + k]) k = j;
    if (k > 0) {
      const double t = diag[i]; diag[i] = diag[k + i]; diag[k + i] = t;
      for (int r = 0; r < 3; r++) { const double u = Q[r * 3 + i]; Q[r * 3 + i] = Q[r * 3 + k + i]; Q[r * 3 + k + i] = u; }
    }
  }
  for (int i = 0; i < 3; i++) ev[i] = diag[i];
  for (int i = 0; i < 9; i++) V[i] = Q[i];
}

// Eigen::JacobiSVD<Matrix3d>(A, ComputeFullU | ComputeFullV) as Eigen 3.3 executes it for a square real matrix (two-sided Jacobi on A / max|A|, sweeps
// over (p, q), q < p, real_2x2_jacobi_svd + makeJacobi, signs, descending order): what fast_gicp's calculate_covariances calls on every
// neighbourhood covariance.  The same sequence of IEEE operations as the CPU checker's generic restatement (the 6 x 6 Newton solve uses it too, laid
// out across a wave in solve6.h), contraction off.  U, V row-major, sv descending.  dgs_params.gicp_cov_jacobi_svd = 1 (default).
__device__ inline void jacobi_svd3_d(const double* A, double* U, double* V, double* sv) {
#pragma clang fp contract(off)
  constexpr int N = 3;
  const double precision = 2.0 * 2.220446049250313e-16, consider_as_zero = 2.2250738585072014e-308;
  double scale = 0.0;
  for (int i = 0; i < N * N; i++) { const double a = fabs(A[i]); if (a > scale) scale = a; }
  if (scale == 0.0) scale = 1.0;
  double W[N * N];
  for (int i = 0; i < N * N; i++) { W[i] = A[i] / scale; U[i] = V[i] = (i % (N + 1) == 0) ? 1.0 : 0.0; }
  double max_diag = 0.0;
  for (int i = 0; i < N; i++) { const double a = fabs(W[i * (N + 1)]); if (a > max_diag) max_diag = a; }
  int sweeps = 0;
  bool finished = false;
  while (!finished && sweeps < 64) {
    finished = true;
    sweeps++;
#pragma unroll
    for (int p = 1; p < N; p++)
#pragma unroll
      for (int q = 0; q < p; q++) {
        const double pm = precision * max_diag;
        const double threshold = consider_as_zero > pm ? consider_as_zero : pm;
        if (!(fabs(W[p * N + q]) > threshold || fabs(W[q * N + p]) > threshold)) continue;
        finished = false;
        double m00 = W[p * N + p], m01 = W[p * N + q], m10 = W[q * N + p], m11 = W[q * N + q];
        const double t = m00 + m11, d = m10 - m01;
        double r1c, r1s;
        if (fabs(d) < consider_as_zero) { r1s = 0.0; r1c = 1.0; }
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
          if (deno < consider_as_zero) { jrc = 1.0; jrs = 0.0; }
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
        if (!(jlc == 1.0 && jls == 0.0)) {
#pragma unroll
          for (int i = 0; i < N; i++) {
            const double xi = W[p * N + i], yi = W[q * N + i];
            W[p * N + i] = jlc * xi + jls * yi;
            W[q * N + i] = -jls * xi + jlc * yi;
          }
#pragma unroll
          for (int i = 0; i < N; i++) {
            const double xi = U[i * N + p], yi = U[i * N + q];
            U[i * N + p] = jlc * xi + jls * yi;
            U[i * N + q] = -jls * xi + jlc * yi;
          }
        }
        if (!(jrc == 1.0 && -jrs == 0.0)) {
          const double c = jrc, s = -jrs;
#pragma unroll
          for (int i = 0; i < N; i++) {
            const double xi = W[i * N + p], yi = W[i * N + q];
            W[i * N + p] = c * xi + s * yi;
            W[i * N + q] = -s * xi + c * yi;
          }
#pragma unroll
          for (int i = 0; i < N; i++) {
            const double xi = V[i * N + p], yi = V[i * N + q];
            V[i * N + p] = c * xi + s * yi;
            V[i * N + q] = -s * xi + c * yi;
          }
        }
        const double app = fabs(W[p * N + p]), aqq = fabs(W[q * N + q]);
        const double mx = app < aqq ? aqq : app;
        if (max_diag < mx) max_diag = mx;
      }
  }
  for (int i = 0; i < N; i++) {
    const double a = W[i * (N + 1)];
    sv[i] = fabs(a);
    if (a < 0.0)
      for (int k = 0; k < N; k++) U[k * N + i] = -U[k * N + i];
  }
  for (int i = 0; i < N; i++) sv[i] *= scale;
  for (int i = 0; i < N; i++) {
    int pos = 0;
    double best = sv[i];
    for (int k = 1; k < N - i; k++)
      if (sv[i + k] > best) { best = sv[i + k]; pos = k; }
    if (best == 0.0) break;
    if (pos) {
      pos += i;
      { const double t = sv[i]; sv[i] = sv[pos]; sv[pos] = t; }
      for (int k = 0; k < N; k++) {
        { const double t = U[k * N + i]; U[k * N + i] = U[k * N + pos]; U[k * N + pos] = t; }
        { const double t = V[k * N + i]; V[k * N + i] = V[k * N + pos]; V[k * N + pos] = t; }
      }
    }
  }
}

}  // namespace dgs
