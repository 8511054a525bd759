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

}  // namespace dgs
