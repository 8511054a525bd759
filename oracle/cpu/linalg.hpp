// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/README.md).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call this.
//
// Small dense f64 linear algebra used by the CPU restatement of NDT_OMP / FAST_GICP.  It replaces the
// Eigen calls the upstream libraries make (Eigen is absent from this image):
//   SelfAdjointEigenSolver<Matrix3d>   -> sym_eig3   (cyclic Jacobi, ascending eigenvalues)
//   Matrix3d::inverse()                -> inv3       (cofactor form, as Eigen's fixed-size 3x3)
//   JacobiSVD<Matrix<double,6,6>>::solve -> svd_solve6 (one-sided Jacobi SVD, Eigen's default rank threshold)
//   LDLT<Matrix<double,6,6>>::solve    -> ldlt_solve6 (Bunch-Kaufman-free diagonal-pivoted LDL^T, as Eigen)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <algorithm>

namespace orc {

// ---- 3x3 (row-major double[9]) ------------------------------------------------------------------
inline void mat3_mul(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
  std::memcpy(C, T, sizeof(T));
}

inline bool inv3(const double* A, double* Ai) {
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

// Symmetric 3x3 eigen-decomposition, cyclic Jacobi on the lower triangle's symmetric completion.
// evals ascending, evecs column k (V[r*3+k]) is the unit eigenvector of evals[k].
inline void sym_eig3(const double* Ain, double* evals, double* V) {
  double A[9];
  // SelfAdjointEigenSolver reads the lower triangle only
  A[0] = Ain[0]; A[4] = Ain[4]; A[8] = Ain[8];
  A[3] = A[1] = Ain[3]; A[6] = A[2] = Ain[6]; A[7] = A[5] = Ain[7];
  for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double dia = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-34 * dia || off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        const double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {  // A <- A * J
          const double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {  // A <- J^T * A
          const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  double ev[3] = {A[0], A[4], A[8]};
  int idx[3] = {0, 1, 2};
  std::sort(idx, idx + 3, [&](int a, int b) { return ev[a] < ev[b]; });
  double Vs[9];
  for (int k = 0; k < 3; k++) {
    evals[k] = ev[idx[k]];
    for (int r = 0; r < 3; r++) Vs[r * 3 + k] = V[r * 3 + idx[k]];
  }
  std::memcpy(V, Vs, sizeof(Vs));
}

// ---- exp(float) with a platform-independent value ------------------------------------------------------
// Upstream calls std::exp(float), whose last bit depends on the libm at hand (glibc's expf is within 0.502 ulp and
// even differs between its own FMA / non-FMA ifunc variants).  The restatement therefore defines it as this fixed
// sequence of IEEE double operations (no contraction), rounded once to float: accurate to ~1e-16 before the
// rounding, i.e. the correctly rounded expf except for ~1e-9 of the arguments, and bit-reproducible on any IEEE
// machine -- the device library carries the same sequence, so strict-order evaluations can be compared bit for bit.
inline float det_expf(float xf) {
  const double x = static_cast<double>(xf);
  if (x != x) return xf;
  if (x < -104.0) return 0.0f;  // below half the smallest subnormal float
  if (x > 89.0) return std::numeric_limits<float>::infinity();
  const double kd = std::floor(x * 1.4426950408889634 + 0.5);  // round(x / ln 2)
  const double r = (x - kd * 0x1.62e42fefa38p-1) - kd * 0x1.ef35793c7673p-45;  // ln 2 split hi / lo; |r| <= 0.3466
  double p = 1.0 / 6227020800.0;  // Taylor to r^13 / 13!: truncation 6e-18 relative
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  const int64_t k = static_cast<int64_t>(kd);
  const uint64_t bits = static_cast<uint64_t>(k + 1023) << 52;  // 2^k, k in [-151, 129]: a normal double
  double s;
  std::memcpy(&s, &bits, sizeof(s));
  return static_cast<float>(p * s);
}

// ---- 6x6 (row-major double[36]) -----------------------------------------------------------------
// x = pinv(A) b through a one-sided (Hestenes) Jacobi SVD; singular values <= 6*eps*s_max are dropped,
// which is Eigen::JacobiSVD's default threshold (SVDBase::threshold(): diagSize * epsilon).
inline void svd_solve6(const double* A, const double* b, double* x) {
  const int n = 6;
  double U[36], V[36];
  std::memcpy(U, A, sizeof(U));  // columns of U converge to u_k * s_k
  for (int i = 0; i < 36; i++) V[i] = (i % 7 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < n; k++) {
          alpha += U[k * n + p] * U[k * n + p];
          beta += U[k * n + q] * U[k * n + q];
          gamma += U[k * n + p] * U[k * n + q];
        }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-17 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < n; k++) {
          const double up = U[k * n + p], uq = U[k * n + q];
          U[k * n + p] = c * up - s * uq;
          U[k * n + q] = s * up + c * uq;
          const double vp = V[k * n + p], vq = V[k * n + q];
          V[k * n + p] = c * vp - s * vq;
          V[k * n + q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double sv[6], smax = 0;
  for (int j = 0; j < n; j++) {
    double s2 = 0;
    for (int k = 0; k < n; k++) s2 += U[k * n + j] * U[k * n + j];
    sv[j] = std::sqrt(s2);
    smax = std::max(smax, sv[j]);
  }
  const double thr = std::max(smax * 6.0 * std::numeric_limits<double>::epsilon(), std::numeric_limits<double>::min());
  for (int i = 0; i < n; i++) x[i] = 0.0;
  for (int j = 0; j < n; j++) {
    if (!(sv[j] > thr)) continue;
    double ub = 0;  // (u_j . b) / s_j  with u_j = U[:,j]/s_j
    for (int k = 0; k < n; k++) ub += U[k * n + j] * b[k];
    const double coef = ub / (sv[j] * sv[j]);
    for (int i = 0; i < n; i++) x[i] += V[i * n + j] * coef;
  }
}

// LDL^T with symmetric diagonal pivoting (the algorithm behind Eigen::LDLT), solve A x = b.
inline void ldlt_solve6(const double* Ain, const double* b, double* x) {
  const int n = 6;
  double A[36];
  std::memcpy(A, Ain, sizeof(A));
  int perm[6] = {0, 1, 2, 3, 4, 5};
  for (int k = 0; k < n; k++) {
    int piv = k;
    double best = std::fabs(A[k * n + k]);
    for (int i = k + 1; i < n; i++)
      if (std::fabs(A[i * n + i]) > best) { best = std::fabs(A[i * n + i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < n; j++) std::swap(A[k * n + j], A[piv * n + j]);
      for (int i = 0; i < n; i++) std::swap(A[i * n + k], A[i * n + piv]);
      std::swap(perm[k], perm[piv]);
    }
    const double d = A[k * n + k];
    if (d == 0.0) continue;
    double col[6];
    for (int i = k + 1; i < n; i++) col[i] = A[i * n + k];
    for (int i = k + 1; i < n; i++) {
      const double l = col[i] / d;
      for (int j = k + 1; j <= i; j++) {
        A[i * n + j] -= l * col[j];
        A[j * n + i] = A[i * n + j];
      }
      A[i * n + k] = l;
    }
  }
  double y[6];
  for (int i = 0; i < n; i++) y[i] = b[perm[i]];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) y[i] -= A[i * n + j] * y[j];
  for (int i = 0; i < n; i++) y[i] = (A[i * n + i] != 0.0) ? y[i] / A[i * n + i] : 0.0;
  for (int i = n - 1; i >= 0; i--)
    for (int j = i + 1; j < n; j++) y[i] -= A[j * n + i] * y[j];
  for (int i = 0; i < n; i++) x[perm[i]] = y[i];
}

}  // namespace orc
