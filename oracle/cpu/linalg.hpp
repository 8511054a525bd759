// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/README.md).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call this.
//
// Small dense f64 linear algebra used by the CPU restatement of NDT_OMP / FAST_GICP.  It replaces the
// Eigen calls the upstream libraries make (Eigen is absent from this image):
//   SelfAdjointEigenSolver<Matrix3d>   -> eigen_selfadjoint3 (Eigen's tridiagonalisation + implicit QR, restated; round 4) / sym_eig3 (cyclic Jacobi, rounds 1-3)
//   Matrix3d::inverse()                -> inv3       (cofactor form, as Eigen's fixed-size 3x3)
//   JacobiSVD<Matrix<double,6,6>>::solve -> jsvd_solve6 (Eigen's own two-sided Jacobi sequence, restated; the default) or
//                                          svd_solve6 (one-sided Hestenes Jacobi, rounds 1-3's stand-in; NdtParams::newton_solver = 0)
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

// ---- Eigen::SelfAdjointEigenSolver<Matrix3d>::compute(A, ComputeEigenvectors) restated (Eigen 3.3.x; [UPSTREAM-RECALL]) --------------------
// What pclomp's VoxelGridCovariance runs on every voxel covariance (`eigensolver.compute(leaf.cov_)`): NOT the closed form (computeDirect) but the
// generic path -- the lower triangle divided by its largest |entry|, the 3 x 3 Householder tridiagonalisation written out in
// tridiagonalization_inplace_selector<MatrixType, 3, false>, implicit symmetric QR steps with Wilkinson's shift (tridiagonal_qr_step: Givens
// rotations by JacobiRotation::makeGivens, the eigenvector matrix updated by applyOnTheRight) until every sub-diagonal entry is negligible
// (|e_i| <= 2 eps (|d_i| + |d_i+1|) or <= DBL_MIN), eigenvalues scaled back, selection sort ascending with the columns swapped along.
// Operation for operation as recalled from SelfAdjointEigenSolver.h / Tridiagonalization.h / Jacobi.h of Eigen 3.3.7 (Ubuntu 20.04's); not
// checkable here (no Eigen in the image).  evals ascending; V row-major, column k = eigenvector of evals[k].  Returns the QR steps taken.
inline int eigen_selfadjoint3(const double* Ain, double* evals, double* V) {
  // mat = lower triangle of A (the strict upper part is never read), scaled into [-1, 1]
  double m00 = Ain[0], m10 = Ain[3], m11 = Ain[4], m20 = Ain[6], m21 = Ain[7], m22 = Ain[8];
  double scale = std::fabs(m00);
  for (double v : {m10, m11, m20, m21, m22}) scale = std::fabs(v) > scale ? std::fabs(v) : scale;   // cwiseAbs().maxCoeff() of the lower-triangular copy (zeros above)
  if (scale == 0.0) scale = 1.0;
  m00 /= scale; m10 /= scale; m11 /= scale; m20 /= scale; m21 /= scale; m22 /= scale;
  double diag[3], sub[2], Q[9];
  const double tol = std::numeric_limits<double>::min();
  diag[0] = m00;
  const double v1norm2 = m20 * m20;
  if (v1norm2 <= tol) {
    diag[1] = m11; diag[2] = m22; sub[0] = m10; sub[1] = m21;
    for (int i = 0; i < 9; i++) Q[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    const double beta = std::sqrt(m10 * m10 + v1norm2);
    const double invBeta = 1.0 / beta;
    const double m01 = m10 * invBeta, m02 = m20 * invBeta;
    const double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
    diag[1] = m11 + m02 * q;
    diag[2] = m22 - m02 * q;
    sub[0] = beta;
    sub[1] = m21 - m01 * q;
    Q[0] = 1; Q[1] = 0; Q[2] = 0; Q[3] = 0; Q[4] = m01; Q[5] = m02; Q[6] = 0; Q[7] = m02; Q[8] = -m01;
  }
  // computeFromTridiagonal_impl
  const int n = 3, maxIterations = 30;
  int end = n - 1, start = 0, iter = 0;
  const double considerAsZero = std::numeric_limits<double>::min(), precision = 2.0 * std::numeric_limits<double>::epsilon();
  while (end > 0) {
    for (int i = start; i < end; ++i)
      if (std::fabs(sub[i]) <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])) * precision || std::fabs(sub[i]) <= considerAsZero) sub[i] = 0.0;
    while (end > 0 && sub[end - 1] == 0.0) end--;
    if (end <= 0) break;
    iter++;
    if (iter > maxIterations * n) break;
    start = end - 1;
    while (start > 0 && sub[start - 1] != 0.0) start--;
    // tridiagonal_qr_step(diag, sub, start, end, Q, n)
    const double td = (diag[end - 1] - diag[end]) * 0.5;
    const double e = sub[end - 1];
    double mu = diag[end];
    if (td == 0.0) {
      mu -= std::fabs(e);
    } else {
      const double e2 = e * e;
      // numext::hypot(td, e): p = max(|td|, |e|), p * sqrt(1 + (min / p)^2)
      const double at = std::fabs(td), ae = std::fabs(e);
      const double p = at > ae ? at : ae;
      double h = 0.0;
      if (p != 0.0) {
        const double qp = (at > ae ? ae : at) / p;
        h = p * std::sqrt(1.0 + qp * qp);
      }
      if (e2 == 0.0) mu -= (e / (td + (td > 0.0 ? 1.0 : -1.0))) * (e / h);
      else mu -= e2 / (td + (td > 0.0 ? h : -h));
    }
    double x = diag[start] - mu;
    double z = sub[start];
    for (int k = start; k < end; ++k) {
      // JacobiRotation::makeGivens(x, z)
      double c, sn;
      if (z == 0.0) {
        c = x < 0.0 ? -1.0 : 1.0; sn = 0.0;
      } else if (x == 0.0) {
        c = 0.0; sn = z < 0.0 ? 1.0 : -1.0;
      } else if (std::fabs(x) > std::fabs(z)) {
        const double t = z / x;
        double u = std::sqrt(1.0 + t * t);
        if (x < 0.0) u = -u;
        c = 1.0 / u; sn = -t * c;
      } else {
        const double t = x / z;
        double u = std::sqrt(1.0 + t * t);
        if (z < 0.0) u = -u;
        sn = -1.0 / u; c = -t * sn;
      }
      // T = G' T G
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
      // Q = Q * G: columns k, k + 1 (applyOnTheRight(k, k + 1, rot): x_i' = c x_i - s y_i, y_i' = s x_i + c y_i)
      for (int i = 0; i < 3; i++) {
        const double xi = Q[i * 3 + k], yi = Q[i * 3 + k + 1];
        Q[i * 3 + k] = c * xi - sn * yi;
        Q[i * 3 + k + 1] = sn * xi + c * yi;
      }
    }
  }
  for (int i = 0; i < 3; i++) diag[i] *= scale;
  // selection sort, ascending, eigenvector columns swapped along (minCoeff: the FIRST smallest of the remaining segment)
  for (int i = 0; i < n - 1; ++i) {
    int k = 0;
    for (int j = 1; j < n - i; j++)
      if (diag[i + j] < diag[i + k]) k = j;
    if (k > 0) {
      std::swap(diag[i], diag[k + i]);
      for (int r = 0; r < 3; r++) std::swap(Q[r * 3 + i], Q[r * 3 + k + i]);
    }
  }
  for (int i = 0; i < 3; i++) evals[i] = diag[i];
  std::memcpy(V, Q, sizeof(Q));
  return iter;
}

// ---- exp(float), rounds 1-3's stand-in (NdtParams::exp_libm = 0) ---------------------------------------
// A fixed sequence of IEEE double operations (no contraction), rounded once to float: accurate to ~1e-16 before the
// rounding, i.e. the correctly rounded expf except for ~1e-9 of the arguments.  It is NOT what upstream's std::exp(float)
// returns (glibc's expf is within 0.502 ulp, not correctly rounded): since round 4 the default is glibc_expf below, and
// this one is the "other exp" of the sensitivity twins (oracle.py ndt_band).
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

// ---- std::exp(float) as glibc computes it ----------------------------------------------------------------
// glibc >= 2.27 (sysdeps/ieee754/flt-32/e_expf.c, Szabolcs Nagy's algorithm; the reference's Docker image is Ubuntu 20.04 / glibc 2.31,
// this image has 2.35): x * N / ln2 = k + r with N = 32, 2^(k/N) from a 32-entry table of correctly rounded doubles, a degree-3
// polynomial in r, everything in double, ONE rounding to float.  The table below was generated from 2^(i/32) at 80 decimal digits; the
// polynomial and the scaled constants are glibc's __exp2f_data.  On an x86-64 CPU with FMA glibc dispatches to a build of the same source
// compiled with -mfma, where r = fma(N / ln2, x, -k) (the one contraction that changes a float result: 2 of the 2.24e9 floats in
// [-104, 88] differ from the unfused form); this restatement spells the fused form out.  PINNED: tests/test_oracle_round4.py compares it with
// the container's own expf on EVERY float in [-104, 0] -- the whole range NDT's exponent can take -- bit for bit (0 differences; measured
// once over [-104, 88] as well).  The device library carries the same sequence of IEEE double operations (csrc/common.h).
inline const uint64_t* glibc_exp2f_table() {
  static const uint64_t T[32] = {
      0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
      0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
      0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
      0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
      0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
      0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};   // asuint64(2^(i/32)) - (i << 47)
  return T;
}
inline float glibc_expf(float xf) {
  if (xf != xf) return xf;
  if (xf > 0x1.62e42ep6f) return std::numeric_limits<float>::infinity();   // overflow (x > log(0x1p128))
  if (xf < -0x1.9fe368p6f) return 0.0f;                                      // underflow to zero (x < log(0x1p-150))
  constexpr double N = 32.0;
  constexpr double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
  constexpr double kShift = 0x1.8p+52, kInvLn2N = 0x1.71547652b82fep+0 * N;
  const double xd = static_cast<double>(xf);
  double kd = std::fma(kInvLn2N, xd, kShift);   // round(x * N / ln2) in the low bits of the shifted double
  uint64_t ki;
  std::memcpy(&ki, &kd, sizeof(ki));
  kd -= kShift;
  const double r = std::fma(kInvLn2N, xd, -kd);
  const uint64_t t = glibc_exp2f_table()[ki % 32] + (ki << (52 - 5));
  double s;
  std::memcpy(&s, &t, sizeof(s));
  const double z = std::fma(C0, r, C1);
  const double r2 = r * r;
  double y = std::fma(C2, r, 1.0);
  y = std::fma(z, r2, y);
  y = y * s;
  return static_cast<float>(y);
}

// ---- std::exp(double) as glibc computes it (PCL's updateHessian: `exp(-gauss_d2 * x^T C x / 2)` in double) ---------------
// glibc >= 2.28 (sysdeps/ieee754/dbl-64/e_exp.c, the same author's algorithm): x N / ln2 = k + r with N = 128, 2^(k/N) = scale (1 + tail) from a
// 2 x 128 table, a degree-5 polynomial, the result scale + scale * tmp.  Table generated here from 2^(i/128) at 120 decimal digits (scale = the
// nearest double, tail = the relative remainder rounded to double; its first entries are the published ones); constants = glibc's __exp_data.
// The contractions of the -mfma build are spelled out (which of the source's a * b + c are fused was settled against the image's libm: the
// range reduction, both polynomial halves, the accumulation of tmp and the final scale + scale * tmp are; the k < 0 special case keeps its
// product apart because the source uses it twice).  PINNED statistically, not exhaustively: tests/test_oracle_round4.py compares it with the
// container's exp on 2e8 doubles drawn over [-760, 720] (dense in NDT's range, through the subnormal results to underflow, tiny arguments, up to
// overflow): 0 differences (4e8 when it was written).  The device library carries the same sequence (csrc/ndt_strict.h).
inline const uint64_t* glibc_exp_table() {
  static const uint64_t T[256] = {
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
      0x3c77893b4d91cd9dull, 0x3fefe7c1819e90d8ull, 0x3c5305c14160cc89ull, 0x3feff3c22b8f71f1ull};   // [2 i] = asuint64(tail_i), [2 i + 1] = asuint64(scale_i) - (i << 45)
  return T;
}
inline double glibc_exp(double x) {
  constexpr double N = 128.0;
  constexpr double kInvLn2N = 0x1.71547652b82fep0 * N, kNegLn2hiN = -0x1.62e42fefa0000p-8, kNegLn2loN = -0x1.cf79abc9e3b3ap-47, kShift = 0x1.8p52;
  constexpr double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  uint64_t xb;
  std::memcpy(&xb, &x, sizeof(xb));
  uint32_t abstop = static_cast<uint32_t>(xb >> 52) & 0x7ffu;
  bool special = false;
  if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {                 // |x| < 2^-54 or |x| >= 512 (top12(0x1p-54) = 0x3c9, top12(512.0) = 0x408)
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;     // tiny
    if (abstop >= 0x409u) {                                 // |x| >= 1024
      if (xb == 0xfff0000000000000ull) return 0.0;
      if (abstop >= 0x7ffu) return 1.0 + x;                 // inf / nan
      return (xb >> 63) ? 0.0 : std::numeric_limits<double>::infinity();
    }
    special = true;                                         // 512 <= |x| < 1024: the scale may leave the normal range
  }
  double kd = std::fma(kInvLn2N, x, kShift);
  uint64_t ki;
  std::memcpy(&ki, &kd, sizeof(ki));
  kd -= kShift;
  const double r = std::fma(kd, kNegLn2loN, std::fma(kd, kNegLn2hiN, x));
  const uint64_t idx = 2 * (ki % 128);
  const uint64_t* T = glibc_exp_table();
  double tail;
  std::memcpy(&tail, &T[idx], sizeof(tail));
  uint64_t sbits = T[idx + 1] + (ki << (52 - 7));
  const double r2 = r * r;
  double tmp = std::fma(r2, std::fma(r, C3, C2), tail + r);
  tmp = std::fma(r2 * r2, std::fma(r, C5, C4), tmp);
  double scale;
  if (special) {
    if ((ki & 0x80000000ull) == 0) {                        // k > 0
      sbits -= 1009ull << 52;
      std::memcpy(&scale, &sbits, sizeof(scale));
      return 0x1p1009 * std::fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;                                 // k < 0: care in the subnormal range
    std::memcpy(&scale, &sbits, sizeof(scale));
    const double st = scale * tmp;
    double y = scale + st;
    if (y < 1.0) {
      double lo = scale - y + st;
      const double hi = 1.0 + y;
      lo = 1.0 - hi + y + lo;
      y = (hi + lo) - 1.0;
      if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022 * y;
  }
  std::memcpy(&scale, &sbits, sizeof(scale));
  return std::fma(scale, tmp, scale);
}

// std::exp(double) with a platform-independent value, for the double-precision computeHessian / updateHessian pass: the same
// reduction and Taylor polynomial as det_expf, not rounded to float -- within ~2 ulp of exp(x), the same bits on any IEEE machine
// (the device carries the same sequence).  The scaling by 2^k is split in two so that results in the subnormal range round once.
inline double det_exp(double x) {
  if (x != x) return x;
  if (x < -746.0) return 0.0;
  if (x > 710.0) return std::numeric_limits<double>::infinity();
  const double kd = std::floor(x * 1.4426950408889634 + 0.5);
  const double r = (x - kd * 0x1.62e42fefa38p-1) - kd * 0x1.ef35793c7673p-45;
  double p = 1.0 / 6227020800.0;
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
  const int64_t k1 = k / 2, k2 = k - k1;   // |k| <= 1077: both halves are normal powers of two
  const uint64_t b1 = static_cast<uint64_t>(k1 + 1023) << 52, b2 = static_cast<uint64_t>(k2 + 1023) << 52;
  double s1, s2;
  std::memcpy(&s1, &b1, sizeof(s1));
  std::memcpy(&s2, &b2, sizeof(s2));
  return (p * s1) * s2;
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

// x = A^+ b the way Eigen::JacobiSVD<Matrix<double,6,6>>(A, ComputeFullU | ComputeFullV).solve(b) forms it -- what ndt_omp's
// computeTransformation calls for the Newton step (SURVEY.md App. A "Outer loop") -- restated from the published Eigen 3.3 source
// (Eigen/src/SVD/JacobiSVD.h, Eigen/src/Jacobi/Jacobi.h; [UPSTREAM-RECALL], not on disk):
//   * TWO-SIDED Jacobi on the work matrix W = A / max|A_ij|: sweeps over (p, q), p = 1..5, q = 0..p-1; a 2x2 block is treated when
//     |W_pq| or |W_qp| exceeds max(DBL_MIN, 2 eps * maxDiagEntry); real_2x2_jacobi_svd (first a rotation that makes the block
//     symmetric, then makeJacobi on it), W <- J_left W, U <- U J_left^T, W <- W J_right, V <- V J_right, maxDiagEntry updated; the
//     iteration ends with the first sweep that treats no block (the real Eigen loop has no sweep cap; 64 here, never reached);
//   * singular values |W_ii| * scale, columns of U negated where W_ii < 0, sorted in descending order (selection by the first
//     maximum of the tail, column swaps in U and V);
//   * rank: singular values below max(s_0 * 6 eps, DBL_MIN) dropped from the END (SVDBase::rank);
//   * solve: tmp = U(:, :rank)^T b, tmp_j *= 1 / s_j (asDiagonal().inverse()), x = V(:, :rank) tmp.
// What cannot be known offline is Eigen's order of additions inside those two small matrix-vector products (its SSE2 packet
// reductions); they are written here as plain left-to-right sums.  Every other operation follows the source operation for
// operation, each individually rounded (-ffp-contract=off; the device carries the same sequence, csrc/solve6.h).
// apply_rotation_in_the_plane(x, y, (c, s)):  x' = c x + s y,  y' = -s x + c y  (skipped altogether when c == 1 and s == 0).
struct JsvdStats { int sweeps, rotations; };

// Steps 1-4 of JacobiSVD<Matrix<S, N, N>>::compute(A, ComputeFullU | ComputeFullV) for a square real matrix (row-major S[N * N]):
// U, V (row-major), singular values in descending order; returns the number of non-zero singular values.
template <typename S, int N>
inline int jacobi_svd_square(const S* A, S* U, S* V, S* sv, JsvdStats* stats = nullptr) {
  const S precision = S(2) * std::numeric_limits<S>::epsilon();
  const S consider_as_zero = std::numeric_limits<S>::min();
  S scale = S(0);
  for (int i = 0; i < N * N; i++) { const S a = std::fabs(A[i]); if (a > scale) scale = a; }
  if (scale == S(0)) scale = S(1);
  S W[N * N];
  for (int i = 0; i < N * N; i++) { W[i] = A[i] / scale; U[i] = V[i] = (i % (N + 1) == 0) ? S(1) : S(0); }
  S max_diag = S(0);
  for (int i = 0; i < N; i++) { const S a = std::fabs(W[i * (N + 1)]); if (a > max_diag) max_diag = a; }
  int sweeps = 0, rotations = 0;
  bool finished = false;
  while (!finished && sweeps < 64) {
    finished = true;
    sweeps++;
    for (int p = 1; p < N; p++)
      for (int q = 0; q < p; q++) {
        const S pm = precision * max_diag;
        const S threshold = consider_as_zero > pm ? consider_as_zero : pm;
        if (!(std::fabs(W[p * N + q]) > threshold || std::fabs(W[q * N + p]) > threshold)) continue;
        finished = false;
        rotations++;
        // ---- real_2x2_jacobi_svd(W, p, q, &j_left, &j_right)
        S m00 = W[p * N + p], m01 = W[p * N + q], m10 = W[q * N + p], m11 = W[q * N + q];
        const S t = m00 + m11, d = m10 - m01;
        S r1c, r1s;
        if (std::fabs(d) < consider_as_zero) { r1s = S(0); r1c = S(1); }
        else {
          const S u = t / d;
          const S tmp = std::sqrt(S(1) + u * u);
          r1s = S(1) / tmp;
          r1c = u / tmp;
        }
        if (!(r1c == S(1) && r1s == S(0))) {   // m.applyOnTheLeft(0, 1, rot1)
          const S x0 = m00, y0 = m10, x1 = m01, y1 = m11;
          m00 = r1c * x0 + r1s * y0; m10 = -r1s * x0 + r1c * y0;
          m01 = r1c * x1 + r1s * y1; m11 = -r1s * x1 + r1c * y1;
        }
        // j_right->makeJacobi(m, 0, 1) = makeJacobi(m00, m01, m11)
        S jrc, jrs;
        {
          const S deno = S(2) * std::fabs(m01);
          if (deno < consider_as_zero) { jrc = S(1); jrs = S(0); }
          else {
            const S tau = (m00 - m11) / deno;
            const S w = std::sqrt(tau * tau + S(1));
            const S tt = (tau > S(0)) ? S(1) / (tau + w) : S(1) / (tau - w);
            const S sign_t = tt > S(0) ? S(1) : S(-1);
            const S nn = S(1) / std::sqrt(tt * tt + S(1));
            jrs = -sign_t * (m01 / std::fabs(m01)) * std::fabs(tt) * nn;
            jrc = nn;
          }
        }
        // *j_left = rot1 * j_right->transpose();   transpose() = (c, -s);   (a * b) = (a.c b.c - a.s b.s,  a.c b.s + a.s b.c)
        const S jtc = jrc, jts = -jrs;
        const S jlc = r1c * jtc - r1s * jts;
        const S jls = r1c * jts + r1s * jtc;
        if (!(jlc == S(1) && jls == S(0))) {
          for (int i = 0; i < N; i++) {   // W.applyOnTheLeft(p, q, j_left): rows p, q
            const S xi = W[p * N + i], yi = W[q * N + i];
            W[p * N + i] = jlc * xi + jls * yi;
            W[q * N + i] = -jls * xi + jlc * yi;
          }
          for (int i = 0; i < N; i++) {   // U.applyOnTheRight(p, q, j_left.transpose()): columns p, q rotated by j_left
            const S xi = U[i * N + p], yi = U[i * N + q];
            U[i * N + p] = jlc * xi + jls * yi;
            U[i * N + q] = -jls * xi + jlc * yi;
          }
        }
        if (!(jrc == S(1) && -jrs == S(0))) {   // W / V .applyOnTheRight(p, q, j_right): columns p, q rotated by j_right.transpose() = (c, -s)
          const S c = jrc, s = -jrs;
          for (int i = 0; i < N; i++) {
            const S xi = W[i * N + p], yi = W[i * N + q];
            W[i * N + p] = c * xi + s * yi;
            W[i * N + q] = -s * xi + c * yi;
          }
          for (int i = 0; i < N; i++) {
            const S xi = V[i * N + p], yi = V[i * N + q];
            V[i * N + p] = c * xi + s * yi;
            V[i * N + q] = -s * xi + c * yi;
          }
        }
        const S app = std::fabs(W[p * N + p]), aqq = std::fabs(W[q * N + q]);
        const S mx = app < aqq ? aqq : app;   // numext::maxi
        if (max_diag < mx) max_diag = mx;
      }
  }
  if (stats) { stats->sweeps = sweeps; stats->rotations = rotations; }
  // ---- step 3 / 4: singular values, signs, descending order
  for (int i = 0; i < N; i++) {
    const S a = W[i * (N + 1)];
    sv[i] = std::fabs(a);
    if (a < S(0)) for (int k = 0; k < N; k++) U[k * N + i] = -U[k * N + i];
  }
  for (int i = 0; i < N; i++) sv[i] *= scale;
  int nonzero = N;
  for (int i = 0; i < N; i++) {
    int pos = 0;
    S best = sv[i];
    for (int k = 1; k < N - i; k++) if (sv[i + k] > best) { best = sv[i + k]; pos = k; }
    if (best == S(0)) { nonzero = i; break; }
    if (pos) {
      pos += i;
      std::swap(sv[i], sv[pos]);
      for (int k = 0; k < N; k++) { std::swap(U[k * N + i], U[k * N + pos]); std::swap(V[k * N + i], V[k * N + pos]); }
    }
  }
  return nonzero;
}

inline void jsvd_solve6(const double* A, const double* b, double* x, JsvdStats* stats = nullptr) {
  const int n = 6;
  double U[36], V[36], sv[6];
  const int nonzero = jacobi_svd_square<double, 6>(A, U, V, sv, stats);
  // ---- rank (SVDBase::rank, threshold() = diagSize * epsilon) and solve
  const double pt = sv[0] * (6.0 * std::numeric_limits<double>::epsilon());
  const double premultiplied = pt > std::numeric_limits<double>::min() ? pt : std::numeric_limits<double>::min();
  int r = nonzero - 1;
  while (r >= 0 && sv[r] < premultiplied) --r;
  const int rank = r + 1;
  double tmp[6];
  for (int j = 0; j < rank; j++) {
    double acc = 0.0;
    for (int k = 0; k < n; k++) acc += U[k * n + j] * b[k];
    tmp[j] = (1.0 / sv[j]) * acc;
  }
  for (int i = 0; i < n; i++) {
    double acc = 0.0;
    for (int j = 0; j < rank; j++) acc += V[i * n + j] * tmp[j];
    x[i] = acc;
  }
}

// Eigen::Transform<float, 3, Affine>::rotation() on the linear part of a column-major float 4x4 (what ndt_omp's computeTransformation
// takes the Euler angles of: eig_transformation.rotation().eulerAngles(0, 1, 2)): computeRotationScaling, i.e. the polar factor
// through a 3x3 float JacobiSVD -- x = det(U V^T), U.col(0) /= x, R = U V^T.  [UPSTREAM-RECALL: Eigen/src/Geometry/Transform.h.]
// Output: row-major 3x3.
inline void affine_rotation_f32(const float* T_colmajor16, float* R) {
  float L[9], U[9], V[9], sv[3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) L[r * 3 + c] = T_colmajor16[c * 4 + r];
  jacobi_svd_square<float, 3>(L, U, V, sv);
  auto prod = [&](const float* M, int i, int j) { return M[i * 3 + 0] * V[j * 3 + 0] + M[i * 3 + 1] * V[j * 3 + 1] + M[i * 3 + 2] * V[j * 3 + 2]; };   // (M V^T)(i, j)
  float UVt[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) UVt[i * 3 + j] = prod(U, i, j);
  auto det3h = [&](int a, int b, int c) { return UVt[0 * 3 + a] * (UVt[1 * 3 + b] * UVt[2 * 3 + c] - UVt[1 * 3 + c] * UVt[2 * 3 + b]); };
  const float x = det3h(0, 1, 2) - det3h(1, 0, 2) + det3h(2, 0, 1);
  for (int k = 0; k < 3; k++) U[k * 3 + 0] /= x;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[i * 3 + j] = prod(U, i, j);
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
