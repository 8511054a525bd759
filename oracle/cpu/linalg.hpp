// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/README.md).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call this.
//
// Small dense f64 linear algebra used by the CPU restatement of NDT_OMP / FAST_GICP.  It replaces the
// Eigen calls the upstream libraries make (Eigen is absent from this image):
//   SelfAdjointEigenSolver<Matrix3d>   -> sym_eig3   (cyclic Jacobi, ascending eigenvalues)
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
