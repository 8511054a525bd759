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

}  // namespace dgs
