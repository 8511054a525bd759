// Small double-precision linear algebra executed by one lane (3x3 inverse, symmetric 3x3 eigen-decomposition).
#pragma once
#include <hip/hip_runtime.h>

namespace dgs {

__device__ inline bool inv3_d(const double* A, double* Ai) {
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

// symmetric 3x3 eigen-decomposition (cyclic Jacobi, lower triangle is authoritative), ascending eigenvalues
__device__ inline void sym_eig3_d(const double* Ain, double* ev, double* V) {
  double a00 = Ain[0], a11 = Ain[4], a22 = Ain[8], a01 = Ain[3], a02 = Ain[6], a12 = Ain[7];
  double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 32; sweep++) {
    const double off = a01 * a01 + a02 * a02 + a12 * a12;
    const double dia = a00 * a00 + a11 * a11 + a22 * a22;
    if (off == 0.0 || off <= 1e-34 * dia) break;
    // rotation (0,1)
    if (a01 != 0.0) {
      const double th = (a11 - a00) / (2.0 * a01);
      const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      const double n00 = a00 - t * a01, n11 = a11 + t * a01;
      const double n02 = c * a02 - s * a12, n12 = s * a02 + c * a12;
      a00 = n00; a11 = n11; a01 = 0.0; a02 = n02; a12 = n12;
      for (int k = 0; k < 3; k++) { const double x = v[k * 3 + 0], y = v[k * 3 + 1]; v[k * 3 + 0] = c * x - s * y; v[k * 3 + 1] = s * x + c * y; }
    }
    // rotation (0,2)
    if (a02 != 0.0) {
      const double th = (a22 - a00) / (2.0 * a02);
      const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      const double n00 = a00 - t * a02, n22 = a22 + t * a02;
      const double n01 = c * a01 - s * a12, n12 = s * a01 + c * a12;
      a00 = n00; a22 = n22; a02 = 0.0; a01 = n01; a12 = n12;
      for (int k = 0; k < 3; k++) { const double x = v[k * 3 + 0], y = v[k * 3 + 2]; v[k * 3 + 0] = c * x - s * y; v[k * 3 + 2] = s * x + c * y; }
    }
    // rotation (1,2)
    if (a12 != 0.0) {
      const double th = (a22 - a11) / (2.0 * a12);
      const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      const double n11 = a11 - t * a12, n22 = a22 + t * a12;
      const double n01 = c * a01 - s * a02, n02 = s * a01 + c * a02;
      a11 = n11; a22 = n22; a12 = 0.0; a01 = n01; a02 = n02;
      for (int k = 0; k < 3; k++) { const double x = v[k * 3 + 1], y = v[k * 3 + 2]; v[k * 3 + 1] = c * x - s * y; v[k * 3 + 2] = s * x + c * y; }
    }
  }
  double e[3] = {a00, a11, a22};
  int o0 = 0, o1 = 1, o2 = 2;
  if (e[o0] > e[o1]) { int t = o0; o0 = o1; o1 = t; }
  if (e[o1] > e[o2]) { int t = o1; o1 = o2; o2 = t; }
  if (e[o0] > e[o1]) { int t = o0; o0 = o1; o1 = t; }
  const int o[3] = {o0, o1, o2};
  for (int k = 0; k < 3; k++) {
    ev[k] = e[o[k]];
    for (int r = 0; r < 3; r++) V[r * 3 + k] = v[r * 3 + o[k]];
  }
}


}  // namespace dgs
