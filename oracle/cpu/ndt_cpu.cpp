// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/README.md).
// CPU restatement of pclomp::NormalDistributionsTransform; see ndt_cpu.hpp for provenance.
#include "ndt_cpu.hpp"
#include "linalg.hpp"

#include <cmath>
#include <cstring>
#include <limits>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

// ------------------------------------------------------------------------------------------------
// Eigen 3.3 MatrixBase::eulerAngles(0,1,2) in float (used by computeTransformation to turn the guess
// into the 6-vector p; SURVEY App. A "Pose parameterisation").
void euler_angles_012(const float* T, float res[3]) {
  auto m = [&](int r, int c) { return T[c * 4 + r]; };
  const float kPi = 3.14159265358979323846f;
  // a0=0,a1=1,a2=2 -> odd = 0, i=0, j=1, k=2
  res[0] = std::atan2(m(1, 2), m(2, 2));
  const float c2 = std::sqrt(m(0, 0) * m(0, 0) + m(0, 1) * m(0, 1));
  if (res[0] > 0.0f) {  // (!odd) && res[0] > 0
    res[0] -= kPi;      // res[0] > 0 here
    res[1] = std::atan2(-m(0, 2), -c2);
  } else {
    res[1] = std::atan2(-m(0, 2), c2);
  }
  const float s1 = std::sin(res[0]);
  const float c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * m(2, 0) - c1 * m(1, 0), c1 * m(1, 1) - s1 * m(2, 1));
  res[0] = -res[0];  // !odd -> negate
  res[1] = -res[1];
  res[2] = -res[2];
}

// (Translation<float,3>(p0,p1,p2) * AngleAxis<float>(p3,X) * AngleAxis<float>(p4,Y) * AngleAxis<float>(p5,Z)).matrix()
void pose_to_matrix_f32(const double p[6], float* T) {
  const float rx = static_cast<float>(p[3]), ry = static_cast<float>(p[4]), rz = static_cast<float>(p[5]);
  // Eigen evaluates std::cos(float); a correctly rounded cosf equals the double cosine rounded once, which is what is
  // written here so that the value does not depend on the libm at hand (differs from glibc cosf in ~1e-7 of arguments).
  const float cx = static_cast<float>(std::cos(static_cast<double>(rx))), sx = static_cast<float>(std::sin(static_cast<double>(rx)));
  const float cy = static_cast<float>(std::cos(static_cast<double>(ry))), sy = static_cast<float>(std::sin(static_cast<double>(ry)));
  const float cz = static_cast<float>(std::cos(static_cast<double>(rz))), sz = static_cast<float>(std::sin(static_cast<double>(rz)));
  // R = Rx * Ry * Rz
  const float r00 = cy * cz, r01 = -cy * sz, r02 = sy;
  const float r10 = cx * sz + sx * sy * cz, r11 = cx * cz - sx * sy * sz, r12 = -sx * cy;
  const float r20 = sx * sz - cx * sy * cz, r21 = sx * cz + cx * sy * sz, r22 = cx * cy;
  T[0] = r00; T[1] = r10; T[2] = r20; T[3] = 0.f;
  T[4] = r01; T[5] = r11; T[6] = r21; T[7] = 0.f;
  T[8] = r02; T[9] = r12; T[10] = r22; T[11] = 0.f;
  T[12] = static_cast<float>(p[0]); T[13] = static_cast<float>(p[1]); T[14] = static_cast<float>(p[2]); T[15] = 1.f;
}

// ------------------------------------------------------------------------------------------------
// setInputTarget -> VoxelGridCovariance::filter (SURVEY App. A "Target model")
void NdtCpu::set_target(const float* xyz16, int64_t n) {
  target.assign(xyz16, xyz16 + n * 4);
  nt = n;
  leaves.clear();
  for (int a = 0; a < 3; a++) {
    leaf_size[a] = static_cast<float>(prm.resolution);
    inv_leaf_size[a] = 1.0f / leaf_size[a];
  }
  if (n == 0) return;
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[1], -mn[2]};
  for (int64_t i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) {
      const float v = xyz16[i * 4 + a];
      if (!std::isfinite(xyz16[i * 4]) || !std::isfinite(xyz16[i * 4 + 1]) || !std::isfinite(xyz16[i * 4 + 2])) continue;
      mn[a] = std::min(mn[a], v);
      mx[a] = std::max(mx[a], v);
    }
  for (int a = 0; a < 3; a++) {
    min_b[a] = static_cast<int>(std::floor(mn[a] * inv_leaf_size[a]));
    max_b[a] = static_cast<int>(std::floor(mx[a] * inv_leaf_size[a]));
    div_b[a] = max_b[a] - min_b[a] + 1;
  }
  divb_mul[0] = 1;
  divb_mul[1] = div_b[0];
  divb_mul[2] = static_cast<int64_t>(div_b[0]) * div_b[1];

  // first pass: accumulate n, sum p, sum p p^T in double, in point-index order
  for (int64_t i = 0; i < n; i++) {
    const float* pt = xyz16 + i * 4;
    if (!std::isfinite(pt[0]) || !std::isfinite(pt[1]) || !std::isfinite(pt[2])) continue;
    const int ijk0 = static_cast<int>(std::floor(pt[0] * inv_leaf_size[0]) - static_cast<float>(min_b[0]));
    const int ijk1 = static_cast<int>(std::floor(pt[1] * inv_leaf_size[1]) - static_cast<float>(min_b[1]));
    const int ijk2 = static_cast<int>(std::floor(pt[2] * inv_leaf_size[2]) - static_cast<float>(min_b[2]));
    const int64_t idx = ijk0 * divb_mul[0] + ijk1 * divb_mul[1] + ijk2 * divb_mul[2];
    Leaf& leaf = leaves[idx];
    const double p3[3] = {pt[0], pt[1], pt[2]};
    for (int a = 0; a < 3; a++) {
      leaf.sum[a] += p3[a];
      leaf.centroid[a] += pt[a];
      for (int b = 0; b < 3; b++) leaf.sq[a * 3 + b] += p3[a] * p3[b];
    }
    leaf.nr_points++;
  }

  // second pass: mean, single-pass covariance, eigenvalue clamp, inverse
  for (auto& kv : leaves) {
    Leaf& leaf = kv.second;
    const double np = static_cast<double>(leaf.nr_points);
    for (int a = 0; a < 3; a++) {
      leaf.centroid[a] /= static_cast<float>(leaf.nr_points);
      leaf.mean[a] = leaf.sum[a] / np;
    }
    if (leaf.nr_points < prm.min_points_per_voxel) continue;
    // cov = (sq - 2 * (pt_sum * mean^T)) / n + mean * mean^T ; cov *= (n - 1) / n
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        double c = (leaf.sq[a * 3 + b] - 2.0 * (leaf.sum[a] * leaf.mean[b])) / np + leaf.mean[a] * leaf.mean[b];
        leaf.cov[a * 3 + b] = c * ((np - 1.0) / np);
      }
    double ev[3], V[9];
    if (prm.cov_eigensolver) eigen_selfadjoint3(leaf.cov, ev, V);
    else sym_eig3(leaf.cov, ev, V);
    if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) continue;  // nr_points = -1 upstream
    const double min_ev = prm.min_covar_eigvalue_mult * ev[2];
    if (ev[0] < min_ev) {
      ev[0] = min_ev;
      if (ev[1] < min_ev) ev[1] = min_ev;
      // cov = evecs * diag * evecs.inverse()
      double Vi[9], VD[9];
      inv3(V, Vi);
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) VD[r * 3 + c] = V[r * 3 + c] * ev[c];
      mat3_mul(VD, Vi, leaf.cov);
    }
    for (int a = 0; a < 3; a++) leaf.evals[a] = ev[a];
    inv3(leaf.cov, leaf.icov);
    bool ok = true;
    for (int a = 0; a < 9; a++)
      if (std::isinf(leaf.icov[a])) ok = false;
    leaf.valid = ok;
  }
}

void NdtCpu::set_source(const float* xyz16, int64_t n) {
  source.assign(xyz16, xyz16 + n * 4);
  ns = n;
}

// computeAngleDerivatives (Magnusson 2009 eq. 6.19 / 6.21), double trig, float tables
void NdtCpu::compute_angle_derivatives(const double p[6]) {
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p[3]); sx = std::sin(p[3]); }
  if (std::fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p[4]); sy = std::sin(p[4]); }
  if (std::fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p[5]); sz = std::sin(p[5]); }
  const double J[8][3] = {
      {(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},  // a
      {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},    // b
      {(-sy * cz), sy * sz, cy},                                           // c
      {sx * cy * cz, (-sx * cy * sz), sx * sy},                            // d
      {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},                         // e
      {(-cy * sz), (-cy * cz), 0},                                         // f
      {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},            // g
      {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};            // h
  const double Hh[15][3] = {
      {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},     // a2
      {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},  // a3
      {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},                        // b2
      {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},                        // b3
      {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},            // c2
      {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},            // c3
      // d1: upstream PCL/ndt_omp carry (+sy) here (Magnusson eq. 6.21 as printed); the exact d2/dry2 is (-sy).
      {(-cy * cz), (cy * sz), (prm.fix_hessian_d1 ? -sy : sy)},            // d1
      {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},                        // d2
      {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},                       // d3
      {(sy * sz), (sy * cz), 0},                                           // e1
      {(-sx * cy * sz), (-sx * cy * cz), 0},                               // e2
      {(cx * cy * sz), (cx * cy * cz), 0},                                 // e3
      {(-cy * cz), (cy * sz), 0},                                          // f1
      {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},           // f2
      {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};          // f3
  for (int i = 0; i < 8; i++)
    for (int k = 0; k < 3; k++) { j_ang[i][k] = static_cast<float>(J[i][k]); j_ang_d[i][k] = J[i][k]; }
  for (int i = 0; i < 15; i++)
    for (int k = 0; k < 3; k++) { h_ang[i][k] = static_cast<float>(Hh[i][k]); h_ang_d[i][k] = Hh[i][k]; }
}

// getNeighborhoodAtPoint{1,7,26} / radiusSearch (KDTREE) restated on the voxel map
int NdtCpu::neighbours(const float xt[3], const Leaf** out) const {
  static const int off7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  const int ijk[3] = {static_cast<int>(std::floor(xt[0] / leaf_size[0])), static_cast<int>(std::floor(xt[1] / leaf_size[1])),
                      static_cast<int>(std::floor(xt[2] / leaf_size[2]))};
  int cnt = 0;
  auto visit = [&](int dx, int dy, int dz) -> const Leaf* {
    const int c[3] = {ijk[0] + dx, ijk[1] + dy, ijk[2] + dz};
    for (int a = 0; a < 3; a++)
      if (c[a] < min_b[a] || c[a] > max_b[a]) return nullptr;
    const int64_t idx = (c[0] - min_b[0]) * divb_mul[0] + (c[1] - min_b[1]) * divb_mul[1] + (c[2] - min_b[2]) * divb_mul[2];
    auto it = leaves.find(idx);
    if (it == leaves.end() || !it->second.valid) return nullptr;
    return &it->second;
  };
  switch (prm.search_method) {
    case NDT_DIRECT1: {
      const Leaf* l = visit(0, 0, 0);
      if (l) out[cnt++] = l;
    } break;
    case NDT_DIRECT26:
      for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
          for (int dz = -1; dz <= 1; dz++) {
            const Leaf* l = visit(dx, dy, dz);
            if (l) out[cnt++] = l;
          }
      break;
    case NDT_KDTREE: {
      const float r2 = static_cast<float>(prm.resolution * prm.resolution);
      for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
          for (int dz = -1; dz <= 1; dz++) {
            const Leaf* l = visit(dx, dy, dz);
            if (!l) continue;
            const float ex = l->centroid[0] - xt[0], ey = l->centroid[1] - xt[1], ez = l->centroid[2] - xt[2];
            if (ex * ex + ey * ey + ez * ez < r2) out[cnt++] = l;
          }
    } break;
    default:
    case NDT_DIRECT7:
      for (int k = 0; k < 7; k++) {
        const Leaf* l = visit(off7[k][0], off7[k][1], off7[k][2]);
        if (l) out[cnt++] = l;
      }
      break;
  }
  return cnt;
}

double NdtCpu::derivatives(const double p[6], double g[6], double H[36], bool compute_hessian) {
  float T[16];
  pose_to_matrix_f32(p, T);
  return derivatives_with(T, p, g, H, compute_hessian);
}

// transformPointCloud(T) + computeDerivatives / updateDerivatives (SURVEY App. A "Per point")
double NdtCpu::derivatives_with(const float* T, const double p[6], double g[6], double H[36], bool compute_hessian) {
  evaluations++;
  compute_angle_derivatives(p);
  const float gd2 = static_cast<float>(gauss_d2);
  std::vector<double> scores(ns, 0.0), grads(static_cast<size_t>(ns) * 6, 0.0), hess(static_cast<size_t>(ns) * 36, 0.0);
  int nthreads = prm.num_threads;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
#pragma omp parallel for num_threads(nthreads) schedule(guided, 8)
  for (int64_t idx = 0; idx < ns; idx++) {
    const float* xp = source.data() + idx * 4;
    float xt[3];
    for (int r = 0; r < 3; r++) xt[r] = T[0 * 4 + r] * xp[0] + T[1 * 4 + r] * xp[1] + T[2 * 4 + r] * xp[2] + T[3 * 4 + r];
    const Leaf* nb[27];
    const int nn = neighbours(xt, nb);
    if (nn == 0) continue;
    // computePointDerivatives: float 4x6 gradient (rows 0..2 used), 6 second-derivative 3-vectors
    float pg[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
    float xj[8], xh[15];
    for (int i = 0; i < 8; i++) xj[i] = j_ang[i][0] * xp[0] + j_ang[i][1] * xp[1] + j_ang[i][2] * xp[2];
    pg[1][3] = xj[0]; pg[2][3] = xj[1];
    pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4];
    pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
    // ph[i][j][r]: second derivative of T(p)x w.r.t. p_i p_j (only i,j in 3..5 non-zero)
    float ph[6][6][3];
    std::memset(ph, 0, sizeof(ph));
    if (compute_hessian) {
      for (int i = 0; i < 15; i++) xh[i] = h_ang[i][0] * xp[0] + h_ang[i][1] * xp[1] + h_ang[i][2] * xp[2];
      const float a[3] = {0, xh[0], xh[1]}, b[3] = {0, xh[2], xh[3]}, c[3] = {0, xh[4], xh[5]};
      const float d[3] = {xh[6], xh[7], xh[8]}, e[3] = {xh[9], xh[10], xh[11]}, f[3] = {xh[12], xh[13], xh[14]};
      for (int r = 0; r < 3; r++) {
        ph[3][3][r] = a[r]; ph[3][4][r] = b[r]; ph[3][5][r] = c[r];
        ph[4][3][r] = b[r]; ph[4][4][r] = d[r]; ph[4][5][r] = e[r];
        ph[5][3][r] = c[r]; ph[5][4][r] = e[r]; ph[5][5][r] = f[r];
      }
    }
    double score_pt = 0, g_pt[6] = {0, 0, 0, 0, 0, 0}, h_pt[36];
    std::memset(h_pt, 0, sizeof(h_pt));
    for (int v = 0; v < nn; v++) {
      const Leaf* cell = nb[v];
      float q[3], C[3][3];
      for (int r = 0; r < 3; r++) q[r] = static_cast<float>(static_cast<double>(xt[r]) - cell->mean[r]);
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) C[r][c] = static_cast<float>(cell->icov[r * 3 + c]);
      float qC[3];
      for (int c = 0; c < 3; c++) qC[c] = q[0] * C[0][c] + q[1] * C[1][c] + q[2] * C[2][c];
      const float e_arg = -gd2 * (q[0] * qC[0] + q[1] * qC[1] + q[2] * qC[2]) * 0.5f;
      float e_x_cov_x = prm.exp_libm == 1 ? glibc_expf(e_arg) : (prm.exp_libm == 2 ? std::exp(e_arg) : det_expf(e_arg));
      const float score_inc = static_cast<float>(-gauss_d1 * e_x_cov_x);
      e_x_cov_x = gd2 * e_x_cov_x;
      if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;
      e_x_cov_x = static_cast<float>(e_x_cov_x * gauss_d1);
      float cPG[3][6];
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < 6; k++) cPG[r][k] = C[r][0] * pg[0][k] + C[r][1] * pg[1][k] + C[r][2] * pg[2][k];
      float g6[6];
      for (int k = 0; k < 6; k++) g6[k] = q[0] * cPG[0][k] + q[1] * cPG[1][k] + q[2] * cPG[2][k];
      for (int k = 0; k < 6; k++) g_pt[k] += static_cast<double>(e_x_cov_x * g6[k]);
      if (compute_hessian) {
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++) {
            const float xCH = qC[0] * ph[i][j][0] + qC[1] * ph[i][j][1] + qC[2] * ph[i][j][2];
            const float pcp = pg[0][j] * cPG[0][i] + pg[1][j] * cPG[1][i] + pg[2][j] * cPG[2][i];
            h_pt[i * 6 + j] += e_x_cov_x * (-gd2 * g6[i] * g6[j] + xCH + pcp);
          }
      }
      score_pt += score_inc;
    }
    scores[idx] = score_pt;
    for (int k = 0; k < 6; k++) grads[idx * 6 + k] = g_pt[k];
    if (compute_hessian)
      for (int k = 0; k < 36; k++) hess[idx * 36 + k] = h_pt[k];
  }
  // "ensure that the result is invariant against the summing up order": sequential index-order sum
  double score = 0;
  for (int k = 0; k < 6; k++) g[k] = 0;
  if (compute_hessian)
    for (int k = 0; k < 36; k++) H[k] = 0;
  for (int64_t i = 0; i < ns; i++) {
    score += scores[i];
    for (int k = 0; k < 6; k++) g[k] += grads[i * 6 + k];
    if (compute_hessian)
      for (int k = 0; k < 36; k++) H[k] += hess[i * 36 + k];
  }
  return score;
}

// ---- computeHessian / updateHessian in PCL's double form (NdtParams::hessian_recompute_double) -------------------------------------
// ndt_omp moved computeDerivatives / updateDerivatives to float matrices but kept PCL's computeHessian / updateHessian: a single,
// unthreaded pass over the points that adds every (point, voxel) term straight into the 6x6 Hessian, in double -- x from the float
// point, x' - mean from the float transformed point, the voxel's double inverse covariance, point gradient / Hessian from the double
// angle vectors.  std::exp(double) is libm-dependent in its last bit; the restatement uses det_exp (linalg.hpp).  Dot products and the
// 3x3 * 3 products are written left to right (Eigen's reduction order inside them cannot be known offline).
// The (point, voxel) terms are formed in parallel and then added one by one in upstream's order, so the sums are upstream's sums.
void NdtCpu::hessian_double(const double p[6], double H[36]) {
  float T[16];
  pose_to_matrix_f32(p, T);
  compute_angle_derivatives(p);
  hessian_double_with(T, H);
}

void NdtCpu::hessian_double_with(const float* T, double H[36]) {
  evaluations++;
  for (int k = 0; k < 36; k++) H[k] = 0.0;
  int nthreads = prm.num_threads;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
  constexpr int64_t kChunk = 4096;
  std::vector<double> terms(static_cast<size_t>(kChunk) * 27 * 36);
  std::vector<int> counts(kChunk);
  for (int64_t base = 0; base < ns; base += kChunk) {
    const int64_t m = std::min<int64_t>(kChunk, ns - base);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t li = 0; li < m; li++) {
      const int64_t idx = base + li;
      const float* xp = source.data() + idx * 4;
      float xt[3];
      for (int r = 0; r < 3; r++) xt[r] = T[0 * 4 + r] * xp[0] + T[1 * 4 + r] * xp[1] + T[2 * 4 + r] * xp[2] + T[3 * 4 + r];
      const Leaf* nb[27];
      const int nn = neighbours(xt, nb);
      int cnt = 0;
      if (nn > 0) {
        const double x[3] = {xp[0], xp[1], xp[2]};
        auto dot3 = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
        // computePointDerivatives(x, point_gradient_, point_hessian_) -- the double overload
        double pg[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
        pg[1][3] = dot3(x, j_ang_d[0]); pg[2][3] = dot3(x, j_ang_d[1]);
        pg[0][4] = dot3(x, j_ang_d[2]); pg[1][4] = dot3(x, j_ang_d[3]); pg[2][4] = dot3(x, j_ang_d[4]);
        pg[0][5] = dot3(x, j_ang_d[5]); pg[1][5] = dot3(x, j_ang_d[6]); pg[2][5] = dot3(x, j_ang_d[7]);
        double ph[6][6][3];
        std::memset(ph, 0, sizeof(ph));
        {
          double xh[15];
          for (int i = 0; i < 15; i++) xh[i] = dot3(x, h_ang_d[i]);
          const double a[3] = {0, xh[0], xh[1]}, b[3] = {0, xh[2], xh[3]}, c[3] = {0, xh[4], xh[5]};
          const double d[3] = {xh[6], xh[7], xh[8]}, e[3] = {xh[9], xh[10], xh[11]}, f[3] = {xh[12], xh[13], xh[14]};
          for (int r = 0; r < 3; r++) {
            ph[3][3][r] = a[r]; ph[3][4][r] = b[r]; ph[3][5][r] = c[r];
            ph[4][3][r] = b[r]; ph[4][4][r] = d[r]; ph[4][5][r] = e[r];
            ph[5][3][r] = c[r]; ph[5][4][r] = e[r]; ph[5][5][r] = f[r];
          }
        }
        for (int v = 0; v < nn; v++) {
          const Leaf* cell = nb[v];
          const double* C = cell->icov;
          const double q[3] = {static_cast<double>(xt[0]) - cell->mean[0], static_cast<double>(xt[1]) - cell->mean[1], static_cast<double>(xt[2]) - cell->mean[2]};
          auto Cmul = [&](const double* w, double* out) {   // c_inv * w
            for (int r = 0; r < 3; r++) out[r] = C[r * 3 + 0] * w[0] + C[r * 3 + 1] * w[1] + C[r * 3 + 2] * w[2];
          };
          double Cq[3];
          Cmul(q, Cq);
          const double e_arg = -gauss_d2 * dot3(q, Cq) / 2;
          double e_x_cov_x = gauss_d2 * (prm.exp_libm == 1 ? glibc_exp(e_arg) : (prm.exp_libm == 2 ? std::exp(e_arg) : det_exp(e_arg)));
          if (e_x_cov_x > 1 || e_x_cov_x < 0 || e_x_cov_x != e_x_cov_x) continue;
          e_x_cov_x *= gauss_d1;
          double* out = terms.data() + (static_cast<size_t>(li) * 27 + cnt) * 36;
          cnt++;
          for (int i = 0; i < 6; i++) {
            const double pgi[3] = {pg[0][i], pg[1][i], pg[2][i]};
            double cov_dxd_pi[3];
            Cmul(pgi, cov_dxd_pi);
            for (int j = 0; j < 6; j++) {
              const double pgj[3] = {pg[0][j], pg[1][j], pg[2][j]};
              double Cpgj[3], Cph[3];
              Cmul(pgj, Cpgj);
              Cmul(ph[i][j], Cph);
              out[i * 6 + j] = e_x_cov_x * (-gauss_d2 * dot3(q, cov_dxd_pi) * dot3(q, Cpgj) + dot3(q, Cph) + dot3(pgj, cov_dxd_pi));
            }
          }
        }
      }
      counts[li] = cnt;
    }
    for (int64_t li = 0; li < m; li++)
      for (int v = 0; v < counts[li]; v++) {
        const double* t = terms.data() + (static_cast<size_t>(li) * 27 + v) * 36;
        for (int k = 0; k < 36; k++) H[k] += t[k];
      }
  }
}

// ---- More-Thuente helpers (PCL ndt.hpp restated; More & Thuente 1994, Sun & Yuan 2006) ------------
static inline double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
static inline double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

static double trial_value_selection_mt(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t,
                                       double g_t) {
  if (f_t > f_l) {  // case 1
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    if (std::fabs(a_c - a_l) < std::fabs(a_q - a_l)) return a_c;
    return 0.5 * (a_q + a_c);
  } else if (g_t * g_l < 0) {  // case 2
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    if (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) return a_c;
    return a_s;
  } else if (std::fabs(g_t) <= std::fabs(g_l)) {  // case 3
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
    const double w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double a_t_next = (std::fabs(a_c - a_t) < std::fabs(a_s - a_t)) ? a_c : a_s;
    if (a_t > a_l) return std::min(a_t + 0.66 * (a_u - a_t), a_t_next);
    return std::max(a_t + 0.66 * (a_u - a_t), a_t_next);
  } else {  // case 4
    const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
    const double w = std::sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
  }
}

static bool update_interval_mt(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t,
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

// computeTransformation + computeStepLengthMT (SURVEY App. A "Outer loop" / "More-Thuente")
NdtResult NdtCpu::align(const float* guess, double* trajectory, int* traj_len) {
  NdtResult res;
  std::memset(&res, 0, sizeof(res));
  evaluations = 0;
  int nr_iterations = 0;
  bool converged = false;

  const double c1 = 10.0 * (1.0 - prm.outlier_ratio);
  const double c2 = prm.outlier_ratio / std::pow(prm.resolution, 3);
  const double d3 = -std::log(c2);
  gauss_d1 = -std::log(c1 + c2) - d3;
  gauss_d2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / gauss_d1);

  float final_T[16];
  std::memcpy(final_T, guess, sizeof(final_T));
  float eul[3];
  if (prm.guess_rotation_polar) {   // eig_transformation.rotation().eulerAngles(0, 1, 2): Affine3f::rotation() is the polar factor
    float R[9], G[16];
    affine_rotation_f32(guess, R);
    std::memcpy(G, guess, sizeof(G));
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) G[c * 4 + r] = R[r * 3 + c];
    euler_angles_012(G, eul);
  } else {
    euler_angles_012(guess, eul);
  }
  double p[6] = {guess[12], guess[13], guess[14], eul[0], eul[1], eul[2]};
  double g[6], H[36], delta[6];
  int tl = 0;
  auto push_traj = [&]() {
    if (trajectory) std::memcpy(trajectory + 6 * tl, p, sizeof(double) * 6);
    tl++;
  };
  push_traj();

  double score = derivatives_with(guess, p, g, H, true);

  while (!converged) {
    double neg_g[6];
    for (int k = 0; k < 6; k++) neg_g[k] = -g[k];
    if (prm.newton_solver) jsvd_solve6(H, neg_g, delta); else svd_solve6(H, neg_g, delta);
    double norm = 0;
    for (int k = 0; k < 6; k++) norm += delta[k] * delta[k];
    norm = std::sqrt(norm);
    if (norm == 0 || norm != norm) {
      converged = (norm == norm);
      break;
    }
    for (int k = 0; k < 6; k++) delta[k] /= norm;

    // ---- computeStepLengthMT(p, delta, norm, step_size, eps/2, score, g, H) ----
    const double step_init = norm, step_max = prm.step_size, step_min = prm.transformation_epsilon / 2;
    double a_t;
    {
      const double phi_0 = -score;
      double d_phi_0 = 0;
      for (int k = 0; k < 6; k++) d_phi_0 -= g[k] * delta[k];
      bool zero_step = false;
      if (d_phi_0 >= 0) {
        if (d_phi_0 == 0) {
          zero_step = true;
        } else {
          d_phi_0 *= -1;
          for (int k = 0; k < 6; k++) delta[k] *= -1;
        }
      }
      if (zero_step) {
        a_t = 0;
      } else {
        const int max_step_iterations = prm.mt_max_step_iterations;
        int step_iterations = 0;
        const double mu = 1.e-4, nu = 0.9;
        double a_l = 0, a_u = 0;
        double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu), g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
        double f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu), g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
        // PCL >= 1.8.1 (and SURVEY App. A) initialise with `(step_max - step_min) < 0` (NDT_LS_MORE_THUENTE); older PCL
        // wrote `> 0`, which marks the interval converged whenever step_max > step_min so the trial loop below never
        // runs and every iteration costs one evaluation at the clamped Newton step (NDT_LS_FIXED_STEP).
        bool interval_converged = (prm.line_search == NDT_LS_FIXED_STEP) ? ((step_max - step_min) > 0) : ((step_max - step_min) < 0);
        bool open_interval = true;
        a_t = step_init;
        a_t = std::min(a_t, step_max);
        a_t = std::max(a_t, step_min);
        double x_t[6];
        for (int k = 0; k < 6; k++) x_t[k] = p[k] + delta[k] * a_t;
        pose_to_matrix_f32(x_t, final_T);
        score = derivatives_with(final_T, x_t, g, H, true);
        double phi_t = -score, d_phi_t = 0;
        for (int k = 0; k < 6; k++) d_phi_t -= g[k] * delta[k];
        double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
        while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
          if (open_interval)
            a_t = trial_value_selection_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
          else
            a_t = trial_value_selection_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
          a_t = std::min(a_t, step_max);
          a_t = std::max(a_t, step_min);
          for (int k = 0; k < 6; k++) x_t[k] = p[k] + delta[k] * a_t;
          pose_to_matrix_f32(x_t, final_T);
          score = derivatives_with(final_T, x_t, g, H, false);
          phi_t = -score;
          d_phi_t = 0;
          for (int k = 0; k < 6; k++) d_phi_t -= g[k] * delta[k];
          psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
          d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
          if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
            open_interval = false;
            f_l = f_l + phi_0 - mu * d_phi_0 * a_l;
            g_l = g_l + mu * d_phi_0;
            f_u = f_u + phi_0 - mu * d_phi_0 * a_u;
            g_u = g_u + mu * d_phi_0;
          }
          if (open_interval)
            interval_converged = update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
          else
            interval_converged = update_interval_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
          step_iterations++;
        }
        if (step_iterations) {  // computeHessian(hessian, trans_cloud, x_t) at the accepted point
          res.hessian_recomputes++;
          if (prm.hessian_recompute_double) {
            hessian_double_with(final_T, H);   // angle tables: those of the last trial's computeDerivatives, i.e. of x_t
          } else {   // rounds 1-3: the float pass again (score / gradient are re-derived identically)
            double g2[6];
            derivatives_with(final_T, x_t, g2, H, true);
          }
        }
      }
    }
    const double delta_p_norm = a_t;
    for (int k = 0; k < 6; k++) {
      delta[k] *= delta_p_norm;
      p[k] += delta[k];
    }
    push_traj();
    if (nr_iterations > prm.max_iterations || (nr_iterations && (std::fabs(delta_p_norm) < prm.transformation_epsilon))) converged = true;
    nr_iterations++;
  }

  std::memcpy(res.T, final_T, sizeof(final_T));
  res.converged = converged ? 1 : 0;
  res.iterations = nr_iterations;
  res.evaluations = evaluations;
  res.score = score;
  if (traj_len) *traj_len = tl;
  return res;
}

}  // namespace orc
