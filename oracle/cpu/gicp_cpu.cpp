// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/oracle.py).
// CPU restatement of fast_gicp::FastGICP / LsqRegistration; see gicp_cpu.hpp for provenance.
#include "gicp_cpu.hpp"
#include "linalg.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

int GicpCpu::threads() const {
#ifdef _OPENMP
  return prm.num_threads > 0 ? prm.num_threads : omp_get_max_threads();
#else
  return 1;
#endif
}

void GicpCpu::set_target(const float* xyz16, int64_t n) {
  target.assign(xyz16, xyz16 + n * 4);
  nt = n;
  tree_t.build(target.data(), n);
  cov_t.clear();
}

void GicpCpu::set_source(const float* xyz16, int64_t n) {
  source.assign(xyz16, xyz16 + n * 4);
  ns = n;
  tree_s.build(source.data(), n);
  cov_s.clear();
}

// so3_exp / se3_exp of fast_gicp's so3.hpp
void se3_exp(const double* a, double* T) {
  const double wx = a[0], wy = a[1], wz = a[2];
  const double theta_sq = wx * wx + wy * wy + wz * wz;
  double imag, real;
  if (theta_sq < 1e-10) {
    const double tq = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * tq;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * tq;
  } else {
    const double theta = std::sqrt(theta_sq), half = 0.5 * theta;
    imag = std::sin(half) / theta;
    real = std::cos(half);
  }
  // Eigen::Quaterniond(real, imag*w).toRotationMatrix()
  const double qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  const double theta = std::sqrt(theta_sq);
  double V[9];
  if (theta < 1e-10) {
    std::memcpy(V, R, sizeof(V));
  } else {
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
    mat3_mul(O, O, O2);
    const double c1 = (1.0 - std::cos(theta)) / theta_sq, c2 = (theta - std::sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
  }
  for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[r * 4 + c] = R[r * 3 + c];
    T[r * 4 + 3] = V[r * 3 + 0] * a[3] + V[r * 3 + 1] * a[4] + V[r * 3 + 2] * a[5];
  }
}

// FastGICP::calculate_covariances
void GicpCpu::calc_covariances(const std::vector<float>& cloud, int64_t n, const KdTree& tree, std::vector<double>& covs) {
  const int k = prm.k_correspondences;
  covs.assign(static_cast<size_t>(n) * 9, 0.0);
#pragma omp parallel for num_threads(threads()) schedule(guided, 8)
  for (int64_t i = 0; i < n; i++) {
    std::vector<int> ki(k);
    std::vector<float> kd(k);
    const int found = tree.knn(cloud.data() + i * 4, k, ki.data(), kd.data());
    // neighbors is a 4 x k matrix upstream; with fewer than k points in the cloud the missing columns stay zero
    double mean[3] = {0, 0, 0};
    for (int j = 0; j < found; j++)
      for (int a = 0; a < 3; a++) mean[a] += cloud[static_cast<size_t>(ki[j]) * 4 + a];
    for (int a = 0; a < 3; a++) mean[a] /= k;
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < k; j++) {
      double d[3];
      for (int a = 0; a < 3; a++) d[a] = (j < found ? static_cast<double>(cloud[static_cast<size_t>(ki[j]) * 4 + a]) : 0.0) - mean[a];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) cov[a * 3 + b] += d[a] * d[b];
    }
    for (int a = 0; a < 9; a++) cov[a] /= k;
    double* out = covs.data() + static_cast<size_t>(i) * 9;
    if (prm.regularization == GICP_REG_NONE) {
      std::memcpy(out, cov, sizeof(cov));
    } else if (prm.regularization == GICP_REG_FROBENIUS) {
      const double lambda = 1e-3;
      double C[9], Ci[9];
      for (int a = 0; a < 9; a++) C[a] = cov[a] + ((a % 4 == 0) ? lambda : 0.0);
      inv3(C, Ci);
      double nrm = 0;
      for (int a = 0; a < 9; a++) nrm += Ci[a] * Ci[a];
      nrm = std::sqrt(nrm);
      for (int a = 0; a < 9; a++) Ci[a] /= nrm;
      inv3(Ci, out);
    } else {
      // Eigen::JacobiSVD<Matrix3d> svd(cov, ComputeFullU | ComputeFullV); cov = U * values.asDiagonal() * V^T   (singular values descending)
      double U[9], V[9], sv[3];
      int col[3] = {0, 1, 2};
      if (prm.cov_svd) {
        jacobi_svd_square<double, 3>(cov, U, V, sv);
      } else {   // rounds 1-3: the eigen-decomposition of the symmetric PSD matrix (same factors up to rounding, U = V)
        double ev[3];
        sym_eig3(cov, ev, V);
        std::memcpy(U, V, sizeof(U));
        sv[0] = std::fabs(ev[2]); sv[1] = std::fabs(ev[1]); sv[2] = std::fabs(ev[0]);
        col[0] = 2; col[1] = 1; col[2] = 0;
      }
      double vals[3];
      if (prm.regularization == GICP_REG_PLANE) {
        vals[0] = 1; vals[1] = 1; vals[2] = 1e-3;
      } else if (prm.regularization == GICP_REG_MIN_EIG) {
        for (int a = 0; a < 3; a++) vals[a] = std::max(sv[a], 1e-3);
      } else {
        for (int a = 0; a < 3; a++) vals[a] = std::max(sv[a] / sv[0], 1e-3);
      }
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
          double s = 0;
          for (int a = 0; a < 3; a++) s += U[r * 3 + col[a]] * vals[a] * V[c * 3 + col[a]];
          out[r * 3 + c] = s;
        }
    }
  }
}

void GicpCpu::ensure_covariances() {
  if (cov_s.size() != static_cast<size_t>(ns) * 9) calc_covariances(source, ns, tree_s, cov_s);
  if (cov_t.size() != static_cast<size_t>(nt) * 9) calc_covariances(target, nt, tree_t, cov_t);
}

// FastGICP::update_correspondences: T is a row-major double 4x4 (Eigen::Isometry3d)
void GicpCpu::update_correspondences(const double* T) {
  corr.assign(ns, -1);
  sq_dist.assign(ns, 0.f);
  mahal.assign(static_cast<size_t>(ns) * 9, 0.0);
  float Tf[12];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) Tf[r * 4 + c] = static_cast<float>(T[r * 4 + c]);
  const float dmax = static_cast<float>(prm.max_correspondence_distance);  // corr_dist_threshold_ is a float upstream
  const float dmax2 = dmax * dmax;
#pragma omp parallel for num_threads(threads()) schedule(guided, 8)
  for (int64_t i = 0; i < ns; i++) {
    const float* p = source.data() + i * 4;
    float pt[3];
    for (int r = 0; r < 3; r++) pt[r] = ((Tf[r * 4 + 0] * p[0] + Tf[r * 4 + 1] * p[1]) + Tf[r * 4 + 2] * p[2]) + Tf[r * 4 + 3];
    int ki;
    float kd;
    if (tree_t.knn(pt, 1, &ki, &kd) < 1) continue;
    sq_dist[i] = kd;
    if (!(kd < dmax2)) continue;
    corr[i] = ki;
    // RCR = cov_B + T cov_A T^T (3x3 block), inverse
    const double* CA = cov_s.data() + static_cast<size_t>(i) * 9;
    const double* CB = cov_t.data() + static_cast<size_t>(ki) * 9;
    double R[9], RC[9], RCR[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) R[r * 3 + c] = T[r * 4 + c];
    mat3_mul(R, CA, RC);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) RCR[r * 3 + c] = CB[r * 3 + c] + (RC[r * 3 + 0] * R[c * 3 + 0] + RC[r * 3 + 1] * R[c * 3 + 1] + RC[r * 3 + 2] * R[c * 3 + 2]);
    inv3(RCR, mahal.data() + static_cast<size_t>(i) * 9);
  }
}

double GicpCpu::linearize(const double* T, double* H, double* b) {
  evaluations++;
  ensure_covariances();
  update_correspondences(T);
  const int nth = threads();
  std::vector<double> Hs(static_cast<size_t>(nth) * 36, 0.0), bs(static_cast<size_t>(nth) * 6, 0.0);
  double sum_errors = 0.0;
#pragma omp parallel for num_threads(nth) reduction(+ : sum_errors) schedule(guided, 8)
  for (int64_t i = 0; i < ns; i++) {
    const int j = corr[i];
    if (j < 0) continue;
    const float* pa = source.data() + i * 4;
    const float* pb = target.data() + static_cast<size_t>(j) * 4;
    double ta[3], e[3];
    for (int r = 0; r < 3; r++) ta[r] = T[r * 4 + 0] * pa[0] + T[r * 4 + 1] * pa[1] + T[r * 4 + 2] * pa[2] + T[r * 4 + 3];
    for (int r = 0; r < 3; r++) e[r] = static_cast<double>(pb[r]) - ta[r];
    const double* M = mahal.data() + static_cast<size_t>(i) * 9;
    double Me[3];
    for (int r = 0; r < 3; r++) Me[r] = M[r * 3 + 0] * e[0] + M[r * 3 + 1] * e[1] + M[r * 3 + 2] * e[2];
    sum_errors += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
    // dtdx0 = [ skew(T a) | -I ]  (3 x 6)
    const double J[3][6] = {{0, -ta[2], ta[1], -1, 0, 0}, {ta[2], 0, -ta[0], 0, -1, 0}, {-ta[1], ta[0], 0, 0, 0, -1}};
    double MJ[3][6];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 6; c++) MJ[r][c] = M[r * 3 + 0] * J[0][c] + M[r * 3 + 1] * J[1][c] + M[r * 3 + 2] * J[2][c];
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    double* Ht = Hs.data() + static_cast<size_t>(t) * 36;
    double* bt = bs.data() + static_cast<size_t>(t) * 6;
    for (int r = 0; r < 6; r++) {
      for (int c = 0; c < 6; c++) Ht[r * 6 + c] += J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c];
      bt[r] += J[0][r] * Me[0] + J[1][r] * Me[1] + J[2][r] * Me[2];
    }
  }
  for (int k = 0; k < 36; k++) H[k] = 0;
  for (int k = 0; k < 6; k++) b[k] = 0;
  for (int t = 0; t < nth; t++) {
    for (int k = 0; k < 36; k++) H[k] += Hs[static_cast<size_t>(t) * 36 + k];
    for (int k = 0; k < 6; k++) b[k] += bs[static_cast<size_t>(t) * 6 + k];
  }
  return sum_errors;
}

double GicpCpu::compute_error(const double* T) {
  evaluations++;
  double sum_errors = 0.0;
#pragma omp parallel for num_threads(threads()) reduction(+ : sum_errors) schedule(guided, 8)
  for (int64_t i = 0; i < ns; i++) {
    const int j = corr[i];
    if (j < 0) continue;
    const float* pa = source.data() + i * 4;
    const float* pb = target.data() + static_cast<size_t>(j) * 4;
    double e[3];
    for (int r = 0; r < 3; r++) e[r] = static_cast<double>(pb[r]) - (T[r * 4 + 0] * pa[0] + T[r * 4 + 1] * pa[1] + T[r * 4 + 2] * pa[2] + T[r * 4 + 3]);
    const double* M = mahal.data() + static_cast<size_t>(i) * 9;
    double Me[3];
    for (int r = 0; r < 3; r++) Me[r] = M[r * 3 + 0] * e[0] + M[r * 3 + 1] * e[1] + M[r * 3 + 2] * e[2];
    sum_errors += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
  }
  return sum_errors;
}

static void mat4_mul(const double* A, const double* B, double* C) {
  double T[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) T[r * 4 + c] = A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c] + A[r * 4 + 2] * B[2 * 4 + c] + A[r * 4 + 3] * B[3 * 4 + c];
  std::memcpy(C, T, sizeof(T));
}

// LsqRegistration::computeTransformation / step_lm / step_gn / is_converged
GicpResult GicpCpu::align(const float* guess) {
  GicpResult res;
  std::memset(&res, 0, sizeof(res));
  evaluations = 0;
  ensure_covariances();
  double x0[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) x0[r * 4 + c] = static_cast<double>(guess[c * 4 + r]);
  double lm_lambda = -1.0;
  bool converged = false;
  int nr_iterations = 0;
  double last_error = 0;
  auto is_converged = [&](const double* delta) {
    double rmax = 0, tmax = 0;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) rmax = std::max(rmax, std::fabs(delta[r * 4 + c] - (r == c ? 1.0 : 0.0)) / prm.rotation_epsilon);
      tmax = std::max(tmax, std::fabs(delta[r * 4 + 3]) / prm.transformation_epsilon);
    }
    return std::max(rmax, tmax) < 1;
  };
  for (int i = 0; i < prm.max_iterations && !converged; i++) {
    nr_iterations = i;
    double H[36], b[6], delta[16];
    bool ok = false;
    const double y0 = linearize(x0, H, b);
    last_error = y0;
    if (prm.optimizer == GICP_OPT_GN) {
      double nb[6], d[6];
      for (int k = 0; k < 6; k++) nb[k] = -b[k];
      ldlt_solve6(H, nb, d);
      se3_exp(d, delta);
      mat4_mul(delta, x0, x0);
      ok = true;
    } else {
      if (lm_lambda < 0.0) {
        double m = 0;
        for (int k = 0; k < 6; k++) m = std::max(m, std::fabs(H[k * 6 + k]));
        lm_lambda = prm.lm_init_lambda_factor * m;
      }
      double nu = 2.0;
      for (int t = 0; t < prm.lm_max_iterations; t++) {
        double Hl[36], nb[6], d[6], xi[16];
        std::memcpy(Hl, H, sizeof(Hl));
        for (int k = 0; k < 6; k++) { Hl[k * 6 + k] += lm_lambda; nb[k] = -b[k]; }
        ldlt_solve6(Hl, nb, d);
        se3_exp(d, delta);
        mat4_mul(delta, x0, xi);
        const double yi = compute_error(xi);
        double denom = 0;
        for (int k = 0; k < 6; k++) denom += d[k] * (lm_lambda * d[k] - b[k]);
        const double rho = (y0 - yi) / denom;
        if (rho < 0) {
          if (is_converged(delta)) { ok = true; break; }
          lm_lambda = nu * lm_lambda;
          nu = 2 * nu;
          continue;
        }
        std::memcpy(x0, xi, sizeof(xi));
        lm_lambda = lm_lambda * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
        last_error = yi;
        ok = true;
        break;
      }
    }
    if (!ok) break;  // "lm not converged!!"
    converged = is_converged(delta);
  }
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) res.T[c * 4 + r] = static_cast<float>(x0[r * 4 + c]);
  res.converged = converged ? 1 : 0;
  res.iterations = nr_iterations;
  res.evaluations = evaluations;
  res.error = last_error;
  return res;
}

// pcl::Registration::getFitnessScore + the inlier count of scan_matching_odometry_nodelet.cpp:321-332
double fitness_score(const float* target, int64_t nt, const float* source, int64_t ns, const float* T, double max_range, double inlier_sq,
                     int64_t* n_used, int64_t* n_inliers) {
  KdTree tree;
  tree.build(target, nt);
  double sum = 0;
  int64_t nr = 0, inl = 0;
  for (int64_t i = 0; i < ns; i++) {
    const float* p = source + i * 4;
    float pt[3];
    for (int r = 0; r < 3; r++) pt[r] = ((T[0 * 4 + r] * p[0] + T[1 * 4 + r] * p[1]) + T[2 * 4 + r] * p[2]) + T[3 * 4 + r];
    int ki;
    float kd;
    if (tree.knn(pt, 1, &ki, &kd) < 1) continue;
    if (kd <= max_range) {  // squared distance against max_range, as PCL does
      sum += kd;
      nr++;
    }
    if (kd < inlier_sq) inl++;
  }
  if (n_used) *n_used = nr;
  if (n_inliers) *n_inliers = inl;
  return nr > 0 ? sum / static_cast<double>(nr) : DBL_MAX;
}

}  // namespace orc
