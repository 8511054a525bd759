// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/README.md).
#include "oracle_api.h"
#include "linalg.hpp"
#include "ndt_cpu.hpp"
#include "gicp_cpu.hpp"
#include "vgicp_cpu.hpp"
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

extern "C" {

void orc_ndt_default_params(orc_ndt_params* p) {
  NdtParams d;
  p->resolution = d.resolution;
  p->step_size = d.step_size;
  p->outlier_ratio = d.outlier_ratio;
  p->transformation_epsilon = d.transformation_epsilon;
  p->min_covar_eigvalue_mult = d.min_covar_eigvalue_mult;
  p->max_iterations = d.max_iterations;
  p->search_method = d.search_method;
  p->min_points_per_voxel = d.min_points_per_voxel;
  p->line_search = d.line_search;
  p->mt_max_step_iterations = d.mt_max_step_iterations;
  p->num_threads = d.num_threads;
  p->fix_hessian_d1 = d.fix_hessian_d1;
  p->exp_libm = d.exp_libm;
  p->newton_solver = d.newton_solver;
  p->hessian_recompute_double = d.hessian_recompute_double;
  p->guess_rotation_polar = d.guess_rotation_polar;
  p->cov_eigensolver = d.cov_eigensolver;
}

void* orc_ndt_create(const orc_ndt_params* p) {
  NdtParams d;
  d.resolution = p->resolution;
  d.step_size = p->step_size;
  d.outlier_ratio = p->outlier_ratio;
  d.transformation_epsilon = p->transformation_epsilon;
  d.min_covar_eigvalue_mult = p->min_covar_eigvalue_mult;
  d.max_iterations = p->max_iterations;
  d.search_method = p->search_method;
  d.min_points_per_voxel = p->min_points_per_voxel;
  d.line_search = p->line_search;
  d.mt_max_step_iterations = p->mt_max_step_iterations;
  d.num_threads = p->num_threads;
  d.fix_hessian_d1 = p->fix_hessian_d1;
  d.exp_libm = p->exp_libm;
  d.newton_solver = p->newton_solver;
  d.hessian_recompute_double = p->hessian_recompute_double;
  d.guess_rotation_polar = p->guess_rotation_polar;
  d.cov_eigensolver = p->cov_eigensolver;
  return new NdtCpu(d);
}
void orc_ndt_destroy(void* h) { delete static_cast<NdtCpu*>(h); }
void orc_ndt_set_target(void* h, const float* xyz16, int64_t n) { static_cast<NdtCpu*>(h)->set_target(xyz16, n); }
void orc_ndt_set_source(void* h, const float* xyz16, int64_t n) { static_cast<NdtCpu*>(h)->set_source(xyz16, n); }

void orc_ndt_align(void* h, const float* guess16, orc_result* out, double* trajectory, int32_t* traj_len) {
  int tl = 0;
  NdtResult r = static_cast<NdtCpu*>(h)->align(guess16, trajectory, &tl);
  std::memcpy(out->T, r.T, sizeof(r.T));
  out->converged = r.converged;
  out->iterations = r.iterations;
  out->evaluations = r.evaluations;
  out->pad = r.hessian_recomputes;   // diagnostics: computeHessian passes among the evaluations
  out->score = r.score;
  if (traj_len) *traj_len = tl;
}

double orc_ndt_derivatives(void* h, const double* p6, const float* T16, double* g6, double* H36, int32_t compute_hessian) {
  NdtCpu* n = static_cast<NdtCpu*>(h);
  // the gaussian constants are (re)computed by align(); make single evaluations self-contained
  const double c1 = 10.0 * (1.0 - n->prm.outlier_ratio);
  const double c2 = n->prm.outlier_ratio / std::pow(n->prm.resolution, 3);
  const double d3 = -std::log(c2);
  n->gauss_d1 = -std::log(c1 + c2) - d3;
  n->gauss_d2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / n->gauss_d1);
  if (T16) return n->derivatives_with(T16, p6, g6, H36, compute_hessian != 0);
  return n->derivatives(p6, g6, H36, compute_hessian != 0);
}

int64_t orc_ndt_voxels(void* h, int64_t* keys, int32_t* counts, int32_t* valid, double* mean3, double* cov9, double* icov9) {
  NdtCpu* n = static_cast<NdtCpu*>(h);
  int64_t k = 0;
  if (keys)
    for (auto& kv : n->leaves) {
      keys[k] = kv.first;
      counts[k] = kv.second.nr_points;
      valid[k] = kv.second.valid ? 1 : 0;
      std::memcpy(mean3 + 3 * k, kv.second.mean, sizeof(double) * 3);
      if (kv.second.valid) {
        std::memcpy(cov9 + 9 * k, kv.second.cov, sizeof(double) * 9);
        std::memcpy(icov9 + 9 * k, kv.second.icov, sizeof(double) * 9);
      } else {
        std::memset(cov9 + 9 * k, 0, sizeof(double) * 9);
        std::memset(icov9 + 9 * k, 0, sizeof(double) * 9);
      }
      k++;
    }
  return static_cast<int64_t>(n->leaves.size());
}

void orc_ndt_grid(void* h, int32_t* min_b3, int32_t* max_b3, int32_t* div_b3) {
  NdtCpu* n = static_cast<NdtCpu*>(h);
  for (int a = 0; a < 3; a++) {
    min_b3[a] = n->min_b[a];
    max_b3[a] = n->max_b[a];
    div_b3[a] = n->div_b[a];
  }
}

void orc_euler_angles_012(const float* T16, float* out3) { euler_angles_012(T16, out3); }
void orc_pose_to_matrix_f32(const double* p6, float* T16) { pose_to_matrix_f32(p6, T16); }
void orc_svd_solve6(const double* A, const double* b, double* x) { svd_solve6(A, b, x); }
void orc_jsvd_solve6(const double* A, const double* b, double* x, int32_t* sr) {
  JsvdStats st{0, 0};
  jsvd_solve6(A, b, x, &st);
  if (sr) { sr[0] = st.sweeps; sr[1] = st.rotations; }
}
void orc_affine_rotation_f32(const float* T16, float* R9) { affine_rotation_f32(T16, R9); }
void orc_ndt_hessian_double(void* h, const double* p6, double* H36) {
  NdtCpu* n = static_cast<NdtCpu*>(h);
  const double c1 = 10.0 * (1.0 - n->prm.outlier_ratio);   // as orc_ndt_derivatives: single evaluations are self-contained
  const double c2 = n->prm.outlier_ratio / std::pow(n->prm.resolution, 3);
  const double d3 = -std::log(c2);
  n->gauss_d1 = -std::log(c1 + c2) - d3;
  n->gauss_d2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / n->gauss_d1);
  n->hessian_double(p6, H36);
}
double orc_det_exp(double x) { return det_exp(x); }
float orc_glibc_expf(float x) { return glibc_expf(x); }
double orc_glibc_exp(double x) { return glibc_exp(x); }
// glibc_exp against the host libm's exp on n doubles drawn by a fixed xorshift stream over [-760, 720] (an eighth each positives up to overflow,
// negatives through the subnormal results to underflow and tiny magnitudes; the rest dense in [-60, 0], NDT's range): the number that differ
long long orc_glibc_exp_mismatches(long long n, uint64_t seed, double* first_bad) {
  long long bad = 0;
  double fb = 0.0;
#pragma omp parallel reduction(+ : bad)
  {
    uint64_t s = seed + 977ull * (uint64_t)omp_get_thread_num() + 1ull;
    const int nt = omp_get_num_threads();
    const long long mine = n / nt + 1;
    for (long long i = 0; i < mine; i++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const double u = (double)(s >> 11) / 9007199254740992.0;
      double x;
      switch (s & 7) {
        case 0: x = 720.0 * u; break;
        case 1: x = -760.0 * u; break;
        case 2: { const uint64_t b = 0x3ff0000000000000ull - (((s >> 20) % 60ull) << 52); double t; std::memcpy(&t, &b, 8); x = -t * u; } break;
        default: x = -60.0 * u * u; break;
      }
      const double a = exp(x), m = glibc_exp(x);
      uint64_t ua, um;
      std::memcpy(&ua, &a, 8);
      std::memcpy(&um, &m, 8);
      if (ua != um && !(a != a && m != m)) {
        bad++;
#pragma omp critical
        fb = x;
      }
    }
  }
  if (first_bad) *first_bad = fb;
  return bad;
}
// glibc_expf against the host libm's expf on every float whose bit pattern lies in [first_bits, last_bits] (both ends included): the
// number of floats whose results differ in any bit; *first_bad receives the bit pattern of the first one (or 0xFFFFFFFF)
long long orc_glibc_expf_mismatches(uint32_t first_bits, uint32_t last_bits, uint32_t* first_bad) {
  long long bad = 0;
  uint32_t fb = 0xFFFFFFFFu;
#pragma omp parallel for reduction(+ : bad) reduction(min : fb) schedule(static)
  for (long long b = (long long)first_bits; b <= (long long)last_bits; b++) {
    const uint32_t u = (uint32_t)b;
    float x;
    std::memcpy(&x, &u, 4);
    const float a = expf(x), m = glibc_expf(x);
    uint32_t ua, um;
    std::memcpy(&ua, &a, 4);
    std::memcpy(&um, &m, 4);
    if (ua != um && !(a != a && m != m)) {
      bad++;
      if (u < fb) fb = u;
    }
  }
  if (first_bad) *first_bad = fb;
  return bad;
}
void orc_ldlt_solve6(const double* A, const double* b, double* x) { ldlt_solve6(A, b, x); }
void orc_sym_eig3(const double* A, double* ev, double* V) { sym_eig3(A, ev, V); }
int orc_eigen_selfadjoint3(const double* A, double* ev, double* V) { return eigen_selfadjoint3(A, ev, V); }
int32_t orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_gicp_default_params(orc_gicp_params* p) {
  GicpParams d;
  p->transformation_epsilon = d.transformation_epsilon;
  p->rotation_epsilon = d.rotation_epsilon;
  p->max_correspondence_distance = d.max_correspondence_distance;
  p->lm_init_lambda_factor = d.lm_init_lambda_factor;
  p->max_iterations = d.max_iterations;
  p->k_correspondences = d.k_correspondences;
  p->regularization = d.regularization;
  p->optimizer = d.optimizer;
  p->lm_max_iterations = d.lm_max_iterations;
  p->num_threads = d.num_threads;
  p->cov_svd = d.cov_svd;
  p->pad0 = 0;
}
void* orc_gicp_create(const orc_gicp_params* p) {
  GicpParams d;
  d.transformation_epsilon = p->transformation_epsilon;
  d.rotation_epsilon = p->rotation_epsilon;
  d.max_correspondence_distance = p->max_correspondence_distance;
  d.lm_init_lambda_factor = p->lm_init_lambda_factor;
  d.max_iterations = p->max_iterations;
  d.k_correspondences = p->k_correspondences;
  d.regularization = p->regularization;
  d.optimizer = p->optimizer;
  d.lm_max_iterations = p->lm_max_iterations;
  d.num_threads = p->num_threads;
  d.cov_svd = p->cov_svd;
  return new GicpCpu(d);
}
static GicpParams gicp_params_from(const orc_gicp_params* p) {
  GicpParams d;
  d.transformation_epsilon = p->transformation_epsilon;
  d.rotation_epsilon = p->rotation_epsilon;
  d.max_correspondence_distance = p->max_correspondence_distance;
  d.lm_init_lambda_factor = p->lm_init_lambda_factor;
  d.max_iterations = p->max_iterations;
  d.k_correspondences = p->k_correspondences;
  d.regularization = p->regularization;
  d.optimizer = p->optimizer;
  d.lm_max_iterations = p->lm_max_iterations;
  d.num_threads = p->num_threads;
  d.cov_svd = p->cov_svd;
  return d;
}
/* FAST_VGICP: the returned object is used through the orc_gicp_* entry points */
void* orc_vgicp_create(const orc_gicp_params* p, double resolution, int32_t search_method) {
  return static_cast<GicpCpu*>(new VgicpCpu(gicp_params_from(p), resolution, search_method));
}
int64_t orc_vgicp_voxels(void* h, int32_t* coord3, int32_t* counts, double* mean3, double* cov9) {
  VgicpCpu* v = dynamic_cast<VgicpCpu*>(static_cast<GicpCpu*>(h));
  if (!v) return -1;
  if (!v->map_valid) v->build_voxelmap();
  if (coord3) v->dump_voxels(coord3, counts, mean3, cov9);
  return v->voxel_count();
}
void orc_gicp_destroy(void* h) { delete static_cast<GicpCpu*>(h); }
void orc_gicp_set_target(void* h, const float* xyz16, int64_t n) { static_cast<GicpCpu*>(h)->set_target(xyz16, n); }
void orc_gicp_set_source(void* h, const float* xyz16, int64_t n) { static_cast<GicpCpu*>(h)->set_source(xyz16, n); }
void orc_gicp_align(void* h, const float* guess16, orc_result* out) {
  GicpResult r = static_cast<GicpCpu*>(h)->align(guess16);
  std::memcpy(out->T, r.T, sizeof(r.T));
  out->converged = r.converged;
  out->iterations = r.iterations;
  out->evaluations = r.evaluations;
  out->pad = 0;
  out->score = r.error;
}
double orc_gicp_linearize(void* h, const double* T, double* H36, double* b6) { return static_cast<GicpCpu*>(h)->linearize(T, H36, b6); }
double orc_gicp_compute_error(void* h, const double* T) { return static_cast<GicpCpu*>(h)->compute_error(T); }
void orc_gicp_covariances(void* h, int32_t which, double* out9) {
  GicpCpu* g = static_cast<GicpCpu*>(h);
  g->ensure_covariances();
  const std::vector<double>& c = which ? g->cov_t : g->cov_s;
  std::memcpy(out9, c.data(), c.size() * sizeof(double));
}
void orc_gicp_correspondences(void* h, int32_t* corr, float* sq_dist) {
  GicpCpu* g = static_cast<GicpCpu*>(h);
  for (size_t i = 0; i < g->corr.size(); i++) { corr[i] = g->corr[i]; sq_dist[i] = g->sq_dist[i]; }
}
double orc_fitness_score(const float* target, int64_t nt, const float* source, int64_t ns, const float* T16, double max_range, double inlier_sq,
                         int64_t* n_used, int64_t* n_inliers) {
  return fitness_score(target, nt, source, ns, T16, max_range, inlier_sq, n_used, n_inliers);
}
void orc_knn(const float* cloud, int64_t n, const float* queries, int64_t m, int32_t k, int32_t* idx, float* d2) {
  KdTree t;
  t.build(cloud, n);
#pragma omp parallel for schedule(guided, 8)
  for (int64_t i = 0; i < m; i++) {
    const int found = t.knn(queries + i * 4, k, idx + i * k, d2 + i * k);
    for (int j = found; j < k; j++) { idx[i * k + j] = -1; d2[i * k + j] = INFINITY; }
  }
}
void orc_se3_exp(const double* a6, double* T) { se3_exp(a6, T); }
}
