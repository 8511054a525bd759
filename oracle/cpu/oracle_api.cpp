// ORACLE -- TEST INFRASTRUCTURE ONLY.  Parity unpinned (see oracle/README.md).
#include "oracle_api.h"
#include "linalg.hpp"
#include "ndt_cpu.hpp"
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
  out->pad = 0;
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
void orc_ldlt_solve6(const double* A, const double* b, double* x) { ldlt_solve6(A, b, x); }
void orc_sym_eig3(const double* A, double* ev, double* V) { sym_eig3(A, ev, V); }
int32_t orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
}
