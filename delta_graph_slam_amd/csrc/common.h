// Shared host/device definitions of libdgs_reg.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/dgs_reg.h"

namespace dgs {

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup: one wave per SIMD
constexpr int kAccum = 28;         // score + 6 gradient + 21 upper-triangle Hessian terms
constexpr int kAccumPad = 32;      // partial-sum row stride (doubles)
constexpr int kMaxPartialBlocks = 1024;  // per pair
constexpr int kStrictAccum = 43;   // ndt_strict_order: score + 6 gradient + the full 6x6 Hessian (upstream's is not exactly symmetric)
constexpr int kStrictPad = 48;
constexpr int kTrajCap = 72;

// ---- NDT voxel-Gaussian target model in HBM ------------------------------------------------------------
// One 48-byte record per occupied voxel (three 16-byte loads): the mean stays double because upstream forms
// q = float(double(x') - mean), the inverse covariance is what upstream casts to float.
struct __attribute__((aligned(16))) VoxelRec {
  double mean[3];
  float icov[6];  // xx, xy, xz, yy, yz, zz
};
static_assert(sizeof(VoxelRec) == 48, "VoxelRec must be 48 bytes");

// The same voxel as the UPSTREAM evaluation orders read it (ndt_strict_order >= 1): all nine entries of the inverse covariance
// as upstream casts them to float (after the eigenvalue clamp the matrix is rebuilt as V diag V^-1 and is not exactly symmetric).
// 64 bytes, four aligned 16-byte loads.  (The double inverse covariance of the double-precision computeHessian pass is read from
// the 96-byte table the test hook exposes: dgs_handle::vox_dbg.)
struct __attribute__((aligned(16))) VoxelStrictRec {
  double mean[3];
  float C[9];   // row-major float(icov)
  float pad;
};
static_assert(sizeof(VoxelStrictRec) == 64, "VoxelStrictRec must be 64 bytes");

struct VoxelGrid {
  int min_b[3];
  int max_b[3];
  int div_b[3];
  int mul1, mul2;        // divb_mul = (1, div0, div0*div1)
  float leaf;            // leaf_size (all axes equal: setResolution)
  float inv_leaf;        // 1 / leaf_size as float (VoxelGridCovariance::inverse_leaf_size_)
  const int* cell2vox;   // dense [div0*div1*div2] -> voxel id or -1
  const VoxelRec* vox;   // [n_occupied]
  const float4* centroid;  // float centroid + (w = 1 valid / 0 invalid), KDTREE mode
};

// ---- per-pair NDT optimiser state (device resident; written by the init / solve kernels) ---------------
enum NdtPhase : int {
  PH_INIT_EVAL = 0,   // waiting for derivatives at the initial guess
  PH_MT_FIRST = 1,    // waiting for full derivatives at the first trial step of this iteration
  PH_MT_TRIAL = 2,    // waiting for score + gradient at a More-Thuente trial step
  PH_MT_HESSIAN = 3,  // waiting for the Hessian at the accepted step
  PH_PROBE = 4,       // test hook: reduce only
  PH_DONE = 5,
  PH_SOLVE_PENDING = 6   // upstream order: the iteration has been closed, its Newton step is left to ndt_strict_solve_kernel (ndt_strict.h)
};

// Optimiser state proper: the solve kernel copies it into registers, advances it with every lane of one wave computing
// the same values, and lane 0 writes it back.
struct NdtSolver {
  int phase;
  int nr_iterations;
  int evaluations;
  int converged;
  int step_iterations;
  int interval_converged;
  int open_interval;
  int traj_len;        // test hook: pose after every outer iteration (first kTrajCap entries)
  double p[6];         // current pose (x, y, z, rx, ry, rz)
  double x_t[6];       // pose of the evaluation in flight
  double dir[6];       // unit Newton direction (possibly reversed)
  double score;        // last evaluated score / gradient / Hessian
  double grad[6];
  double hess[36];
  double phi_0, d_phi_0;
  double a_t, a_l, f_l, g_l, a_u, f_u, g_u;
  double step_init;
  // Upstream orders: the trial points of the current line search and what they evaluated to.  More-Thuente's step is clamped to
  // [eps / 2, step_size], so a search often evaluates the SAME pose again -- on the CPU that gives the same doubles and updateIntervalMT's
  // `f_t > f_l` is decided by equality; here the sums of two launches of different composition differ in their last bits (~1e-14), which
  // flipped that test on 3 of ~700 pairs at transformation_epsilon = 0.1 (soak, profiles/r04).  A pose that was evaluated before in this line
  // search takes its earlier value (ndt_advance); reset by begin_iteration.
  static constexpr int kTrialCache = 3;
  int trial_n, trial_next;
  double trial_x[kTrialCache][6];
  double trial_score[kTrialCache];
  double trial_grad[kTrialCache][6];
};

struct NdtPair {
  // -- read by the derivative kernel
  float T[12];         // row-major 3x4 float transform of this evaluation
  float jang[8][3];    // eq. 6.19 tables (float, from double trig)
  float hang[15][3];   // eq. 6.21 tables
  int need_hessian;
  int active;          // 0: every kernel returns immediately for this pair
  int last_launch;     // fused launches: index of the last launch this pair takes part in (INT_MAX while it iterates).  Written by
                       // the pair's closing workgroup DURING a launch without changing what workgroups of that launch read
  int ticket;          // fused launches: slices of this pair that have finished the current launch
  NdtSolver s;
  float final_T[16];   // column-major, = final_transformation_
  double traj[kTrajCap][6];
  // need_hessian == 2 (computeHessian in PCL's double form, dgs_params.ndt_hessian_recompute_double): the double angle vectors
  // j_ang_a_ .. h_ang_f3_ of the evaluation's pose, which upstream keeps beside the float matrices
  double jang_d[8][3];
  double hang_d[15][3];
  // fused launches of the upstream order (ndt_strict.h), two kernels per round.  The first kernel (evaluation kinds 0 / 1) of round r
  // takes a pair when r <= serve[0] or r == serve[2]; the second (kind 2, the double computeHessian pass) when r == serve[1].
  //   serve[0]: raised to r + 1 by the pair's closing in the first kernel of round r when the evaluation it queues is of kind 0 / 1.  The
  //             word only grows, and never past the reading round plus one: every workgroup of the running launch keeps seeing the pair.
  //   serve[1]: set to r by that closing when it queues kind 2: the second kernel of the SAME round serves it.
  //   serve[2]: set to r + lag by the pair's closing in the second kernel of round r.  With lag = 2 the second kernel of round r may run
  //             BESIDE the first kernel of round r + 1 (its own stream): an exact match, so a value written while that launch runs
  //             (r + 2) is read as "not this round" before and after the store alike.
  // The item-compacted kernel (one launch per round) uses serve[0] alone.
  int serve[4];
  // Speculated Newton steps of the upstream order (ndt_strict.h "deferred exact solve"): the closing publishes the next evaluation from the
  // 2-us Gauss-Jordan direction (spec_pending = 1); one wave of the pair's first workgroup in the NEXT launch runs the exact Jacobi-SVD step
  // beside the derivative work and leaves the exact optimiser state here with spec_result = 1 (the exact step yields the very float header
  // that was published: adopt the state) or 2 (it does not, or the exact step ends the registration: discard the evaluation, decide exactly).
  NdtSolver spec_s;
  int spec_pending, spec_result, pad_spec[2];
};

// ---- per-pair GICP optimiser state (fast_gicp::LsqRegistration) -------------------------------
enum GicpPhase : int { GP_LINEARIZE_WAIT = 0, GP_ERROR_WAIT = 1, GP_DONE = 2, GP_PROBE = 3 };

struct GicpSolver {
  int phase, iteration, evaluations, converged, lm_try, pad;
  double x0[12];     // current pose, rows 0..2 of the double 4x4 (Eigen::Isometry3d), row-major 3x4
  double xi[12];     // trial pose delta * x0
  double delta[12];  // last se3_exp(d)
  double H[36], b[6], d[6];
  double y0, yi, lambda, nu;
};

struct GicpPair {
  double Teval[12];  // pose of the evaluation in flight
  int eval_kind;     // 0: update_correspondences + linearize, 1: compute_error on the stored correspondences
  int active;
  GicpSolver s;
  float final_T[16];  // column-major
  int last_launch;    // fused rounds: the pair takes part in the linearize launches numbered <= last_launch (written by its closing workgroup)
  int ticket;         // fused rounds: workgroups of this pair that have published their row in the running launch
};

// ---- FAST_VGICP target model: GaussianVoxelMap (ADDITIVE) ----------------------------------------------------------------
struct VgicpVoxel {   // 96 B, six aligned 16-B loads
  double mean[3];     // sum of the voxel's points / n
  double w;           // sqrt(n): weight of a correspondence with this voxel
  double cov[6];      // sum of the points' regularised covariances / n: xx, xy, xz, yy, yz, zz
  int n, coord[3];    // points in the voxel; voxel coordinate floor(x / resolution - 0.5)
};

struct VgicpMap {
  int min_c[3], div[3];  // coordinate of cell 0 and grid extents (from the target's AABB)
  int mul1, mul2;
  double resolution;
  const int* cell2vox;   // dense [div0 * div1 * div2] -> voxel id or -1
  const VgicpVoxel* vox;
  int n_offsets;         // 1 / 7 / 27 voxels searched per source point
  int search;            // dgs_vgicp_search
};

struct GicpItem {  // one registration of a batch: its source cloud (with index and covariances) and its work arrays
  const float4* src;         // source points in the caller's order
  const float4* src_sorted;  // the same points in their own index's Hilbert order, w = original index
  const double* cov_s;       // 6 doubles per source point
  int* corr;                 // per source point: index of its target correspondence or -1
  float* corr_sq;
  double* mahal;             // 6 doubles per source point
  int n, pad;
};

struct GicpConsts {
  double trans_eps, rot_eps, lm_init_lambda_factor;
  float max_corr_sq;  // corr_dist_threshold_^2 as upstream forms it (float)
  int max_iterations, optimizer, lm_max_iterations, k, regularization;
};

struct NdtConsts {
  double gauss_d1, gauss_d2;
  double step_size, trans_eps;
  int max_iterations, line_search, mt_max_step_iterations, fix_hessian_d1;
  int search_method;
  int strict_order;  // dgs_ndt_strict_order
  int newton_solver;     // upstream orders: 1 = Eigen's two-sided JacobiSVD sequence (solve6.h jsvd_solve6_wave), 0 = one-sided Hestenes Jacobi
  int hessian_double;    // upstream orders: computeStepLengthMT's closing computeHessian in PCL's double form (evaluation kind 2)
  int exp_libm;          // upstream orders: updateDerivatives' std::exp(float) as glibc computes it (glibc_expf_dev), 0 = det_expf (rounds 1-3)
};

struct NdtInit {  // host -> device per pair, per align
  float guess[16];  // column-major
  double p0[6];
};

// Deals the workgroups of a launch evenly to the pairs of a batch for which pred(pair index) holds (the pairs that still
// iterate).  Every wave derives the same mapping: one strided load + ballot per 64 pairs, no inter-block traffic.
// Returns false when this workgroup has nothing to do.  gridDim.x must be >= the number of pairs.
#ifdef __HIPCC__
template <class Pred>
__device__ inline bool deal_workgroup(const int n_pairs, const int cap_blocks, Pred pred, int& pair, int& slice, int& blocks_per_pair, int* n_active_out = nullptr,
                                      const int block_id_in = -1, const int grid_in = -1) {
  const int block_id = block_id_in >= 0 ? block_id_in : (int)blockIdx.x, grid = grid_in >= 0 ? grid_in : (int)gridDim.x;   // a kernel may keep some workgroups for other work
  const int lane_id = threadIdx.x & 63;
  int n_active = 0;
  for (int c0 = 0; c0 < n_pairs; c0 += 64) {
    const int pi = c0 + lane_id;
    const bool a = (pi < n_pairs) && pred(pi);
    n_active += __popcll(__ballot(a));
  }
  if (n_active_out) *n_active_out = n_active;
  if (n_active == 0) return false;
  blocks_per_pair = max(1, min(grid / n_active, cap_blocks));
  const int rank = block_id / blocks_per_pair;
  slice = block_id % blocks_per_pair;
  if (rank >= n_active) return false;
  int found = -1, seen = 0;
  for (int c0 = 0; c0 < n_pairs && found < 0; c0 += 64) {
    const int pi = c0 + lane_id;
    const bool a = (pi < n_pairs) && pred(pi);
    unsigned long long m = __ballot(a);
    const int cnt = __popcll(m);
    if (rank < seen + cnt) {
      for (int k = rank - seen; k > 0; k--) m &= m - 1ull;  // drop the (rank - seen) lowest set bits
      found = c0 + __ffsll((long long)m) - 1;
    }
    seen += cnt;
  }
  pair = __builtin_amdgcn_readfirstlane(found);
  return true;
}
#endif

// ---- in-launch hand-off of partial rows to a pair's closing workgroup (fused NDT / GICP launches) ------------------------------------
// Default = the WRITE-THROUGH form of the agent-scope recipe: every byte of a row is stored with an agent-scope atomic store
// (sc1: written through to memory, never left dirty in this XCD's L2), the storing wave drains its stores (s_waitcnt vmcnt(0)), a
// workgroup barrier, then ONE lane takes the pair's ticket with an agent-scope atomic add; the workgroup whose add came last reads
// the rows with agent-scope atomic loads (sc1: never served from a line this XCD cached earlier).  Coherence is per access, so no
// cache-wide write-back / invalidate is paid.  tests/test_isa_handoff.py checks that the compiled kernels contain exactly this
// sequence (sc1 stores, the drain, the barrier before the atomic, sc1 loads behind it), so a compiler change cannot silently
// reorder it.  -DDGS_HANDOFF_FENCES (`make fences`) builds the textbook form instead -- plain row stores / loads, the ticket an
// acq_rel RMW, i.e. buffer_wbl2 + buffer_inv around it -- for A/B runs (bit-equal results, slower: DESIGN.md).
#ifdef __HIPCC__
#ifdef DGS_HANDOFF_FENCES
constexpr bool kHandoffFences = true;
#else
constexpr bool kHandoffFences = false;
#endif
__device__ __forceinline__ void handoff_store_row(double* p, double v) {
  if (kHandoffFences) *p = v;
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void handoff_drain_stores() {   // by the storing wave, before the workgroup barrier in front of the ticket
  if (!kHandoffFences) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ double handoff_load_row(const double* p) {
  return kHandoffFences ? *p : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane per workgroup, behind a workgroup barrier; true for the workgroup that took the pair's last ticket of this launch
__device__ __forceinline__ bool handoff_take_ticket(int* ticket, int n_slices) {
  const int t = kHandoffFences ? __hip_atomic_fetch_add(ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                               : __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool last = t == n_slices - 1;
  if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch (ordered by the kernel boundary)
  return last;
}
#endif

// ---- error handling --------------------------------------------------------------------------------------
#define DGS_HIP_TRY(h, expr)                                                                      \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                               \
      return DGS_ERR_HIP;                                                                         \
    }                                                                                             \
  } while (0)

template <typename T>
struct DevBuf {
  T* ptr = nullptr;
  size_t cap = 0;  // elements
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    size_t want = n + n / 4 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

// ---- device helpers --------------------------------------------------------------------------------------
// Individually rounded float ops.  hipcc contracts a*b+c into FMA by default (-ffp-contract=fast-honor-pragmas) and
// HIP's __fmul_rn/__fadd_rn are plain operators, so the pragma is what keeps these two roundings apart.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
// pcl::transformPointCloud row in float: ((m0 x + m1 y) + m2 z) + m3, every step rounded
__device__ __forceinline__ float affine_row_rn(float m0, float m1, float m2, float m3, float x, float y, float z) {
  return add_rn(add_rn(add_rn(mul_rn(m0, x), mul_rn(m1, y)), mul_rn(m2, z)), m3);
}
// exp(float) as a fixed sequence of IEEE double operations rounded once to float (no contraction): platform independent, within
// ~1e-16 of exp before the rounding.  ndt_strict_order evaluations use it so that a CPU run of the same sequence (the checker
// carries its own statement of it) can be compared bit for bit; the default path uses the hardware exponential.
__device__ __forceinline__ float det_expf(float xf) {
#pragma clang fp contract(off)
  const double x = (double)xf;
  if (x != x) return xf;
  if (x < -104.0) return 0.0f;  // below half the smallest subnormal float
  if (x > 89.0) return __builtin_inff();
  const double kd = floor(x * 1.4426950408889634 + 0.5);  // round(x / ln 2)
  const double r = (x - kd * 0x1.62e42fefa38p-1) - kd * 0x1.ef35793c7673p-45;  // ln 2 split hi / lo; |r| <= 0.3466
  double p = 1.0 / 6227020800.0;  // Taylor to r^13 / 13!
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
  const long long k = (long long)kd;
  const double s = __longlong_as_double((k + 1023) << 52);  // 2^k, k in [-151, 129]: a normal double
  return (float)(p * s);
}
// std::exp(float) as glibc >= 2.27 computes it on an FMA-capable x86-64 (sysdeps/ieee754/flt-32/e_expf.c: x N / ln2 = k + r, N = 32, a table of
// 2^(i/32), a cubic in r, all in double, one rounding to float): the exponential of upstream's updateDerivatives bit for bit -- the CPU checker
// carries the same sequence and compares IT with its libm on every float in [-104, 0].  `tab`: the 32 table words (kGlibcExp2fTab, or a copy in LDS).
__device__ __forceinline__ float glibc_expf_dev(float xf, const unsigned long long* __restrict__ tab) {
#pragma clang fp contract(off)
  if (xf != xf) return xf;
  if (xf > 0x1.62e42ep6f) return __builtin_inff();
  if (xf < -0x1.9fe368p6f) return 0.0f;
  constexpr double N = 32.0;
  constexpr double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
  constexpr double kShift = 0x1.8p+52, kInvLn2N = 0x1.71547652b82fep+0 * N;
  const double xd = (double)xf;
  double kd = __builtin_fma(kInvLn2N, xd, kShift);
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= kShift;
  const double r = __builtin_fma(kInvLn2N, xd, -kd);
  const double s = __longlong_as_double((long long)(tab[ki & 31ull] + (ki << 47)));
  const double z = __builtin_fma(C0, r, C1);
  const double r2 = r * r;
  double y = __builtin_fma(C2, r, 1.0);
  y = __builtin_fma(z, r2, y);
  y = y * s;
  return (float)y;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Wave64 sum through DPP (no LDS crossbar traffic): row_shr 1/2/4/8 inside each 16-lane row, then row_bcast 15 / 31
// across rows.  The total lands in lane 63 only; the order of additions is fixed, so the result is reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_f64<0x111, 0xf>(v);  // row_shr:1
  v += dpp_f64<0x112, 0xf>(v);  // row_shr:2
  v += dpp_f64<0x114, 0xf>(v);  // row_shr:4
  v += dpp_f64<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row sum
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  return v;
}

}  // namespace dgs
