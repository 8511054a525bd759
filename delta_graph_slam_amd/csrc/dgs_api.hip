// C ABI of libdgs_reg.so (include/dgs_reg.h).  Thin: argument checks, device/stream plumbing, uploads, and
// dispatch into the kernels' host drivers.  No exception ever crosses this boundary.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <new>

#include "handle.h"

using namespace dgs;

namespace dgs {

int ensure_pinned(dgs_handle* h, size_t bytes) {
  if (bytes <= h->pinned_bytes) return DGS_OK;
  if (h->pinned) (void)hipHostFree(h->pinned);
  h->pinned = nullptr;
  h->pinned_bytes = 0;
  const size_t want = std::max<size_t>(bytes * 2, 1 << 16);
  hipError_t e = hipHostMalloc(&h->pinned, want, hipHostMallocDefault);
  if (e != hipSuccess) {
    h->err = std::string("hipHostMalloc: ") + hipGetErrorString(e);
    return DGS_ERR_HIP;
  }
  h->pinned_bytes = want;
  return DGS_OK;
}

int ensure_poll_events(dgs_handle* h) {
  for (int k = 0; k < 2; k++)
    if (!h->ev_poll[k]) DGS_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_poll[k], hipEventDisableTiming));
  return DGS_OK;
}

// The side stream is created together with the handle's own stream (dgs_create), not at first use: HIP deals streams to a small number
// of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order, and two streams on one hardware queue run one after the
// other.  Created back to back the two get neighbouring queues; created lazily -- after RCCL had made its streams for a dgs_group --
// the side stream landed on the main stream's queue and the index build it should hide (0.9-1.2 ms) ran in line with the iterations
// (measured: 3.48 ms per 32-candidate step against 2.36 ms; scripts/dbg_group_time.py).
static int side_create(dgs_handle* h) {
  if (h->side_stream) return DGS_OK;
  // lowest priority: the side stream's small builds fill what the main stream's launches leave free, never the other way round
  int prio_low = 0, prio_high = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
  DGS_HIP_TRY(h, hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, prio_low));
  DGS_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  DGS_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  // the computeHessian stream of the upstream NDT order: same priority as the main stream, its own hardware queue (created third)
  if (h->prm.method == DGS_METHOD_NDT) {
    DGS_HIP_TRY(h, hipStreamCreateWithFlags(&h->hd_stream, hipStreamNonBlocking));
    for (int k = 0; k < dgs_handle::kHdEvents; k++) {
      DGS_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_hd_a[k], hipEventDisableTiming));
      DGS_HIP_TRY(h, hipEventCreateWithFlags(&h->ev_hd_b[k], hipEventDisableTiming));
    }
  }
  return DGS_OK;
}

int side_fork(dgs_handle* h) {
  if (side_create(h) != DGS_OK) return DGS_ERR_HIP;
  DGS_HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
  DGS_HIP_TRY(h, hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
  return DGS_OK;
}

// the deferred build of the target's NN index (dgs_align_batch): enqueue it on the side stream now
int side_build_now(dgs_handle* h) {
  if (!h->side_build_deferred) return DGS_OK;
  h->side_build_deferred = false;
  int rs = ensure_target_index(h, h->side_stream);
  if (rs == DGS_OK && hipEventRecord(h->ev_join, h->side_stream) == hipSuccess) h->side_pending = true;
  if (rs != DGS_OK) { (void)hipStreamSynchronize(h->side_stream); h->tgt->bvh.valid = false; h->tgt_grid.valid = false; }
  return rs;
}

int side_join(dgs_handle* h) {
  if (h->side_build_deferred && side_build_now(h) != DGS_OK) return DGS_ERR_HIP;   // nobody took it up (an early error return)
  if (!h->side_pending) return DGS_OK;
  h->side_pending = false;
  DGS_HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
  return DGS_OK;
}

int prof_begin(dgs_handle* h, int kernel_id, hipStream_t st) {
  Profiler& p = h->prof;
  if (!p.enabled) return -1;
  if (p.next_free == p.pool.size()) {
    EventPair ep;
    if (hipEventCreate(&ep.start) != hipSuccess || hipEventCreate(&ep.stop) != hipSuccess) return -1;
    p.pool.push_back(ep);
  }
  const int slot = (int)p.next_free++;
  (void)hipEventRecord(p.pool[slot].start, st ? st : h->stream);
  (void)kernel_id;
  return slot;
}

void prof_end(dgs_handle* h, int kernel_id, int slot, hipStream_t st) {
  if (slot < 0) return;
  Profiler& p = h->prof;
  (void)hipEventRecord(p.pool[slot].stop, st ? st : h->stream);
  p.pending[kernel_id].push_back(slot);
}

static void prof_collect(dgs_handle* h) {
  Profiler& p = h->prof;
  (void)hipStreamSynchronize(h->stream);
  if (h->hd_stream) (void)hipStreamSynchronize(h->hd_stream);
  for (int k = 0; k < DGS_K_COUNT; k++) {
    for (int slot : p.pending[k]) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.pool[slot].start, p.pool[slot].stop) == hipSuccess) {
        p.total_ms[k] += ms;
        p.launches[k] += 1;
      }
    }
    p.pending[k].clear();
  }
  p.next_free = 0;
}

// out = T * in  (K8 transform_cloud; pcl::transformPointCloud semantics, pad lane set to 1)
__global__ __launch_bounds__(kBlock) void transform_kernel(const float4* __restrict__ in, float4* __restrict__ out, int64_t n, float t0, float t1,
                                                           float t2, float t3, float t4, float t5, float t6, float t7, float t8, float t9,
                                                           float t10, float t11) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 x = in[i];
  float4 y;
  y.x = t0 * x.x + t1 * x.y + t2 * x.z + t3;
  y.y = t4 * x.x + t5 * x.y + t6 * x.z + t7;
  y.z = t8 * x.x + t9 * x.y + t10 * x.z + t11;
  y.w = 1.f;
  out[i] = y;
}

int transform_cloud(dgs_handle* h, const float4* in, float4* out, int64_t n, const float* T) {
  if (n == 0) return DGS_OK;
  int slot = prof_begin(h, DGS_K_TRANSFORM);
  hipLaunchKernelGGL(transform_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, in, out, n, T[0], T[4], T[8], T[12],
                     T[1], T[5], T[9], T[13], T[2], T[6], T[10], T[14]);
  prof_end(h, DGS_K_TRANSFORM, slot);
  return DGS_OK;
}

}  // namespace dgs

// LoopDetector::find_candidates: one workgroup walks the keyframes 1,024 at a time and writes the indices that pass both tests in
// keyframe order (wave ballots + a running base: an ordered compaction without atomics)
__global__ __launch_bounds__(1024) void find_candidates_kernel(const double* __restrict__ accum, const double* __restrict__ xy, const long long n,
                                                               const double new_accum, const double nx, const double ny, const double accum_thresh,
                                                               const double dist_thresh, int* __restrict__ out, const long long capacity,
                                                               long long* __restrict__ n_out) {
#pragma clang fp contract(off)
  __shared__ int wave_count[16];
  __shared__ long long base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (long long i0 = 0; i0 < n; i0 += 1024) {
    const long long i = i0 + threadIdx.x;
    bool keep = false;
    if (i < n) {
      const double dx = xy[2 * i] - nx, dy = xy[2 * i + 1] - ny;
      keep = !(new_accum - accum[i] < accum_thresh) && !(sqrt(dx * dx + dy * dy) > dist_thresh);   // the reference's two `continue`s, negated
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_count[wave] = __popcll(m);
    __syncthreads();
    long long pos = base;
    for (int w = 0; w < wave; w++) pos += wave_count[w];
    pos += __popcll(m & ((1ull << lane) - 1ull));
    if (keep && pos < capacity) out[pos] = (int)i;
    __syncthreads();
    if (threadIdx.x == 0) {
      long long t = 0;
      for (int w = 0; w < 16; w++) t += wave_count[w];
      base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = base;
}

static int set_device(dgs_handle* h) {
  DGS_HIP_TRY(h, hipSetDevice(h->device));
  return DGS_OK;
}

static int upload_cloud(dgs_handle* h, DevBuf<float4>& buf, const float* xyz16, int64_t n, int on_device) {
  if (n == 0) return DGS_OK;
  DGS_HIP_TRY(h, buf.reserve((size_t)n));
  DGS_HIP_TRY(h, hipMemcpyAsync(buf.ptr, xyz16, (size_t)n * sizeof(float4), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  // neither a host nor a device pointer is retained past return: the caller may free or overwrite its buffer at once (a device
  // buffer handed in by a framework's caching allocator is recycled as soon as its tensor dies)
  DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));
  return DGS_OK;
}

// a handle borrows at most two dgs_cloud objects (target, source); both sides keep a back reference so that either may go first
static void unbind(dgs_handle* h, dgs_cloud*& slot) {
  if (!slot) return;
  dgs_cloud* c = slot;
  slot = nullptr;
  if (h->tgt_cloud != c && h->src_cloud != c) c->users.erase(std::remove(c->users.begin(), c->users.end(), h), c->users.end());
}
static void bind(dgs_handle* h, dgs_cloud*& slot, dgs_cloud* c) {
  if (slot == c) return;
  unbind(h, slot);
  if (std::find(c->users.begin(), c->users.end(), h) == c->users.end()) c->users.push_back(h);
  slot = c;
}

namespace dgs {
int cloud_clone_to(dgs_handle* h, const dgs_cloud* src, dgs_cloud** out) {
  *out = nullptr;
  DGS_HIP_TRY(h, hipSetDevice(h->device));
  dgs_cloud* c = new (std::nothrow) dgs_cloud();
  if (!c) return DGS_ERR_HIP;
  c->device = h->device;
  c->st.n = src->st.n;
  hipError_t e = hipSuccess;
  if (src->st.n > 0) {
    e = c->st.pts.reserve((size_t)src->st.n);
    const size_t bytes = (size_t)src->st.n * sizeof(float4);
    if (e == hipSuccess)
      e = (src->device == h->device) ? hipMemcpyAsync(c->st.pts.ptr, src->st.pts.ptr, bytes, hipMemcpyDeviceToDevice, h->stream)
                                     : hipMemcpyPeerAsync(c->st.pts.ptr, h->device, src->st.pts.ptr, src->device, bytes, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  }
  if (e != hipSuccess) {
    h->err = std::string("cloud_clone_to: ") + hipGetErrorString(e);
    c->st.release();
    delete c;
    return DGS_ERR_HIP;
  }
  *out = c;
  return DGS_OK;
}
}  // namespace dgs

static thread_local std::string g_create_error;  // dgs_last_error(NULL): why the last dgs_create on this thread failed

extern "C" {

int dgs_abi_version(void) { return DGS_ABI_VERSION; }

int dgs_params_init(dgs_params* p, int32_t method) {
  if (!p) return DGS_ERR_INVALID_ARGUMENT;
  std::memset(p, 0, sizeof(*p));
  p->struct_size = sizeof(dgs_params);
  p->method = method;
  p->device = -1;
  p->num_threads = 0;
  p->transformation_epsilon = 0.01;
  p->maximum_iterations = 64;
  p->ndt_search_method = DGS_NDT_DIRECT7;
  p->ndt_resolution = 0.5;
  p->ndt_step_size = 0.1;
  p->ndt_outlier_ratio = 0.55;
  p->ndt_min_covar_eigvalue_mult = 0.01;
  p->ndt_min_points_per_voxel = 6;
  p->ndt_line_search = DGS_NDT_LS_MORE_THUENTE;
  p->ndt_mt_max_step_iterations = 10;
  p->ndt_fix_hessian_d1 = 0;
  p->ndt_strict_order = DGS_NDT_ORDER_UPSTREAM;   // the order that reproduces a CPU run of upstream; FAST is the caller's choice
  p->ndt_newton_solver = 1;
  p->ndt_hessian_recompute_double = 1;
  p->ndt_guess_rotation_polar = 1;
  p->ndt_exp_glibc = 1;
  p->ndt_cov_eigensolver = 1;
  p->gicp_cov_jacobi_svd = 0;   // see dgs_reg.h: available, not the default
  p->gicp_max_correspondence_distance = 2.5;
  p->gicp_rotation_epsilon = 2e-3;
  p->gicp_lm_init_lambda_factor = 1e-9;
  p->gicp_correspondence_randomness = 20;
  p->gicp_regularization = DGS_GICP_REG_PLANE;
  p->gicp_optimizer = DGS_GICP_OPT_LEVENBERG_MARQUARDT;
  p->gicp_lm_max_iterations = 10;
  p->vgicp_search_method = DGS_VGICP_DIRECT1;
  p->vgicp_resolution = 1.0;
  if (method != DGS_METHOD_NDT && method != DGS_METHOD_GICP && method != DGS_METHOD_VGICP) return DGS_ERR_INVALID_ARGUMENT;
  return DGS_OK;
}

int dgs_create(const dgs_params* params, dgs_handle** out) {
  if (!params || !out || params->struct_size != sizeof(dgs_params)) return DGS_ERR_INVALID_ARGUMENT;
  if (params->method != DGS_METHOD_NDT && params->method != DGS_METHOD_GICP && params->method != DGS_METHOD_VGICP) return DGS_ERR_INVALID_ARGUMENT;
  if (params->method == DGS_METHOD_VGICP && (!(params->vgicp_resolution > 0) || params->vgicp_search_method < 0 || params->vgicp_search_method > DGS_VGICP_DIRECT27))
    return DGS_ERR_INVALID_ARGUMENT;
  if (!(params->ndt_resolution > 0) || params->maximum_iterations < 0 || params->gicp_correspondence_randomness < 1) return DGS_ERR_INVALID_ARGUMENT;
  if (params->ndt_strict_order < DGS_NDT_ORDER_FAST || params->ndt_strict_order > DGS_NDT_ORDER_UPSTREAM_SEQUENTIAL) return DGS_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  dgs_handle* h = new (std::nothrow) dgs_handle();
  if (!h) return DGS_ERR_HIP;
  h->prm = *params;
  int dev = params->device;
  hipError_t e = hipSuccess;
  if (dev < 0) e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipSetDevice(dev);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {  // no usable HIP device: there is no CPU fallback
    g_create_error = std::string("dgs_create(device=") + std::to_string(params->device) + "): " + hipGetErrorString(e);
    delete h;
    return DGS_ERR_HIP;
  }
  h->device = dev;
  h->own_stream = true;
  (void)side_create(h);   // now, next to the main stream (see side_create); a failure here is retried at first use
  if (kExperiments)   // the product library carries neither the grid index nor the packed-FP32 kernel (measured losers: `make experiments`)
    if (const char* e = std::getenv("DGS_NN_GRID")) { h->grid_mode = std::atoi(e); h->grid_levels = std::max(1, std::atoi(e)); }
  if (const char* e = std::getenv("DGS_NDT_FUSED")) h->ndt_fused = std::atoi(e) != 0;
  if (const char* e = std::getenv("DGS_NDT_STRICT_KERNEL")) h->strict_kernel = std::atoi(e) == 2 ? 2 : 3;
  if (const char* e = std::getenv("DGS_NDT_HD_OVERLAP")) h->hd_overlap = std::atoi(e) != 0;
  if (const char* e = std::getenv("DGS_NDT_SOLVE_MIN_ACTIVE")) h->solve_min_active = std::max(0, std::atoi(e));
  if (const char* e = std::getenv("DGS_NDT_SPECULATE")) h->ndt_speculate = std::atoi(e) != 0;
  if (const char* e = std::getenv("DGS_NDT_FIXED_SLICES")) h->ndt_fixed_slices = std::atoi(e) != 0;
  if (const char* e = std::getenv("DGS_NDT_QUEUE")) h->ndt_queue_mode = std::atoi(e);
  if (const char* e = std::getenv("DGS_NDT_SCHEDULE")) h->ndt_schedule = std::atoi(e) != 0;
  if (const char* e = std::getenv("DGS_NDT_QUEUE_MIN_PAIRS")) h->ndt_queue_min_pairs = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("DGS_EARLY_FITNESS")) h->early_fitness_enabled = std::atoi(e) != 0;
  if (const char* e = std::getenv("DGS_EARLY_FITNESS_LDS_KB")) h->early_fit.lds_kb = std::max(0, std::atoi(e));
  if (const char* e = std::getenv("DGS_EARLY_FITNESS_MAX_ACTIVE")) h->early_fit.max_active = std::max(0, std::atoi(e));
  if (const char* e = std::getenv("DGS_EARLY_FITNESS_MIN_PAIRS")) h->early_fit.min_pairs = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("DGS_NN_KD")) h->nn_kd = std::atoi(e) != 0;
  if (const char* e = std::getenv("DGS_NN_KD_ALL")) h->nn_kd_all = std::atoi(e) != 0;
  if (const char* e = std::getenv("DGS_NN_KD_MIN_QUERIES")) h->nn_kd_min_queries = std::atoll(e);
  if (const char* e = std::getenv("DGS_KNN_PARTS")) { const int v = std::atoi(e); h->knn_parts = (v == 1 || v == 2 || v == 4 || v == 8) ? v : 0; }
  if (const char* e = std::getenv("DGS_KNN_LEAF")) h->knn_leaf = std::atoi(e) != 0;
  if (const char* e = std::getenv("DGS_KNN_MIN_WAVES")) h->knn_min_waves = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("DGS_KNN_ROUNDS")) h->knn_rounds = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("DGS_GICP_FUSED")) h->gicp_fused = std::atoi(e) != 0;
  if (kExperiments)
    if (const char* e = std::getenv("DGS_NDT_PACK2")) h->ndt_pack2 = std::atoi(e) != 0;
  if (const char* e = std::getenv("DGS_NN_GRID_FACTOR")) h->grid_spacing_factor = std::max(0.5f, (float)std::atof(e));
  std::memset(h->final_T, 0, sizeof(h->final_T));
  h->final_T[0] = h->final_T[5] = h->final_T[10] = h->final_T[15] = 1.f;
  *out = h;
  return DGS_OK;
}

void dgs_destroy(dgs_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  unbind(h, h->tgt_cloud);
  unbind(h, h->src_cloud);
  h->own_target.release(); h->own_source.release();
  for (int k = 0; k < 2; k++)
    if (h->ev_poll[k]) (void)hipEventDestroy(h->ev_poll[k]);
  if (h->hd_stream) {
    (void)hipStreamSynchronize(h->hd_stream);
    (void)hipStreamDestroy(h->hd_stream);
    for (int k = 0; k < dgs_handle::kHdEvents; k++) {
      if (h->ev_hd_a[k]) (void)hipEventDestroy(h->ev_hd_a[k]);
      if (h->ev_hd_b[k]) (void)hipEventDestroy(h->ev_hd_b[k]);
    }
  }
  if (h->side_stream) {
    (void)hipStreamSynchronize(h->side_stream);
    (void)hipStreamDestroy(h->side_stream);
    (void)hipEventDestroy(h->ev_fork);
    (void)hipEventDestroy(h->ev_join);
  }
  h->batch_slab.release();
  for (auto& c : h->batch_clouds) c.release();
  h->gitems.release(); h->vvox.release(); h->vcell2vox.release();
  h->cell2vox.release(); h->vox.release(); h->vox_centroid.release(); h->vox_dbg.release(); h->vox_strict.release(); h->vox_count.release(); h->vox_valid.release();
  h->key_in.release(); h->key_out.release(); h->val_in.release(); h->val_out.release(); h->run_keys.release();
  h->run_counts.release(); h->run_offsets.release(); h->dev_scalars.release(); h->vg_run_keys.release(); h->vg_scalars.release(); h->minmax_partial.release(); h->cub_temp.release();
  h->pairs.release(); h->inits.release(); h->partials.release(); h->done_counter.release(); h->ndt_queue.release(); h->ndt_ring.release(); h->pair_blocks.release(); h->src_ptrs.release(); h->src_sizes.release();
  h->nn_partials.release(); h->scratch_cloud.release(); h->strict_rows.release(); h->strict_totals.release(); h->tgt_grid.release(); h->aux_grid.release();
  h->fc_in.release(); h->fc_out.release(); h->fc_cnt.release();
  h->aux_cloud1.release(); h->aux_cloud2.release(); h->aux_out.release(); h->aux_bvh.sorted.release(); h->aux_bvh.node_lo.release(); h->aux_bvh.node_hi.release();
  h->aux_bvh.keys.release(); h->aux_bvh.keys_alt.release(); h->aux_bvh.vals.release(); h->aux_bvh.vals_alt.release();
  h->corr.release(); h->corr_sq.release(); h->mahal.release(); h->gpairs.release();
  for (auto& ep : h->prof.pool) { (void)hipEventDestroy(ep.start); (void)hipEventDestroy(ep.stop); }
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->done_flags) (void)hipHostFree(h->done_flags);
  if (h->fit_host) (void)hipHostFree(h->fit_host);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char* dgs_last_error(const dgs_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dgs_set_stream(dgs_handle* h, void* hip_stream) {
  if (!h) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  if (hip_stream) {
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    h->own_stream = false;
  } else {
    DGS_HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  return DGS_OK;
}

int dgs_synchronize(dgs_handle* h) {
  if (!h) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));
  return DGS_OK;
}

int dgs_set_input_target(dgs_handle* h, const float* xyz16, int64_t n, int32_t on_device) {
  if (!h || n < 0 || (n > 0 && !xyz16) || n > INT32_MAX) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  h->have_target = false;
  h->vmap_valid = false;
  h->tgt_grid.valid = false;
  unbind(h, h->tgt_cloud);
  h->tgt = &h->own_target;
  h->tgt->invalidate();
  h->tgt->n = n;
  h->nt = n;
  int rc = upload_cloud(h, h->own_target.pts, xyz16, n, on_device);
  if (rc) return rc;
  if (h->prm.method == DGS_METHOD_NDT) {
    rc = ndt_build_target(h);
    if (rc) return rc;
  }
  h->have_target = true;
  return DGS_OK;
}

int dgs_set_input_source(dgs_handle* h, const float* xyz16, int64_t n, int32_t on_device) {
  if (!h || n < 0 || (n > 0 && !xyz16) || n > INT32_MAX) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  h->have_source = false;
  unbind(h, h->src_cloud);
  h->src = &h->own_source;
  h->src->invalidate();
  h->src->n = n;
  h->ns = n;
  int rc = upload_cloud(h, h->own_source.pts, xyz16, n, on_device);
  if (rc) return rc;
  h->have_source = true;
  return DGS_OK;
}

// ---- device-resident cloud objects (keyframe cache) ---------------------------------------------------
int dgs_cloud_create(dgs_handle* h, const float* xyz16, int64_t n, int32_t on_device, dgs_cloud** out) {
  if (!h || !out || n < 0 || (n > 0 && !xyz16) || n > INT32_MAX) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  *out = nullptr;
  if (set_device(h)) return DGS_ERR_HIP;
  dgs_cloud* c = new (std::nothrow) dgs_cloud();
  if (!c) return DGS_ERR_HIP;
  c->device = h->device;
  c->st.n = n;
  int rc = upload_cloud(h, c->st.pts, xyz16, n, on_device);
  if (rc != DGS_OK) {
    c->st.release();
    delete c;
    return rc;
  }
  *out = c;
  return DGS_OK;
}

void dgs_cloud_destroy(dgs_cloud* c) {
  if (!c) return;
  for (dgs_handle* h : c->users) {  // handles still pointing here fall back to "no input set"
    if (h->tgt_cloud == c) { h->tgt_cloud = nullptr; h->tgt = &h->own_target; h->nt = 0; h->have_target = false; h->tgt_grid.valid = false; }
    if (h->src_cloud == c) { h->src_cloud = nullptr; h->src = &h->own_source; h->ns = 0; h->have_source = false; }
  }
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  c->st.release();
  delete c;
}

int64_t dgs_cloud_size(const dgs_cloud* c) { return c ? c->st.n : 0; }

int dgs_set_input_target_cloud(dgs_handle* h, dgs_cloud* c) {
  if (!h || !c || c->device != h->device) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  h->have_target = false;
  h->vmap_valid = false;
  h->tgt_grid.valid = false;
  bind(h, h->tgt_cloud, c);
  h->tgt = &c->st;
  h->nt = c->st.n;
  if (h->prm.method == DGS_METHOD_NDT) {
    int rc = ndt_build_target(h);
    if (rc) return rc;
  }
  h->have_target = true;
  return DGS_OK;
}

int dgs_set_input_source_cloud(dgs_handle* h, dgs_cloud* c) {
  if (!h || !c || c->device != h->device) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  bind(h, h->src_cloud, c);
  h->src = &c->st;
  h->ns = c->st.n;
  h->have_source = true;
  return DGS_OK;
}

static void fail_result(dgs_result* r, const float* guess, int status) {
  const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::memcpy(r->final_transformation, guess ? guess : ident, sizeof(float) * 16);
  r->converged = 0;
  r->iterations = 0;
  r->evaluations = 0;
  r->status = status;
  r->score = 0.0;
  r->fitness = NAN;
}

int dgs_align(dgs_handle* h, const float* guess16, dgs_result* out, float* aligned, int32_t aligned_on_device) {
  if (!h || !out) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  fail_result(out, guess16, DGS_OK);
  if (set_device(h)) { out->status = DGS_ERR_HIP; return DGS_ERR_HIP; }
  if (!h->have_target || h->nt == 0) { out->status = DGS_ERR_NO_TARGET; h->err = "no input target dataset was given"; return DGS_ERR_NO_TARGET; }
  if (!h->have_source || h->ns == 0) { out->status = DGS_ERR_NO_SOURCE; h->err = "no input source dataset was given"; return DGS_ERR_NO_SOURCE; }
  int rc;
  if (h->prm.method == DGS_METHOD_NDT) {
    const float4* src = h->src->pts.ptr;
    const int n = (int)h->ns;
    rc = ndt_align_pairs(h, 1, &src, &n, guess16, out);
  } else {
    rc = gicp_align(h, guess16, out);
  }
  if (rc != DGS_OK) {
    fail_result(out, guess16, rc);
    return rc;
  }
  std::memcpy(h->final_T, out->final_transformation, sizeof(h->final_T));
  h->have_result = true;
  if (aligned) {
    float4* dst = reinterpret_cast<float4*>(aligned);
    if (!aligned_on_device) {
      DGS_HIP_TRY(h, h->scratch_cloud.reserve((size_t)h->ns));
      dst = h->scratch_cloud.ptr;
    }
    transform_cloud(h, h->src->pts.ptr, dst, h->ns, h->final_T);
    if (!aligned_on_device) {
      DGS_HIP_TRY(h, hipMemcpyAsync(aligned, dst, (size_t)h->ns * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
      DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
  }
  return DGS_OK;
}

int dgs_get_fitness_score(dgs_handle* h, double max_range, double* score) {
  if (!h || !score) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (!h->have_target || h->nt == 0) return DGS_ERR_NO_TARGET;
  if (!h->have_source || h->ns == 0) return DGS_ERR_NO_SOURCE;
  double sum = 0;
  int64_t cnt = 0, inl = 0;
  int rc = nn_fitness(h, h->src->pts.ptr, h->ns, h->final_T, max_range, 0.0, &sum, &cnt, &inl);
  if (rc) return rc;
  *score = cnt > 0 ? sum / (double)cnt : DBL_MAX;
  return DGS_OK;
}

int dgs_get_inlier_fraction(dgs_handle* h, double max_sq_dist, double* fraction) {
  if (!h || !fraction) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (!h->have_target || h->nt == 0) return DGS_ERR_NO_TARGET;
  if (!h->have_source || h->ns == 0) return DGS_ERR_NO_SOURCE;
  double sum = 0;
  int64_t cnt = 0, inl = 0;
  int rc = nn_fitness(h, h->src->pts.ptr, h->ns, h->final_T, DBL_MAX, max_sq_dist, &sum, &cnt, &inl);
  if (rc) return rc;
  *fraction = (double)inl / (double)h->ns;
  return DGS_OK;
}

int dgs_nearest_search_target(dgs_handle* h, const float* queries, int64_t m, int32_t on_device, int32_t* indices, float* sq_dists) {
  if (!h || m < 0 || (m > 0 && (!queries || !indices || !sq_dists))) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (!h->have_target || h->nt == 0) return DGS_ERR_NO_TARGET;
  if (m == 0) return DGS_OK;
  const float4* dq = reinterpret_cast<const float4*>(queries);
  int32_t* didx = indices;
  float* dsq = sq_dists;
  DevBuf<float4> q;
  DevBuf<int32_t> bi;
  DevBuf<float> bd;
  int rc = DGS_OK;
  if (!on_device) {
    if (q.reserve(m) != hipSuccess || bi.reserve(m) != hipSuccess || bd.reserve(m) != hipSuccess) { h->err = "hipMalloc failed"; rc = DGS_ERR_HIP; }
    if (!rc && hipMemcpyAsync(q.ptr, queries, (size_t)m * sizeof(float4), hipMemcpyHostToDevice, h->stream) != hipSuccess) { h->err = "hipMemcpyAsync failed"; rc = DGS_ERR_HIP; }
    dq = q.ptr; didx = bi.ptr; dsq = bd.ptr;
  }
  if (!rc) rc = nn_search(h, dq, m, didx, dsq);
  if (!rc && !on_device) {
    if (hipMemcpyAsync(indices, didx, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipMemcpyAsync(sq_dists, dsq, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess) { h->err = "hipMemcpyAsync failed"; rc = DGS_ERR_HIP; }
  }
  (void)hipStreamSynchronize(h->stream);
  q.release(); bi.release(); bd.release();
  return rc;
}

int dgs_nn_fitness_distances(dgs_handle* h, const float* queries, int64_t m, int32_t on_device, float* sq_dists) {
  if (!h || m < 0 || (m > 0 && (!queries || !sq_dists))) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (!h->have_target || h->nt == 0) return DGS_ERR_NO_TARGET;
  if (h->grid_mode == 0) return DGS_ERR_UNSUPPORTED;
  if (m == 0) return DGS_OK;
  h->use_grid = true;
  if (side_join(h) != DGS_OK) return DGS_ERR_HIP;
  int rc = ensure_target_index(h);
  if (rc) return rc;
  const float4* dq = reinterpret_cast<const float4*>(queries);
  float* dsq = sq_dists;
  DevBuf<float4> q;
  DevBuf<float> bd;
  if (!on_device) {
    if (q.reserve(m) != hipSuccess || bd.reserve(m) != hipSuccess) { h->err = "hipMalloc failed"; rc = DGS_ERR_HIP; }
    if (!rc && hipMemcpyAsync(q.ptr, queries, (size_t)m * sizeof(float4), hipMemcpyHostToDevice, h->stream) != hipSuccess) { h->err = "hipMemcpyAsync failed"; rc = DGS_ERR_HIP; }
    dq = q.ptr; dsq = bd.ptr;
  }
  if (!rc) rc = nn_grid_search(h, h->tgt_grid, h->tgt->bvh, dq, m, dsq);
  if (!rc && !on_device && hipMemcpyAsync(sq_dists, dsq, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, h->stream) != hipSuccess) { h->err = "hipMemcpyAsync failed"; rc = DGS_ERR_HIP; }
  (void)hipStreamSynchronize(h->stream);
  q.release(); bd.release();
  return rc;
}

// FAST_GICP over a batch of sources (loop_detector.hpp:137-156): batched align, then one fitness launch for all candidates
static int gicp_batch(dgs_handle* h, int n, CloudState* const* cs, const float* guesses16, int compute_fitness, double fitness_max_range,
                      dgs_result* results) {
  struct KdScope { dgs_handle* h; ~KdScope() { h->batch_kd = false; } } kd_scope{h};
  // an index built for the target of this batch is k-d ordered when the batch is large enough to repay the slower build, which is on the
  // critical path here (measured per tick over resident 65,536-point keyframes, k-d / Hilbert: 4 candidates 1.38 / 1.17 ms, 8: 1.77 / 1.72,
  // 12: 2.08 / 2.16, 32: 3.30 / 4.23)
  static const int kd_min = std::getenv("DGS_GICP_KD_MIN_CANDIDATES") ? std::atoi(std::getenv("DGS_GICP_KD_MIN_CANDIDATES")) : 10;
  h->batch_kd = h->nn_kd && n >= kd_min;
  int rc = gicp_align_batch(h, n, cs, guesses16, results);
  if (rc == DGS_OK && compute_fitness) {
    DGS_HIP_TRY(h, h->src_ptrs.reserve(n));
    DGS_HIP_TRY(h, h->src_sizes.reserve(n));
    std::vector<const float4*> ptrs(n);
    std::vector<int> sz(n);
    int max_n = 0;
    for (int i = 0; i < n; i++) { ptrs[i] = cs[i]->pts.ptr; sz[i] = (int)cs[i]->n; max_n = std::max(max_n, sz[i]); }
    DGS_HIP_TRY(h, hipMemcpyAsync(h->src_ptrs.ptr, ptrs.data(), sizeof(void*) * n, hipMemcpyHostToDevice, h->stream));
    DGS_HIP_TRY(h, hipMemcpyAsync(h->src_sizes.ptr, sz.data(), sizeof(int) * n, hipMemcpyHostToDevice, h->stream));
    DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));  // ptrs / sz are pageable host memory going out of scope
    std::vector<double> sums(n);
    std::vector<int64_t> cnts(n), inl(n);
    size_t stride = 0;
    const float* dT = gicp_final_transforms(h, &stride);
    rc = nn_fitness_batch(h, n, h->src_ptrs.ptr, h->src_sizes.ptr, max_n, dT, stride, fitness_max_range, 0.0, sums.data(), cnts.data(), inl.data());
    if (rc == DGS_OK)
      for (int i = 0; i < n; i++) results[i].fitness = cnts[i] > 0 ? sums[i] / (double)cnts[i] : DBL_MAX;
  }
  if (rc == DGS_OK) {
    for (int i = 0; i < n; i++)
      if (cs[i]->n <= 0) fail_result(&results[i], guesses16 ? guesses16 + 16 * i : nullptr, DGS_ERR_NO_SOURCE);
  }
  return rc;
}

int dgs_align_batch(dgs_handle* h, int32_t n, const float* const* sources, const int64_t* sizes, int32_t on_device, const float* guesses16,
                    int32_t compute_fitness, double fitness_max_range, dgs_result* results) {
  if (!h || n < 0 || (n > 0 && (!sources || !sizes || !results))) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (n == 0) return DGS_OK;
  for (int i = 0; i < n; i++) fail_result(&results[i], guesses16 ? guesses16 + 16 * i : nullptr, DGS_OK);
  if (!h->have_target || h->nt == 0) {
    for (int i = 0; i < n; i++) results[i].status = DGS_ERR_NO_TARGET;
    h->err = "no input target dataset was given";
    return DGS_ERR_NO_TARGET;
  }
  if (h->prm.method != DGS_METHOD_NDT) {
    // FAST_GICP: every candidate needs its own index + covariances; they live in per-slot CloudStates the handle re-uses
    if (h->batch_clouds.size() < (size_t)n) h->batch_clouds.resize(n);
    std::vector<CloudState*> cs(n);
    for (int i = 0; i < n; i++) {
      if (sizes[i] < 0 || sizes[i] > INT32_MAX) return DGS_ERR_INVALID_ARGUMENT;
      CloudState& c = h->batch_clouds[i];
      c.invalidate();
      c.n = sources[i] ? sizes[i] : 0;
      int rc = upload_cloud(h, c.pts, sources[i], c.n, on_device);
      if (rc) return rc;
      cs[i] = &c;
    }
    return gicp_batch(h, n, cs.data(), guesses16, compute_fitness, fitness_max_range, results);
  }
  // stage sources on the device when they come from the host (one contiguous slab, kept by the handle between calls)
  std::vector<const float4*> ptrs(n);
  std::vector<int> sz(n);
  DevBuf<float4>& slab = h->batch_slab;
  int64_t total = 0;
  for (int i = 0; i < n; i++) {
    if (sizes[i] < 0 || sizes[i] > INT32_MAX || (sizes[i] > 0 && !sources[i])) return DGS_ERR_INVALID_ARGUMENT;
    sz[i] = (int)sizes[i];
    total += sizes[i];
  }
  if (!on_device) {
    DGS_HIP_TRY(h, slab.reserve((size_t)std::max<int64_t>(total, 1)));
    int64_t off = 0;
    for (int i = 0; i < n; i++) {
      ptrs[i] = slab.ptr + off;
      if (sz[i]) DGS_HIP_TRY(h, hipMemcpyAsync(slab.ptr + off, sources[i], (size_t)sz[i] * sizeof(float4), hipMemcpyHostToDevice, h->stream));
      off += sz[i];
    }
  } else {
    for (int i = 0; i < n; i++) ptrs[i] = reinterpret_cast<const float4*>(sources[i]);
  }
  // the fitness pass needs the target's NN index only after the last iteration: build it on the side stream meanwhile
  // (a dozen tiny launches, 0.15 ms on the critical path otherwise); kernels timed one by one stay on one stream
  h->use_grid = grid_wanted(h, total);
  // an index built for this batch is k-d ordered: the build (a sort per level) hides behind the iterations, the fitness pass over
  // n x 65,536 queries is twice as fast as over the Hilbert order (nn_bvh.hip)
  struct KdScope { dgs_handle* h; ~KdScope() { h->batch_kd = false; } } kd_scope{h};
  // ... from ~6 candidates of 65,536 points on: below that the build (~0.9 ms of side-stream time) outlasts the iterations it should hide
  // behind (measured per tick, k-d / Hilbert: 1 candidate 0.74 / 0.59 ms, 4: 0.80 / 0.77, 8: 1.16 / 1.24, 32: 2.26 / 2.59)
  h->batch_kd = h->nn_kd && compute_fitness && total >= h->nn_kd_min_queries;
  // The side stream forks HERE (it depends on the target only), but its launches are enqueued by ndt_align_pairs after the first
  // chunks of iteration launches: enqueueing a dozen launches costs the host ~50 us during which the main stream would sit empty.
  if (compute_fitness && (!h->tgt->bvh.valid || (h->use_grid && !h->tgt_grid.valid)) && !h->prof.enabled) {
    int rs = side_fork(h);
    if (rs != DGS_OK) return rs;
    h->side_build_deferred = true;
  }
  int max_n = 0;
  for (int i = 0; i < n; i++) max_n = std::max(max_n, sz[i]);
  // the walk of a finished candidate starts while the others still iterate (ndt_align_pairs); kernels timed one by one stay in line
  h->early_fit.on = h->early_fitness_enabled && compute_fitness && !h->prof.enabled && !h->use_grid && n <= 65535;
  h->early_fit.enqueued = false;
  h->early_fit.max_range = fitness_max_range;
  h->early_fit.max_n = max_n;
  int rc = ndt_align_pairs(h, n, ptrs.data(), sz.data(), guesses16, results);
  h->early_fit.on = false;
  if (side_join(h) != DGS_OK && rc == DGS_OK) rc = DGS_ERR_HIP;  // whatever happened above, nothing stays pending
  if (rc == DGS_OK && compute_fitness && h->early_fit.enqueued) {
    // walked and totalled inside ndt_align_pairs, whose export synchronised the stream
    std::vector<double> sums(n);
    std::vector<int64_t> cnts(n), inl(n);
    nn_fitness_read(h, n, sums.data(), cnts.data(), inl.data());
    for (int i = 0; i < n; i++) results[i].fitness = cnts[i] > 0 ? sums[i] / (double)cnts[i] : DBL_MAX;
  } else if (rc == DGS_OK && compute_fitness) {
    // getFitnessScore for every candidate in one launch; transforms are read from the optimiser state in HBM
    std::vector<double> sums(n);
    std::vector<int64_t> cnts(n), inl(n);
    const float* dT = reinterpret_cast<const float*>(reinterpret_cast<const char*>(h->pairs.ptr) + offsetof(NdtPair, final_T));
    rc = nn_fitness_batch(h, n, h->src_ptrs.ptr, h->src_sizes.ptr, max_n, dT, sizeof(NdtPair), fitness_max_range, 0.0, sums.data(), cnts.data(),
                          inl.data());
    if (rc == DGS_OK)
      for (int i = 0; i < n; i++) results[i].fitness = cnts[i] > 0 ? sums[i] / (double)cnts[i] : DBL_MAX;
  }
  if (rc == DGS_OK) {
    for (int i = 0; i < n; i++)
      if (sz[i] == 0)  // PCL refuses an empty source (initCompute fails): not converged, transform = guess
        fail_result(&results[i], guesses16 ? guesses16 + 16 * i : nullptr, DGS_ERR_NO_SOURCE);
  }
  (void)hipStreamSynchronize(h->stream);
  return rc;
}

int dgs_align_batch_clouds(dgs_handle* h, int32_t n, dgs_cloud* const* sources, const float* guesses16, int32_t compute_fitness,
                           double fitness_max_range, dgs_result* results) {
  if (!h || n < 0 || (n > 0 && (!sources || !results))) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (n == 0) return DGS_OK;
  for (int i = 0; i < n; i++)
    if (!sources[i] || sources[i]->device != h->device) return DGS_ERR_INVALID_ARGUMENT;
  if (h->prm.method == DGS_METHOD_NDT) {
    // the batched NDT path already takes device pointers: hand it the resident clouds, nothing is copied
    std::vector<const float*> ptrs(n);
    std::vector<int64_t> sizes(n);
    for (int i = 0; i < n; i++) { ptrs[i] = reinterpret_cast<const float*>(sources[i]->st.pts.ptr); sizes[i] = sources[i]->st.n; }
    return dgs_align_batch(h, n, ptrs.data(), sizes.data(), 1, guesses16, compute_fitness, fitness_max_range, results);
  }
  // FAST_GICP: one batched LM loop; each resident cloud keeps its index and covariances across calls
  for (int i = 0; i < n; i++) fail_result(&results[i], guesses16 ? guesses16 + 16 * i : nullptr, DGS_OK);
  if (!h->have_target || h->nt == 0) {
    for (int i = 0; i < n; i++) results[i].status = DGS_ERR_NO_TARGET;
    h->err = "no input target dataset was given";
    return DGS_ERR_NO_TARGET;
  }
  std::vector<CloudState*> cs(n);
  for (int i = 0; i < n; i++) cs[i] = &sources[i]->st;
  return gicp_batch(h, n, cs.data(), guesses16, compute_fitness, fitness_max_range, results);
}

int dgs_profile_enable(dgs_handle* h, int32_t enable) {
  if (!h) return DGS_ERR_INVALID_ARGUMENT;
  if (set_device(h)) return DGS_ERR_HIP;
  prof_collect(h);
  h->prof.enabled = enable != 0;
  return DGS_OK;
}

int dgs_profile_get(dgs_handle* h, int32_t kernel_id, double* total_ms, int64_t* launches) {
  if (!h || kernel_id < 0 || kernel_id >= DGS_K_COUNT) return DGS_ERR_INVALID_ARGUMENT;
  if (set_device(h)) return DGS_ERR_HIP;
  prof_collect(h);
  if (total_ms) *total_ms = h->prof.total_ms[kernel_id];
  if (launches) *launches = h->prof.launches[kernel_id];
  return DGS_OK;
}

int dgs_profile_reset(dgs_handle* h) {
  if (!h) return DGS_ERR_INVALID_ARGUMENT;
  if (set_device(h)) return DGS_ERR_HIP;
  prof_collect(h);
  for (int k = 0; k < DGS_K_COUNT; k++) { h->prof.total_ms[k] = 0; h->prof.launches[k] = 0; }
  return DGS_OK;
}

int dgs_get_counts(dgs_handle* h, int64_t out[8]) {
  if (!h || !out) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  for (int k = 0; k < 8; k++) out[k] = 0;
  out[0] = h->nt;
  out[1] = h->ns;
  if (h->prm.method == DGS_METHOD_NDT && h->have_target && h->grid_cells > 0) {
    if (h->counts_stale) {
      int sc[2] = {0, 0};
      DGS_HIP_TRY(h, hipMemcpyAsync(sc, h->dev_scalars.ptr, sizeof(sc), hipMemcpyDeviceToHost, h->stream));
      DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));
      h->n_occupied = sc[0];
      h->n_valid = sc[1];
      h->counts_stale = false;
    }
    out[2] = h->n_valid;
    out[3] = h->n_occupied;
    out[4] = h->grid_cells;
  }
  out[5] = h->last_evaluations;
  return DGS_OK;
}

int dgs_ndt_derivatives(dgs_handle* h, const double* p6, const float* T16, double* score, double* grad6, double* hess36) {
  if (!h || !p6 || !score || !grad6 || !hess36) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (h->prm.method != DGS_METHOD_NDT) return DGS_ERR_UNSUPPORTED;
  if (!h->have_target || h->nt == 0) return DGS_ERR_NO_TARGET;
  if (!h->have_source || h->ns == 0) return DGS_ERR_NO_SOURCE;
  return ndt_probe(h, p6, T16, score, grad6, hess36);
}

int dgs_ndt_hessian_double(dgs_handle* h, const double* p6, double* hess36) {
  if (!h || !p6 || !hess36) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (h->prm.method != DGS_METHOD_NDT || h->prm.ndt_strict_order == DGS_NDT_ORDER_FAST || !h->prm.ndt_hessian_recompute_double) return DGS_ERR_UNSUPPORTED;
  if (!h->have_target || h->nt == 0) return DGS_ERR_NO_TARGET;
  if (!h->have_source || h->ns == 0) return DGS_ERR_NO_SOURCE;
  double score, g6[6];
  return ndt_probe(h, p6, nullptr, &score, g6, hess36, 2);
}

int dgs_find_loop_candidates(dgs_handle* h, const double* accum_distance, const double* xy, int64_t n, int32_t on_device, double new_accum_distance,
                             const double* new_xy, double accum_distance_thresh, double distance_thresh, int32_t* indices, int64_t capacity, int64_t* n_out) {
  if (!h || !n_out || !new_xy || n < 0 || capacity < 0 || (n > 0 && (!accum_distance || !xy)) || (capacity > 0 && !indices) || n > INT32_MAX) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  *n_out = 0;
  if (n == 0) return DGS_OK;
  // staging in the handle (grown geometrically, never per call): [accum n | xy 2n] doubles, [indices cap] ints, one count.  One stream
  // synchronisation per call: the count comes back through the pinned block, the indices behind it in the same stream order.
  dgs::DevBuf<double>& din = h->fc_in;
  dgs::DevBuf<int>& dout = h->fc_out;
  dgs::DevBuf<long long>& dcnt = h->fc_cnt;
  const double* d_acc = accum_distance;
  const double* d_xy = xy;
  int* d_idx = indices;
  const int64_t cap = on_device ? capacity : std::min<int64_t>(capacity, n);   // (indices are int32 keyframe numbers: n <= INT32_MAX was checked)
  if (dcnt.reserve(1) != hipSuccess) { h->err = "hipMalloc failed"; return DGS_ERR_HIP; }
  if (ensure_pinned(h, 8192) != DGS_OK) return DGS_ERR_HIP;
  long long* count = reinterpret_cast<long long*>(reinterpret_cast<char*>(h->pinned) + 4096);
  *count = 0;
  if (!on_device) {
    if (din.reserve((size_t)3 * n) != hipSuccess || dout.reserve((size_t)std::max<int64_t>(cap, 1)) != hipSuccess) { h->err = "hipMalloc failed"; return DGS_ERR_HIP; }
    DGS_HIP_TRY(h, hipMemcpyAsync(din.ptr, accum_distance, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    DGS_HIP_TRY(h, hipMemcpyAsync(din.ptr + n, xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice, h->stream));
    d_acc = din.ptr; d_xy = din.ptr + n; d_idx = dout.ptr;
  }
  hipLaunchKernelGGL(find_candidates_kernel, dim3(1), dim3(1024), 0, h->stream, d_acc, d_xy, (long long)n, new_accum_distance, new_xy[0], new_xy[1],
                     accum_distance_thresh, distance_thresh, d_idx, (long long)cap, dcnt.ptr);
  DGS_HIP_TRY(h, hipMemcpyAsync(count, dcnt.ptr, sizeof(long long), hipMemcpyDeviceToHost, h->stream));
  // the kernel writes at most `cap` indices whatever it counts: all of them travel now (a few hundred keyframes), the count says how many are meant
  if (!on_device && cap > 0) DGS_HIP_TRY(h, hipMemcpyAsync(indices, d_idx, sizeof(int32_t) * (size_t)cap, hipMemcpyDeviceToHost, h->stream));
  DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (hipGetLastError() != hipSuccess) { h->err = "find_candidates_kernel failed"; return DGS_ERR_HIP; }
  *n_out = *count;
  if (*count > capacity) { h->err = "indices buffer too small for the candidates"; return DGS_ERR_INVALID_ARGUMENT; }
  return DGS_OK;
}

int dgs_calc_fitness_score(dgs_handle* h, const float* cloud1, int64_t n1, const float* cloud2, int64_t n2, int32_t on_device,
                           const float* relpose16, double max_range, double* score) {
  if (!h || !score || n1 < 0 || n2 < 0 || (n1 > 0 && !cloud1) || (n2 > 0 && !cloud2) || n1 > INT32_MAX || n2 > INT32_MAX) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  *score = DBL_MAX;
  if (n1 == 0 || n2 == 0) return DGS_OK;  // no neighbour / no query: "nr == 0" branch of the reference
  const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  int rc = upload_cloud(h, h->aux_cloud1, cloud1, n1, on_device);
  if (rc == DGS_OK) rc = upload_cloud(h, h->aux_cloud2, cloud2, n2, on_device);
  if (rc == DGS_OK) rc = bvh_build(h, h->aux_bvh, h->aux_cloud1.ptr, n1);
  const bool aux_grid = grid_wanted(h, n2);
  if (rc == DGS_OK && aux_grid) rc = nn_grid_build(h, h->aux_grid, h->aux_bvh, h->aux_cloud1.ptr, n1);
  if (rc != DGS_OK) return rc;
  // stage pointer / size / transform exactly like the single-pair fitness path, but against the auxiliary index
  hipStream_t st = h->stream;
  DGS_HIP_TRY(h, h->src_ptrs.reserve(1));
  DGS_HIP_TRY(h, h->src_sizes.reserve(1));
  DGS_HIP_TRY(h, h->inits.reserve(1));
  if (ensure_pinned(h, 8192) != DGS_OK) return DGS_ERR_HIP;
  char* base = reinterpret_cast<char*>(h->pinned) + 2048;
  const float4* src = h->aux_cloud2.ptr;
  const int ni = (int)n2;
  std::memcpy(base, &src, sizeof(void*));
  std::memcpy(base + 16, &ni, sizeof(int));
  std::memcpy(base + 64, relpose16 ? relpose16 : ident, sizeof(float) * 16);
  DGS_HIP_TRY(h, hipMemcpyAsync(h->src_ptrs.ptr, base, sizeof(void*), hipMemcpyHostToDevice, st));
  DGS_HIP_TRY(h, hipMemcpyAsync(h->src_sizes.ptr, base + 16, sizeof(int), hipMemcpyHostToDevice, st));
  DGS_HIP_TRY(h, hipMemcpyAsync(h->inits.ptr, base + 64, sizeof(float) * 16, hipMemcpyHostToDevice, st));
  double sum = 0;
  int64_t cnt = 0, inl = 0;
  rc = nn_fitness_batch_on(h, h->aux_bvh, (aux_grid && h->aux_grid.valid) ? &h->aux_grid : nullptr, 1, h->src_ptrs.ptr, h->src_sizes.ptr, ni, reinterpret_cast<const float*>(h->inits.ptr), 64, max_range, 0.0,
                           &sum, &cnt, &inl);
  if (rc != DGS_OK) return rc;
  *score = cnt > 0 ? sum / (double)cnt : DBL_MAX;
  return DGS_OK;
}

static int voxel_filter_call(int approx, dgs_handle* h, const float* in_xyz16, int64_t n, int32_t in_on_device, float leaf_size, float* out_xyz16,
                          int64_t out_capacity, int32_t out_on_device, int64_t* n_out) {
  if (!h || !n_out || n < 0 || (n > 0 && (!in_xyz16 || !out_xyz16)) || n > INT32_MAX || !(leaf_size > 0) || out_capacity < 0) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  *n_out = 0;
  if (n == 0) return DGS_OK;
  const float4* din = reinterpret_cast<const float4*>(in_xyz16);
  if (!in_on_device) {
    int rc = upload_cloud(h, h->aux_cloud1, in_xyz16, n, 0);
    if (rc) return rc;
    din = h->aux_cloud1.ptr;
  }
  float4* dout = reinterpret_cast<float4*>(out_xyz16);
  int64_t cap = out_capacity;
  if (!out_on_device) {
    DGS_HIP_TRY(h, h->aux_out.reserve((size_t)n));
    dout = h->aux_out.ptr;
    cap = n;
  }
  int64_t m = 0;
  int rc = approx ? approx_voxel_grid_filter(h, din, n, leaf_size, dout, cap, &m) : voxel_grid_filter(h, din, n, leaf_size, dout, cap, &m);
  if (rc) return rc;
  *n_out = m;
  if (m > out_capacity) {
    h->err = "output buffer too small for the filtered cloud";
    return DGS_ERR_INVALID_ARGUMENT;
  }
  if (!out_on_device && m > 0) {
    DGS_HIP_TRY(h, hipMemcpyAsync(out_xyz16, dout, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    DGS_HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return DGS_OK;
}

int dgs_voxel_grid_filter(dgs_handle* h, const float* in_xyz16, int64_t n, int32_t in_on_device, float leaf_size, float* out_xyz16,
                          int64_t out_capacity, int32_t out_on_device, int64_t* n_out) {
  return voxel_filter_call(0, h, in_xyz16, n, in_on_device, leaf_size, out_xyz16, out_capacity, out_on_device, n_out);
}

int dgs_approx_voxel_grid_filter(dgs_handle* h, const float* in_xyz16, int64_t n, int32_t in_on_device, float leaf_size, float* out_xyz16,
                                 int64_t out_capacity, int32_t out_on_device, int64_t* n_out) {
  return voxel_filter_call(1, h, in_xyz16, n, in_on_device, leaf_size, out_xyz16, out_capacity, out_on_device, n_out);
}

int dgs_gicp_get_covariances(dgs_handle* h, int32_t which, double* cov9) {
  if (!h || !cov9) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (h->prm.method != DGS_METHOD_GICP && h->prm.method != DGS_METHOD_VGICP) return DGS_ERR_UNSUPPORTED;
  if (which ? (!h->have_target || h->nt == 0) : (!h->have_source || h->ns == 0)) return which ? DGS_ERR_NO_TARGET : DGS_ERR_NO_SOURCE;
  return gicp_covariances(h, which, cov9, which ? h->nt : h->ns);
}

int dgs_vgicp_get_voxels(dgs_handle* h, int64_t capacity, int32_t* coord3, int32_t* counts, double* mean3, double* cov9, int64_t* n_voxels) {
  if (!h || !n_voxels || capacity < 0) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (h->prm.method != DGS_METHOD_VGICP) return DGS_ERR_UNSUPPORTED;
  if (set_device(h)) return DGS_ERR_HIP;
  if (!h->have_target) return DGS_ERR_NO_TARGET;
  return vgicp_voxels(h, capacity, coord3, counts, mean3, cov9, n_voxels);
}

int dgs_gicp_linearize(dgs_handle* h, const double* T16_rowmajor, int32_t error_only, double* error, double* hess36, double* b6) {
  if (!h || !T16_rowmajor || !error || (!error_only && (!hess36 || !b6))) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (h->prm.method != DGS_METHOD_GICP && h->prm.method != DGS_METHOD_VGICP) return DGS_ERR_UNSUPPORTED;
  if (!h->have_target || h->nt == 0) return DGS_ERR_NO_TARGET;
  if (!h->have_source || h->ns == 0) return DGS_ERR_NO_SOURCE;
  return gicp_probe(h, T16_rowmajor, error_only, error, hess36, b6);
}

int dgs_ndt_get_trajectory(dgs_handle* h, int32_t pair, double* poses6, int32_t* len) {
  if (!h || !poses6 || !len || pair < 0) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (h->prm.method != DGS_METHOD_NDT || (size_t)pair >= h->pairs.cap) return DGS_ERR_UNSUPPORTED;
  int n = 0;
  int rc = ndt_trajectory(h, pair, poses6, &n);
  *len = n;
  return rc;
}

int dgs_ndt_get_voxels(dgs_handle* h, int64_t* n, int64_t* keys, int32_t* counts, int32_t* valid, double* mean3, double* icov9) {
  if (!h || !n) return DGS_ERR_INVALID_ARGUMENT;
  h->err.clear();
  if (set_device(h)) return DGS_ERR_HIP;
  if (h->prm.method != DGS_METHOD_NDT) return DGS_ERR_UNSUPPORTED;
  if (!h->have_target) return DGS_ERR_NO_TARGET;
  int64_t c[8];
  int rc = dgs_get_counts(h, c);
  if (rc) return rc;
  const int64_t nv = c[3];
  *n = nv;
  if (!keys || nv == 0) return DGS_OK;
  std::vector<uint32_t> k32(nv);
  std::vector<double> dbg((size_t)nv * 12);
  DGS_HIP_TRY(h, hipMemcpy(k32.data(), h->run_keys.ptr, nv * sizeof(uint32_t), hipMemcpyDeviceToHost));
  DGS_HIP_TRY(h, hipMemcpy(counts, h->vox_count.ptr, nv * sizeof(int), hipMemcpyDeviceToHost));
  DGS_HIP_TRY(h, hipMemcpy(valid, h->vox_valid.ptr, nv * sizeof(int), hipMemcpyDeviceToHost));
  DGS_HIP_TRY(h, hipMemcpy(dbg.data(), h->vox_dbg.ptr, (size_t)nv * 12 * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < nv; i++) {
    keys[i] = (k32[i] == 0xFFFFFFFFu) ? -1 : (int64_t)k32[i];
    for (int a = 0; a < 3; a++) mean3[i * 3 + a] = dbg[i * 12 + a];
    for (int a = 0; a < 9; a++) icov9[i * 9 + a] = dbg[i * 12 + 3 + a];
  }
  return DGS_OK;
}

}  // extern "C"
