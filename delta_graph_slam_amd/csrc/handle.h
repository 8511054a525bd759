// dgs_handle: per-registration-object state (one per pcl::Registration instance on the reference side).
#pragma once
#include "common.h"

namespace dgs {

struct EventPair {
  hipEvent_t start, stop;
};

struct Profiler {
  bool enabled = false;
  std::vector<EventPair> pool;                 // all events ever created
  std::vector<int> pending[DGS_K_COUNT];       // indices into pool, recorded but not yet read
  size_t next_free = 0;
  double total_ms[DGS_K_COUNT] = {0};
  int64_t launches[DGS_K_COUNT] = {0};
};

// Exact-NN index over a point cloud (nn_bvh.hip): Hilbert-sorted points, implicit complete 8-ary tree of AABBs.
struct Bvh {
  int64_t n = 0;            // points
  int leaves = 0;           // number of leaf slots (power of two), each covers kLeafSize sorted points
  int levels = 0;
  DevBuf<float4> sorted;    // points in Hilbert order, w = original index (bit-cast int)
  DevBuf<float4> node_lo;   // AABB min per node (heap order, root = 0); w unused
  DevBuf<float4> node_hi;
  DevBuf<uint32_t> keys, keys_alt;
  DevBuf<uint32_t> vals, vals_alt;
  DevBuf<unsigned> kd_bbox;   // k-d order build: boxes of the ranges of the current level
  bool kd = false;          // points are in k-d (median split) order instead of Hilbert order
  bool valid = false;
};

// Sparse 64-ary voxel hierarchy over a target cloud for one-lane-per-query exact 1-NN distances (nn_grid.hip).  Belongs to a
// handle (rebuilt at every setInputTarget: one sort of the target plus a few passes), not to a cached cloud: the dense tables
// (one entry per COARSE cell) are a fixed 8 MB budget per handle whatever the cloud.
constexpr int64_t kNnGridL2Cells = 8192;   // L2 cells (16 x 16 x 16 fine cells each) the tables are sized for
struct NnGridParams {   // decided on the device (grid_params_kernel): no host round trip between the tree build and this one
  float org[3];         // corner of fine cell (0, 0, 0)
  float c, inv_c;       // fine cell size; a coarse cell is 4 x 4 x 4 fine cells, an L2 cell 4 x 4 x 4 coarse cells
  int n2[3];            // L2 cells per axis
  int n;
};
struct __attribute__((aligned(16))) NnCoarse {
  unsigned long long mask;   // occupied fine cells, bit x | y << 2 | z << 4
  int base;                  // rank (in key order) of the first occupied fine cell: index into cstart
  int pad;
};
struct NnGrid {
  DevBuf<float4> sorted;                   // target points ordered by (L2 cell, coarse sub-cell, fine sub-cell)
  DevBuf<NnCoarse> coarse;                 // [L2 cell * 64 + coarse sub-cell]
  DevBuf<unsigned long long> occ2;         // [L2 cell] occupied coarse cells
  DevBuf<int> cstart;                      // first point of every occupied fine cell, in key order; entry [runs] = n
  DevBuf<uint32_t> keys, keys_alt, vals, vals_alt, run_keys;   // run_keys is kept: the next build clears exactly those cells
  DevBuf<int> run_counts, scalars;         // scalars[0] = number of runs
  DevBuf<NnGridParams> params;
  DevBuf<unsigned> hist;
  // query side: squared NN distance per query of a batch, the two queues of still-open queries {query number, best so far}
  DevBuf<float> dist, q_best;
  DevBuf<unsigned> q_items;
  DevBuf<int> q_count;
  DevBuf<double> hook;   // 128-byte staging block of the test hook
  int64_t n = 0, n_prev = 0;
  bool valid = false;
  void release() {
    sorted.release(); coarse.release(); occ2.release(); cstart.release(); keys.release(); keys_alt.release(); vals.release(); vals_alt.release();
    run_keys.release(); run_counts.release(); scalars.release(); params.release(); hist.release();
    dist.release(); q_best.release(); q_items.release(); q_count.release(); hook.release();
    n = n_prev = 0; valid = false;
  }
};

// A cloud resident in HBM with everything derived from the points alone: the exact-NN index and (GICP) the regularised
// k-NN covariances.  Owned by a handle (setInputTarget / setInputSource copies) or by a dgs_cloud object that outlives
// many registrations (keyframe clouds of the loop detector).
struct CloudState {
  DevBuf<float4> pts;
  int64_t n = 0;
  Bvh bvh;
  DevBuf<double> cov;  // 6 doubles per point: xx, xy, xz, yy, yz, zz
  bool cov_valid = false;
  int cov_k = 0, cov_reg = -1;
  void invalidate() { bvh.valid = false; cov_valid = false; }
  void release() {
    pts.release(); cov.release();
    bvh.sorted.release(); bvh.node_lo.release(); bvh.node_hi.release();
    bvh.keys.release(); bvh.keys_alt.release(); bvh.vals.release(); bvh.vals_alt.release();
    bvh.kd_bbox.release();
    n = 0;
    invalidate();
  }
};

}  // namespace dgs

struct dgs_handle;
struct dgs_cloud {
  dgs::CloudState st;
  int device = 0;
  std::vector<dgs_handle*> users;  // handles whose tgt/src point at st: detached again when the cloud is destroyed first
};

struct dgs_handle {
  dgs_params prm;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev_poll[2] = {nullptr, nullptr};  // chunk-boundary events of the optimiser loops (created on first use)
  // side stream: small builds that the main stream does not need yet (the target's NN index while the batch iterates)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // third stream: the double-precision computeHessian launches of the upstream order run beside the main stream's launches (ndt_strict.h)
  hipStream_t hd_stream = nullptr;
  static constexpr int kHdEvents = 32;
  hipEvent_t ev_hd_a[kHdEvents] = {}, ev_hd_b[kHdEvents] = {};
  int strict_kernel = 3;              // DGS_NDT_STRICT_KERNEL: upstream-order kernels: 3 item-compacted (one launch per round), 2 lane-per-point (two launches per round)
  bool hd_overlap = true;             // DGS_NDT_HD_OVERLAP=0: the computeHessian launch of a round in line with the round's first launch
  bool ndt_speculate = true;          // DGS_NDT_SPECULATE=0: upstream order, item-compacted kernel: the Newton step's Jacobi SVD in the closing workgroup instead of speculated
  bool ndt_fixed_slices = false;     // DGS_NDT_FIXED_SLICES=1: upstream order, item-compacted kernel: a pair's slices are a function of its own size (ndt_strict.h strict_slices_of)
                                      // from the Gauss-Jordan direction and verified beside the next launch (NdtPair::spec_s)
  int solve_min_active = 0;           // DGS_NDT_SOLVE_MIN_ACTIVE (default 0 = the Newton step stays in the closing workgroup): item-compacted upstream-order kernel:
                                      // with at least this many pairs in a launch the Newton steps go to ndt_strict_solve_kernel on the third stream.  Measured on the
                                      // bench step: 6.58-6.65 ms with 2 / 4 / 8 / 16 against 6.24 ms in line -- a pair sits out a round per iteration, and the closings'
                                      // ~45 us chains were already hidden behind the other pairs' derivative work
  bool side_pending = false;
  bool side_build_deferred = false;   // forked, launches still to be enqueued (side_build_now)
  std::string err;

  // clouds (pcl::PointXYZ layout): the handle's own copies, or borrowed dgs_cloud objects
  dgs::CloudState own_target, own_source;
  dgs::CloudState* tgt = &own_target;
  dgs::CloudState* src = &own_source;
  dgs_cloud* tgt_cloud = nullptr;  // non-null while tgt / src borrow a dgs_cloud
  dgs_cloud* src_cloud = nullptr;
  int64_t nt = 0, ns = 0;
  bool have_target = false, have_source = false;

  // ---- NDT target model
  dgs::VoxelGrid grid{};
  dgs::DevBuf<int> cell2vox;
  dgs::DevBuf<dgs::VoxelRec> vox;
  dgs::DevBuf<float4> vox_centroid;
  dgs::DevBuf<double> vox_dbg;       // per occupied voxel: mean[3], icov[9]  (test hook; double-precision computeHessian pass)
  dgs::DevBuf<dgs::VoxelStrictRec> vox_strict;   // per occupied voxel: mean, float(icov) 3 x 3 (upstream evaluation orders)
  dgs::DevBuf<int> vox_count;        // points per occupied voxel
  dgs::DevBuf<int> vox_valid;
  dgs::DevBuf<uint32_t> key_in, key_out, val_in, val_out, run_keys;
  dgs::DevBuf<int> run_counts, run_offsets;
  dgs::DevBuf<int> dev_scalars;      // [0]=num_runs, [1]=n_valid
  dgs::DevBuf<uint32_t> vg_run_keys; // dgs_voxel_grid_filter's own runs / scalars: the NDT model's (above) stay what setInputTarget made them
  dgs::DevBuf<int> vg_scalars;
  dgs::DevBuf<float> minmax_partial; // block partials + final 6 floats
  dgs::DevBuf<unsigned char> cub_temp;
  int64_t grid_cells = 0;
  int64_t n_occupied = 0, n_valid = 0;  // filled lazily by counts query
  int64_t n_occupied_bound = 0;         // target points of the voxel model = an upper bound of its occupied voxels (known without a device round trip)
  bool counts_stale = true;

  // ---- NDT optimiser
  dgs::DevBuf<dgs::NdtPair> pairs;
  dgs::DevBuf<dgs::NdtInit> inits;
  dgs::DevBuf<double> partials;       // [pair][block][kAccumPad]
  dgs::DevBuf<int> done_counter;      // [0] = finished pairs
  int* done_flags = nullptr;          // pinned host memory, device-visible: one "finished" flag per pair of the running fused NDT batch
  int done_flags_cap = 0;
  dgs::DevBuf<int> ndt_queue;         // queue kernel: 16 control ints, then one 64-byte line per pair (its queue word)
  dgs::DevBuf<unsigned char> ndt_ring;   // queue kernel: one 384-byte record slot per pair and round (ndt_align.hip, kQueueSlotBytes)
  int ndt_ring_rounds = 0;
  int ndt_queue_mode = 0;             // DGS_NDT_QUEUE=1 (experiments build): the persistent queue kernel instead of one launch per evaluation (measured slower)
  bool ndt_schedule = false;          // DGS_NDT_SCHEDULE=1 (tests): the launch-per-evaluation path cuts every round like the queue kernel would
  int ndt_queue_min_pairs = 2;        // DGS_NDT_QUEUE_MIN_PAIRS: batches smaller than this keep the launch-per-evaluation path
  dgs::DevBuf<int> pair_blocks;       // slices the last derivative launch gave each pair
  dgs::DevBuf<double> strict_rows;    // ndt_strict_order 2: per-point totals, [pair][43][max_n] (column-major per pair)
  dgs::DevBuf<double> strict_totals;  // ndt_strict_order 1/2: [pair][kStrictPad] sums of one evaluation
  dgs::DevBuf<const float4*> src_ptrs;
  dgs::DevBuf<int> src_sizes;
  dgs::NdtConsts consts{};
  int64_t last_evaluations = 0;
  dgs::DevBuf<int> knn_nbr;           // k-NN sets of the cloud whose covariances are being made: [position in Hilbert order * 32 + slot]
  dgs::DevBuf<int> knn_stats;         // debug build (-DDGS_KNN_STATS): waves, waves on the cooperative path, candidate leaves
  bool nn_kd = true;                  // DGS_NN_KD=0: Hilbert order also for the loop batch's target index
  int64_t nn_kd_min_queries = 6 * 65536;   // DGS_NN_KD_MIN_QUERIES: source points of a batch from which its target index is built k-d ordered
  bool nn_kd_all = false;             // DGS_NN_KD_ALL=1: every target index is k-d ordered (tests of the k-d build through the single-query hooks)
  bool batch_kd = false;              // set by dgs_align_batch* around its work: the target index it builds is k-d ordered
  int knn_parts = 0;                  // DGS_KNN_PARTS: waves per leaf in gicp_knn_leaf_kernel (0 = by cloud size)
  bool knn_leaf = true;               // DGS_KNN_LEAF=0: the per-query k-NN walk (gicp_knn_kernel) instead of the wave-per-leaf search
  int knn_min_waves = 4096;           // DGS_KNN_MIN_WAVES: ... but never fewer waves than this (4 per SIMD)
  int knn_rounds = 8;                 // DGS_KNN_ROUNDS: rounds of 8 adjacent queries per wave in gicp_knn_kernel (1 = no warm bounds)
  bool gicp_fused = true;             // DGS_GICP_FUSED=0: the optimiser step of FAST_GICP / FAST_VGICP as its own launch (gicp_solve_kernel)
  bool ndt_pack2 = false;             // DGS_NDT_PACK2=1: DIRECT7 derivatives with two points per lane on packed FP32 (A/B measurements)
  bool ndt_fused = true;              // DGS_NDT_FUSED=0 at dgs_create: (derivatives, solve) launch pairs instead of fused launches

  // pinned host staging
  void* pinned = nullptr;
  size_t pinned_bytes = 0;

  // last result (single-pair API)
  float final_T[16];
  bool have_result = false;

  dgs::DevBuf<double> nn_partials;
  int nn_bpp = 0;                   // rows (workgroups) per pair of the fitness batch being prepared / walked
  double* fit_host = nullptr;       // pinned: the totals of the last fitness batch ({sum, count, inliers, 0} per pair)
  int fit_host_cap = 0;
  // dgs_align_batch -> ndt_align_pairs: walk the fitness of the candidates that have finished on the side stream while the rest iterate
  struct EarlyFitness {
    bool on = false;        // wanted for the batch being aligned
    bool enqueued = false;  // ndt_align_pairs has enqueued every pair's walk, the totals and their copy: read fit_host after its sync
    double max_range = 0.0;
    int max_n = 0;
    int lds_kb = 0;         // DGS_EARLY_FITNESS_LDS_KB: occupancy cap of the side-stream walks (workgroups per CU = 160 / this; 0 = none)
    int max_active = 1 << 30;   // DGS_EARLY_FITNESS_MAX_ACTIVE: side-stream walks start once at most this many pairs still iterate
    int min_pairs = 8;      // DGS_EARLY_FITNESS_MIN_PAIRS: a side-stream launch waits until this many finished pairs are ready
  } early_fit;
  // DGS_EARLY_FITNESS=1.  Off by default: measured on the bench workload (scripts/ab_early_fitness.sh, profiles/r03/early_fitness_ab.txt)
  // it never beat the plain order -- 2.19-2.22 ms per step against 2.15 at best (8+ pairs per launch, no occupancy cap), 2.4-3.3 ms with
  // the cap, 4.3 ms with a launch per finished pair.  A walk's workgroup lives ~100 us whatever the launch holds, so (i) a launch per
  // few pairs leaves the chip as empty as the tail it was meant to fill, (ii) a large launch floods every slot and the iteration
  // launch that arrives next waits one workgroup lifetime (35 -> 90-120 us in the trace), (iii) capped to 1-2 workgroups per CU the
  // walk is several times slower and the end of the batch waits for it.
  bool early_fitness_enabled = false;
  const double* nn_out = nullptr;   // device: {sum, count, inliers, 0} per pair of the last fitness batch (inside nn_partials; read by dgs_group's record kernel)
  dgs::DevBuf<float4> scratch_cloud;
  dgs::NnGrid tgt_grid, aux_grid;   // fitness-pass index over the current target / over cloud1 of dgs_calc_fitness_score
  int grid_mode = 0;                // DGS_NN_GRID at dgs_create: 0 (default) tree walk only, >= 1 grid pass in front of it, -1 grid pass for big batches.
                                    // Measured on the 32 x 65,536 bench step: the grid pass cuts the fitness kernels from 0.83 to 0.69 ms, but
                                    // its build and queue traffic give the gain back (2.99 vs 3.00 ms per step), so it stays opt-in.
  bool use_grid = false;            // decision for the current target (grid_wanted)
  int grid_levels = 1;              // DGS_NN_GRID=2: also the coarse-block pass between the fine-block pass and the tree
  float grid_spacing_factor = 6.f;  // DGS_NN_GRID_FACTOR: fine cell = factor x 2^floor(log2(median spacing))

  // ---- calc_fitness_score between two arbitrary clouds (InformationMatrixCalculator): own buffers, the registration's
  // target / source / result are left untouched
  dgs::DevBuf<float4> aux_cloud1, aux_cloud2, aux_out;
  // dgs_find_loop_candidates: staging kept across calls ([accum n | xy 2n], indices, count) -- the call is made every graph update
  dgs::DevBuf<double> fc_in;
  dgs::DevBuf<int> fc_out;
  dgs::DevBuf<long long> fc_cnt;
  dgs::Bvh aux_bvh;

  // ---- GICP (fast_gicp::FastGICP): k-NN covariances of both clouds, correspondences, Mahalanobis matrices
  dgs::DevBuf<int> corr;
  dgs::DevBuf<float> corr_sq;
  dgs::DevBuf<double> mahal;                   // 6 doubles per source point
  dgs::DevBuf<dgs::GicpPair> gpairs;
  dgs::DevBuf<dgs::GicpItem> gitems;
  // FAST_VGICP target model
  dgs::VgicpMap vmap{};
  dgs::DevBuf<dgs::VgicpVoxel> vvox;
  dgs::DevBuf<int> vcell2vox;
  bool vmap_valid = false;
  int64_t vmap_voxels = 0;
  dgs::DevBuf<float4> batch_slab;              // host sources of dgs_align_batch staged as one slab (kept across calls)
  std::vector<dgs::CloudState> batch_clouds;   // index + covariances of sources handed to dgs_align_batch as raw arrays
  dgs::GicpConsts gconsts{};

  dgs::Profiler prof;
};

namespace dgs {

// profiling wrappers (dgs_api.hip)
int prof_begin(dgs_handle* h, int kernel_id, hipStream_t st = nullptr);   // st: default the handle's stream
void prof_end(dgs_handle* h, int kernel_id, int slot, hipStream_t st = nullptr);
int ensure_pinned(dgs_handle* h, size_t bytes);
int ensure_poll_events(dgs_handle* h);
int side_fork(dgs_handle* h);   // side_stream continues from what the main stream has enqueued so far
int side_join(dgs_handle* h);   // the main stream waits for what side_fork()'s work (no-op when nothing is pending)
int side_build_now(dgs_handle* h);   // enqueues the deferred build of the target's NN index on the side stream

// ndt_voxel.hip
int ndt_build_target(dgs_handle* h);
int voxel_grid_filter(dgs_handle* h, const float4* in, int64_t n, float leaf, float4* out, int64_t out_capacity, int64_t* n_out);
int approx_voxel_grid_filter(dgs_handle* h, const float4* in, int64_t n, float leaf, float4* out, int64_t out_capacity, int64_t* n_out);   // pcl::ApproximateVoxelGrid
int cloud_minmax(dgs_handle* h, const float4* pts, int64_t n, float out6[6]);
int cloud_minmax_device(dgs_handle* h, const float4* pts, int64_t n, float** d_out6, hipStream_t st = nullptr);
// ndt_align.hip
int ndt_align_pairs(dgs_handle* h, int n_pairs, const float4* const* d_src_ptrs_host, const int* sizes_host,
                    const float* guesses16, dgs_result* results);
int ndt_trajectory(dgs_handle* h, int pair, double* out, int* len);
int ndt_probe(dgs_handle* h, const double* p6, const float* T16, double* score, double* g6, double* H36, int kind = 1);   // kind 2: the double-precision computeHessian pass
// nn_bvh.hip
int bvh_build(dgs_handle* h, Bvh& bvh, const float4* pts, int64_t n, hipStream_t st = nullptr, bool kd_order = false);  // st: default the handle's stream; kd_order: median-split order (slower build, faster queries)
int nn_fitness(dgs_handle* h, const float4* src, int64_t n, const float* T16, double max_range, double inlier_sq,
               double* sum, int64_t* count, int64_t* inliers);
// batched: device arrays of source pointers / sizes, device transforms (column-major 16 floats every T_stride_bytes)
int nn_fitness_batch(dgs_handle* h, int n_pairs, const float4* const* d_src_ptrs, const int* d_sizes, int max_size, const float* d_T,
                     size_t T_stride_bytes, double max_range, double inlier_sq, double* sums, int64_t* counts, int64_t* inliers);
int nn_fitness_batch_on(dgs_handle* h, const Bvh& index, NnGrid* grid, int n_pairs, const float4* const* d_src_ptrs, const int* d_sizes, int max_size,
                        const float* d_T, size_t T_stride_bytes, double max_range, double inlier_sq, double* sums, int64_t* counts, int64_t* inliers);
int nn_fitness_prepare(dgs_handle* h, int n_pairs, int max_size, bool grid);
// ids: `count` pair indices (< 65,536) to walk, or nullptr for pairs 0 .. count - 1
void nn_fitness_enqueue(dgs_handle* h, hipStream_t st, const Bvh& index, const int* ids, int count, const float4* const* d_src_ptrs, const int* d_sizes,
                        const float* d_T, size_t T_stride_bytes, double max_range, double inlier_sq, int background_lds_kb);
int nn_fitness_totals_enqueue(dgs_handle* h, int n_pairs);
void nn_fitness_read(const dgs_handle* h, int n_pairs, double* sums, int64_t* counts, int64_t* inliers);
int nn_search(dgs_handle* h, const float4* queries, int64_t m, int32_t* d_idx, float* d_sq);
int ensure_target_index(dgs_handle* h, hipStream_t st = nullptr);   // tree (+ grid when h->use_grid) over the current target
// the grid pass pays for its build (one more sort of the target) from ~4 x 65,536 queries on: fitness of a candidate batch
inline bool grid_wanted(const dgs_handle* h, int64_t queries) { return h->grid_mode > 0 || (h->grid_mode < 0 && queries >= 262144); }
// nn_grid.hip -- the one-lane-per-query voxel-hierarchy fitness index: measured slower than the 8-lane tree walk, so it
// is part of the EXPERIMENTS build only (`make experiments` -> libdgs_reg_exp.so, -DDGS_EXPERIMENTS); the product library does not
// carry its kernels and ignores DGS_NN_GRID.
#ifdef DGS_EXPERIMENTS
constexpr bool kExperiments = true;
int nn_grid_build(dgs_handle* h, NnGrid& G, const Bvh& bvh, const float4* pts, int64_t n, hipStream_t st = nullptr);
int nn_grid_launch_fitness(dgs_handle* h, NnGrid& G, const Bvh& index, int n_pairs, const float4* const* d_src_ptrs, const int* d_sizes, int max_n, const float* d_T,
                           size_t T_stride_bytes, float max_range, float inlier_sq, double* partial, int bpp);
int nn_grid_search(dgs_handle* h, NnGrid& G, const Bvh& index, const float4* queries, int64_t m, float* d_sq);
#else
constexpr bool kExperiments = false;
inline int nn_grid_build(dgs_handle*, NnGrid&, const Bvh&, const float4*, int64_t, hipStream_t = nullptr) { return DGS_ERR_UNSUPPORTED; }
inline int nn_grid_launch_fitness(dgs_handle*, NnGrid&, const Bvh&, int, const float4* const*, const int*, int, const float*, size_t, float, float, double*, int) { return DGS_ERR_UNSUPPORTED; }
inline int nn_grid_search(dgs_handle*, NnGrid&, const Bvh&, const float4*, int64_t, float*) { return DGS_ERR_UNSUPPORTED; }
#endif
// gicp.hip
int vgicp_build_map(dgs_handle* h);  // vgicp_voxel.hip: needs the target covariances
int vgicp_voxels(dgs_handle* h, int64_t capacity, int32_t* coord3, int32_t* counts, double* mean3, double* cov9, int64_t* n_voxels);
int gicp_ensure_target_covariance(dgs_handle* h);
int gicp_align(dgs_handle* h, const float* guess16, dgs_result* out);
int gicp_align_batch(dgs_handle* h, int n, CloudState* const* srcs, const float* guesses16, dgs_result* out);
const float* gicp_final_transforms(dgs_handle* h, size_t* stride_bytes);
int gicp_covariances(dgs_handle* h, int which, double* host_out6, int64_t n);
int gicp_probe(dgs_handle* h, const double* T16_rowmajor, int error_only, double* err, double* H36, double* b6);
// transform
int transform_cloud(dgs_handle* h, const float4* in, float4* out, int64_t n, const float* T16_colmajor_host);
// dgs_api.hip: a copy of `src` (points only) on dst_h's device, device to device (peer copy over xGMI when the devices differ)
int cloud_clone_to(dgs_handle* dst_h, const dgs_cloud* src, dgs_cloud** out);

}  // namespace dgs
