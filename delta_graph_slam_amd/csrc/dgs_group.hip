// dgs_group: LoopDetector::matching's candidate loop (/root/reference/include/hdl_graph_slam/loop_detector.hpp:119-173) over several
// GPUs of ONE process -- what the nodelet can link, since the reference runs loop detection inside the nodelet manager under
// main_thread_mutex (apps/delta_graph_slam_nodelet.cpp:797,816).
//
// MI355X design: one dgs_handle, one host thread and one stream per device; candidate c goes to member c mod G;
// the target is uploaded to every member in parallel (G host->device copies over G PCIe links beat one copy + a broadcast for
// a 1 MB cloud); every member runs its share as one batched launch sequence (dgs_align_batch) with no data-path collective.  The
// one exchange step is an ncclAllGather (RCCL over xGMI, communicators from ncclCommInitAll) of fixed 96-byte result records;
// the arg-min then runs in ORIGINAL candidate order, so loop_detector.hpp:149's tie rule (a later candidate replaces an earlier
// one on an equal score) holds whatever the device count.  RCCL is loaded with dlopen at dgs_group_create: libdgs_reg.so itself
// does not link it, and a group falls back to gathering on the host when the library is missing or a device is listed twice
// (the one-GPU rehearsal).  No exception crosses the boundary; a member's failure is reported per candidate.
#include <dlfcn.h>

#include <atomic>
#include <cfloat>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <set>
#include <thread>

#include "handle.h"

namespace {

// ---- the six RCCL entry points, resolved at run time ---------------------------------------------------------------------------
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;   // optional: dgs_group_rccl_ranks
  bool ok() const { return lib != nullptr; }
};
constexpr int kNcclUint8 = 1;   // ncclDataType_t::ncclUint8 (rccl.h)

Rccl load_rccl() {
  Rccl r;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.lib) break;
  }
  if (!r.lib) return r;
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
  r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
  if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd) {
    dlclose(r.lib);
    r.lib = nullptr;
  }
  return r;
}

// the exchanged record: what loop_detector.hpp:145-155 reads per candidate, fixed size
struct Record {
  double fitness, score;
  float T[16];
  int32_t candidate, converged, iterations, evaluations, status, pad;
};
static_assert(sizeof(Record) == 104, "record layout");
constexpr size_t kRecordBytes = 128;   // padded: a whole number of 16-byte lines per record

// one persistent worker per member: runs the closures the group hands it, in order
struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, stop = false;
  void start() {
    th = std::thread([this] {
      for (;;) {
        std::function<void()> j;
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [this] { return has_job || stop; });
          if (stop) return;
          j = std::move(job);
          has_job = false;
        }
        j();
        {
          std::lock_guard<std::mutex> lk(m);
          done = true;
        }
        cv.notify_all();
      }
    });
  }
  void submit(std::function<void()> j) {
    {
      std::lock_guard<std::mutex> lk(m);
      job = std::move(j);
      has_job = true;
      done = false;
    }
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] { return done; });
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

}  // namespace

struct dgs_group {
  std::vector<dgs_handle*> members;
  std::vector<int> devices;
  std::vector<Worker*> workers;
  Rccl rccl;
  std::vector<void*> comms;          // one ncclComm_t per member (empty: host gather)
  std::vector<void*> d_send, d_recv; // per member: its records / everybody's records
  std::vector<void*> d_stage;        // per member: {candidate numbers | guesses} of its share, uploaded BEFORE its batch starts
  std::vector<void*> h_stage;        // pinned staging per member
  size_t cap_per_member = 0;         // records each member's buffers hold
  bool used_rccl = false;
  std::string err;
};

// KeyFrame::cloud (keyframe.hpp:51: set once, never written again) resident on the group's devices: one dgs_cloud per member
// that holds a copy.  Outlives any number of ticks; may outlive the group (the copies are plain dgs_cloud objects).
struct dgs_group_cloud {
  std::vector<dgs_cloud*> copy;      // [member] or nullptr
  int64_t n = 0;
};

namespace {

// keeps the CALLER's current HIP device what it was: the group's functions visit every member's device on the caller's thread
struct DeviceScope {
  int prev = -1;
  DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void free_buffers(dgs_group* g) {
  for (size_t k = 0; k < g->members.size(); k++) {
    (void)hipSetDevice(g->devices[k]);
    if (k < g->d_send.size() && g->d_send[k]) (void)hipFree(g->d_send[k]);
    if (k < g->d_recv.size() && g->d_recv[k]) (void)hipFree(g->d_recv[k]);
    if (k < g->d_stage.size() && g->d_stage[k]) (void)hipFree(g->d_stage[k]);
    if (k < g->h_stage.size() && g->h_stage[k]) (void)hipHostFree(g->h_stage[k]);
  }
  g->d_send.clear(); g->d_recv.clear(); g->d_stage.clear(); g->h_stage.clear();
  g->cap_per_member = 0;
}

constexpr size_t kStageBytes = 4 + 64;   // per candidate: its number in the caller's list, its guess (16 floats)

bool ensure_buffers(dgs_group* g, size_t per_member) {
  if (per_member <= g->cap_per_member) return true;
  free_buffers(g);
  const size_t G = g->members.size();
  g->d_send.assign(G, nullptr); g->d_recv.assign(G, nullptr); g->d_stage.assign(G, nullptr); g->h_stage.assign(G, nullptr);
  const size_t want = per_member + per_member / 2 + 8;
  for (size_t k = 0; k < G; k++) {
    if (hipSetDevice(g->devices[k]) != hipSuccess || hipMalloc(&g->d_send[k], want * kRecordBytes) != hipSuccess ||
        hipMalloc(&g->d_recv[k], want * kRecordBytes * G) != hipSuccess || hipMalloc(&g->d_stage[k], want * kStageBytes) != hipSuccess ||
        hipHostMalloc(&g->h_stage[k], std::max(want * kRecordBytes * G, want * kStageBytes), hipHostMallocDefault) != hipSuccess) {
      g->err = "dgs_group: buffer allocation failed";
      free_buffers(g);
      return false;
    }
  }
  g->cap_per_member = want;
  return true;
}

void fail_share(std::vector<dgs_result>& res, const float* gs, int rc) {
  const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (size_t j = 0; j < res.size(); j++) {
    std::memcpy(res[j].final_transformation, gs ? gs + 16 * j : ident, sizeof(float) * 16);
    res[j].converged = 0; res[j].iterations = 0; res[j].evaluations = 0; res[j].status = rc;
    res[j].score = 0.0; res[j].fitness = NAN;
  }
}

// The dealt batch: `ids[k]` = the candidates of member k in the order it registers them; `run(k, results_of_member)` registers them.
int run_dealt_batch(dgs_group* g, int n, const std::vector<std::vector<int>>& ids, const float* guesses16, int32_t compute_fitness,
                    const std::function<int(int, const float*, dgs_result*)>& run, dgs_result* results, int32_t* best_index, double* best_score) {
  const int G = (int)g->members.size();
  int per = 1;
  for (int k = 0; k < G; k++) per = std::max(per, (int)ids[k].size());
  std::vector<std::vector<float>> gs(G);
  std::vector<std::vector<dgs_result>> res(G);
  std::vector<int> rcs(G, DGS_OK);
  for (int k = 0; k < G; k++) {
    res[k].resize(ids[k].size());
    if (guesses16)
      for (int c : ids[k]) gs[k].insert(gs[k].end(), guesses16 + 16 * (size_t)c, guesses16 + 16 * (size_t)c + 16);
  }
  const bool have_buffers = ensure_buffers(g, (size_t)per);
  // ---- every member registers its share as one batch on its own thread / stream, then writes its records on the device
  for (int k = 0; k < G; k++) {
    auto job = [k, guesses16, &ids, &gs, &res, &rcs, &run] {
      const float* gk = guesses16 ? gs[k].data() : nullptr;
      if (!ids[k].empty()) rcs[k] = run(k, gk, res[k].data());
      if (rcs[k] != DGS_OK) fail_share(res[k], gk, rcs[k]);   // reported per candidate as "not converged" (the reference skips them, loop_detector.hpp:149)
    };
    if (g->workers[k]) g->workers[k]->submit(job); else job();
  }
  for (int k = 0; k < G; k++)
    if (g->workers[k]) g->workers[k]->wait();
  for (int k = 0; k < G; k++)
    if (rcs[k] != DGS_OK) g->err = "member " + std::to_string(k) + " (device " + std::to_string(g->devices[k]) + "): " + dgs_last_error(g->members[k]);
  // Every member's records are made from the result array its batch call filled anyway (dgs_align_batch ends with one synchronisation and
  // the results on the host): 128 bytes per candidate, staged through pinned memory into the all-gather.  (Round 3 rebuilt the same
  // records on the device with a kernel of its own behind that synchronisation -- a memset, a copy and a launch that bought nothing.)  A
  // member whose batch failed contributes "failed" records: the collective still needs its contribution.
  std::vector<Record> all((size_t)G * per);
  for (auto& r : all) { std::memset(&r, 0, sizeof(r)); r.candidate = -1; }
  auto fill_host = [&](int k, Record* dst) {
    for (size_t j = 0; j < res[k].size(); j++) {
      Record& r = dst[j];
      const dgs_result& a = res[k][j];
      r.fitness = a.fitness; r.score = a.score;
      std::memcpy(r.T, a.final_transformation, sizeof(r.T));
      r.candidate = ids[k][j];
      r.converged = a.converged; r.iterations = a.iterations; r.evaluations = a.evaluations; r.status = a.status;
    }
  };
  bool ok = have_buffers;
  for (int k = 0; k < G && ok; k++) {
    char* hs = static_cast<char*>(g->h_stage[k]);
    std::vector<Record> mine(per);
    for (auto& r : mine) { std::memset(&r, 0, sizeof(r)); r.candidate = -1; }
    fill_host(k, mine.data());
    std::memset(hs, 0, (size_t)per * kRecordBytes);
    for (int j = 0; j < per; j++) std::memcpy(hs + (size_t)j * kRecordBytes, &mine[j], sizeof(Record));
    ok = hipSetDevice(g->devices[k]) == hipSuccess &&
         hipMemcpyAsync(g->d_send[k], hs, (size_t)per * kRecordBytes, hipMemcpyHostToDevice, g->members[k]->stream) == hipSuccess;
  }
  // ---- the exchange step: fixed-size records, all-gathered over RCCL (xGMI) when the group has communicators; the members' streams
  // carry it right behind their record kernels
  bool gathered = false;
  if (ok && !g->comms.empty()) {
    ok = g->rccl.GroupStart() == 0;
    for (int k = 0; k < G && ok; k++)
      ok = g->rccl.AllGather(g->d_send[k], g->d_recv[k], (size_t)per * kRecordBytes, kNcclUint8, g->comms[k], g->members[k]->stream) == 0;
    ok = (g->rccl.GroupEnd() == 0) && ok;
    if (ok) {   // member 0 holds everybody's records after the collective: one device->host copy
      char* hs = static_cast<char*>(g->h_stage[0]);
      ok = hipSetDevice(g->devices[0]) == hipSuccess &&
           hipMemcpyAsync(hs, g->d_recv[0], (size_t)G * per * kRecordBytes, hipMemcpyDeviceToHost, g->members[0]->stream) == hipSuccess;
      for (int k = 0; k < G && ok; k++) ok = hipSetDevice(g->devices[k]) == hipSuccess && hipStreamSynchronize(g->members[k]->stream) == hipSuccess;
      if (ok) {
        for (size_t j = 0; j < (size_t)G * per; j++) std::memcpy(&all[j], hs + j * kRecordBytes, sizeof(Record));
        gathered = true;
        g->used_rccl = true;
      }
    }
    if (!ok) g->err = "dgs_group: RCCL all-gather failed, gathered on the host instead";
  } else if (ok) {
    // no communicators (a device listed twice, or no RCCL): every member's records come back from its device buffer directly
    for (int k = 0; k < G && ok; k++)
      ok = hipSetDevice(g->devices[k]) == hipSuccess &&
           hipMemcpyAsync(g->h_stage[k], g->d_send[k], (size_t)per * kRecordBytes, hipMemcpyDeviceToHost, g->members[k]->stream) == hipSuccess;
    for (int k = 0; k < G && ok; k++) ok = hipSetDevice(g->devices[k]) == hipSuccess && hipStreamSynchronize(g->members[k]->stream) == hipSuccess;
    if (ok) {
      for (int k = 0; k < G; k++)
        for (int j = 0; j < per; j++) std::memcpy(&all[(size_t)k * per + j], static_cast<char*>(g->h_stage[k]) + (size_t)j * kRecordBytes, sizeof(Record));
      gathered = true;
    }
  }
  if (!gathered)   // last resort (allocation / copy failure): the members' host-side results
    for (int k = 0; k < G; k++) fill_host(k, all.data() + (size_t)k * per);
  // ---- results back in ORIGINAL candidate order, then the arg-min of loop_detector.hpp:126-156
  for (const Record& r : all) {
    if (r.candidate < 0 || r.candidate >= n) continue;
    dgs_result& o = results[r.candidate];
    std::memcpy(o.final_transformation, r.T, sizeof(r.T));
    o.converged = r.converged; o.iterations = r.iterations; o.evaluations = r.evaluations; o.status = r.status;
    o.score = r.score; o.fitness = r.fitness;
  }
  double best = DBL_MAX;
  int bi = -1;
  if (compute_fitness)
    for (int c = 0; c < n; c++) {
      if (!results[c].converged || results[c].fitness > best || results[c].fitness != results[c].fitness) continue;   // "score > best_score" skips; ties: the later wins
      best = results[c].fitness;
      bi = c;
    }
  if (best_index) *best_index = bi;
  if (best_score) *best_score = best;
  return DGS_OK;
}

}  // namespace

extern "C" {

int dgs_group_create(const dgs_params* params, const int32_t* devices, int32_t n_devices, dgs_group** out) {
  if (!params || !out || !devices || n_devices < 1 || n_devices > 64) return DGS_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  DeviceScope keep;
  dgs_group* g = new (std::nothrow) dgs_group();
  if (!g) return DGS_ERR_HIP;
  int rc = DGS_OK;
  for (int k = 0; k < n_devices && rc == DGS_OK; k++) {
    dgs_params p = *params;
    p.device = devices[k];
    dgs_handle* h = nullptr;
    rc = dgs_create(&p, &h);
    if (rc == DGS_OK) {
      g->members.push_back(h);
      g->devices.push_back(h->device);
    }
  }
  if (rc != DGS_OK) {
    for (dgs_handle* h : g->members) dgs_destroy(h);
    delete g;
    return rc;
  }
  for (int k = 0; k < n_devices; k++) {
    Worker* w = new (std::nothrow) Worker();
    if (w) w->start();
    g->workers.push_back(w);
  }
  // RCCL communicators over the members' devices; a device listed twice (one-GPU rehearsal) or a missing library: host gather
  std::set<int> distinct(g->devices.begin(), g->devices.end());
  if ((int)distinct.size() == n_devices) {
    g->rccl = load_rccl();
    if (g->rccl.ok()) {
      g->comms.assign(n_devices, nullptr);
      const int e = g->rccl.CommInitAll(g->comms.data(), n_devices, g->devices.data());
      if (e != 0) {
        g->err = std::string("ncclCommInitAll: ") + (g->rccl.GetErrorString ? g->rccl.GetErrorString(e) : "failed") + " (gathering on the host instead)";
        g->comms.clear();
      }
    }
  }
  *out = g;
  return DGS_OK;
}

void dgs_group_destroy(dgs_group* g) {
  if (!g) return;
  DeviceScope keep;
  for (Worker* w : g->workers)
    if (w) { w->shutdown(); delete w; }
  for (size_t k = 0; k < g->comms.size(); k++)
    if (g->comms[k]) { (void)hipSetDevice(g->devices[k]); (void)g->rccl.CommDestroy(g->comms[k]); }
  free_buffers(g);
  for (dgs_handle* h : g->members) dgs_destroy(h);
  // the RCCL library stays loaded: unloading a runtime that owns device state is not worth the risk at process exit
  delete g;
}

const char* dgs_group_last_error(const dgs_group* g) { return g ? g->err.c_str() : ""; }
int32_t dgs_group_size(const dgs_group* g) { return g ? (int32_t)g->members.size() : 0; }
int32_t dgs_group_uses_rccl(const dgs_group* g) { return (g && !g->comms.empty()) ? 1 : 0; }
// ranks of the group's RCCL communicator as RCCL itself reports them (ncclCommCount on member 0's communicator); 0 without RCCL
int32_t dgs_group_rccl_ranks(const dgs_group* g) {
  if (!g || g->comms.empty() || !g->rccl.CommCount) return 0;
  int n = 0;
  if (g->rccl.CommCount(g->comms[0], &n) != 0) return 0;
  return n;
}
dgs_handle* dgs_group_member(dgs_group* g, int32_t k) { return (g && k >= 0 && (size_t)k < g->members.size()) ? g->members[k] : nullptr; }

// runs job(k) for every member on its worker thread and waits; returns the first failing member's status
static int for_each_member(dgs_group* g, const std::function<int(int)>& job, const std::function<bool(int)>& wanted = nullptr) {
  const size_t G = g->members.size();
  std::vector<int> rcs(G, DGS_OK);
  for (size_t k = 0; k < G; k++) {
    if (wanted && !wanted((int)k)) continue;
    auto j = [&rcs, &job, k] { rcs[k] = job((int)k); };
    if (g->workers[k]) g->workers[k]->submit(j); else j();
  }
  for (size_t k = 0; k < G; k++)
    if (g->workers[k]) g->workers[k]->wait();
  for (size_t k = 0; k < G; k++)
    if (rcs[k] != DGS_OK) {
      g->err = "member " + std::to_string(k) + " (device " + std::to_string(g->devices[k]) + "): " + dgs_last_error(g->members[k]);
      return rcs[k];
    }
  return DGS_OK;
}

int dgs_group_set_input_target(dgs_group* g, const float* xyz16, int64_t n) {
  if (!g || n < 0 || (n > 0 && !xyz16)) return DGS_ERR_INVALID_ARGUMENT;
  g->err.clear();
  DeviceScope keep;
  return for_each_member(g, [g, xyz16, n](int k) { return dgs_set_input_target(g->members[k], xyz16, n, 0); });
}

int dgs_group_align_batch(dgs_group* g, int32_t n, const float* const* sources, const int64_t* sizes, const float* guesses16, int32_t compute_fitness,
                          double fitness_max_range, dgs_result* results, int32_t* best_index, double* best_score) {
  if (!g || n < 0 || (n > 0 && (!sources || !sizes || !results))) return DGS_ERR_INVALID_ARGUMENT;
  g->err.clear();
  g->used_rccl = false;
  if (best_index) *best_index = -1;
  if (best_score) *best_score = DBL_MAX;
  if (n == 0) return DGS_OK;
  DeviceScope keep;
  const int G = (int)g->members.size();
  // ---- deal: candidate c -> member c mod G
  std::vector<std::vector<int>> ids(G);
  std::vector<std::vector<const float*>> src(G);
  std::vector<std::vector<int64_t>> sz(G);
  for (int c = 0; c < n; c++) {
    ids[c % G].push_back(c);
    src[c % G].push_back(sources[c]);
    sz[c % G].push_back(sizes[c]);
  }
  auto run = [&](int k, const float* gk, dgs_result* out) {
    return dgs_align_batch(g->members[k], (int32_t)src[k].size(), src[k].data(), sz[k].data(), 0, gk, compute_fitness, fitness_max_range, out);
  };
  return run_dealt_batch(g, n, ids, guesses16, compute_fitness, run, results, best_index, best_score);
}

// ---- keyframe clouds resident on the group's devices -----------------------------------------------------------------------------
int dgs_group_cloud_create(dgs_group* g, const float* xyz16, int64_t n, int32_t owner, dgs_group_cloud** out) {
  if (!g || !out || n < 0 || (n > 0 && !xyz16) || owner < -1) return DGS_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  g->err.clear();
  DeviceScope keep;
  const int G = (int)g->members.size();
  dgs_group_cloud* c = new (std::nothrow) dgs_group_cloud();
  if (!c) return DGS_ERR_HIP;
  c->copy.assign(G, nullptr);
  c->n = n;
  const int only = owner < 0 ? -1 : owner % G;
  const int rc = for_each_member(g, [g, c, xyz16, n](int k) { return dgs_cloud_create(g->members[k], xyz16, n, 0, &c->copy[k]); },
                                 [only](int k) { return only < 0 || k == only; });
  if (rc != DGS_OK) {
    dgs_group_cloud_destroy(c);
    return rc;
  }
  *out = c;
  return DGS_OK;
}

void dgs_group_cloud_destroy(dgs_group_cloud* c) {
  if (!c) return;
  DeviceScope keep;
  for (dgs_cloud* m : c->copy)
    if (m) dgs_cloud_destroy(m);
  delete c;
}

int64_t dgs_group_cloud_size(const dgs_group_cloud* c) { return c ? c->n : 0; }

int32_t dgs_group_cloud_copies(const dgs_group_cloud* c) {
  int32_t m = 0;
  if (c)
    for (dgs_cloud* x : c->copy) m += x ? 1 : 0;
  return m;
}

int dgs_group_cloud_trim(dgs_group* g, dgs_group_cloud* c, int32_t owner) {
  if (!g || !c || c->copy.size() != g->members.size()) return DGS_ERR_INVALID_ARGUMENT;
  const int G = (int)c->copy.size();
  int keep = owner >= 0 ? owner % G : -1;
  if (keep < 0 || !c->copy[keep]) {
    keep = -1;
    for (int k = 0; k < G && keep < 0; k++)
      if (c->copy[k]) keep = k;
  }
  if (keep < 0) return DGS_ERR_INVALID_ARGUMENT;
  DeviceScope scope;
  for (int k = 0; k < G; k++)
    if (k != keep && c->copy[k]) {
      dgs_cloud_destroy(c->copy[k]);
      c->copy[k] = nullptr;
    }
  return DGS_OK;
}

int dgs_group_set_input_target_cloud(dgs_group* g, dgs_group_cloud* c) {
  if (!g || !c || c->copy.size() != g->members.size()) return DGS_ERR_INVALID_ARGUMENT;
  g->err.clear();
  DeviceScope keep;
  int holder = -1;
  for (size_t k = 0; k < c->copy.size() && holder < 0; k++)
    if (c->copy[k]) holder = (int)k;
  if (holder < 0) return DGS_ERR_INVALID_ARGUMENT;
  // the new keyframe is every member's target: members without a copy take one from a holder, device to device (xGMI), and keep it
  return for_each_member(g, [g, c, holder](int k) {
    if (!c->copy[k]) {
      const int rc = dgs::cloud_clone_to(g->members[k], c->copy[holder], &c->copy[k]);
      if (rc != DGS_OK) return rc;
    }
    return dgs_set_input_target_cloud(g->members[k], c->copy[k]);
  });
}

int dgs_group_align_batch_clouds(dgs_group* g, int32_t n, dgs_group_cloud* const* sources, const float* guesses16, int32_t compute_fitness,
                                 double fitness_max_range, dgs_result* results, int32_t* best_index, double* best_score) {
  if (!g || n < 0 || (n > 0 && (!sources || !results))) return DGS_ERR_INVALID_ARGUMENT;
  g->err.clear();
  g->used_rccl = false;
  if (best_index) *best_index = -1;
  if (best_score) *best_score = DBL_MAX;
  if (n == 0) return DGS_OK;
  DeviceScope keep;
  const int G = (int)g->members.size();
  // ---- deal: candidate c -> member c mod G when that member holds the keyframe, else the keyframe's first holder (its owner)
  std::vector<std::vector<int>> ids(G);
  std::vector<std::vector<dgs_cloud*>> cl(G);
  for (int c = 0; c < n; c++) {
    const dgs_group_cloud* s = sources[c];
    if (!s || (int)s->copy.size() != G) return DGS_ERR_INVALID_ARGUMENT;
    int k = c % G;
    if (!s->copy[k]) {
      k = -1;
      for (int j = 0; j < G && k < 0; j++)
        if (s->copy[j]) k = j;
      if (k < 0) return DGS_ERR_INVALID_ARGUMENT;
    }
    ids[k].push_back(c);
    cl[k].push_back(s->copy[k]);
  }
  auto run = [&](int k, const float* gk, dgs_result* out) {
    return dgs_align_batch_clouds(g->members[k], (int32_t)cl[k].size(), cl[k].data(), gk, compute_fitness, fitness_max_range, out);
  };
  return run_dealt_batch(g, n, ids, guesses16, compute_fitness, run, results, best_index, best_score);
}

int32_t dgs_group_last_gather_used_rccl(const dgs_group* g) { return (g && g->used_rccl) ? 1 : 0; }

}  // extern "C"
