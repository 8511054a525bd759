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
  std::vector<void*> h_stage;        // pinned staging per member
  size_t cap_per_member = 0;         // records each member's buffers hold
  bool used_rccl = false;
  std::string err;
};

namespace {

void free_buffers(dgs_group* g) {
  for (size_t k = 0; k < g->members.size(); k++) {
    (void)hipSetDevice(g->devices[k]);
    if (k < g->d_send.size() && g->d_send[k]) (void)hipFree(g->d_send[k]);
    if (k < g->d_recv.size() && g->d_recv[k]) (void)hipFree(g->d_recv[k]);
    if (k < g->h_stage.size() && g->h_stage[k]) (void)hipHostFree(g->h_stage[k]);
  }
  g->d_send.clear(); g->d_recv.clear(); g->h_stage.clear();
  g->cap_per_member = 0;
}

bool ensure_buffers(dgs_group* g, size_t per_member) {
  if (per_member <= g->cap_per_member) return true;
  free_buffers(g);
  const size_t G = g->members.size();
  g->d_send.assign(G, nullptr); g->d_recv.assign(G, nullptr); g->h_stage.assign(G, nullptr);
  const size_t want = per_member + per_member / 2 + 8;
  for (size_t k = 0; k < G; k++) {
    if (hipSetDevice(g->devices[k]) != hipSuccess || hipMalloc(&g->d_send[k], want * kRecordBytes) != hipSuccess ||
        hipMalloc(&g->d_recv[k], want * kRecordBytes * G) != hipSuccess || hipHostMalloc(&g->h_stage[k], want * kRecordBytes * G, hipHostMallocDefault) != hipSuccess) {
      g->err = "dgs_group: buffer allocation failed";
      free_buffers(g);
      return false;
    }
  }
  g->cap_per_member = want;
  return true;
}

}  // namespace

extern "C" {

int dgs_group_create(const dgs_params* params, const int32_t* devices, int32_t n_devices, dgs_group** out) {
  if (!params || !out || !devices || n_devices < 1 || n_devices > 64) return DGS_ERR_INVALID_ARGUMENT;
  *out = nullptr;
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
dgs_handle* dgs_group_member(dgs_group* g, int32_t k) { return (g && k >= 0 && (size_t)k < g->members.size()) ? g->members[k] : nullptr; }

int dgs_group_set_input_target(dgs_group* g, const float* xyz16, int64_t n) {
  if (!g || n < 0 || (n > 0 && !xyz16)) return DGS_ERR_INVALID_ARGUMENT;
  g->err.clear();
  const size_t G = g->members.size();
  std::vector<int> rcs(G, DGS_OK);
  for (size_t k = 0; k < G; k++) {
    auto job = [g, k, xyz16, n, &rcs] { rcs[k] = dgs_set_input_target(g->members[k], xyz16, n, 0); };
    if (g->workers[k]) g->workers[k]->submit(job); else job();
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

int dgs_group_align_batch(dgs_group* g, int32_t n, const float* const* sources, const int64_t* sizes, const float* guesses16, int32_t compute_fitness,
                          double fitness_max_range, dgs_result* results, int32_t* best_index, double* best_score) {
  if (!g || n < 0 || (n > 0 && (!sources || !sizes || !results))) return DGS_ERR_INVALID_ARGUMENT;
  g->err.clear();
  g->used_rccl = false;
  if (best_index) *best_index = -1;
  if (best_score) *best_score = DBL_MAX;
  if (n == 0) return DGS_OK;
  const int G = (int)g->members.size();
  const int per = (n + G - 1) / G;
  // ---- deal: candidate c -> member c mod G; every member registers its share as one batch on its own thread / stream
  std::vector<std::vector<const float*>> src(G);
  std::vector<std::vector<int64_t>> sz(G);
  std::vector<std::vector<float>> gs(G);
  std::vector<std::vector<dgs_result>> res(G);
  std::vector<int> rcs(G, DGS_OK);
  for (int c = 0; c < n; c++) {
    const int k = c % G;
    src[k].push_back(sources[c]);
    sz[k].push_back(sizes[c]);
    if (guesses16) gs[k].insert(gs[k].end(), guesses16 + 16 * (size_t)c, guesses16 + 16 * (size_t)c + 16);
  }
  for (int k = 0; k < G; k++) {
    res[k].resize(src[k].size());
    auto job = [g, k, &src, &sz, &gs, &res, &rcs, guesses16, compute_fitness, fitness_max_range] {
      if (src[k].empty()) return;
      rcs[k] = dgs_align_batch(g->members[k], (int32_t)src[k].size(), src[k].data(), sz[k].data(), 0, guesses16 ? gs[k].data() : nullptr, compute_fitness,
                               fitness_max_range, res[k].data());
    };
    if (g->workers[k]) g->workers[k]->submit(job); else job();
  }
  for (int k = 0; k < G; k++)
    if (g->workers[k]) g->workers[k]->wait();
  // a member that failed as a whole reports its candidates as not converged (the reference skips them, loop_detector.hpp:149)
  for (int k = 0; k < G; k++)
    if (rcs[k] != DGS_OK) {
      g->err = "member " + std::to_string(k) + " (device " + std::to_string(g->devices[k]) + "): " + dgs_last_error(g->members[k]);
      for (size_t j = 0; j < res[k].size(); j++) {
        const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        std::memcpy(res[k][j].final_transformation, guesses16 ? gs[k].data() + 16 * j : ident, sizeof(float) * 16);
        res[k][j].converged = 0; res[k][j].iterations = 0; res[k][j].evaluations = 0; res[k][j].status = rcs[k];
        res[k][j].score = 0.0; res[k][j].fitness = NAN;
      }
    }
  // ---- the exchange step: fixed-size records, all-gathered over RCCL (xGMI) when the group has communicators
  std::vector<Record> all((size_t)G * per);
  for (auto& r : all) { std::memset(&r, 0, sizeof(r)); r.candidate = -1; }
  auto fill = [&](int k, Record* dst) {
    for (size_t j = 0; j < res[k].size(); j++) {
      Record& r = dst[j];
      const dgs_result& a = res[k][j];
      r.fitness = a.fitness; r.score = a.score;
      std::memcpy(r.T, a.final_transformation, sizeof(r.T));
      r.candidate = (int32_t)(k + (int)j * G);
      r.converged = a.converged; r.iterations = a.iterations; r.evaluations = a.evaluations; r.status = a.status;
    }
  };
  bool gathered = false;
  if (!g->comms.empty() && ensure_buffers(g, (size_t)per)) {
    bool ok = true;
    for (int k = 0; k < G && ok; k++) {
      char* hs = static_cast<char*>(g->h_stage[k]);
      std::memset(hs, 0, (size_t)per * kRecordBytes);
      std::vector<Record> mine(per);
      for (auto& r : mine) { std::memset(&r, 0, sizeof(r)); r.candidate = -1; }
      fill(k, mine.data());
      for (int j = 0; j < per; j++) std::memcpy(hs + (size_t)j * kRecordBytes, &mine[j], sizeof(Record));
      ok = hipSetDevice(g->devices[k]) == hipSuccess &&
           hipMemcpyAsync(g->d_send[k], hs, (size_t)per * kRecordBytes, hipMemcpyHostToDevice, g->members[k]->stream) == hipSuccess;
    }
    if (ok) {
      ok = g->rccl.GroupStart() == 0;
      for (int k = 0; k < G && ok; k++)
        ok = g->rccl.AllGather(g->d_send[k], g->d_recv[k], (size_t)per * kRecordBytes, kNcclUint8, g->comms[k], g->members[k]->stream) == 0;
      ok = (g->rccl.GroupEnd() == 0) && ok;
    }
    // member 0 holds everybody's records after the collective: one device->host copy
    if (ok) {
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
  }
  if (!gathered)
    for (int k = 0; k < G; k++) fill(k, all.data() + (size_t)k * per);
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

int32_t dgs_group_last_gather_used_rccl(const dgs_group* g) { return (g && g->used_rccl) ? 1 : 0; }

}  // extern "C"
