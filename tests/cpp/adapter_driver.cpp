// Compile / link / run test of include/dgs/hip_registration.hpp against the PCL-shape stubs (tests/stub_pcl).
// It exercises the adapter exactly the way the reference does: through pcl::Registration<PointT,PointT>::Ptr returned by a
// factory fed with rosparam-style values (registrations.cpp), with the odometry call sequence
// (scan_matching_odometry_nodelet.cpp:180-228) and the loop detector's candidate loop (loop_detector.hpp:124-156).
// usage: adapter_driver <method> <clouds.bin> [--devices 0,1,...]
//   clouds.bin: int32 n_clouds, then per cloud int32 n + n*4 floats; cloud 0 = target.
//   --devices: additionally run the candidate loop as ONE dgs_group_align_batch over the listed devices (single process, one
//   host thread + stream per device, RCCL all-gather of the records) and print its records.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include <dgs/registrations_hip.hpp>

using PointT = pcl::PointXYZ;

// the JSON answer is collected and written as ONE last line: libraries underneath (RCCL at communicator set-up) may print too
static std::string g_json;
static void jprintf(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_json += buf;
}

struct FakeNodeHandle {  // stands in for ros::NodeHandle::param<T>(name, default)
  std::map<std::string, std::string> s;
  std::map<std::string, double> d;
  template <typename T>
  T param(const std::string& name, const T& def) const;
};
template <> std::string FakeNodeHandle::param<std::string>(const std::string& n, const std::string& def) const { auto it = s.find(n); return it == s.end() ? def : it->second; }
template <> double FakeNodeHandle::param<double>(const std::string& n, const double& def) const { auto it = d.find(n); return it == d.end() ? def : it->second; }
template <> int FakeNodeHandle::param<int>(const std::string& n, const int& def) const { auto it = d.find(n); return it == d.end() ? def : (int)it->second; }

static std::vector<pcl::PointCloud<PointT>::Ptr> read_clouds(const char* path) {
  std::vector<pcl::PointCloud<PointT>::Ptr> out;
  FILE* f = std::fopen(path, "rb");
  if (!f) return out;
  int n_clouds = 0;
  if (std::fread(&n_clouds, 4, 1, f) != 1) n_clouds = 0;
  for (int c = 0; c < n_clouds; c++) {
    int n = 0;
    if (std::fread(&n, 4, 1, f) != 1) break;
    pcl::PointCloud<PointT>::Ptr cloud(new pcl::PointCloud<PointT>());
    cloud->points.resize(n);
    if (n && std::fread(cloud->points.data(), 16, n, f) != (size_t)n) break;
    cloud->width = n;
    out.push_back(cloud);
  }
  std::fclose(f);
  return out;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FakeNodeHandle pnh;
  pnh.s["registration_method"] = argv[1];
  pnh.d["reg_resolution"] = 1.0;
  pnh.d["reg_max_correspondence_distance"] = 2.0;
  const std::string method = pnh.param<std::string>("registration_method", "NDT_OMP");
  pcl::Registration<PointT, PointT>::Ptr registration = dgs::select_hip_registration<PointT>(method, pnh);
  if (!registration) { std::printf("{\"error\": \"unknown method\"}\n"); return 3; }
  auto clouds = read_clouds(argv[2]);
  if (clouds.size() < 2) { std::printf("{\"error\": \"no clouds\"}\n"); return 4; }

  registration->setInputTarget(clouds[0]);                              // loop_detector.hpp:124 / smo:180
  double best_score = DBL_MAX;
  int best = -1;
  jprintf("{\"candidates\": [");
  pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
  for (size_t c = 1; c < clouds.size(); c++) {
    registration->setInputSource(clouds[c]);                           // :138
    registration->align(*aligned, Eigen::Matrix4f::Identity());        // :145
    auto* hip = dynamic_cast<dgs::HipRegistration<PointT, PointT>*>(registration.get());
    const double score = hip->getFitnessScore(DBL_MAX);                // :148 (device)
    const double score_pcl = registration->getFitnessScore(DBL_MAX);   // the same call through the base pointer (CPU kd-tree)
    const bool conv = registration->hasConverged();
    const Eigen::Matrix4f T = registration->getFinalTransformation();
    jprintf("%s{\"converged\": %d, \"score\": %.17g, \"score_pcl\": %.17g, \"inliers\": %.17g, \"n_aligned\": %zu, \"T\": [", c > 1 ? ", " : "", conv ? 1 : 0,
                score, score_pcl, hip->getInlierFraction(0.25), aligned->size());
    for (int k = 0; k < 16; k++) jprintf("%s%.9g", k ? ", " : "", T.data()[k]);
    jprintf("], \"error\": \"%s\"}", hip->lastError().c_str());
    if (!conv || score > best_score) continue;                         // :149
    best_score = score;
    best = (int)c;
  }
  jprintf("], \"best\": %d", best);
  {
    // setKeepPclTree(false): the base kd-tree is parked on a sentinel, so the base-pointer score is "none" (DBL_MAX), never stale
    auto* hip = dynamic_cast<dgs::HipRegistration<PointT, PointT>*>(registration.get());
    hip->setKeepPclTree(false);
    registration->setInputTarget(clouds[0]);
    registration->setInputSource(clouds[1]);
    registration->align(*aligned, Eigen::Matrix4f::Identity());
    jprintf(", \"parked_base_score\": %.17g, \"parked_device_score\": %.17g", registration->getFitnessScore(DBL_MAX), hip->getFitnessScore(DBL_MAX));
  }
  // ---- the same candidate loop, sharded over several devices of this process (what the nodelet links for loop closure)
  std::vector<int32_t> devices;
  for (int a = 3; a + 1 < argc; a++)
    if (std::string(argv[a]) == "--devices")
      for (const char* p = argv[a + 1]; *p;) {
        devices.push_back((int32_t)std::strtol(p, const_cast<char**>(&p), 10));
        if (*p == ',') p++;
      }
  if (!devices.empty()) {
    dgs_params prm;
    dgs_params_init(&prm, method == "NDT_HIP" ? DGS_METHOD_NDT : method == "FAST_GICP_HIP" ? DGS_METHOD_GICP : DGS_METHOD_VGICP);
    prm.ndt_resolution = prm.vgicp_resolution = pnh.param<double>("reg_resolution", 0.5);
    prm.gicp_max_correspondence_distance = pnh.param<double>("reg_max_correspondence_distance", 2.5);
    dgs_group* group = nullptr;
    const int rc = dgs_group_create(&prm, devices.data(), (int32_t)devices.size(), &group);
    jprintf(", \"group\": {\"create\": %d", rc);
    if (rc == DGS_OK) {
      const int n = (int)clouds.size() - 1;
      std::vector<const float*> src(n);
      std::vector<int64_t> sizes(n);
      for (int c = 0; c < n; c++) {
        src[c] = reinterpret_cast<const float*>(clouds[c + 1]->points.data());
        sizes[c] = (int64_t)clouds[c + 1]->points.size();
      }
      std::vector<dgs_result> res(n);
      int32_t bi = -1;
      double bs = 0;
      int rt = dgs_group_set_input_target(group, reinterpret_cast<const float*>(clouds[0]->points.data()), (int64_t)clouds[0]->points.size());  // :124
      int ra = rt == DGS_OK ? dgs_group_align_batch(group, n, src.data(), sizes.data(), nullptr, 1, DBL_MAX, res.data(), &bi, &bs) : rt;     // :137-156
      jprintf(", \"status\": %d, \"size\": %d, \"rccl\": %d, \"best\": %d, \"candidates\": [", ra, dgs_group_size(group), dgs_group_last_gather_used_rccl(group),
                  bi >= 0 ? bi + 1 : -1);
      for (int c = 0; c < n && ra == DGS_OK; c++) {
        jprintf("%s{\"converged\": %d, \"score\": %.17g, \"T\": [", c ? ", " : "", res[c].converged, res[c].fitness);
        for (int k = 0; k < 16; k++) jprintf("%s%.9g", k ? ", " : "", res[c].final_transformation[k]);
        jprintf("]}");
      }
      jprintf("]");
      dgs_group_destroy(group);
    }
    jprintf("}");
  }
  jprintf("}");
  std::printf("\n%s\n", g_json.c_str());
  return 0;
}
