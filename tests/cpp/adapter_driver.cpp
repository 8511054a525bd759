// Compile / link / run test of include/dgs/hip_registration.hpp against the PCL-shape stubs (tests/stub_pcl).
// It exercises the adapter exactly the way the reference does: through pcl::Registration<PointT,PointT>::Ptr returned by a
// factory fed with rosparam-style values (registrations.cpp), with the odometry call sequence
// (scan_matching_odometry_nodelet.cpp:180-228) and the loop detector's candidate loop (loop_detector.hpp:124-156).
// usage: adapter_driver <method> <clouds.bin>   (clouds.bin: int32 n_clouds, then per cloud int32 n + n*4 floats; cloud 0 = target)
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include <dgs/registrations_hip.hpp>

using PointT = pcl::PointXYZ;

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
  std::printf("{\"candidates\": [");
  pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
  for (size_t c = 1; c < clouds.size(); c++) {
    registration->setInputSource(clouds[c]);                           // :138
    registration->align(*aligned, Eigen::Matrix4f::Identity());        // :145
    auto* hip = dynamic_cast<dgs::HipRegistration<PointT, PointT>*>(registration.get());
    const double score = hip->getFitnessScore(DBL_MAX);                // :148 (device)
    const double score_pcl = registration->getFitnessScore(DBL_MAX);   // the same call through the base pointer (CPU kd-tree)
    const bool conv = registration->hasConverged();
    const Eigen::Matrix4f T = registration->getFinalTransformation();
    std::printf("%s{\"converged\": %d, \"score\": %.17g, \"score_pcl\": %.17g, \"inliers\": %.17g, \"n_aligned\": %zu, \"T\": [", c > 1 ? ", " : "", conv ? 1 : 0,
                score, score_pcl, hip->getInlierFraction(0.25), aligned->size());
    for (int k = 0; k < 16; k++) std::printf("%s%.9g", k ? ", " : "", T.data()[k]);
    std::printf("], \"error\": \"%s\"}", hip->lastError().c_str());
    if (!conv || score > best_score) continue;                         // :149
    best_score = score;
    best = (int)c;
  }
  std::printf("], \"best\": %d}\n", best);
  return 0;
}
