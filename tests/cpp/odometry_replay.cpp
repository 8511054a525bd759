// C++ replay of ScanMatchingOdometryNodelet::matching (/root/reference/apps/scan_matching_odometry_nodelet.cpp:173-270) through the
// drop-in: the registration is held as pcl::Registration<PointT,PointT>::Ptr exactly as the nodelet holds it (:392), built by the
// factory branch (include/dgs/registrations_hip.hpp), and driven with the nodelet's own call sequence -- first frame becomes
// the keyframe (:174-182), then setInputSource (:185), align(prev_trans * msf_delta) (:218), the !hasConverged early-out
// (:222-226), transform thresholding (:231-241), the keyframe switch on translation / angle / time (:249-260).  ROS, tf and the
// down-sampling filter are outside the path (clouds arrive filtered); Eigen's 4x4 products are written out below because the
// test image has no Eigen (tests/stub_pcl carries storage only).
// usage: odometry_replay <method> <clouds.bin> <keyframe_delta_trans> <keyframe_delta_angle> <keyframe_delta_time>
//   clouds.bin: int32 n_frames, then per frame int32 n + n*4 floats; frame k is stamped 0.1 * k s.  One JSON line per frame.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include <dgs/registrations_hip.hpp>

using PointT = pcl::PointXYZ;
using M4 = Eigen::Matrix4f;

struct FakeNodeHandle {
  std::map<std::string, std::string> s;
  std::map<std::string, double> d;
  template <typename T>
  T param(const std::string& name, const T& def) const;
};
template <> std::string FakeNodeHandle::param<std::string>(const std::string& n, const std::string& def) const { auto it = s.find(n); return it == s.end() ? def : it->second; }
template <> double FakeNodeHandle::param<double>(const std::string& n, const double& def) const { auto it = d.find(n); return it == d.end() ? def : it->second; }
template <> int FakeNodeHandle::param<int>(const std::string& n, const int& def) const { auto it = d.find(n); return it == d.end() ? def : (int)it->second; }

static M4 mul(const M4& a, const M4& b) {
  M4 c;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float v = 0.f;
      for (int k = 0; k < 4; k++) v += a(i, k) * b(k, j);
      c(i, j) = v;
    }
  return c;
}
// inverse of a rigid transform [R t; 0 1] = [R^T  -R^T t; 0 1]  (prev_trans.inverse() at :232 is applied to rigid transforms only)
static M4 rigid_inverse(const M4& a) {
  M4 c = M4::Identity();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) c(i, j) = a(j, i);
  for (int i = 0; i < 3; i++) c(i, 3) = -(c(i, 0) * a(0, 3) + c(i, 1) * a(1, 3) + c(i, 2) * a(2, 3));
  return c;
}
static float trans_norm(const M4& a) { return std::sqrt(a(0, 3) * a(0, 3) + a(1, 3) * a(1, 3) + a(2, 3) * a(2, 3)); }
// w of Eigen::Quaternionf(R) (Shepperd's form, as Eigen's QuaternionBase::operator=(rotation matrix))
static float quat_w(const M4& a) {
  const float t = a(0, 0) + a(1, 1) + a(2, 2);
  if (t > 0.f) return 0.5f * std::sqrt(t + 1.0f);
  int i = 0;
  if (a(1, 1) > a(0, 0)) i = 1;
  if (a(2, 2) > a(i, i)) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  const float s = std::sqrt(a(i, i) - a(j, j) - a(k, k) + 1.0f);
  return (a(k, j) - a(j, k)) * 0.5f / s;
}
static float clamped_acos(float w) { return std::acos(w < -1.f ? -1.f : (w > 1.f ? 1.f : w)); }

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  FakeNodeHandle pnh;
  pnh.s["registration_method"] = argv[1];
  pnh.d["reg_resolution"] = 1.0;
  pnh.d["reg_max_correspondence_distance"] = 2.0;
  pnh.d["reg_transformation_epsilon"] = 0.1;       // launch/delta_graph_slam.launch:60-69
  const double keyframe_delta_trans = std::atof(argv[3]), keyframe_delta_angle = std::atof(argv[4]), keyframe_delta_time = std::atof(argv[5]);
  const bool transform_thresholding = false;
  const double max_acceptable_trans = 1.0, max_acceptable_angle = 1.0;
  pcl::Registration<PointT, PointT>::Ptr registration =
      dgs::select_hip_registration<PointT>(pnh.param<std::string>("registration_method", "NDT_OMP"), pnh);   // :105
  if (!registration) return 3;
  FILE* f = std::fopen(argv[2], "rb");
  if (!f) return 4;
  int n_frames = 0;
  if (std::fread(&n_frames, 4, 1, f) != 1) return 4;

  pcl::PointCloud<PointT>::ConstPtr keyframe;
  M4 keyframe_pose = M4::Identity(), prev_trans = M4::Identity();
  double keyframe_stamp = 0.0;
  int n_keyframes = 0;
  pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
  for (int k = 0; k < n_frames; k++) {
    int n = 0;
    if (std::fread(&n, 4, 1, f) != 1) break;
    pcl::PointCloud<PointT>::Ptr cloud(new pcl::PointCloud<PointT>());
    cloud->points.resize(n);
    if (n && std::fread(cloud->points.data(), 16, n, f) != (size_t)n) break;
    cloud->width = n;
    const double stamp = 0.1 * k;
    M4 odom = M4::Identity(), trans = M4::Identity();
    int converged = -1, switched = 0;
    if (!keyframe) {                                                     // :174-182
      prev_trans = M4::Identity();
      keyframe_pose = M4::Identity();
      keyframe_stamp = stamp;
      keyframe = cloud;
      registration->setInputTarget(keyframe);
      n_keyframes = 1;
    } else {
      registration->setInputSource(cloud);                               // :185
      const M4 msf_delta = M4::Identity();                               // no IMU / wheel odometry in the replay (:190-215)
      registration->align(*aligned, mul(prev_trans, msf_delta));         // :218
      converged = registration->hasConverged() ? 1 : 0;
      trans = registration->getFinalTransformation();
      if (!converged) {                                                  // :222-226
        odom = mul(keyframe_pose, prev_trans);
      } else {
        odom = mul(keyframe_pose, trans);                                // :228-229
        bool rejected = false;
        if (transform_thresholding) {                                    // :231-241
          const M4 delta = mul(rigid_inverse(prev_trans), trans);
          if (trans_norm(delta) > max_acceptable_trans || clamped_acos(quat_w(delta)) > max_acceptable_angle) {
            odom = mul(keyframe_pose, prev_trans);
            rejected = true;
          }
        }
        if (!rejected) {
          prev_trans = trans;                                            // :243-244
          const double delta_trans = trans_norm(trans), delta_angle = clamped_acos(quat_w(trans)), delta_time = stamp - keyframe_stamp;
          if (delta_trans > keyframe_delta_trans || delta_angle > keyframe_delta_angle || delta_time > keyframe_delta_time) {   // :249-260
            keyframe = cloud;
            registration->setInputTarget(keyframe);
            keyframe_pose = odom;
            keyframe_stamp = stamp;
            prev_trans = M4::Identity();
            n_keyframes++;
            switched = 1;
          }
        }
      }
    }
    std::printf("{\"frame\": %d, \"converged\": %d, \"keyframe_switch\": %d, \"n_keyframes\": %d, \"T\": [", k, converged, switched, n_keyframes);
    for (int q = 0; q < 16; q++) std::printf("%s%.9g", q ? ", " : "", trans.data()[q]);
    std::printf("], \"odom\": [");
    for (int q = 0; q < 16; q++) std::printf("%s%.9g", q ? ", " : "", odom.data()[q]);
    std::printf("]}\n");
  }
  std::fclose(f);
  return 0;
}
