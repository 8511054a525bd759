// Factory branches for hdl_graph_slam::select_registration_method (src/hdl_graph_slam/registrations.cpp:22-124).
// Usage inside the reference factory (see INTEGRATION.md for the full patch):
//
//   #include <dgs/registrations_hip.hpp>
//   ...
//   if (auto reg = dgs::select_hip_registration<PointT>(registration_method, pnh)) return reg;
//
// `Params` is anything with `template <class T> T param(const std::string&, const T&)` -- ros::NodeHandle in the nodelets.
#pragma once

#include <iostream>
#include <string>

#include "hip_registration.hpp"

namespace dgs {

template <typename PointT, typename Params>
typename pcl::Registration<PointT, PointT>::Ptr select_hip_registration(const std::string& registration_method, Params& pnh) {
  using Reg = HipRegistration<PointT, PointT>;
  if (registration_method == "FAST_GICP_HIP") {
    std::cout << "registration: FAST_GICP_HIP" << std::endl;
    typename pcl::Registration<PointT, PointT>::Ptr base(new Reg(DGS_METHOD_GICP));
    Reg* gicp = static_cast<Reg*>(base.get());
    gicp->setNumThreads(pnh.template param<int>("reg_num_threads", 0));                                   // registrations.cpp:30
    gicp->setTransformationEpsilon(pnh.template param<double>("reg_transformation_epsilon", 0.01));       // :31
    gicp->setMaximumIterations(pnh.template param<int>("reg_maximum_iterations", 64));                    // :32
    gicp->setMaxCorrespondenceDistance(pnh.template param<double>("reg_max_correspondence_distance", 2.5));  // :33
    gicp->setCorrespondenceRandomness(pnh.template param<int>("reg_correspondence_randomness", 20));      // :34
    return base;
  }
  if (registration_method == "FAST_VGICP_HIP") {
    std::cout << "registration: FAST_VGICP_HIP" << std::endl;
    typename pcl::Registration<PointT, PointT>::Ptr base(new Reg(DGS_METHOD_VGICP));
    Reg* vgicp = static_cast<Reg*>(base.get());
    vgicp->setNumThreads(pnh.template param<int>("reg_num_threads", 0));                                  // registrations.cpp:51
    vgicp->setResolution(static_cast<float>(pnh.template param<double>("reg_resolution", 1.0)));          // :52
    vgicp->setTransformationEpsilon(pnh.template param<double>("reg_transformation_epsilon", 0.01));      // :53
    vgicp->setMaximumIterations(pnh.template param<int>("reg_maximum_iterations", 64));                   // :54
    vgicp->setCorrespondenceRandomness(pnh.template param<int>("reg_correspondence_randomness", 20));     // :55
    return base;
  }
  if (registration_method == "NDT_HIP") {
    const double ndt_resolution = pnh.template param<double>("reg_resolution", 0.5);                      // :93
    const std::string nn_search_method = pnh.template param<std::string>("reg_nn_search_method", "DIRECT7");  // :103
    std::cout << "registration: NDT_HIP " << nn_search_method << " " << ndt_resolution << std::endl;
    typename pcl::Registration<PointT, PointT>::Ptr base(new Reg(DGS_METHOD_NDT));
    Reg* ndt = static_cast<Reg*>(base.get());
    ndt->setNumThreads(pnh.template param<int>("reg_num_threads", 0));                                    // :102
    ndt->setTransformationEpsilon(pnh.template param<double>("reg_transformation_epsilon", 0.01));        // :110
    ndt->setMaximumIterations(pnh.template param<int>("reg_maximum_iterations", 64));                     // :111
    ndt->setResolution(static_cast<float>(ndt_resolution));                                                // :112
    ndt->setNeighborhoodSearchMethod(nn_search_method == "KDTREE" ? DGS_NDT_KDTREE : nn_search_method == "DIRECT1" ? DGS_NDT_DIRECT1 : DGS_NDT_DIRECT7);  // :113-119
    return base;
  }
  return typename pcl::Registration<PointT, PointT>::Ptr();
}

}  // namespace dgs
