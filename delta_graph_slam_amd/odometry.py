"""Frame-to-keyframe scan-matching odometry: the call sequence of ScanMatchingOdometryNodelet::matching over the HIP
registration, without the ROS shell.

Mirrors /root/reference/apps/scan_matching_odometry_nodelet.cpp:
  initialize_params   :64-106   keyframe_delta_trans / _angle / _time, transform_thresholding, max_acceptable_*
  matching            :173-270  first frame -> keyframe + setInputTarget; else setInputSource, align(prev_trans * msf_delta),
                                reject on !hasConverged (:222-226) or implausible jump (:231-241), keyframe switch (:249-260)
  publish_scan_matching_status :309-346  matching_error = getFitnessScore(), inlier_fraction (d^2 < 0.25)
Down-sampling (:83-103,155-165): VOXELGRID runs on the device (pcl::VoxelGrid centroid filter) so a raw scan
uploaded once stays in HBM through align; APPROX_VOXELGRID (pcl::ApproximateVoxelGrid, :90-96) likewise; NONE passes the cloud through.

The frame loop is inherently sequential (frame t's guess is frame t-1's result), so this path does not shard:
"replicas only"  -- one stream per GPU if several robots / bags are processed.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional

import numpy as np

__all__ = ["ScanMatchingOdometry", "ScanMatchingStatus"]


@dataclass
class ScanMatchingStatus:
    """msg/ScanMatchingStatus.msg:1-8 (the fields the registration feeds)"""
    has_converged: bool
    matching_error: float
    inlier_fraction: float
    relative_pose: np.ndarray


def _quat_w(R: np.ndarray) -> float:
    """w of Eigen::Quaternionf(R) (positive branch used for small rotations; general Shepperd form)."""
    t = float(np.trace(R))
    if t > 0:
        return 0.5 * float(np.sqrt(t + 1.0))
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = float(np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0))
    return float((R[k, j] - R[j, k]) * 0.5 / s)


class ScanMatchingOdometry:
    def __init__(self, registration=None, params: Optional[dict] = None, device: Optional[int] = None):
        pr = dict(params or {})
        self.keyframe_delta_trans = float(pr.get("keyframe_delta_trans", 0.25))
        self.keyframe_delta_angle = float(pr.get("keyframe_delta_angle", 0.15))
        self.keyframe_delta_time = float(pr.get("keyframe_delta_time", 1.0))
        self.transform_thresholding = bool(pr.get("transform_thresholding", False))
        self.max_acceptable_trans = float(pr.get("max_acceptable_trans", 1.0))
        self.max_acceptable_angle = float(pr.get("max_acceptable_angle", 1.0))
        # the reference defaults to VOXELGRID 0.1 m (:83-84); the mirror defaults to NONE so callers that already filtered are not
        # filtered twice -- pass downsample_method explicitly to reproduce the nodelet
        self.downsample_method = str(pr.get("downsample_method", "NONE"))
        self.downsample_resolution = float(pr.get("downsample_resolution", 0.1))
        if registration is None:
            from .registration import select_registration_method
            registration = select_registration_method(pr, device=device)
        self.registration = registration
        self.keyframe: Any = None
        self.keyframe_pose = np.eye(4, dtype=np.float32)
        self.keyframe_stamp = 0.0
        self.prev_time: Optional[float] = None
        self.prev_trans = np.eye(4, dtype=np.float32)
        self.last_status: Optional[ScanMatchingStatus] = None
        self.n_keyframes = 0

    def downsample(self, cloud):
        """scan_matching_odometry_nodelet.cpp:155-165"""
        if self.downsample_method == "VOXELGRID":
            return self.registration.voxel_grid_filter(cloud, self.downsample_resolution)
        if self.downsample_method == "APPROX_VOXELGRID":
            return self.registration.voxel_grid_filter(cloud, self.downsample_resolution, approximate=True)
        return cloud

    def matching(self, stamp: float, cloud, msf_delta: Optional[np.ndarray] = None, want_status: bool = False) -> np.ndarray:
        """Returns odom (4x4 float32): the pose of this frame in the odometry frame."""
        reg = self.registration
        if self.keyframe is None:
            self.prev_time = None
            self.prev_trans = np.eye(4, dtype=np.float32)
            self.keyframe_pose = np.eye(4, dtype=np.float32)
            self.keyframe_stamp = stamp
            self.keyframe = self.downsample(cloud)
            reg.setInputTarget(self.keyframe)
            self.n_keyframes = 1
            return np.eye(4, dtype=np.float32)

        filtered = self.downsample(cloud)
        reg.setInputSource(filtered)
        delta = np.eye(4, dtype=np.float32) if msf_delta is None else np.asarray(msf_delta, np.float32)
        reg.align((self.prev_trans @ delta).astype(np.float32))

        if want_status:  # publish_scan_matching_status, only when someone listens (:310)
            self.last_status = ScanMatchingStatus(reg.hasConverged(), reg.getFitnessScore(), reg.getInlierFraction(0.5 * 0.5),
                                                  reg.getFinalTransformation())

        if not reg.hasConverged():
            return (self.keyframe_pose @ self.prev_trans).astype(np.float32)       # "ignore this frame"

        trans = reg.getFinalTransformation()
        odom = (self.keyframe_pose @ trans).astype(np.float32)

        if self.transform_thresholding:
            d = np.linalg.inv(self.prev_trans) @ trans
            dx = float(np.linalg.norm(d[:3, 3]))
            da = float(np.arccos(np.clip(_quat_w(d[:3, :3]), -1.0, 1.0)))
            if dx > self.max_acceptable_trans or da > self.max_acceptable_angle:
                return (self.keyframe_pose @ self.prev_trans).astype(np.float32)   # "too large transform"

        self.prev_time = stamp
        self.prev_trans = trans

        delta_trans = float(np.linalg.norm(trans[:3, 3]))
        delta_angle = float(np.arccos(np.clip(_quat_w(trans[:3, :3]), -1.0, 1.0)))
        delta_time = stamp - self.keyframe_stamp
        if delta_trans > self.keyframe_delta_trans or delta_angle > self.keyframe_delta_angle or delta_time > self.keyframe_delta_time:
            self.keyframe = filtered
            reg.setInputTarget(filtered)
            self.keyframe_pose = odom
            self.keyframe_stamp = stamp
            self.prev_time = stamp
            self.prev_trans = np.eye(4, dtype=np.float32)
            self.n_keyframes += 1
        return odom
