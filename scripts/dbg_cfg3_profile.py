# cfg3 odometry stream (FAST_GICP) for a rocprofv3 kernel trace: where a frame's time goes.  usage: rocprofv3 --kernel-trace --stats ... -- python3 scripts/dbg_cfg3_profile.py
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
from delta_graph_slam_amd.odometry import ScanMatchingOdometry
clouds, poses = synth.vlp16_stream(n_frames=60)
d = [torch.from_numpy(c).cuda() for c in clouds]
odo = ScanMatchingOdometry(Registration("FAST_GICP", gicp_max_correspondence_distance=2.0, transformation_epsilon=0.1),
                           dict(keyframe_delta_trans=1.0, keyframe_delta_angle=1.0, keyframe_delta_time=1e9))
lat = []
for k, c in enumerate(d):
    t0 = time.perf_counter(); odo.matching(0.1 * k, c); lat.append(time.perf_counter() - t0)
print('frame ms p50 %.3f' % (1e3 * np.median(lat[1:])), 'keyframes', odo.n_keyframes)
