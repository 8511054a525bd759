"""Workload for profiling the k-NN covariance pass alone (rocprofv3 --kernel-trace --stats / --pmc): FastGICP::calculate_covariances
of 65,536-point HDL-64E-shaped clouds and 26k-point VLP-16 frames, repeated."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from delta_graph_slam_amd import synth, _lib as L
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=3, n_points=65536, seed=40, distinct_scans=3)
frames = synth.vlp16_stream(n_frames=3)[0]
reg = Registration("FAST_GICP", gicp_max_correspondence_distance=2.0)
reg.profile_enable(True)
for name, clouds in (("hdl64", [tgt] + list(sources)), ("vlp16", frames)):
    dev = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda() for c in clouds]
    reg.profile_reset()
    n_clouds = 0
    for rep in range(int(os.environ.get("KNN_REPS", "4"))):
        for c in dev:
            reg.setInputTarget(c)
            reg.gicp_covariances("target")      # index build + k-NN + covariances of this cloud
            n_clouds += 1
    ms, n = reg.profile_get(L.K_GICP_COVARIANCE)
    print(name, len(clouds[0]), "covariance pass (k-NN + covariances) avg ms per cloud %.4f over %d clouds" % (ms / max(n, 1), n), flush=True)
