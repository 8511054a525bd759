"""Loop tick with FAST_GICP over 32 resident candidate keyframes (the reference's recommended method, README.md:231) for a
`rocprofv3 --kernel-trace --stats` run: where the 2.5 ms go.  usage: rocprofv3 --kernel-trace --stats ... -- python3 scripts/dbg_gicp_tick_profile.py [method]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
method = sys.argv[1] if len(sys.argv) > 1 else "FAST_GICP"
tgt, sources, guesses, _ = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=32)
dev = torch.device("cuda", 0)
cands = []
for c, G in enumerate(guesses):
    est = np.eye(3); est[:2, :2] = G[:2, :2]; est[:2, 2] = G[:2, 3]
    cands.append(KeyFrame(torch.from_numpy(sources[c]).to(dev), est, 0.0, c + 1))
reg = Registration(method, gicp_max_correspondence_distance=2.0) if method != "NDT_OMP" else Registration(method, ndt_resolution=1.0)
det = LoopDetector({"fitness_score_thresh": 1e9}, registration=reg, cache_clouds=True)
kfs = [KeyFrame(torch.from_numpy(tgt).to(dev), np.eye(3), 100.0, 1000 + k) for k in range(4)]   # a new keyframe per tick: its index + covariances are built inside the tick
for k in range(3):
    det.matching(cands, KeyFrame(kfs[0].cloud, np.eye(3), 100.0, 2000 + k))
torch.cuda.synchronize()
lat = []
for k in range(20):
    kf = KeyFrame(kfs[k % 4].cloud, np.eye(3), 100.0, 3000 + k)
    t0 = time.perf_counter(); det.matching(cands, kf); lat.append(time.perf_counter() - t0)
print(method, "tick ms p50 %.3f" % (1e3 * np.median(lat)), "evaluations", reg.counts().get("evaluations"))
