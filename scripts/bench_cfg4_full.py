#!/usr/bin/env python3
"""cfg4 at full size on ONE GPU: 1 target + 256 candidate sources of 65,536 points (16 distinct ray-cast scans re-used with
their own guesses), one LoopDetector.matching per method; candidates/s with resident clouds."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
from delta_graph_slam_amd.registration import Registration
from delta_graph_slam_amd.transforms import transform3Dto2D
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=N, n_points=65536, seed=40, distinct_scans=16)
dev = [torch.from_numpy(s).cuda() for s in sources[:16]]
new = KeyFrame(torch.from_numpy(tgt).cuda(), np.eye(3), 100.0, 100000)
kfs = [KeyFrame(dev[i % 16], transform3Dto2D(np.asarray(g, np.float32)).astype(np.float64), 0.0, i) for i, g in enumerate(guesses)]
for method, kw in (("NDT_OMP", dict(ndt_resolution=1.0)), ("FAST_GICP", dict(gicp_max_correspondence_distance=2.0)), ("FAST_VGICP", dict(vgicp_resolution=1.0))):
    det = LoopDetector({"fitness_score_thresh": 1e9}, Registration(method, **kw), cache_clouds=True)
    det.matching(kfs, new)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); loop = det.matching(kfs, new); ts.append(time.perf_counter() - t0)
    rec = det.last_records
    ok = int((rec[:, 1] > 0.5).sum())
    err = [np.linalg.norm(rec[c, 4:20].reshape(4, 4)[:3, 3] - gts[c][:3, 3]) for c in range(N)]
    print(json.dumps({"method": method, "candidates": N, "ms_per_tick": 1e3 * float(np.median(ts)), "candidates_per_s": N / float(np.median(ts)),
                      "converged": ok, "median_translation_error_vs_truth_m": float(np.median(err)), "best_score": loop.score if loop else None}), flush=True)
