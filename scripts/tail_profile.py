#!/usr/bin/env python3
"""Workload for the batch-tail table: the bench step (32 candidates x 65,536 points, NDT 1.0 m DIRECT7) run STEPS
times with resident clouds; writes the per-pair evaluation counts of one step to gpurun_out/tail_evals.json.  Run it under
`rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 scripts/tail_profile.py`, then scripts/tail_fit.py <dir>."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
STEPS = int(os.environ.get("TAIL_STEPS", "12"))
tgt, sources, guesses, _ = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=32)
from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
dev = torch.device("cuda", 0)
new_kf = KeyFrame(torch.from_numpy(tgt).to(dev), np.eye(3), 100.0, 0)
cands = []
for c, G in enumerate(guesses):
    est = np.eye(3); est[:2, :2] = G[:2, :2]; est[:2, 2] = G[:2, 3]
    cands.append(KeyFrame(torch.from_numpy(sources[c]).to(dev), est, 0.0, c + 1))
reg = Registration("NDT_OMP", ndt_resolution=1.0)
det = LoopDetector({"fitness_score_thresh": 1e9}, registration=reg)
for _ in range(STEPS):
    det.matching(cands, new_kf)
torch.cuda.synchronize()
reg.setInputTarget(new_kf.cloud)
res = reg.align_batch([k.cloud for k in cands], LoopDetector.guesses_for(new_kf, cands))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"evaluations": [r["evaluations"] for r in res], "steps": STEPS + 1}, open(os.path.join(ROOT, "gpurun_out", "tail_evals.json"), "w"))
