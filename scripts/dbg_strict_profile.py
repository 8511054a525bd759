# upstream-order NDT (ndt_strict_order = 1) on the bench batch, for a rocprofv3 kernel trace: which kernel the 15 ms per step are.
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=32)
d = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in sources]
r = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=1)
r.setInputTarget(torch.from_numpy(tgt).cuda())
for _ in range(3):
    res = r.align_batch(d, guesses)
print('evaluations', sum(x['evaluations'] for x in res))
