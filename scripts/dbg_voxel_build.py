# setInputTarget (NDT voxel table build) on the bench target, for a rocprofv3 kernel trace.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=2, n_points=65536, seed=40)
d = torch.from_numpy(tgt).cuda()
reg = Registration("NDT_OMP", ndt_resolution=1.0)
for _ in range(3): reg.setInputTarget(d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): reg.setInputTarget(d)
torch.cuda.synchronize()
print('setInputTarget ms %.4f' % (1e3 * (time.perf_counter() - t0) / 20))
