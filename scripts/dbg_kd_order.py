"""Fitness pass of a 32-candidate NDT batch over the Hilbert ordered target index (DGS_NN_KD=0) and over the k-d ordered one
(default), kernel time from the handle's profiler.  (Round 2 first injected a host-computed k-d order to measure the idea:
0.875 -> 0.449 ms; the build on the device gives the same kernel time.)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from delta_graph_slam_amd import _lib as L, synth
from delta_graph_slam_amd.registration import Registration
import torch

tgt, sources, guesses, gts = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=8)
d = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in sources]
G = [g.astype(np.float32) for g in gts]
for label, kd in (('hilbert', '0'), ('k-d', '1')):
    os.environ['DGS_NN_KD'] = kd
    reg = Registration("NDT_OMP", ndt_resolution=1.0, maximum_iterations=0)
    t = torch.from_numpy(tgt).cuda()
    reg.setInputTarget(t)
    reg.align_batch(d, G)
    reg.profile_enable(True)
    reg.profile_reset()
    for _ in range(5):
        reg.setInputTarget(t)          # a new target every tick, as in the loop detector: the index is rebuilt
        res = reg.align_batch(d, G)
    ms, n = reg.profile_get(L.K_NN_SEARCH)
    print(label, 'fitness kernel ms/call %.4f' % (ms / n), 'mean fitness %.9f' % np.mean([r['fitness'] for r in res]), flush=True)
