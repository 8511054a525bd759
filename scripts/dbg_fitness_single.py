# getFitnessScore of ONE 65,536-point pair by launch shape (DGS_NN_BLOCKS): rounds of warm bounds against more waves
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
tgt, src, _ = synth.kitti_pair()
reg = Registration("NDT_OMP", ndt_resolution=1.0)
reg.setInputTarget(torch.from_numpy(tgt).cuda()); reg.setInputSource(torch.from_numpy(src).cuda()); reg.align()
for _ in range(5): f = reg.getFitnessScore()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): f = reg.getFitnessScore()
print('blocks', os.environ.get('DGS_NN_BLOCKS', '8192'), 'fitness ms %.4f' % (1e3 * (time.perf_counter() - t0) / 50), 'score %.9f' % f)
