"""GPU scratch (make -C delta_graph_slam_amd/csrc dbg; DGS_REG_LIB=delta_graph_slam_amd/libdgs_reg_dbg.so): queue kernel counters per batch."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, _ = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=8)
import torch
dev = [torch.from_numpy(s).cuda() for s in sources]
r = Registration("NDT_OMP", ndt_resolution=1.0)
r.setInputTarget(torch.from_numpy(tgt).cuda())
for k in range(3):
    t0 = time.perf_counter(); res = r.align_batch(dev, guesses, compute_fitness=False); t1 = time.perf_counter()
    print('batch ms', 1e3 * (t1 - t0), 'evaluations', sum(x['evaluations'] for x in res), flush=True)
