"""Times the batched fitness kernel alone: 32 candidates at their ground-truth poses, NDT with 0 iterations (so the
final transform is the guess).  usage: python scripts/dbg_fitness_batch.py [lib suffix, e.g. _old]"""
import sys, numpy as np
sys.path.insert(0, '.')
from delta_graph_slam_amd import _lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = L.LIB_PATH.replace('libdgs_reg.so', 'libdgs_reg%s.so' % sys.argv[1])
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
import torch
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=8)
reg = Registration("NDT_OMP", ndt_resolution=1.0, maximum_iterations=0)
reg.setInputTarget(torch.from_numpy(tgt).cuda())
rng = np.random.default_rng(1)
for label, srcs in (("scan order", sources), ("shuffled", [s[rng.permutation(s.shape[0])] for s in sources])):
    d = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in srcs]
    G = [g.astype(np.float32) for g in gts]
    reg.align_batch(d, G)
    reg.profile_enable(True)
    reg.profile_reset()
    for _ in range(5):
        res = reg.align_batch(d, G)
    ms, n = reg.profile_get(L.K_NN_SEARCH)
    print(sys.argv[1:] or 'current', label, 'fitness kernel ms/call %.4f' % (ms / n), 'mean fitness %.6f' % np.mean([r['fitness'] for r in res]))
