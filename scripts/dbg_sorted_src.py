"""Experiment: does spatial (Morton) ordering of the source points speed up ndt_derivatives / nn_fitness?"""
import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from delta_graph_slam_amd import synth, _lib as L
from delta_graph_slam_amd.registration import Registration

def morton_order(xyz, cell=0.25):
    q = np.floor((xyz - xyz.min(0)) / cell).astype(np.uint64)
    def spread(v):
        v = v & 0x1FFFFF
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    code = (spread(q[:,0]) << 2) | (spread(q[:,1]) << 1) | spread(q[:,2])
    return np.argsort(code, kind='stable')

P=32
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=P, n_points=65536, seed=40, distinct_scans=8)
for mode in ('as generated', 'morton sorted', 'shuffled'):
    if mode == 'morton sorted': srcs = [s[morton_order(s[:,:3])] for s in sources]
    elif mode == 'shuffled': srcs = [s[np.random.default_rng(1).permutation(len(s))] for s in sources]
    else: srcs = sources
    dsrc = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in srcs]
    reg = Registration("NDT_OMP", ndt_resolution=1.0)
    reg.setInputTarget(torch.from_numpy(tgt).cuda())
    reg.align_batch(dsrc, guesses)
    reg.profile_enable(True); reg.profile_reset()
    t0=time.perf_counter()
    for _ in range(3): res = reg.align_batch(dsrc, guesses)
    dt=(time.perf_counter()-t0)/3
    d_ms,d_n = reg.profile_get(L.K_NDT_DERIVATIVES); n_ms,n_n = reg.profile_get(L.K_NN_SEARCH); s_ms,s_n=reg.profile_get(L.K_NDT_SOLVE)
    print('%-14s step %.2f ms | deriv %.2f ms/step (%.1f us/launch) | nn %.2f ms/step | solve %.2f' % (mode, dt*1e3, d_ms/3, 1e3*d_ms/d_n, n_ms/3, s_ms/3), 'iters', sum(r['iterations'] for r in res))
