import sys, time, numpy as np
sys.path.insert(0,'.')
from delta_graph_slam_amd import synth, _lib as L
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=2, n_points=65536, seed=40, distinct_scans=2)
reg = Registration("NDT_OMP", ndt_resolution=1.0)
reg.setInputTarget(tgt)
reg.profile_enable(True)
def timed(q, label):
    reg.nearestKSearch(q[:1024].copy())
    reg.profile_reset()
    for _ in range(3): idx, sq = reg.nearestKSearch(q)
    ms, n = reg.profile_get(L.K_NN_SEARCH)
    print(label, 'ms/call', ms/n, 'mean d', float(np.sqrt(sq).mean()), 'max d', float(np.sqrt(sq).max()))
timed(tgt, 'self')
from tests.helpers import f32_transform
for c in range(2):
    q = np.ones_like(sources[c]); q[:, :3] = f32_transform(gts[c].astype(np.float32), sources[c]); timed(q, 'src%d@gt' % c)
    q = np.ones_like(sources[c]); q[:, :3] = f32_transform(guesses[c], sources[c]); timed(q, 'src%d@guess' % c)
rng = np.random.default_rng(0)
q = np.ones((65536,4),np.float32); q[:,:3] = rng.uniform(-50,50,(65536,3)); timed(q, 'uniform box')
q = tgt.copy(); q[:,:3] += rng.normal(0,0.05,(65536,3)).astype(np.float32); timed(q, 'self+5cm noise')
q = tgt.copy(); q[:,2] += 30; timed(q, 'self shifted 30m up')
