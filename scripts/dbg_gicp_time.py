import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from delta_graph_slam_amd import synth, _lib as L
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=8, n_points=65536, seed=40, distinct_scans=8)
reg = Registration("FAST_GICP", gicp_max_correspondence_distance=2.0)
dt = torch.from_numpy(tgt).cuda(); ds=[torch.from_numpy(s).cuda() for s in sources]
reg.setInputTarget(dt)
for c in range(2): reg.setInputSource(ds[c]); reg.align(guesses[c])
reg.profile_enable(True); reg.profile_reset()
t0=time.perf_counter()
its=[]; 
for c in range(8):
    reg.setInputSource(ds[c]); reg.align(guesses[c]); its.append((reg.last_result.iterations, reg.last_result.evaluations, reg.hasConverged())); f=reg.getFitnessScore()
dt_=time.perf_counter()-t0
print('per candidate ms', 1e3*dt_/8, its)
for k,name in ((L.K_GICP_COVARIANCE,'cov'),(L.K_NN_SEARCH,'nn'),(L.K_GICP_LINEARIZE,'lin')):
    ms,n=reg.profile_get(k); print(name, 'total ms %.3f launches %d avg us %.1f'%(ms,n,1e3*ms/max(n,1)))
t0=time.perf_counter(); res=reg.align_batch(ds, guesses); print('batch api ms per cand', 1e3*(time.perf_counter()-t0)/8)
from tests.helpers import pose_error
print([round(pose_error(r['T'], g)[0],3) for r,g in zip(res,gts)])
