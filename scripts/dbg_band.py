import sys, numpy as np
sys.path.insert(0,'.')
np.set_printoptions(linewidth=220, precision=6, suppress=True)
from delta_graph_slam_amd import synth, _lib as L
from delta_graph_slam_amd.registration import Registration
from oracle import oracle as orc
from tests.helpers import pose_error
P=32
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=P, n_points=65536, seed=40, distinct_scans=8)
reg = Registration("NDT_OMP", ndt_resolution=1.0)
reg.setInputTarget(tgt)
res = reg.align_batch(sources, guesses, compute_fitness=False)
o1 = orc.NdtOracle(resolution=1.0); o1.set_target(tgt)
o2 = orc.NdtOracle(resolution=1.0, perturbed=True); o2.set_target(tgt)
rows=[]
for c in range(P):
    reg.setInputSource(sources[c]); reg.align(guesses[c]); Ts = reg.getFinalTransformation(); its=reg.last_result.iterations
    o1.set_source(sources[c]); r1=o1.align(guesses[c])
    o2.set_source(sources[c]); r2=o2.align(guesses[c])
    eb=pose_error(res[c]['T'], r1['T']); es=pose_error(Ts, r1['T']); eo=pose_error(r2['T'], r1['T'])
    rows.append((eb[0],es[0],eo[0],eb[1],es[1],eo[1]))
    print(c, 'it', res[c]['iterations'], its, r1['iterations'], r2['iterations'], 'batch %.1e single %.1e band %.1e | rot %.1e %.1e %.1e'%(eb[0],es[0],eo[0],eb[1],es[1],eo[1]))
r=np.array(rows); print('rmse', np.sqrt((r**2).mean(0))); print('max', r.max(0))
