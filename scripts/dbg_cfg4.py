import sys, numpy as np
sys.path.insert(0,'.')
np.set_printoptions(linewidth=220, precision=7, suppress=True)
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
from oracle import oracle as orc
from tests.helpers import pose_error
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=6, n_points=16384, seed=77, distinct_scans=3)
reg = Registration("NDT_OMP", ndt_resolution=1.0); reg.setInputTarget(tgt)
o = orc.NdtOracle(resolution=1.0); o.set_target(tgt)
for c in range(6):
    G = guesses[c]
    reg.setInputSource(sources[c]); reg.align(G); tg = reg.ndt_trajectory(0)
    o.set_source(sources[c]); ro = o.align(G); to = ro['trajectory']
    n=min(len(tg),len(to))
    e = pose_error(reg.getFinalTransformation(), ro['T'])
    print(c, 'it', reg.last_result.iterations, ro['iterations'], 'ev', reg.last_result.evaluations, ro['evaluations'], 'err %.2e %.2e'%e)
    if e[0] > 1e-4:
        print(' dp per iter', np.abs(tg[:n]-to[:n]).max(1))
        print(' p0 gpu', tg[0]); print(' p0 cpu', to[0])
        # derivative check at p0 with guess transform
        sg, gg, Hg = reg.ndt_derivatives(to[0], T=G)
        so, go, Ho = o.derivatives(to[0], T=G)
        print(' eval@guess: score', sg, so, 'g rel', np.abs(gg-go).max()/np.abs(go).max(), 'H rel', np.abs(Hg-Ho).max()/np.abs(Ho).max())
        for k in range(1, min(n,4)):
            sg, gg, Hg = reg.ndt_derivatives(to[k]); so, go, Ho = o.derivatives(to[k])
            print(' eval@traj[%d]: score %.6f %.6f g rel %.2e H rel %.2e'%(k, sg, so, np.abs(gg-go).max()/np.abs(go).max(), np.abs(Hg-Ho).max()/np.abs(Ho).max()))
