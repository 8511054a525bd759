import sys, numpy as np
sys.path.insert(0,'.')
np.set_printoptions(linewidth=220, precision=6, suppress=True)
from delta_graph_slam_amd import synth, _lib as L
from delta_graph_slam_amd.registration import Registration
from oracle import oracle as orc
from tests.helpers import pose_error
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=8, n_points=65536, seed=40, distinct_scans=8)
reg = Registration("NDT_OMP", ndt_resolution=1.0)
reg.setInputTarget(tgt)
o = orc.NdtOracle(resolution=1.0); o.set_target(tgt)
for c in range(8):
    reg.setInputSource(sources[c]); reg.align(guesses[c]); tg = reg.ndt_trajectory(0)
    o.set_source(sources[c]); ro = o.align(guesses[c]); to = ro['trajectory']
    n = min(len(tg), len(to))
    print(c, 'iters', reg.last_result.iterations, ro['iterations'], 'evals', reg.last_result.evaluations, ro['evaluations'], 'err', pose_error(reg.getFinalTransformation(), ro['T']))
    print('  per-iter max|dp|:', np.abs(tg[:n]-to[:n]).max(1))
