"""GPU scratch: candidate 2 of the seed-77 6 x 16,384 batch (tests/test_fullsize_gpu.py) -- where does the fast order leave the oracle?"""
import sys, numpy as np
sys.path.insert(0, '.')
np.set_printoptions(linewidth=220, precision=7, suppress=True)
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
from oracle import oracle as orc
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=6, n_points=16384, seed=77, distinct_scans=3)
c = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for mode in (0, 1):
    reg = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=mode)
    reg.setInputTarget(tgt)
    reg.setInputSource(sources[c]); reg.align(guesses[c]); tg = reg.ndt_trajectory(0)
    o = orc.NdtOracle(resolution=1.0); o.set_target(tgt); o.set_source(sources[c]); ro = o.align(guesses[c]); to = ro['trajectory']
    n = min(len(tg), len(to))
    print('mode', mode, 'iters', reg.last_result.iterations, ro['iterations'], 'evals', reg.last_result.evaluations, ro['evaluations'], 'err', orc.pose_error(reg.getFinalTransformation(), ro['T']))
    print('  per-iter max|dp|:', np.abs(tg[:n] - to[:n]).max(1))
    print('  gpu end', tg[-1], '\n  cpu end', to[-1], '\n  truth', gts[c][:3, 3])
for tw in ((True, 0, 0), (False, 1, 0), (False, 0, 1), (False, 0, -1), (False, 0, 2), (False, 0, -2), (False, 0, 3), (False, 0, 5), (False, 0, -7)):
    o = orc.NdtOracle(resolution=1.0, perturbed=tw[0], exp_libm=tw[1]); o.set_target(tgt); o.set_source(sources[c])
    r = o.align(orc._ulp_shift(guesses[c], tw[2]))
    print('twin', tw, 'iters', r['iterations'], 'evals', r['evaluations'], 'err vs base', orc.pose_error(r['T'], ro['T']))
print('---- batch of 6')
reg = Registration("NDT_OMP", ndt_resolution=1.0)
reg.setInputTarget(tgt)
res = reg.align_batch(sources, guesses)
o = orc.NdtOracle(resolution=1.0); o.set_target(tgt)
for k in range(6):
    o.set_source(sources[k]); ro = o.align(guesses[k])
    fo = orc.fitness_score(tgt, sources[k], ro['T'])[0]
    tg = reg.ndt_trajectory(k); to = ro['trajectory']; n = min(len(tg), len(to))
    print(k, 'iters', res[k]['iterations'], ro['iterations'], 'evals', res[k]['evaluations'], ro['evaluations'], 'err', orc.pose_error(res[k]['T'], ro['T']), 'fitness', res[k]['fitness'], fo, 'score', res[k]['score'], ro['score'])
    if k == c:
        print('  per-iter max|dp|:', np.abs(tg[:n] - to[:n]).max(1))
        print('  gpu end', tg[-1], '\n  cpu end', to[-1], '\n  truth', gts[k][:3, 3])
