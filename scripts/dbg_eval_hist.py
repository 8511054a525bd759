"""Distribution of evaluations per pair in the bench workload, and the time of a step without / with the fitness."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from delta_graph_slam_amd import synth, _lib as L
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=int(sys.argv[1]) if len(sys.argv) > 1 else 8)
reg = Registration("NDT_OMP", ndt_resolution=1.0)
dt = torch.from_numpy(tgt).cuda(); ds = [torch.from_numpy(s).cuda() for s in sources]
reg.setInputTarget(dt)
for fit in (False, True):
    reg.align_batch(ds, guesses, compute_fitness=fit)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); res = reg.align_batch(ds, guesses, compute_fitness=fit); ts.append(time.perf_counter() - t0)
    print('fitness' if fit else 'align only', 'ms per batch %.3f' % (1e3 * np.median(ts)))
ev = sorted(r['evaluations'] for r in res)
print('evaluations per pair (sorted):', ev, 'mean %.1f max %d' % (np.mean(ev), max(ev)))
reg.profile_enable(True); reg.profile_reset(); reg.align_batch(ds, guesses, compute_fitness=True)
for k, name in ((L.K_NDT_DERIVATIVES, 'deriv'), (L.K_NDT_SOLVE, 'solve'), (L.K_NN_SEARCH, 'fitness')):
    ms, n = reg.profile_get(k); print(name, 'total ms %.3f launches %d' % (ms, n))
