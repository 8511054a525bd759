"""Does splitting a 32-candidate NDT batch over S handles / host threads (S concurrent streams) beat one batch?"""
import sys, time, numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, '.')
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=8)
dt = torch.from_numpy(tgt).cuda(); ds = [torch.from_numpy(s).cuda() for s in sources]
for S in (1, 2, 3, 4):
    regs = [Registration("NDT_OMP", ndt_resolution=1.0) for _ in range(S)]
    pool = ThreadPoolExecutor(S)
    def work(k):
        regs[k].setInputTarget(dt)
        return regs[k].align_batch(ds[k::S], guesses[k::S], compute_fitness=True)
    def step():
        return list(pool.map(work, range(S)))
    for _ in range(3): step()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); out = step(); ts.append(time.perf_counter() - t0)
    conv = sum(r['converged'] for o in out for r in o)
    print('streams', S, 'ms per 32-candidate step incl. setInputTarget %.3f' % (1e3 * np.median(ts)), 'converged', conv)
