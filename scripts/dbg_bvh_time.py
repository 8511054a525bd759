import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=1, n_points=65536, seed=40, distinct_scans=1)
d = torch.from_numpy(tgt).cuda(); q = torch.from_numpy(sources[0]).cuda()
for method in ("FAST_GICP",):
    reg = Registration(method)
    def a():
        reg.setInputTarget(d); torch.cuda.synchronize()
    def b():
        reg.setInputTarget(d); reg.nearestKSearch(sources[0][:8]); torch.cuda.synchronize()
    def c():
        reg.setInputTarget(d); reg.gicp_covariances("target"); torch.cuda.synchronize()
    for name, f in (("setInputTarget (copy only)", a), ("+ index build + 8-query search", b), ("+ index + covariances", c)):
        for _ in range(3): f()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        print(name, "%.3f ms" % (1e3 * np.median(ts)))
