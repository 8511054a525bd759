"""GPU scratch: where a group step spends its time (one member)."""
import sys, time, threading, numpy as np
sys.path.insert(0, '.')
import torch
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import RegistrationGroup, Registration
tgt, sources, guesses, _ = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=8)
def t(fn, n=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
which = sys.argv[1]
if which == 'single_thread':
    r = Registration("NDT_OMP", ndt_resolution=1.0)
    cl = [r.make_cloud(s) for s in sources]
    out = []
    def step1():
        r.setInputTarget(tgt); r.align_batch(cl, guesses)
    th = threading.Thread(target=lambda: out.append(t(step1))); th.start(); th.join()
    print('single handle, host target, resident sources, from a side thread:', out[0])
    print('the same from the main thread:', t(step1))
else:
    devs = [0] if which == 'rccl' else [0, 0]
    g = RegistrationGroup("NDT_OMP", devices=devs, ndt_resolution=1.0)
    kf = [g.make_cloud(s, owner=i) for i, s in enumerate(sources)]
    def step():
        g.setInputTarget(tgt); g.align_batch(kf, guesses)
    print('group', devs, 'uses_rccl', g.uses_rccl, 'step', t(step))
