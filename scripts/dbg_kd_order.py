"""Experiment: the fitness pass over a k-d ordered (median-split, disjoint boxes) target index instead of the Hilbert order.
The order is computed on the host and injected through the instrumented build (make stats; DGS_BVH_ORDER_FILE)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['DGS_REG_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'delta_graph_slam_amd', 'libdgs_reg_stats.so')
from delta_graph_slam_amd import _lib as L, synth
from delta_graph_slam_amd.registration import Registration
import torch


def kd_order(P, leaf=8):
    n = len(P)
    order = np.arange(n)
    segs = [(0, n)]
    nl = 1
    while nl * leaf < n:
        nl *= 2
    for _ in range(int(np.log2(nl))):
        new = []
        for a, b in segs:
            sub = order[a:b]
            pts = P[sub]
            ax = int(np.argmax(pts.max(0) - pts.min(0)))
            order[a:b] = sub[np.argsort(pts[:, ax], kind='stable')]
            mid = a + (b - a + 1) // 2
            new += [(a, mid), (mid, b)]
        segs = new
    return order


tgt, sources, guesses, gts = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=8)
d = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in sources]
G = [g.astype(np.float32) for g in gts]
path = '/tmp/kd_order.u32'
kd_order(tgt[:, :3].astype(np.float64)).astype(np.uint32).tofile(path)
for label, env in (('hilbert', None), ('k-d', path)):
    if env:
        os.environ['DGS_BVH_ORDER_FILE'] = env
    reg = Registration("NDT_OMP", ndt_resolution=1.0, maximum_iterations=0)
    reg.setInputTarget(torch.from_numpy(tgt).cuda())
    reg.align_batch(d, G)
    reg.profile_enable(True)
    reg.profile_reset()
    for _ in range(5):
        res = reg.align_batch(d, G)
    ms, n = reg.profile_get(L.K_NN_SEARCH)
    print(label, 'fitness kernel ms/call %.4f' % (ms / n), 'mean fitness %.9f' % np.mean([r['fitness'] for r in res]), flush=True)
