"""Times the batched fitness kernel alone: 32 candidates at their ground-truth poses, NDT with 0 iterations (so the
final transform is the guess).  usage: python scripts/dbg_fitness_batch.py [lib suffix, e.g. _old]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from delta_graph_slam_amd import _lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = L.LIB_PATH.replace('libdgs_reg.so', 'libdgs_reg%s.so' % sys.argv[1])
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
import torch
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=8)
reg = Registration("NDT_OMP", ndt_resolution=1.0, maximum_iterations=0)
reg.setInputTarget(torch.from_numpy(tgt).cuda())
rng = np.random.default_rng(1)


def hilbert30(P):
    """30-bit Hilbert index of the rows of P (Skilling's transpose), as the device index uses"""
    lo = P.min(0); ext = (P.max(0) - lo).max()
    X = np.clip(((P - lo) * (1023.999 / ext)).astype(np.uint32), 0, 1023).T.copy()
    Q = 1 << 9
    while Q > 1:
        Pm = Q - 1
        for i in range(3):
            hit = (X[i] & Q) != 0
            X[0] = np.where(hit, X[0] ^ Pm, X[0])
            t = np.where(hit, 0, (X[0] ^ X[i]) & Pm)
            X[0] ^= t; X[i] ^= t
        Q >>= 1
    X[1] ^= X[0]; X[2] ^= X[1]
    t = np.zeros_like(X[0]); Q = 1 << 9
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t); Q >>= 1
    X ^= t
    out = np.zeros(len(P), np.uint64)
    for b in range(10):
        for a in range(3):
            out |= ((X[a].astype(np.uint64) >> b) & 1) << (3 * b + 2 - a)
    return out


for label, srcs in (("scan order", sources), ("hilbert order", [s[np.argsort(hilbert30(s[:, :3].astype(np.float64)), kind='stable')] for s in sources]),
                    ("shuffled", [s[rng.permutation(s.shape[0])] for s in sources])):
    d = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in srcs]
    G = [g.astype(np.float32) for g in gts]
    reg.align_batch(d, G)
    reg.profile_enable(True)
    reg.profile_reset()
    for _ in range(5):
        res = reg.align_batch(d, G)
    ms, n = reg.profile_get(L.K_NN_SEARCH)
    print(sys.argv[1:] or 'current', 'coop', os.environ.get('DGS_NN_COOP', '0'), label, 'fitness kernel ms/call %.4f' % (ms / n), 'mean fitness %.6f' % np.mean([r['fitness'] for r in res]))
