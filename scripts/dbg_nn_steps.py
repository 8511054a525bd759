"""Node / leaf visit counts of the NN traversal (needs `make -C delta_graph_slam_amd/csrc dbg`).  A wave takes 64 consecutive
queries, 8 adjacent ones per round: round 0 is unbounded, rounds 1..7 start from the warm bound of the previous round."""
import sys, numpy as np
sys.path.insert(0, '.')
from delta_graph_slam_amd import _lib as L
L.LIB_PATH = L.LIB_PATH.replace('libdgs_reg.so', 'libdgs_reg_dbg.so')
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
from tests.helpers import f32_transform
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=1, n_points=65536, seed=40, distinct_scans=1)
reg = Registration("NDT_OMP", ndt_resolution=1.0)
reg.setInputTarget(tgt)
q = np.ones_like(sources[0]); q[:, :3] = f32_transform(gts[0].astype(np.float32), sources[0])
rng = np.random.default_rng(0)
shuf = q[rng.permutation(q.shape[0])]
step = np.linalg.norm(np.diff(q[:, :3], axis=0), axis=1)
print('consecutive-point distance in the source: median %.3f m, p90 %.3f m' % (np.median(step), np.percentile(step, 90)))
for name, qq in (('self', tgt), ('src@gt', q), ('src@gt shuffled', shuf)):
    qq = np.tile(qq, (8, 1))      # 8 x 65,536 queries -> the search kernel runs its 8 rounds per wave
    code, sq = reg.nearestKSearch(qq)
    wasted, nodes, leaves = code // 1000000, (code // 1000) % 1000, code % 1000
    print('%-16s revisits that found nothing left: mean %.1f of %.1f node visits' % (name, wasted.mean(), nodes.mean()))
    pos = (np.arange(qq.shape[0]) // 8) % 8
    for label, sel in (('unbounded (round 0)', pos == 0), ('warm (rounds 1-7)', pos > 0)):
        nd, lv = nodes[sel], leaves[sel]
        print('%-16s %-22s nodes mean %.1f p50 %d p99 %d max %d | leaves mean %.1f p99 %d max %d' % (
            name, label, nd.mean(), np.median(nd), np.percentile(nd, 99), nd.max(), lv.mean(), np.percentile(lv, 99), lv.max()))
    # what a wave pays: the slowest of its 8 groups at every run position
    w = nodes[: nodes.size // 64 * 64].reshape(-1, 8, 8)      # [wave, round, group]
    print('%-16s wave cost (sum over rounds of max over groups) / 8 queries: %.1f node steps; mean-based %.1f' % (
        name, w.max(axis=2).sum(axis=1).mean() / 8, nodes.mean()))
