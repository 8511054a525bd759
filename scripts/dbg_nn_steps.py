import sys, numpy as np
sys.path.insert(0,'.')
from delta_graph_slam_amd import _lib as L
L.LIB_PATH = L.LIB_PATH.replace('libdgs_reg.so','libdgs_reg_dbg.so')
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
from tests.helpers import f32_transform
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=1, n_points=65536, seed=40, distinct_scans=1)
reg = Registration("NDT_OMP", ndt_resolution=1.0)
reg.setInputTarget(tgt)
q = np.ones_like(sources[0]); q[:, :3] = f32_transform(gts[0].astype(np.float32), sources[0])
for name,qq in (('self',tgt),('src@gt',q)):
    code, sq = reg.nearestKSearch(qq)
    nodes, leaves = code//1000, code%1000
    print(name,'nodes mean %.1f p50 %d p99 %d max %d | leaves mean %.1f p99 %d max %d'%(nodes.mean(), np.median(nodes), np.percentile(nodes,99), nodes.max(), leaves.mean(), np.percentile(leaves,99), leaves.max()))
