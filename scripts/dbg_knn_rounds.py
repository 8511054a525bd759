# k-NN covariance pass: wave-per-leaf search (DGS_KNN_LEAF=1, default) against the per-query walk with 1 / 4 warm-bound rounds.
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from delta_graph_slam_amd import synth, _lib as L
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, gts = synth.loop_batch(n_candidates=4, n_points=65536, seed=40, distinct_scans=4)
frames = synth.vlp16_stream(n_frames=4)[0]
for name, clouds in (('hdl64 65,536', [tgt] + list(sources)), ('vlp16', [np.ascontiguousarray(f) for f in frames])):
    for leaf, rounds, minw in ((1, 1, 0), (1, 2, 0), (1, 4, 0), (1, 8, 0), (1, 0, 0)):
        os.environ['DGS_KNN_LEAF'] = str(leaf); os.environ['DGS_KNN_PARTS'] = str(rounds) if leaf else '0'; os.environ['DGS_KNN_ROUNDS'] = str(max(rounds, 1)); os.environ['DGS_KNN_MIN_WAVES'] = str(max(minw, 1))
        reg = Registration("FAST_GICP", gicp_max_correspondence_distance=2.0)
        dev = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).cuda() for c in clouds]
        reg.setInputTarget(dev[0]); reg.setInputSource(dev[1]); reg.align(np.eye(4, dtype=np.float32))
        reg.profile_enable(True); reg.profile_reset()
        for c in dev[1:] + dev[1:]:
            reg.setInputTarget(c); reg.setInputSource(dev[0]); reg.align(np.eye(4, dtype=np.float32))
        ms, n = reg.profile_get(L.K_GICP_COVARIANCE)
        print(name, len(clouds[0]), 'leaf', leaf, 'rounds', rounds, 'min waves', minw, 'avg ms per cloud %.3f' % (ms / max(n, 1)), 'launches', n, flush=True)
