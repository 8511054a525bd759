# FAST_GICP loop tick over resident keyframe clouds by number of candidates: k-d ordered target index (built on the main stream at the
# first search) against Hilbert ordered.
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
from delta_graph_slam_amd.registration import Registration
from delta_graph_slam_amd.transforms import transform3Dto2D
new_cloud, cands, guesses, _ = synth.loop_batch(n_candidates=32, n_points=65536, distinct_scans=8)
for n in (2, 4, 8, 12, 16, 32):
    row = {}
    for kd in ('1', '0'):
        os.environ['DGS_NN_KD'] = kd
        os.environ['DGS_GICP_KD_MIN_CANDIDATES'] = '1'
        det = LoopDetector({"fitness_score_thresh": 10.0}, Registration("FAST_GICP", gicp_max_correspondence_distance=2.0), cache_clouds=True)
        kfs = [KeyFrame(c.copy(), transform3Dto2D(np.asarray(g, np.float32)).astype(np.float64), float(i), i) for i, (c, g) in enumerate(zip(cands[:n], guesses[:n]))]
        ts = []
        for k in range(6):
            new = KeyFrame(new_cloud, np.eye(3), accum_distance=100.0, id=10_000 + k)   # a new keyframe every tick: index and covariances rebuilt
            t0 = time.perf_counter()
            det.register_shard(kfs, new)
            ts.append(time.perf_counter() - t0)
        row[kd] = 1e3 * float(np.median(ts[1:]))
    print('candidates', n, 'k-d %.3f ms' % row['1'], 'hilbert %.3f ms' % row['0'], flush=True)
