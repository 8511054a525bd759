"""Host side of one bench step (LoopDetector.matching over 32 resident candidates): cProfile over 300 steps -- where the ~0.14 ms per
step that is not kernel time goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
tgt, sources, guesses, _ = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=32)
dev = torch.device("cuda", 0)
new_kf = KeyFrame(torch.from_numpy(tgt).to(dev), np.eye(3), 100.0, 0)
cands = []
for c, G in enumerate(guesses):
    est = np.eye(3); est[:2, :2] = G[:2, :2]; est[:2, 2] = G[:2, 3]
    cands.append(KeyFrame(torch.from_numpy(sources[c]).to(dev), est, 0.0, c + 1))
reg = Registration("NDT_OMP", ndt_resolution=1.0)
det = LoopDetector({"fitness_score_thresh": 1e9}, registration=reg)
for _ in range(20):
    det.matching(cands, new_kf)
t0 = time.perf_counter()
for _ in range(300):
    det.matching(cands, new_kf)
print("ms per step %.4f" % ((time.perf_counter() - t0) / 300 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    det.matching(cands, new_kf)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
