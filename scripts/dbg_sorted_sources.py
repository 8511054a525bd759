"""GPU scratch: does the ORDER of the source points matter to the NDT derivative kernel?  The bench step with the candidate clouds as
generated (voxel-filter order), Morton-ordered at several cell sizes, and shuffled."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from delta_graph_slam_amd import synth, _lib as L
from delta_graph_slam_amd.registration import Registration

def morton(p, cell):
    q = np.floor((p[:, :3] - p[:, :3].min(0)) / cell).astype(np.uint64)
    def spread(v):
        v &= 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)

tgt, sources, guesses, _ = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=32)
dt = torch.from_numpy(tgt).cuda()
rng = np.random.default_rng(0)
for name, fn in (("as generated", lambda s: s), ("morton 2.0 m", lambda s: s[np.argsort(morton(s, 2.0), kind="stable")]), ("morton 1.0 m", lambda s: s[np.argsort(morton(s, 1.0), kind="stable")]),
                 ("morton 0.5 m", lambda s: s[np.argsort(morton(s, 0.5), kind="stable")]), ("morton 0.25 m", lambda s: s[np.argsort(morton(s, 0.25), kind="stable")]),
                 ("shuffled", lambda s: s[rng.permutation(len(s))])):
    dev = [torch.from_numpy(np.ascontiguousarray(fn(s))).cuda() for s in sources]
    r = Registration("NDT_OMP", ndt_resolution=1.0)
    r.setInputTarget(dt)
    r.align_batch(dev, guesses)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        r.setInputTarget(dt); res = r.align_batch(dev, guesses)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 30
    r.profile_enable(True); r.profile_reset()
    for _ in range(10):
        r.setInputTarget(dt); r.align_batch(dev, guesses)
    ms, n = r.profile_get(L.K_NDT_DERIVATIVES); msf, nf = r.profile_get(L.K_NN_SEARCH)
    print("%-14s step %.3f ms  derivatives %.3f ms/step (%.2f us per launch)  fitness %.3f ms  evaluations %d" % (name, 1e3 * t, ms / 10, 1e3 * ms / n, msf / 10, sum(x["evaluations"] for x in res)), flush=True)
