"""GPU scratch: the fast order on the three bench shards -- pairs outside the 1e-4 m / 1e-5 rad gate, with the oracle's band from 6 and
from 34 twins (calibration of tests/test_parity_gate_gpu.py)."""
import sys, time, json, numpy as np
sys.path.insert(0, '.')
from delta_graph_slam_amd.registration import Registration
from oracle import oracle as orc
from tests.helpers import oracle_shard, sequential_best
wide = [(True, 0, 0), (False, 1, 0)] + [(False, 0, k) for k in range(-16, 17) if k]
for seed in (40, 1040, 2040):
    t0 = time.time()
    tgt, sources, guesses, ref, fit = oracle_shard(orc, seed)
    f = Registration("NDT_OMP", ndt_resolution=1.0); f.setInputTarget(tgt)
    fast = f.align_batch(sources, guesses)
    err = np.array([orc.pose_error(fast[c]["T"], ref[c]["T"]) for c in range(32)])
    ok = (err[:, 0] <= 1e-4) & (err[:, 1] <= 1e-5)
    b_ref = sequential_best([x["converged"] for x in ref], fit); b_gpu = sequential_best([x["converged"] for x in fast], [x["fitness"] for x in fast])
    print(json.dumps({"seed": seed, "inside": int(ok.sum()), "best_ref": b_ref, "best_gpu": b_gpu, "max_fit_rel": float(max(abs(fast[c]["fitness"] - fit[c]) / fit[c] for c in range(32))), "t": time.time() - t0}), flush=True)
    for c in np.nonzero(~ok)[0]:
        _, b6t, b6r = orc.ndt_band(tgt, sources[c], guesses[c], resolution=1.0)
        _, bwt, bwr = orc.ndt_band(tgt, sources[c], guesses[c], twins=wide, resolution=1.0)
        print(json.dumps({"seed": seed, "pair": int(c), "err": [float(err[c, 0]), float(err[c, 1])], "band6": [b6t, b6r], "band34": [bwt, bwr],
                          "fit_rel": abs(fast[c]["fitness"] - fit[c]) / fit[c], "evals": [fast[c]["evaluations"], ref[c]["evaluations"]]}), flush=True)
