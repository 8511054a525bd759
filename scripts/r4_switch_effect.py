#!/usr/bin/env python3
"""How many of the 96 bench-shard pairs (seeds 40 / 1040 / 2040, 32 x 65,536 points) each of round 4's five oracle switches moves past
north_star's gate (1e-4 m / 1e-5 rad), each switched on ALONE against the all-off oracle of rounds 1-3, and all together.  CPU only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from delta_graph_slam_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import pose_error  # noqa: E402

OFF = dict(newton_solver=0, hessian_recompute_double=0, guess_rotation_polar=0, exp_libm=0, cov_eigensolver=0)
CONFIGS = [("all off (rounds 1-3)", OFF), ("newton_solver", dict(OFF, newton_solver=1)), ("hessian_recompute_double", dict(OFF, hessian_recompute_double=1)),
           ("guess_rotation_polar", dict(OFF, guess_rotation_polar=1)), ("exp_libm (glibc's expf)", dict(OFF, exp_libm=1)), ("cov_eigensolver (Eigen's tridiagonal QR)", dict(OFF, cov_eigensolver=1)), ("all on (round 4 default)", {})]
only = os.environ.get("SWITCHES")
if only:   # e.g. SWITCHES="exp_libm" : the all-off baseline + the named rows
    CONFIGS = [c for c in CONFIGS if c[0].startswith("all off") or any(c[0].startswith(o) for o in only.split(","))]
points = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tot = {n: [0, 0, 0.0] for n, _ in CONFIGS}
for seed in (40, 1040, 2040):
    tgt, sources, guesses, _ = synth.loop_batch(n_candidates=32, n_points=points, seed=seed, distinct_scans=32)
    runs = {}
    for name, kw in CONFIGS:
        o = orc.NdtOracle(resolution=1.0, **kw)
        o.set_target(tgt)
        res = []
        for c in range(32):
            o.set_source(sources[c])
            res.append(o.align(guesses[c]))
        runs[name] = res
    base = runs["all off (rounds 1-3)"]
    for name, _ in CONFIGS:
        e = np.array([pose_error(runs[name][c]["T"], base[c]["T"]) for c in range(32)])
        moved = (e[:, 0] > 1e-4) | (e[:, 1] > 1e-5)
        bit = sum(int(np.array_equal(runs[name][c]["T"], base[c]["T"])) for c in range(32))
        row = {"seed": seed, "switch": name, "pairs_moved_past_the_gate": int(moved.sum()), "pairs": [int(c) for c in np.nonzero(moved)[0]], "bit_equal_to_all_off": bit,
               "max_m": float(e[:, 0].max()), "evaluations": int(sum(r["evaluations"] for r in runs[name])), "hessian_recomputes": int(sum(r["hessian_recomputes"] for r in runs[name]))}
        tot[name][0] += int(moved.sum()); tot[name][1] += bit; tot[name][2] = max(tot[name][2], float(e[:, 0].max()))
        print(json.dumps(row), flush=True)
print(json.dumps({"total_of_96": {n: {"moved_past_the_gate": v[0], "bit_equal_to_all_off": v[1], "max_m": v[2]} for n, v in tot.items()}}), flush=True)
