#!/usr/bin/env python3
"""Writes tests/golden/fast_order_gate.json: on each of bench.py's three shards (seeds 40 / 1040 / 2040, 32 x 65,536 points) the pairs on which the
product's FAST NDT order (dgs_params.ndt_strict_order = 0) ends outside north_star's gate (1e-4 m / 1e-5 rad) of the oracle, their errors, and the
oracle's own 34-twin band on each (FMA build, the other exp, the float32 guess moved by +-1 .. +-16 ulps).  Needs an MI355X and the oracle.
tests/test_parity_gate_gpu.py asserts the SET of pairs and ONE times the band."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from delta_graph_slam_amd import synth  # noqa: E402
from delta_graph_slam_amd.registration import Registration  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import pose_error  # noqa: E402

TOL_M, TOL_RAD = 1e-4, 1e-5
TWINS = ((True, 0, 0), (False, 1, 0)) + tuple((False, 0, k) for k in range(-16, 17) if k)
out = {"generator": "scripts/make_fast_order_gate.py on an MI355X box: the product's FAST NDT order (library expf, N accumulated directly) against the oracle with its "
                    "defaults (Eigen's JacobiSVD sequence, PCL's double computeHessian, the polar-factor guess, glibc's expf) on bench.py's shards; oracle_band = the "
                    "oracle's own 34-twin band (FMA build, the other exp, the float32 guess moved by +-1 .. +-16 ulps)",
       "tolerance": [TOL_M, TOL_RAD], "shards": {}}
for seed in (40, 1040, 2040):
    tgt, sources, guesses, _ = synth.loop_batch(n_candidates=32, n_points=65536, seed=seed, distinct_scans=32)
    o = orc.NdtOracle(resolution=1.0)
    o.set_target(tgt)
    To = []
    for c in range(32):
        o.set_source(sources[c])
        To.append(o.align(guesses[c])["T"])
    r = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=0, **({"lib_path": os.environ["DGS_GATE_LIB"]} if os.environ.get("DGS_GATE_LIB") else {}))
    r.setInputTarget(tgt)
    res = r.align_batch(list(sources), guesses)
    err = np.array([pose_error(res[c]["T"], To[c]) for c in range(32)])
    outside = [int(c) for c in np.nonzero((err[:, 0] > TOL_M) | (err[:, 1] > TOL_RAD))[0]]
    shard = {"pairs_inside": 32 - len(outside), "outside": {}}
    for c in outside:
        if os.environ.get("DGS_GATE_NO_BANDS"):
            print(seed, c, err[c], flush=True)
            continue
        _, bt, br = orc.ndt_band(tgt, sources[c], guesses[c], twins=TWINS, resolution=1.0)
        shard["outside"][str(c)] = {"error_m_rad": [float(err[c, 0]), float(err[c, 1])], "oracle_band_m_rad": [float(bt), float(br)]}
        print(seed, c, err[c], bt, br, flush=True)
    out["shards"][str(seed)] = shard
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "fast_order_gate.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
