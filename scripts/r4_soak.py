#!/usr/bin/env python3
"""Soak of the upstream-order launch structure (ndt_strict_order = 1: item-compacted kernel, speculated Newton steps, one launch per round)
against the validation order (2: index-order sums, no speculation -- every evaluation bit-identical to the CPU restatement's) on the device:
loop batches of other seeds / sizes / resolutions / searches, ragged and tiny sources.  Prints one line per batch; a pair whose transform,
iteration count or convergence flag differs is listed (evaluation counts may differ by the 1e-14 association of the sums: counted)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delta_graph_slam_amd import _lib as L  # noqa: E402
from delta_graph_slam_amd import synth  # noqa: E402
from delta_graph_slam_amd.registration import Registration  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=24)
ap.add_argument("--seed", type=int, default=9000)
ap.add_argument("--only", type=int, default=-1, help="run this batch only (the others are generated and skipped: same random stream)")
ap.add_argument("--oracle", action="store_true", help="with --only: the CPU oracle's iterations / evaluations on that batch instead of the device (no GPU needed)")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
bad = 0
tot = 0
ev_diff = 0
for b in range(a.batches):
    kind = ["loop", "planar", "indoor"][b % 3]
    res = float(rng.choice([0.5, 1.0, 2.0]))
    search = str(rng.choice(["DIRECT7", "DIRECT7", "DIRECT1", "DIRECT26", "KDTREE"]))
    if kind == "loop":
        n = int(rng.choice([4096, 16384, 65536]))
        c = int(rng.choice([3, 8, 17, 32]))
        tgt, sources, guesses, _ = synth.loop_batch(n_candidates=c, n_points=n, seed=int(rng.integers(1, 1 << 30)), distinct_scans=min(c, 8))
        sources = list(sources)
    elif kind == "planar":
        n = int(rng.choice([300, 1024, 2048, 5000, 16384]))
        tgt, src, _ = synth.planar_pair(n=n)
        sources = [src, src[: max(8, n // 2)], src[: max(4, n // 7)], src[:3]]
        guesses = np.stack([synth.make_transform(rng.uniform(-0.2, 0.2, 3), rng.uniform(-0.03, 0.03, 3)).astype(np.float32) for _ in sources])
    else:
        tgt, src, _ = synth.indoor_pair(n=int(rng.choice([20000, 60000])))
        sources = [src, src[::2], src[::5]]
        guesses = np.stack([synth.make_transform(rng.uniform(-0.1, 0.1, 3), rng.uniform(-0.02, 0.02, 3)).astype(np.float32) for _ in sources])
    k = int(rng.integers(0, len(sources)))
    sources[k] = sources[k][: max(1, len(sources[k]) - int(rng.integers(0, 70)))]       # ragged
    if a.only >= 0 and b != a.only:
        continue
    if a.oracle:
        from oracle import oracle as orc
        o = orc.NdtOracle(resolution=res, search_method=search)
        o.set_target(tgt)
        rows = []
        for c, s_ in enumerate(sources):
            o.set_source(s_)
            r_ = o.align(guesses[c])
            rows.append((int(r_["iterations"]), int(r_["evaluations"]), bool(r_["converged"])))
        print(json.dumps({"batch": b, "kind": kind, "resolution": res, "search": search, "oracle_iterations_evaluations_converged": rows}), flush=True)
        continue
    kw = dict(ndt_resolution=res, ndt_search_method=L.NDT_SEARCH[search])
    out = {}
    for order in (1, 2):
        r = Registration("NDT_OMP", ndt_strict_order=order, **kw)
        r.setInputTarget(tgt)
        out[order] = r.align_batch(sources, guesses)
    diffs = []
    for c, (x, y) in enumerate(zip(out[1], out[2])):
        tot += 1
        if not np.array_equal(x["T"], y["T"]) or x["iterations"] != y["iterations"] or x["converged"] != y["converged"]:
            diffs.append((c, int(x["iterations"]), int(y["iterations"]), bool(x["converged"]), bool(y["converged"]), float(np.abs(x["T"] - y["T"]).max())))
        elif x["evaluations"] != y["evaluations"]:
            ev_diff += 1
    bad += len(diffs)
    print(json.dumps({"batch": b, "kind": kind, "resolution": res, "search": search, "pairs": len(sources), "points": int(len(sources[0])),
                      "iterations": [int(x["iterations"]) for x in out[2]], "differ": diffs}), flush=True)
print(json.dumps({"pairs": tot, "pairs_that_differ": bad, "pairs_with_other_evaluation_count_only": ev_diff}))
