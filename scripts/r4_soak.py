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

def batches(n_batches, seed, vary_params=False):
    """The soak's batches, one after the other: (b, kind, res, search, tgt, sources, guesses, kw) -- a fixed function of (seed, vary_params)."""
    rng = np.random.default_rng(seed)
    for b in range(n_batches):
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
        kw = dict(ndt_resolution=res, ndt_search_method=L.NDT_SEARCH[search])
        if vary_params:   # the optimiser's own switches too: line-search form, iteration cap, stop tolerance, step size
            kw.update(ndt_line_search=int(rng.choice([1, 1, 0])), maximum_iterations=int(rng.choice([1, 3, 10, 64])),
                      transformation_epsilon=float(rng.choice([1e-4, 0.01, 0.1])), ndt_step_size=float(rng.choice([0.1, 0.05, 0.5])))
        yield b, kind, res, search, tgt, sources, guesses, kw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--seed", type=int, default=9000)
    ap.add_argument("--vary-params", action="store_true", help="also draw ndt_line_search / maximum_iterations / transformation_epsilon / ndt_step_size per batch")
    ap.add_argument("--variants", action="store_true", help="with --only: the batch under every launch structure of order 1, against the oracle")
    ap.add_argument("--oracle-on-difference", action="store_true", help="run the CPU oracle on every pair that differs between the orders")
    ap.add_argument("--only", type=str, default="", help="run these batches only, comma-separated (the others are generated and skipped: same random stream)")
    ap.add_argument("--oracle", action="store_true", help="with --only: the CPU oracle's iterations / evaluations on that batch instead of the device (no GPU needed)")
    a = ap.parse_args()
    only = set(int(x) for x in a.only.split(",") if x != "")
    bad = 0
    tot = 0
    ev_diff = 0
    for b, kind, res, search, tgt, sources, guesses, kw in batches(a.batches, a.seed, a.vary_params):
        if only and b not in only:   # (skipped AFTER every random draw of the batch: the stream stays the one of the full run)
            continue
        if a.oracle:
            from oracle import oracle as orc
            o = orc.NdtOracle(resolution=res, search_method=search, line_search=kw.get("ndt_line_search", 1), max_iterations=kw.get("maximum_iterations", 64),
                              transformation_epsilon=kw.get("transformation_epsilon", 0.01), step_size=kw.get("ndt_step_size", 0.1))
            o.set_target(tgt)
            rows = []
            for c, s_ in enumerate(sources):
                o.set_source(s_)
                r_ = o.align(guesses[c])
                rows.append((int(r_["iterations"]), int(r_["evaluations"]), bool(r_["converged"])))
            print(json.dumps({"batch": b, "kind": kind, "resolution": res, "search": search, "oracle_iterations_evaluations_converged": rows}), flush=True)
            continue
        if a.variants:   # one batch under every launch structure of order 1, against order 2 and the oracle
            from oracle import oracle as orc
            okw = dict(resolution=res, search_method=search, line_search=kw.get("ndt_line_search", 1), max_iterations=kw.get("maximum_iterations", 64),
                       transformation_epsilon=kw.get("transformation_epsilon", 0.01), step_size=kw.get("ndt_step_size", 0.1))
            o = orc.NdtOracle(**okw)
            o.set_target(tgt)
            ref = []
            for c, s_ in enumerate(sources):
                o.set_source(s_)
                ref.append(o.align(guesses[c]))
            for name, env, order in (("order 2", {}, 2), ("order 1", {}, 1), ("order 1, no speculation", {"DGS_NDT_SPECULATE": "0"}, 1),
                                     ("order 1, lane-per-point kernels", {"DGS_NDT_STRICT_KERNEL": "2"}, 1), ("order 1, unfused", {"DGS_NDT_FUSED": "0"}, 1)):
                old_env = {k: os.environ.get(k) for k in env}
                os.environ.update(env)
                try:
                    r = Registration("NDT_OMP", ndt_strict_order=order, **kw)
                finally:
                    for k, v in old_env.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                r.setInputTarget(tgt)
                got = r.align_batch(sources, guesses)
                bad = [(c, int(x["iterations"]), int(y["iterations"]), int(x["evaluations"]), int(y["evaluations"]), float(np.abs(x["T"] - y["T"]).max()))
                       for c, (x, y) in enumerate(zip(got, ref)) if not np.array_equal(x["T"], y["T"]) or x["iterations"] != y["iterations"]]
                print(json.dumps({"batch": b, "structure": name, "pairs_that_differ_from_the_oracle (pair, it, oracle it, ev, oracle ev, max |dT|)": bad}), flush=True)
            continue
        out = {}
        for order in (1, 2):
            r = Registration("NDT_OMP", ndt_strict_order=order, **kw)
            r.setInputTarget(tgt)
            out[order] = r.align_batch(sources, guesses)
        diffs = []
        ev_only = []
        for c, (x, y) in enumerate(zip(out[1], out[2])):
            tot += 1
            if not np.array_equal(x["T"], y["T"]) or x["iterations"] != y["iterations"] or x["converged"] != y["converged"]:
                diffs.append((c, int(x["iterations"]), int(y["iterations"]), bool(x["converged"]), bool(y["converged"]), float(np.abs(x["T"] - y["T"]).max())))
            elif x["evaluations"] != y["evaluations"]:
                ev_diff += 1
                ev_only.append((c, int(x["evaluations"]), int(y["evaluations"])))
        if diffs and a.oracle_on_difference:   # which of the two runs is the CPU's?  (the history of the process matters: the batches before this one ran too)
            from oracle import oracle as orc
            o = orc.NdtOracle(resolution=res, search_method=search, line_search=kw.get("ndt_line_search", 1), max_iterations=kw.get("maximum_iterations", 64),
                              transformation_epsilon=kw.get("transformation_epsilon", 0.01), step_size=kw.get("ndt_step_size", 0.1))
            o.set_target(tgt)
            for d_ in diffs:
                c = d_[0]
                o.set_source(sources[c])
                ro = o.align(guesses[c])
                print(json.dumps({"batch": b, "pair": c, "oracle": [int(ro["iterations"]), int(ro["evaluations"])],
                                  "order 1": [int(out[1][c]["iterations"]), int(out[1][c]["evaluations"]), bool(np.array_equal(out[1][c]["T"], ro["T"]))],
                                  "order 2": [int(out[2][c]["iterations"]), int(out[2][c]["evaluations"]), bool(np.array_equal(out[2][c]["T"], ro["T"]))]}), flush=True)
        bad += len(diffs)
        print(json.dumps({"batch": b, "kind": kind, "resolution": res, "search": search, "pairs": len(sources), "points": int(len(sources[0])),
                          "iterations": [int(x["iterations"]) for x in out[2]], "params": {k: v for k, v in kw.items() if k not in ("ndt_resolution", "ndt_search_method")}, "differ": diffs, "other_evaluation_count_only (pair, order 1, order 2)": ev_only}), flush=True)
    print(json.dumps({"pairs": tot, "pairs_that_differ": bad, "pairs_with_other_evaluation_count_only": ev_diff}))



if __name__ == "__main__":
    main()
