#!/usr/bin/env python3
"""GPU scratch check of dgs_params.ndt_strict_order (1: upstream operation order, 2: + sequential index-order sums)
against the CPU oracle: voxel table, single evaluations, full aligns on the bench workload.  Prints one JSON line per
experiment.  Usage: python scripts/strict_check.py [--pairs 32] [--points 65536]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from delta_graph_slam_amd import synth  # noqa: E402
from delta_graph_slam_amd.registration import Registration  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import pose_error  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--eps", type=float, default=0.01)
    ap.add_argument("--distinct-scans", type=int, default=8)
    ap.add_argument("--seed", type=int, default=40, help="40 = the bench workload of rank 0; 1040, 2040, ... = the other ranks' shards")
    a = ap.parse_args()
    tgt, sources, guesses, gts = synth.loop_batch(n_candidates=a.pairs, n_points=a.points, seed=a.seed, distinct_scans=min(a.distinct_scans, a.pairs))
    o = orc.NdtOracle(resolution=1.0, transformation_epsilon=a.eps)
    o.set_target(tgt)
    vo = o.voxels()
    regs = {m: Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=m, transformation_epsilon=a.eps) for m in (0, 1, 2)}
    for m, r in regs.items():
        r.setInputTarget(tgt)
    vg = regs[2].ndt_voxels()
    ok = vo["valid"]
    print(json.dumps({"voxels": int(len(vo["keys"])), "keys_equal": bool(np.array_equal(vo["keys"], vg["keys"])),
                      "valid_equal": bool(np.array_equal(vo["valid"], vg["valid"])),
                      "mean_bit_equal": bool(np.array_equal(vo["mean"][ok], vg["mean"][ok])),
                      "icov_bit_equal_voxels": int(np.all(vo["icov"][ok] == vg["icov"][ok], axis=(1, 2)).sum()), "valid_voxels": int(ok.sum())}))
    # single evaluations along the oracle's trajectory of pair 0
    o.set_source(sources[0])
    ro = o.align(guesses[0])
    for m, r in regs.items():
        r.setInputSource(sources[0])
        worst = [0.0, 0.0, 0.0]
        exact = 0
        for p in ro["trajectory"][:6]:
            so, go, Ho = o.derivatives(p)
            sg, gg, Hg = r.ndt_derivatives(p)
            worst[0] = max(worst[0], abs(so - sg) / abs(so))
            worst[1] = max(worst[1], np.abs(go - gg).max() / np.abs(go).max())
            worst[2] = max(worst[2], np.abs(Ho - Hg).max() / np.abs(Ho).max())
            exact += int(so == sg and np.array_equal(go, gg) and np.array_equal(Ho, Hg))
        print(json.dumps({"mode": m, "eval_rel_err_score_grad_hess": worst, "bit_exact_evaluations": exact, "of": 6}))
    # full batch aligns
    t0 = time.perf_counter()
    To = []
    for c in range(a.pairs):
        o.set_source(sources[c])
        To.append(o.align(guesses[c]))
    t_cpu = time.perf_counter() - t0
    for m, r in regs.items():
        r.align_batch(sources, guesses)
        t0 = time.perf_counter()
        res = r.align_batch(sources, guesses)
        dt = time.perf_counter() - t0
        err = np.array([pose_error(res[c]["T"], To[c]["T"]) for c in range(a.pairs)])
        same_it = sum(int(res[c]["iterations"] == To[c]["iterations"] and res[c]["evaluations"] == To[c]["evaluations"]) for c in range(a.pairs))
        bit = sum(int(np.array_equal(res[c]["T"], To[c]["T"])) for c in range(a.pairs))
        within = int(((err[:, 0] <= 1e-4) & (err[:, 1] <= 1e-5)).sum())
        print(json.dumps({"mode": m, "pairs": a.pairs, "eps": a.eps, "ms_batch": 1e3 * dt, "cpu_s": t_cpu, "within_tol": within, "bit_equal_T": bit,
                          "same_iterations_and_evaluations": same_it, "max_dt": float(err[:, 0].max()), "max_dr": float(err[:, 1].max()),
                          "rms_dt": float(np.sqrt(np.mean(err[:, 0] ** 2))), "rms_dr": float(np.sqrt(np.mean(err[:, 1] ** 2)))}))


if __name__ == "__main__":
    main()
