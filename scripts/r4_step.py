#!/usr/bin/env python3
"""Bench-shaped step (LoopDetector.matching over 32 resident 65,536-point candidates) in a chosen NDT evaluation order; prints ms per step
and the handle's own per-kernel event timings.  Usage: python scripts/r4_step.py [--order 1] [--steps 20] [--kw name=value ...]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--seed", type=int, default=40)
    ap.add_argument("--profile", action="store_true", help="second pass with the handle's HIP-event profiler on")
    ap.add_argument("--evals", action="store_true", help="also print the evaluations of every pair (one more batch, host clouds)")
    ap.add_argument("--lib", default=None, help="another build of the library (e.g. delta_graph_slam_amd/libdgs_reg_ab.so)")
    ap.add_argument("--kw", nargs="*", default=[])
    a = ap.parse_args()
    import torch
    from delta_graph_slam_amd import _lib as L
    from delta_graph_slam_amd import synth
    from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
    from delta_graph_slam_amd.registration import Registration
    kw = dict(ndt_resolution=1.0, ndt_strict_order=a.order)
    for item in a.kw:
        k, v = item.split("=")
        kw[k] = float(v) if "." in v else int(v)
    dev = torch.device("cuda", 0)
    tgt, sources, guesses, gts = synth.loop_batch(n_candidates=a.pairs, n_points=a.points, seed=a.seed, distinct_scans=a.pairs)
    new_kf = KeyFrame(cloud=torch.from_numpy(tgt).to(dev), estimate=np.eye(3), accum_distance=100.0, id=0)
    cands = []
    for c in range(a.pairs):
        est = np.eye(3)
        est[:2, :2] = guesses[c][:2, :2]
        est[:2, 2] = guesses[c][:2, 3]
        cands.append(KeyFrame(cloud=torch.from_numpy(sources[c]).to(dev), estimate=est, accum_distance=0.0, id=1 + c))
    if a.lib:
        kw["lib_path"] = os.path.join(ROOT, a.lib) if not os.path.isabs(a.lib) else a.lib
    reg = Registration("NDT_OMP", device=0, **kw)
    kw.pop("lib_path", None)
    det = LoopDetector({"fitness_score_thresh": 1e9}, registration=reg)
    for _ in range(a.warmup):
        det.matching(cands, new_kf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        det.matching(cands, new_kf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"order": a.order, "kw": kw, "ms_per_step": 1e3 * dt / a.steps, "registrations_per_s": a.pairs * a.steps / dt, "evaluations": reg.counts()["evaluations"]}
    if a.profile:
        reg.profile_enable(True)
        reg.profile_reset()
        for _ in range(a.steps):
            det.matching(cands, new_kf)
        for name, k in (("ndt_derivatives", L.K_NDT_DERIVATIVES), ("ndt_solve", L.K_NDT_SOLVE), ("nn", L.K_NN_SEARCH), ("voxel", L.K_NDT_VOXEL_BUILD)):
            ms, n = reg.profile_get(k)
            out[name] = {"ms_per_step": ms / a.steps, "launches_per_step": n / a.steps, "us_per_launch": 1e3 * ms / max(n, 1)}
        reg.profile_enable(False)
    if a.evals:   # evaluations per pair: how many pairs are still iterating in launch k of a step
        reg.setInputTarget(tgt)
        res = reg.align_batch(list(sources), guesses)
        ev = sorted(x["evaluations"] for x in res)
        out["evaluations_per_pair_sorted"] = ev
        out["pairs_active_in_launch"] = [sum(e > k for e in ev) for k in range(max(ev))]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
