#!/usr/bin/env python3
"""GPU scratch check of round 4's upstream-order path (ndt_strict.h, solve6.h jsvd) against the CPU oracle: single evaluations of the
three kinds, the polar guess, full aligns in orders 1 / 2 with the round-4 switches on and off, and the 32-pair batch with timings.
One JSON line per experiment."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from delta_graph_slam_amd import synth  # noqa: E402
from delta_graph_slam_amd import _lib as L  # noqa: E402
from delta_graph_slam_amd.registration import Registration  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import pose_error  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--seed", type=int, default=40)
    ap.add_argument("--skip-order2-batch", action="store_true")
    a = ap.parse_args()
    sw_new = dict(ndt_newton_solver=1, ndt_hessian_recompute_double=1, ndt_guess_rotation_polar=1)
    sw_old = dict(ndt_newton_solver=0, ndt_hessian_recompute_double=0, ndt_guess_rotation_polar=0)
    o_new = dict(newton_solver=1, hessian_recompute_double=1, guess_rotation_polar=1)
    o_old = dict(newton_solver=0, hessian_recompute_double=0, guess_rotation_polar=0)

    # ---- single evaluations, every search method
    tgt, src, _ = synth.planar_pair(n=16384)
    for search in ("DIRECT7", "DIRECT1", "DIRECT26", "KDTREE"):
        o = orc.NdtOracle(resolution=1.0, search_method=search)
        o.set_target(tgt)
        o.set_source(src)
        regs = {m: Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=m, ndt_search_method=L.NDT_SEARCH[search]) for m in (1, 2)}
        for r in regs.values():
            r.setInputTarget(tgt)
            r.setInputSource(src)
        out = {"search": search}
        for p in ([0.2, -0.05, 0.03, 0.02, -0.03, 0.04], [5.0, 3.0, 0.5, 0.3, -0.2, 1.0]):
            p = np.array(p)
            so, go, Ho = o.derivatives(p)
            Hd = o.hessian_double(p)
            for m, r in regs.items():
                s, g, H = r.ndt_derivatives(p)
                H2 = r.ndt_hessian_double(p)
                out.setdefault(f"order{m}", []).append({"score": abs(s - so) / abs(so), "grad": rel(g, go), "hess": rel(H, Ho), "hess_double": rel(H2, Hd),
                                                        "bit": bool(s == so and np.array_equal(g, go) and np.array_equal(H, Ho)), "bit_hd": bool(np.array_equal(H2, Hd))})
        print(json.dumps(out), flush=True)

    # ---- full aligns, single pairs
    for name, (tg, sr, Tg) in (("cfg1", synth.planar_pair()), ("cfg2", synth.kitti_pair())):
        guess = np.eye(4, dtype=np.float32)
        if name == "cfg2":
            guess = Tg.copy().astype(np.float32)
            guess[0, 3] -= 0.25
            guess[1, 3] += 0.10
        for tag, sw, osw in (("new", sw_new, o_new), ("old", sw_old, o_old)):
            o = orc.NdtOracle(resolution=1.0, **osw)
            o.set_target(tg)
            o.set_source(sr)
            ro = o.align(guess)
            for m in (1, 2):
                for fused in ("1", "0"):
                    os.environ["DGS_NDT_FUSED"] = fused
                    r = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=m, **sw)
                    r.setInputTarget(tg)
                    r.setInputSource(sr)
                    r.align(guess)
                    tg_ = r.ndt_trajectory()
                    n = min(len(tg_), len(ro["trajectory"]))
                    print(json.dumps({"case": name, "switches": tag, "order": m, "fused": fused, "T_equal": bool(np.array_equal(r.getFinalTransformation(), ro["T"])),
                                      "iters": [r.last_result.iterations, ro["iterations"]], "evals": [r.last_result.evaluations, ro["evaluations"]],
                                      "traj_maxdiff": float(np.abs(tg_[:n] - ro["trajectory"][:n]).max()), "traj_first": float(np.abs(tg_[1] - ro["trajectory"][1]).max()) if n > 1 else None,
                                      "p0_diff": float(np.abs(tg_[0] - ro["trajectory"][0]).max())}), flush=True)
    os.environ["DGS_NDT_FUSED"] = "1"

    # ---- the batch
    tgt, sources, guesses, gts = synth.loop_batch(n_candidates=a.pairs, n_points=a.points, seed=a.seed, distinct_scans=a.pairs)
    o = orc.NdtOracle(resolution=1.0)
    o.set_target(tgt)
    t0 = time.perf_counter()
    To = []
    for c in range(a.pairs):
        o.set_source(sources[c])
        To.append(o.align(guesses[c]))
    t_cpu = time.perf_counter() - t0
    print(json.dumps({"oracle_s": t_cpu, "evals": [r["evaluations"] for r in To]}), flush=True)
    for m in (1, 0) + (() if a.skip_order2_batch else (2,)):
        r = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=m)
        r.setInputTarget(tgt)
        r.align_batch(sources, guesses)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            res = r.align_batch(sources, guesses)
            ts.append(time.perf_counter() - t0)
        err = np.array([pose_error(res[c]["T"], To[c]["T"]) for c in range(a.pairs)])
        same_it = sum(int(res[c]["iterations"] == To[c]["iterations"] and res[c]["evaluations"] == To[c]["evaluations"]) for c in range(a.pairs))
        bit = sum(int(np.array_equal(res[c]["T"], To[c]["T"])) for c in range(a.pairs))
        within = int(((err[:, 0] <= 1e-4) & (err[:, 1] <= 1e-5)).sum())
        print(json.dumps({"batch_order": m, "pairs": a.pairs, "ms_batch": [1e3 * t for t in ts], "within_tol": within, "bit_equal_T": bit, "same_iterations_and_evaluations": same_it,
                          "max_dt": float(err[:, 0].max()), "max_dr": float(err[:, 1].max()), "outside": [int(i) for i in np.nonzero(~((err[:, 0] <= 1e-4) & (err[:, 1] <= 1e-5)))[0]]}), flush=True)


if __name__ == "__main__":
    main()
