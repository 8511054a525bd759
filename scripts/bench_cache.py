#!/usr/bin/env python3
"""Loop-closure tick with and without the device-resident keyframe cache.

A tick = LoopDetector.register_shard over the same candidate keyframes (host arrays, as the reference holds them in
KeyFrame::cloud).  Prints ms per tick for: copying calls (upload + index + covariances every tick) vs cached clouds
(second and later ticks).  usage: python scripts/bench_cache.py [--candidates 16] [--points 65536] [--ticks 5]
"""
from __future__ import annotations

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
    ap.add_argument("--candidates", type=int, default=16)
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--ticks", type=int, default=5)
    ap.add_argument("--distinct", type=int, default=4)
    args = ap.parse_args()
    from delta_graph_slam_amd import synth
    from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
    from delta_graph_slam_amd.registration import Registration
    from delta_graph_slam_amd.transforms import transform3Dto2D
    new_cloud, cands, guesses, _ = synth.loop_batch(n_candidates=args.candidates, n_points=args.points, distinct_scans=args.distinct)
    new = KeyFrame(new_cloud, np.eye(3), accum_distance=100.0, id=10_000)
    kfs = [KeyFrame(c.copy(), transform3Dto2D(np.asarray(g, np.float32)).astype(np.float64), float(i), i) for i, (c, g) in enumerate(zip(cands, guesses))]
    for method, kw in (("NDT_OMP", dict(ndt_resolution=1.0)), ("FAST_GICP", dict(gicp_max_correspondence_distance=2.0))):
        out = {"method": method, "candidates": args.candidates, "points": args.points}
        recs = {}
        for label, cache in (("copying", False), ("resident", True)):
            det = LoopDetector({"fitness_score_thresh": 10.0}, Registration(method, **kw), cache_clouds=cache)
            ts = []
            for _ in range(args.ticks + 1):
                t0 = time.perf_counter()
                recs[label] = det.register_shard(kfs, new)
                ts.append(time.perf_counter() - t0)
            out[label + "_first_tick_ms"] = 1e3 * ts[0]
            out[label + "_tick_ms"] = 1e3 * float(np.median(ts[1:]))
        out["identical_records"] = bool(np.array_equal(recs["copying"], recs["resident"], equal_nan=True))
        out["speedup"] = out["copying_tick_ms"] / out["resident_tick_ms"]
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
