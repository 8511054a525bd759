#!/usr/bin/env python3
"""dgs_find_loop_candidates (loop_detector.hpp:83-111 on the device) against the same two tests vectorised on the host, per keyframe count:
where the device call's fixed cost (two small uploads, one 1024-thread workgroup, one synchronisation) is paid back."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delta_graph_slam_amd.registration import Registration  # noqa: E402

r = Registration("NDT_OMP")
rng = np.random.default_rng(5)
for n in (100, 300, 1000, 10000, 100000, 1000000):
    acc = np.sort(rng.uniform(0, 0.5 * n, n))
    xy = rng.uniform(-50, 50, (n, 2))
    new_acc, new_xy = acc[-1] + 1.0, np.array([1.0, 2.0])
    ref = np.nonzero(~(new_acc - acc < 8.0) & ~(np.sqrt(((xy - new_xy) ** 2).sum(1)) > 5.0))[0]
    got = r.find_loop_candidates(acc, xy, new_acc, new_xy, 8.0, 5.0)
    assert np.array_equal(got, ref)
    reps = 200 if n <= 10000 else 20
    t0 = time.perf_counter()
    for _ in range(reps):
        r.find_loop_candidates(acc, xy, new_acc, new_xy, 8.0, 5.0)
    t_dev = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        np.nonzero(~(new_acc - acc < 8.0) & ~(np.sqrt(((xy - new_xy) ** 2).sum(1)) > 5.0))[0]
    t_host = (time.perf_counter() - t0) / reps
    print(json.dumps({"keyframes": n, "candidates": int(len(ref)), "device_call_us": 1e6 * t_dev, "host_numpy_us": 1e6 * t_host}), flush=True)
