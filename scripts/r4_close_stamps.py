#!/usr/bin/env python3
"""Where the closing workgroup of a fused UPSTREAM-order NDT launch spends its time (make -C delta_graph_slam_amd/csrc dbg;
DGS_REG_LIB=delta_graph_slam_amd/libdgs_reg_dbg.so).  100 MHz wall-clock stamps of the last closing each pair ran while in its second iteration."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from delta_graph_slam_amd import synth  # noqa: E402
from delta_graph_slam_amd.registration import Registration  # noqa: E402

tgt, sources, guesses, gts = synth.loop_batch(n_candidates=8, n_points=65536, seed=40, distinct_scans=8)
r = Registration("NDT_OMP", ndt_resolution=1.0, maximum_iterations=64, ndt_strict_order=1)
r.setInputTarget(tgt)
r.align_batch(sources, guesses, compute_fitness=False)
rows = []
for c in range(8):
    t = r.ndt_trajectory(c)
    a, b = t[71], t[70]
    # a: [0] close entry, [1] rows summed, [2] state set, [3] advance done, [4] written, [5] ticket seen; b: [0] solve done, [1] before tables, [2] tables done
    rows.append([a[0] - a[5], a[1] - a[0], a[2] - a[1], b[0] - a[2], b[1] - b[0], b[2] - b[1], a[3] - b[2], a[4] - a[3]])
rows = np.array(rows) * 10.0   # 100 MHz ticks -> ns
names = ["ticket->entry", "state load + row sums", "unpack", "6x6 solve (jsvd) [+ line-search logic before it]", "step set-up", "transform + angle tables (trig)", "rest of advance", "write-back"]
for n_, v in zip(names, np.median(rows, 0)):
    print("%-50s %8.0f ns" % (n_, v))
print("%-50s %8.0f ns" % ("total", np.median(rows.sum(1))))
print(np.round(rows).astype(int))
