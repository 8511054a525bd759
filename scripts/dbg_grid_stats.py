#!/usr/bin/env python3
"""Where the one-lane-per-query grid search (nn_grid.hip) spends its work: needs the instrumented build
(make -C delta_graph_slam_amd/csrc dbg; DGS_REG_LIB=delta_graph_slam_amd/libdgs_reg_dbg.so)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from delta_graph_slam_amd import synth  # noqa: E402
from delta_graph_slam_amd.registration import Registration  # noqa: E402
from tests.helpers import f32_transform  # noqa: E402

tgt, sources, guesses, gts = synth.loop_batch(n_candidates=4, n_points=65536, seed=40, distinct_scans=4)
r = Registration("NDT_OMP", ndt_resolution=1.0)
r.setInputTarget(tgt)
for c in range(4):
    q = np.ones_like(sources[c])
    q[:, :3] = f32_transform(gts[c].astype(np.float32), sources[c])
    v = r.nn_fitness_distances(q)
    cell = v[0]
    v = v[1:].astype(np.int64)
    lvl, cells, pts = v & 3, (v >> 2) & 63, v >> 8
    print("scan %d: fine cell %.3f m | resolved at fine %.1f %% coarse %.1f %% L2 %.1f %% | points scanned mean %.1f p50 %d p90 %d p99 %d max %d | "
          "cells scanned mean %.1f | per-wave max points mean %.1f" %
          (c, cell, 100 * (lvl == 0).mean(), 100 * (lvl == 1).mean(), 100 * (lvl == 2).mean(), pts.mean(), np.percentile(pts, 50), np.percentile(pts, 90),
           np.percentile(pts, 99), pts.max(), cells.mean(), pts[: len(pts) // 64 * 64].reshape(-1, 64).max(1).mean()))
    for L in range(4):
        m = lvl == L
        if m.any():
            print("   level %d: %.1f %% of queries, points mean %.1f, cells mean %.1f" % (L, 100 * m.mean(), pts[m].mean(), cells[m].mean()))
