#!/usr/bin/env python3
"""Per-launch table of the fused NDT launches of one bench step from a rocprofv3 kernel trace: launch number, pairs still
iterating (from the per-pair evaluation counts), duration; least-squares fit t = a + b * active.  usage: tail_fit.py <trace dir> [out.json]"""
import csv, glob, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
ev = json.load(open(os.path.join(ROOT, "gpurun_out", "tail_evals.json")))
evals = np.array(ev["evaluations"])
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# split into steps at every voxel_finalize kernel (one per setInputTarget)
steps, cur = [], None
for s, e, name in rows:
    if "voxel_finalize" in name:
        cur = []
        steps.append(cur)
    elif cur is not None and "ndt_derivatives_kernel" in name:
        cur.append((s, e))
steps = [st for st in steps if st][2:]          # drop warm-up steps
n_launch = int(evals.max())
active = np.array([(evals > l).sum() for l in range(n_launch)])
dur = np.array([[ (st[l][1] - st[l][0]) * 1e-3 for l in range(n_launch)] for st in steps if len(st) >= n_launch])
gap = np.array([[ (st[l + 1][0] - st[l][1]) * 1e-3 for l in range(n_launch - 1)] for st in steps if len(st) >= n_launch])
med = np.median(dur, 0)
A = np.stack([np.ones(n_launch), active], 1)
(a, b), *_ = np.linalg.lstsq(A, med, rcond=None)
tail = active <= 4
out = {"steps_used": int(dur.shape[0]), "launches_per_step": n_launch, "launches_enqueued_per_step": int(np.median([len(st) for st in steps])),
       "fit_us": {"a": float(a), "b_per_active_pair": float(b)},
       "sum_us": float(med.sum()), "tail_launches_le4_active": int(tail.sum()), "tail_sum_us": float(med[tail].sum()),
       "median_gap_us": float(np.median(gap)), "sum_gaps_us": float(np.median(gap, 0).sum()),
       "table": [{"launch": int(l), "active_pairs": int(active[l]), "us": round(float(med[l]), 2)} for l in range(n_launch)]}
print(json.dumps(out))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
