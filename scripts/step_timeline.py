#!/usr/bin/env python3
"""Timeline of ONE bench step from a rocprofv3 kernel trace: every kernel of the step in start order with its queue, start (us from the
step's first kernel) and duration.  Usage: step_timeline.py <trace dir> [step index, default: the one before last]"""
import csv, glob, os, sys
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?")))
rows.sort()
# a step starts at a voxel build kernel of the NDT target (first kernel of setInputTarget)
starts = [i for i, r in enumerate(rows) if "ndt_init" in r[2]]
i0 = starts[which]
i1 = starts[which + 1] if which + 1 < 0 and which + 1 + len(starts) < len(starts) else len(rows)
t0 = rows[i0][0]
prev_end = {}
for s, e, name, q in rows[i0:i1]:
    print("%9.1f us  +%7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name))
