#!/usr/bin/env python3
"""Per-launch durations of the NDT iteration kernels in the LAST step of a rocprofv3 --kernel-trace run (sqlite results.db)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if 'voxel_finalize' in r[0]]
seq = rows[idx[-2]:idx[-1]]
t0 = seq[0][1]
out = []
for name, s, e in seq:
    if 'ndt_strict' in name or 'ndt_derivatives' in name:
        tag = 'S3' if 'strict3' in name else ('H' if 'Lb1ELb1E' in name and 'strict' in name else 'F')
        out.append("%s%.0f" % (tag, (e - s) / 1e3))
print(" ".join(out))
it = [(n_, s_, e_) for n_, s_, e_ in seq if 'ndt_strict' in n_ or 'ndt_derivatives' in n_]
gaps = [(it[k + 1][1] - it[k][2]) / 1e3 for k in range(len(it) - 1)]
print("gaps_us between consecutive iteration launches:", " ".join("%.1f" % g for g in gaps))
print("gap sum %.0f us, median %.1f us" % (sum(gaps), sorted(gaps)[len(gaps) // 2]))
others = {}
for n_, s_, e_ in seq:
    if not ('ndt_strict' in n_ or 'ndt_derivatives' in n_):
        k_ = n_.split('(')[0][-40:]
        others[k_] = others.get(k_, 0) + (e_ - s_) / 1e3
print("other kernels in the step (us):", {k_: round(v_, 1) for k_, v_ in sorted(others.items(), key=lambda x: -x[1])[:8]})
print("launches", len(out), "sum_us %.0f" % sum(float(x.lstrip('SFH3')) if not x.startswith('S3') else float(x[2:]) for x in out), "span_us %.0f" % ((seq[-1][2] - t0) / 1e3))
