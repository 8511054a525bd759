"""Summarise a rocprofv3 --kernel-trace run of scripts/dbg_knn_profile.py: per-kernel average of the covariance pass's kernels."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dgs::", "")[:48]
    acc.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, d in acc.items():
    if any(k in n for k in ("knn", "cov", "hilbert", "bvh", "sort", "gather")):
        print("%-50s calls %4d  avg %8.2f us  first-16 median %8.2f  last-12 median %8.2f" % (n, len(d), sum(d) / len(d), sorted(d[:16])[len(d[:16]) // 2], sorted(d[-12:])[len(d[-12:]) // 2]))
