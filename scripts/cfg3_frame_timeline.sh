cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/c3
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3 -- python3 $GRAFT_REPO_ROOT/scripts/dbg_cfg3_profile.py > $GRAFT_REPO_ROOT/gpurun_out/c3.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/c3 -name "*kernel_stats.csv" | head -n 1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-58s calls %6s total %9.1f us avg %8.2f us  %5s%%" % (r["Name"].split("(")[0][:58], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
python scripts/step_timeline.py gpurun_out/c3 2>/dev/null | head -n 0
tail -n 1 gpurun_out/c3.log
python - <<'PY'
# timeline of one frame (the 30th source BVH build onwards)
import csv,glob
f=sorted(glob.glob("gpurun_out/c3/**/*kernel_trace.csv",recursive=True))[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:56]) for r in csv.DictReader(open(f))]
rows.sort()
idx=[i for i,r in enumerate(rows) if "hilbert" in r[2] or "morton" in r[2].lower()]
starts=[i for i,r in enumerate(rows) if "gicp_knn_leaf" in r[2]]
i0=starts[40]; 
# back up to the frame's first kernel: find previous gap > 100us
j=i0
while j>0 and rows[j][0]-rows[j-1][1] < 60000: j-=1
t0=rows[j][0]
k=j
while k+1<len(rows) and rows[k+1][0]-rows[k][1] < 60000: k+=1
for s,e,n in rows[j:k+1]: print("%8.1f +%6.1f %s"%((s-t0)/1e3,(e-s)/1e3,n))
PY
rm -rf gpurun_out/c3
