cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/gt
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gt -- python3 $GRAFT_REPO_ROOT/scripts/dbg_gicp_tick_profile.py > $GRAFT_REPO_ROOT/gpurun_out/gt.log 2>&1
cd $GRAFT_REPO_ROOT
tail -n 2 gpurun_out/gt.log
python - "$(find gpurun_out/gt -name '*kernel_stats.csv' | head -n 1)" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-58s calls %6s total %9.1f us avg %8.2f us  %5s%%" % (r["Name"].split("(")[0][:58], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
rm -rf gpurun_out/gt
