cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl_on $GRAFT_REPO_ROOT/gpurun_out/tl_off
cd $GRAFT_REPO_ROOT
TAIL_STEPS=6 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_on -- python3 scripts/tail_profile.py > gpurun_out/tl_on.log 2>&1
python scripts/step_timeline.py gpurun_out/tl_on > gpurun_out/timeline_on.txt
find gpurun_out/tl_on -name "*.csv" -size +2M -delete; rm -rf gpurun_out/tl_on
tail -n 70 gpurun_out/timeline_on.txt
