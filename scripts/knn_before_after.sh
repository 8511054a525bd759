# PMC (instruction counts, wave cycles) + kernel trace of the k-NN covariance pass with two builds of the library:
# usage (GPU box, repo root): bash scripts/knn_before_after.sh <before.so> <after.so>
for tag in before after; do
  if [ $tag = before ]; then export DGS_REG_LIB=$PWD/$1; else export DGS_REG_LIB=$PWD/$2; fi
  python scripts/dbg_knn_profile.py > gpurun_out/knn_${tag}_ms.txt 2>&1
  (cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/knn_tr_$tag && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/knn_tr_$tag -- python3 $GRAFT_REPO_ROOT/scripts/dbg_knn_profile.py > /dev/null 2>&1)
  python scripts/knn_trace_summary.py gpurun_out/knn_tr_$tag > gpurun_out/knn_${tag}_kernels.txt; rm -rf gpurun_out/knn_tr_$tag
  PMC_CMD="scripts/dbg_knn_profile.py" NPASS=3 bash scripts/pmc_round.sh knn_$tag > /dev/null 2>&1
  tail -n 2 gpurun_out/knn_${tag}_ms.txt; grep -E "knn_leaf|cov_from" gpurun_out/knn_${tag}_kernels.txt
done
