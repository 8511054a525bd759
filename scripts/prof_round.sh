set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_d
cd $R
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --traffic > gpurun_out/d_bench_with_traffic.json 2> gpurun_out/d_bench_with_traffic.err
echo traffic-done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/d_bench_under_rocprof.json 2> $R/gpurun_out/d_rocprof.err
echo rocprof-done
cd $R
timeout -k 10 600 python scripts/bench_configs.py --frames 100 > gpurun_out/d_configs.jsonl 2> gpurun_out/d_configs.err
echo configs-done
find gpurun_out/prof_d -name "*kernel_stats.csv" | head
