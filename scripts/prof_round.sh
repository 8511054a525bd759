# Round profile on the GPU box: bench line (with PMC traffic), rocprofv3 kernel stats of the same command, all configs.
# usage (from the repo root on the box): bash scripts/prof_round.sh <prefix>     -> files gpurun_out/<prefix>_*
set -e
P=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$P
cd $R
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/${P}_bench.json 2> gpurun_out/${P}_bench.err
echo bench-done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$P -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $R/gpurun_out/${P}_bench_under_rocprof.json 2> $R/gpurun_out/${P}_rocprof.err
echo rocprof-done
cd $R
timeout -k 10 600 python scripts/bench_configs.py --frames 100 > gpurun_out/${P}_configs.jsonl 2> gpurun_out/${P}_configs.err
echo configs-done
timeout -k 10 300 python scripts/bench_cache.py --candidates 32 --distinct 8 > gpurun_out/${P}_cache_ticks.jsonl 2> gpurun_out/${P}_cache.err
echo cache-done
find gpurun_out/prof_$P -name "*kernel_stats.csv"
