# Round profile on the GPU box: bench line (with PMC traffic), rocprofv3 kernel stats of the same command, all configs.
# usage (from the repo root on the box): bash scripts/prof_round.sh <prefix>     -> files gpurun_out/<prefix>_*
P=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$P
cd $R
timeout -k 10 500 python bench.py --steps 100 --warmup 10 > gpurun_out/${P}_bench.json 2> gpurun_out/${P}_bench.err
echo bench-done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$P -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $R/gpurun_out/${P}_bench_under_rocprof.json 2> $R/gpurun_out/${P}_rocprof.err
echo rocprof-done
cd $R
timeout -k 10 600 python scripts/bench_configs.py --frames 100 > gpurun_out/${P}_configs.jsonl 2> gpurun_out/${P}_configs.err
echo configs-done
timeout -k 10 300 python scripts/bench_cache.py --candidates 32 --distinct 8 > gpurun_out/${P}_cache_ticks.jsonl 2> gpurun_out/${P}_cache.err
echo cache-done
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --group --no-traffic --no-cpu-baseline > gpurun_out/${P}_bench_group.json 2> gpurun_out/${P}_bench_group.err
echo group-done
cd /tmp
TAIL_STEPS=12 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tail_$P -- python3 $R/scripts/tail_profile.py > $R/gpurun_out/${P}_tail.log 2>&1
cd $R
python3 scripts/tail_fit.py gpurun_out/tail_$P gpurun_out/${P}_tail_table.json > /dev/null
echo tail-done
# keep the summaries, drop the raw traces (gpurun copies at most 64 MiB back)
cp $(find gpurun_out/prof_$P -name "*kernel_stats.csv" | head -1) gpurun_out/${P}_kernel_stats.csv
rm -rf gpurun_out/prof_$P gpurun_out/tail_$P gpurun_out/traffic
ls -la gpurun_out | head -40
