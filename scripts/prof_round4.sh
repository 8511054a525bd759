# Round-4 profile on the GPU box: the bench line (upstream order timed; PMC traffic; cpu_baseline; parity gate), rocprofv3 kernel stats of the
# same command, SQ counters of the timed kernel, every config in both orders (+ cfg5_batched with traffic), the group form, the fast order.
# usage (repo root on the box): bash scripts/prof_round4.sh <prefix>   -> gpurun_out/<prefix>_*
P=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 700 python bench.py --steps 100 --warmup 10 > gpurun_out/${P}_bench.json 2> gpurun_out/${P}_bench.err
echo bench-done
mkdir -p gpurun_out/prof_$P
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$P -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $R/gpurun_out/${P}_bench_under_rocprof.json 2> $R/gpurun_out/${P}_rocprof.err
echo rocprof-done
cd $R
cp $(find gpurun_out/prof_$P -name "*kernel_stats.csv" | head -1) gpurun_out/${P}_kernel_stats.csv
rm -rf gpurun_out/prof_$P
NPASS=5 PMC_CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic" bash scripts/pmc_round.sh ${P} > gpurun_out/${P}_pmc.log 2>&1
echo pmc-done
cd $R
timeout -k 10 900 python scripts/bench_configs.py --frames 100 > gpurun_out/${P}_configs.jsonl 2> gpurun_out/${P}_configs.err
echo configs-done
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --group --no-traffic --no-cpu-baseline > gpurun_out/${P}_bench_group.json 2> gpurun_out/${P}_bench_group.err
echo group-done
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --order 0 --no-traffic --no-cpu-baseline > gpurun_out/${P}_bench_fast_order.json 2> gpurun_out/${P}_bench_fast.err
echo fast-done
ls -la gpurun_out | grep ${P}_
