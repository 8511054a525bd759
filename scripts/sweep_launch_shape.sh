# Launch-shape sweep of the single-pair NDT align: points per thread x cap, cfg1 / cfg2 / cfg5.
# usage (repo root, on the GPU box): bash scripts/sweep_launch_shape.sh > gpurun_out/launch_shape_sweep.jsonl
for ppt in 1 2 4; do
  for cap in 128 1024; do
    DGS_NDT_PPT=$ppt DGS_NDT_CAP=$cap python scripts/bench_configs.py --only-ndt --no-cpu --reps 20 2>/dev/null | python -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        print(json.dumps({'ppt': $ppt, 'cap': $cap, 'config': d['config'][:30], 'points': d['points'], 'align_ms': round(d['gpu_align_ms'], 4), 'evaluations': d['evaluations'],
                          'launch_us': round(d['roofline']['avg_launch_us'], 2), 'GBps': round(d['roofline']['achieved'], 1), 'frac': round(d['roofline']['frac'], 4),
                          'Gpoints_per_s': round(d['roofline']['points_per_s'] / 1e9, 2)}))
"
  done
done
