# the bench workload (32 candidates per step) over the launch-shape knobs of the experiments build: points per thread, floor of blocks per pair
export DGS_REG_LIB=$PWD/delta_graph_slam_amd/libdgs_reg_exp.so
for cfg in "2 64" "4 64" "4 32" "3 64" "1 64" "2 128"; do set -- $cfg; DGS_NDT_PPT=$1 DGS_NDT_MIN_BLOCKS=$2 python bench.py --steps 60 --no-traffic --no-cpu-baseline > gpurun_out/sh.json 2> gpurun_out/sh.err; python -c "
import json
d=json.loads(open('gpurun_out/sh.json').read().strip().splitlines()[-1])
print('ppt $1 min_blocks $2:', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"; done
