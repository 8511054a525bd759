# A/B of the early fitness walks (DGS_EARLY_FITNESS) on the bench workload; prints registrations/s, ms per step, roofline fraction
# columns: on/off, min pairs per side-stream launch, LDS cap in KB, max pairs still iterating
for cfg in "1 8 40 99" "1 8 54 99" "1 8 80 99" "1 8 0 12" "1 8 40 12" "1 6 54 8" "0 0 0 0"; do set -- $cfg; DGS_EARLY_FITNESS=$1 DGS_EARLY_FITNESS_MIN_PAIRS=$2 DGS_EARLY_FITNESS_LDS_KB=$3 DGS_EARLY_FITNESS_MAX_ACTIVE=$4 python bench.py --steps 100 --no-traffic > gpurun_out/ef.json 2> gpurun_out/ef.err; python -c "
import json,sys
d=json.loads(open('gpurun_out/ef.json').read().strip().splitlines()[-1])
print('early $1 min_pairs $2 lds $3 max_active $4:', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"; done
