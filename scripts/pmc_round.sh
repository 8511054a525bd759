# SQ / cache counters per kernel for the bench workload: separate rocprofv3 --pmc passes (--kernel-trace only, as the pool requires),
# summarised by scripts/pmc_summarise.py.  usage on the GPU box (repo root): bash scripts/pmc_round.sh <prefix> [script args...]
# (default workload: bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic; PMC_CMD="scripts/dbg_strict_profile.py" profiles another script)
P=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$P
rm -rf $OUT && mkdir -p $OUT
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  if [ -n "$NPASS" ] && [ $i -gt $NPASS ]; then break; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $R/${PMC_CMD:-bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic} > $OUT/pass$i.log 2>&1
  echo "pass $i done: $SET"
done
cd $R
python3 scripts/pmc_summarise.py $OUT > gpurun_out/${P}_pmc_per_kernel.json
rm -rf $OUT   # raw per-dispatch counter files: tens of MB per pass
echo summary-done
