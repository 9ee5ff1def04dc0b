#!/bin/bash
# tools/pmc_lds.sh TAG -- LDS counters per kernel (bank conflicts, busy cycles) of the bench command, every kernel alone on one stream
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmclds_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RS_SIDE_STREAM=0
CMD="python $R/bench.py --steps 6 --warmup 2 --cpu-frames 0"
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed rc=$?"
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
grep -A12 "k_ris_lds\|k_spatial_shade" $OUT/summary.txt
