#!/bin/bash
# tools/pmc_mem.sh TAG [LIB] -- texture-addresser / L1 / UTCL1 counters of the bench command.
# Few counters per pass (a request beyond the block's slots makes rocprofv3 abort), each pass under its own timeout.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$2" ] && export RESTIR_HIP_LIB=$R/restir_amd/$2
OUT=$R/gpurun_out/pmcmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 6 --warmup 2 --cpu-frames 0"
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed rc=$?"
done
find $OUT -name "*counter_collection.csv"
