#!/bin/bash
# tools/pmc_cache.sh TAG [ENV...] -- L1 / L2 counters of the bench command per kernel (own rocprofv3 passes, kernels alone on one stream)
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcc_$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
export RS_SIDE_STREAM=0
CMD=${PMC_CMD:-"python $R/bench.py --steps 6 --warmup 2 --cpu-frames 0"}      # PMC_CMD="python $R/tools/bench_denoisers.py eaw": another workload
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TD_LOAD_WAVEFRONT_sum"; do      # at most two TA counters per pass (three abort with "exceeds the capabilities of the hardware")
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed rc=$? (tail: $(tail -2 $OUT/p$i.log | tr '\n' ' '))"
done
for k in ${PMC_KERNELS:-k_shadow k_primary}; do python3 $R/tools/pmc_summary.py $OUT $k; done
