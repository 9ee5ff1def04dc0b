#!/bin/bash
# tools/pmc_quick.sh TAG -- the four counter passes that classify a kernel as VALU- / TA- / TD- / latency-bound
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $R/bench.py --steps 6 --warmup 2 --cpu-frames 0"}
export RS_SIDE_STREAM=0   # counters per kernel: every kernel alone on the library stream
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed rc=$?"
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
