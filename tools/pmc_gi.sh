#!/bin/bash
# tools/pmc_gi.sh TAG -- SQ / TA / L1 counters of the multi-bounce kernels (tools/bench_gi.py: pathTraceDirect, pathTrace, pathTraceIndirect,
# ReSTIRIndirect at depth 4), own rocprofv3 passes with the kernel trace only; condensed by tools/pmc_summary.py into gpurun_out/pmcgi_TAG/summary.txt
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcgi_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_gi.py ${PMC_GI_ARGS:-}"        # e.g. PMC_GI_ARGS="--bistro" for the other scene
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" \
           "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed rc=$?"
done
python3 $R/tools/pmc_summary.py $OUT k_path > $OUT/summary.txt 2>&1
python3 $R/tools/pmc_summary.py $OUT k_wf >> $OUT/summary.txt 2>&1
python3 $R/tools/pmc_summary.py $OUT k_pt_direct >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
