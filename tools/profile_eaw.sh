#!/bin/bash
# tools/profile_eaw.sh TAG -- LeveledEAWFilter (src/denoiser.cu:64-134,463-477) at 1080p on the bench scene's G-buffer under rocprofv3:
# kernel trace + separate counter passes (HBM bytes, L1 / L2, TA, SQ / LDS), condensed per kernel into
# gpurun_out/eaw_TAG/summary.txt (copy to profiles/).  Counter passes never share a run with a trace domain other than the kernel trace.
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/eaw_$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_denoisers.py eaw"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace failed rc=$?"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed rc=$? (tail: $(tail -2 $OUT/p$i.log | tr '\n' ' '))"
done
{
  echo "# $(date -u) LeveledEAWFilter 1080p, tools/profile_eaw.sh $TAG $*"
  grep -h "LeveledEAWFilter" $OUT/trace.log
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Name"])
    if m: print("%-22s %-12s calls %5s  avg %9.1f us  total %9.1f ms" % (m.group(1), (m.group(2) or ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  for k in k_wavelet k_positions; do python3 $R/tools/pmc_summary.py $OUT $k; done
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
