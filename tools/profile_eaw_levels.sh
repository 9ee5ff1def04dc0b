#!/bin/bash
# tools/profile_eaw_levels.sh TAG [--bistro] -- LeveledEAWFilter at 1080p under rocprofv3, three passes (kernel trace; FETCH_SIZE; WRITE_SIZE: counters
# in passes of their own, with the kernel trace only), condensed PER LEVEL by tools/eaw_levels_counters.py into gpurun_out/eawlv_TAG/summary.txt
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/eawlv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_denoisers.py eaw $*"
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace failed rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 || echo "FETCH_SIZE pass failed rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 || echo "WRITE_SIZE pass failed rc=$?"
{ echo "# $(date -u) tools/profile_eaw_levels.sh $TAG $*"; grep -h LeveledEAWFilter $OUT/trace.log; python3 $R/tools/eaw_levels_counters.py $OUT; } > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
