#!/bin/bash
# tools_profile.sh TAG -- on the GPU box: kernel trace + HBM counters of the bench command.
# Separate rocprofv3 passes for FETCH_SIZE and WRITE_SIZE (TCC slots), counters never combined with sys/hip traces.
set -e -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --cpu-frames 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
