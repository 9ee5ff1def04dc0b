#!/bin/bash
# tools/profile.sh TAG [CONFIG] -- on the GPU box: kernel trace + HBM counters of the bench command (CONFIG = bench.py --config, default 3).
# Separate rocprofv3 passes for FETCH_SIZE and WRITE_SIZE (TCC slots), counters never combined with sys/hip traces.
set -e -o pipefail
TAG=${1:-r01}
CONFIG=${2:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --config $CONFIG --steps 20 --warmup 3 --cpu-frames 0 --sustained-frames 200"
# per-kernel durations with every kernel on one stream (what bench.py's roofline / pass_ms section measures: kernels that
# share the CUs with another frame's kernels last longer without doing more work) ...
RS_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
# ... and as the timed region runs them: frames overlapped on the auxiliary streams
# (--only-timed: the process ends after the timed region, the sustained frames and the 64 frames of the in-frame measurement, so every
# launch of the trace belongs to an overlapped frame)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlap -- $CMD --only-timed > $OUT/trace_overlap.log 2>&1
export RS_SIDE_STREAM=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
