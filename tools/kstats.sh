#!/bin/bash
# tools/kstats.sh TAG [ENV...] -- per-kernel average durations of the bench command, every kernel alone on one stream
# (rocprofv3 --kernel-trace --stats); prints the kernel_stats table.  Extra arguments are VAR=value settings.
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kstats_$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
export RS_SIDE_STREAM=0
CMD=${PMC_CMD:-"python $R/bench.py --steps 20 --warmup 3 --cpu-frames 0"}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "k_" not in n: continue
    import re
    m = re.search(r"(k_[a-z_0-9]+)", n)
    print("%-28s calls %5s  avg %9.1f us  min %9.1f  max %9.1f" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
