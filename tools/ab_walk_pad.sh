#!/bin/bash
R=$GRAFT_REPO_ROOT
export RESTIR_HIP_LIB=$R/restir_amd/librestir_pad.so
OUT=$R/gpurun_out/r06/walk_pad.log
: > $OUT
for pad in 0 22000 26000 0 22000; do
  echo "== RS_WALK_LDS_PAD=$pad config 5 N=1" >> $OUT
  RS_WALK_LDS_PAD=$pad WORLDS=1 python $R/tools/strip_period.py 5 2>&1 | grep -E "^config 5 N=1 even" >> $OUT
done
for pad in 0 22000 26000; do
  echo "== RS_WALK_LDS_PAD=$pad config 3 N=1, N=8" >> $OUT
  RS_WALK_LDS_PAD=$pad WORLDS=1,8 ROUNDS=0 python $R/tools/strip_period.py 3 2>&1 | grep -E "^config 3 N=. even" >> $OUT
done
cd /tmp && export TMPDIR=/tmp
for pad in 0 22000; do
  rm -rf /tmp/tr_$pad
  CONFIG=5 RS_WALK_LDS_PAD=$pad rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$pad -- python $R/tools/strip_trace_c.py 1 0 > /tmp/tr_$pad.log 2>&1
  f=$(find /tmp/tr_$pad -name "*kernel_trace.csv" | head -1)
  echo "== GPU-paced trace, config 5 N=1, RS_WALK_LDS_PAD=$pad" >> $OUT
  python $R/tools/strip_trace_report.py $f >> $OUT 2>&1
done
cat $OUT
