cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dbg -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_dbg.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof_dbg/*/*kernel_stats.csv | cut -c1-60,200-400 | head -12
