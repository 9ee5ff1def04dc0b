#!/bin/bash
# tools/ab_eaw.sh lib1 lib2 ... -- interleaved A/B of library builds on the EAW filter (tools/bench_denoisers.py eaw), 3 rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for lib in "$@"; do
    echo "$lib round $round: $(RESTIR_HIP_LIB=$R/restir_amd/$lib python $R/tools/bench_denoisers.py eaw 2>/dev/null | grep LeveledEAWFilter)"
  done
done
