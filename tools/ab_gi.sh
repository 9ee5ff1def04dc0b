#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib round $round"; RESTIR_HIP_LIB=$R/restir_amd/$lib python $R/tools/bench_gi.py 2>/dev/null | grep -v pathTraceDirect
  done
done
