#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for lib in "$@"; do
    for cfg in 3 4; do
    RESTIR_HIP_LIB=$R/restir_amd/$lib python $R/bench.py --config $cfg --steps 30 --warmup 5 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pass_ms']
print('$lib config $cfg round $round ms/frame %.3f spatial %.4f us frac %.3f'%(d['ms_per_step'],p['spatial_shade']*1e3,d['roofline']['frac']))"
    done
  done
done
