#!/bin/bash
# tools/ab_env.sh "VAR=1" ... -- interleaved A/B of one library under different environments ("" = none)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for e in "$@"; do
    env $e python $R/bench.py --steps 60 --warmup 10 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pass_ms']
print('[$e] round $round ms/frame %.3f  primary %.3f ris %.3f shadow %.3f spatial %.4f  Mrays/s %.0f'%(d['ms_per_step'],p['primary'],p['ris'],p['shadow_temporal'],p['spatial_shade'],d['value']))"
  done
done
