#!/bin/bash
# tools/ab.sh lib1 lib2 ... -- interleaved A/B of library builds: 3 rounds of bench.py per variant, pass times only
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for lib in "$@"; do
    RESTIR_HIP_LIB=$R/restir_amd/$lib python $R/bench.py --steps 60 --warmup 10 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pass_ms']
print('$lib round $round ms/frame %.3f  primary %.3f ris %.3f shadow %.3f spatial %.4f  Mrays/s %.0f'%(d['ms_per_step'],p['primary'],p['ris'],p['shadow_temporal'],p['spatial_shade'],d['value']))"
  done
done
