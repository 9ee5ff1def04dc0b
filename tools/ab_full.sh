#!/bin/bash
# tools/ab_full.sh lib1 lib2 ... -- interleaved A/B of library builds: config 3 overlapped / synchronous / per pass, config 5, multi-bounce kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for lib in "$@"; do
    RESTIR_HIP_LIB=$R/restir_amd/$lib python $R/bench.py --steps 60 --warmup 10 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pass_ms']
print('$lib round $round config 3: ms/frame %.3f sync %.3f gbuffer %.3f primary %.3f ris %.3f shadow %.3f spatial %.4f'%(d['ms_per_step'],d['ms_per_frame_synchronous'],p['gbuffer'],p['primary'],p['ris'],p['shadow_temporal'],p['spatial_shade']))"
  done
done
for lib in "$@"; do
  RESTIR_HIP_LIB=$R/restir_amd/$lib python $R/bench.py --config 5 --steps 40 --warmup 10 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pass_ms']
print('$lib config 5: ms/frame %.3f sync %.3f gbuffer %.3f primary %.3f ris %.3f shadow %.3f'%(d['ms_per_step'],d['ms_per_frame_synchronous'],p['gbuffer'],p['primary'],p['ris'],p['shadow_temporal']))"
  echo "$lib multi-bounce:"; RESTIR_HIP_LIB=$R/restir_amd/$lib python $R/tools/bench_gi.py 2>/dev/null | grep -v amdgpu
done
