#!/bin/bash
# tools/ab_denoise_stream.sh OUT -- config 5 (Bistro-class + EAW) with the filter on the library stream (DENOISE_STREAM=0) against the denoise
# stream (1): tools/strip_period.py at N = 1 and on the eight balanced strips of round 5's split, interleaved, STREAM_LEVEL 1 and 2.
OUT=${1:-gpurun_out/r06/ab_denoise_stream.log}
ROWS8=${ROWS8:-160,144,128,104,96,136,144,168}
: > "$OUT"
for rep in 1 2; do
  for ds in ${MODES:-0 1}; do
    for lvl in ${LEVELS:-1}; do
      echo "== rep $rep DENOISE_STREAM=$ds STREAM_LEVEL=$lvl N=1" >> "$OUT"
      DENOISE_STREAM=$ds STREAM_LEVEL=$lvl WORLDS=1 python tools/strip_period.py 5 2>&1 | grep -E "^config|internal" >> "$OUT" || exit 1
      echo "== rep $rep DENOISE_STREAM=$ds STREAM_LEVEL=$lvl N=8 rows $ROWS8" >> "$OUT"
      DENOISE_STREAM=$ds STREAM_LEVEL=$lvl WORLDS=8 ROWS=$ROWS8 python tools/strip_period.py 5 2>&1 | grep -E "^config|internal" >> "$OUT" || exit 1
    done
  done
done
