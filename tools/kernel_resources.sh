#!/bin/bash
# tools/kernel_resources.sh [file.hip ...] -- VGPRs, SGPRs, scratch bytes and LDS of every kernel of the given sources (default: all of
# restir_amd/csrc/*.hip), from the gfx950 code object's metadata: the device side compiled alone with the library's flags (no GPU needed).
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/restir_amd/csrc
FILES=${@:-*.hip}
for f in $FILES; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $EXTRA \
    --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -c $f -o /tmp/kr_$$.co 2>/dev/null || { echo "$f: compile failed"; continue; }
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kr_$$.co | awk -v F=$f '
    /\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {scr=$2} /\.sgpr_count:/ {sg=$2}
    /\.vgpr_count:/ {vg=$2; printf "%-14s vgpr %3d sgpr %3d scratch %4d lds %6d  %s\n", F, vg, sg, scr, lds, name}'
  rm -f /tmp/kr_$$.co
done
