#!/bin/bash
# tools/build_variant.sh NAME "EXTRA flags" [git-rev] -- builds restir_amd/librestir_NAME.so from a private copy of the
# sources (of the working tree, or of a git revision) with extra compiler flags: A/B libraries for tools/ab.sh.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; REV=$3
B=/tmp/bv/$NAME
rm -rf $B && mkdir -p $B/restir_amd/csrc $B/include
if [ -n "$REV" ]; then
  git -C $R archive $REV restir_amd/csrc include | tar -x -C $B
else
  cp $R/restir_amd/csrc/*.hip $R/restir_amd/csrc/*.cpp $R/restir_amd/csrc/*.h $R/restir_amd/csrc/Makefile $B/restir_amd/csrc/
  cp $R/include/*.h $B/include/
fi
make -C $B/restir_amd/csrc -j8 EXTRA="$EXTRA" OUT=$R/restir_amd/librestir_$NAME.so VIEWER= RCCLCHK= RCCLRANKS= LOOPBACK= >/dev/null
ls -la $R/restir_amd/librestir_$NAME.so
