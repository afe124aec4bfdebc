#!/bin/bash
# A/B of alternative builds of the library on the loss-kernel bench: tools/r02/bce_ab.sh "<lib suffixes>" [bench args]
cd "$(dirname "$0")/../.."
libs=$1; shift
for l in $libs; do
  echo "== $l"
  if [ "$l" = default ]; then python tools/bce_bench.py "$@" 2>&1 | grep -v Warn | tail -4
  else GAE_HIP_LIB=$PWD/build/exp/libgae_hip_$l.so python tools/bce_bench.py "$@" 2>&1 | grep -v Warn | tail -4; fi
done
