#!/bin/bash
# A/B on one box: libnmfk_hip_old.so (HEAD) against the working build, gather form (NMFK_SP_BLK=0), alternating
set -u
OLD=$PWD/nmfk.jl_amd/libnmfk_hip_old.so
NEW=$PWD/nmfk.jl_amd/libnmfk_hip.so
export NMFK_SP_BLK=0
for rng in "32 16 2" "32 16 17" "16 16 9" "8 16 2"; do
  for rep in 1 2; do
    for lib in $OLD $NEW; do
      echo -n "$(basename $lib) [$rng]: "
      NMFK_HIP_LIB=$lib timeout -k 10 200 python3 scripts/bench_sparse.py 50 $rng 2>&1 | tail -1 | cut -c1-60 || exit 1
    done
  done
done
echo "blocked W half-step (new build):"
NMFK_SP_BLK=1 timeout -k 10 200 python3 scripts/bench_sparse.py 50 32 16 17 2>&1 | tail -1 | cut -c1-60
