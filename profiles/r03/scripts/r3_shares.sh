#!/bin/bash
# Round 3: a rank's share of the bench sweep at N = 2, 4, 8 (fixed budget), default geometry against forced ones.
cd $(dirname $0)/..
for N in 8 4 2; do
  TAG="default" timeout -k 10 100 python scripts/share_bench.py $N 400
  TAG="H half-step: shared staging (TARGET_WGS=64)" NMFK_TARGET_WGS=64 timeout -k 10 100 python scripts/share_bench.py $N 400
  TAG="no resident form" NMFK_HYB_RES=0 timeout -k 10 100 python scripts/share_bench.py $N 400
  TAG="round-2 schedule" NMFK_HYB_SMALL=0 NMFK_HYB_RES=0 timeout -k 10 100 python scripts/share_bench.py $N 400
done
