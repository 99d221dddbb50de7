#!/bin/bash
# sparse cfg4: parity tests (gather form, then the blocked form forced onto the small cases), ms/iter of both forms
set -u
mkdir -p gpurun_out
NMFK_SP_BLK=2 timeout -k 10 400 python -m pytest tests -m gpu -x -q -k sparse > gpurun_out/sp_tests_blk.log 2>&1 || { tail -40 gpurun_out/sp_tests_blk.log; exit 1; }
tail -2 gpurun_out/sp_tests_blk.log
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k sparse > gpurun_out/sp_tests.log 2>&1 || { tail -30 gpurun_out/sp_tests.log; exit 1; }
tail -2 gpurun_out/sp_tests.log
for rng in "32 16 2" "32 16 17" "16 16 9"; do
  for blk in 1 0; do
    echo -n "NMFK_SP_BLK=$blk [$rng]: "
    NMFK_SP_BLK=$blk timeout -k 10 200 python3 scripts/bench_sparse.py 50 $rng 2>&1 | tail -1 | cut -c1-110 || exit 1
  done
done
