#!/bin/bash
# sparse cfg4: parity tests (library's choice, then the blocked form forced onto the small cases, then the gather form only),
# ms/iter of both forms at 200 iterations (the D2H of the results diluted), kernel trace of the blocked form
set -u
mkdir -p gpurun_out
for mode in "" 2 0; do
  NMFK_SP_BLK=$mode timeout -k 10 500 python -m pytest tests -m gpu -x -q -k sparse > gpurun_out/sp_tests_$mode.log 2>&1 || { tail -40 gpurun_out/sp_tests_$mode.log; exit 1; }
  echo "NMFK_SP_BLK='$mode': $(tail -1 gpurun_out/sp_tests_$mode.log)"
done
for rng in "32 16 2" "32 16 17" "16 16 9" "8 16 2"; do
  for blk in 1 0; do
    echo "NMFK_SP_BLK=$blk [kmax R kmin = $rng]:"
    NMFK_SP_BLK=$blk timeout -k 10 300 python3 scripts/bench_sparse.py 200 $rng 2>&1 | grep -E "step|units" || exit 1
  done
done
