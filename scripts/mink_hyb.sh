#!/bin/bash
# sub-sweeps (32 restarts per rank): packed-VALU kernel vs split-operand MFMA half-step
for r in "9 16" "11 16" "5 8" "2 8" "2 10"; do
  set -- $r
  echo "k=$1:$2"; NMFK_HYB=0 python scripts/microbench.py 100 $1 $2 32 | cut -c1-70
  NMFK_HYB_MINK=5 python scripts/microbench.py 100 $1 $2 32 | cut -c1-70
done
