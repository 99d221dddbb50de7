#!/bin/bash
# Known hazard: the fp32 merged packed-VALU kernel (one iteration at rank 2) beside register-only loops of different
# matrix instructions in another process (tools/hazard/burner.hip modes 0, 1, 6, 7, 8, 9)
cd $(dirname $0)/../..
export SHOW=0 TAIL=1 REPS=${REPS:-80} SECS=22
for mode in 0 1 6 7 8 9; do
  bash tools/hazard/dbg_first_diff.sh $mode "merged fp32 kernel" NMFK_HYB=0 NMFK_MERGE=1 KS=2 ITERS=1
done
