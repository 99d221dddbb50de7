#!/bin/bash
# does the automatic schedule hurt on sweeps other than the bench's?  (GPU time of the MU loop, 200 iterations)
f() { python scripts/microbench.py 200 $1 $2 $3 | sed 's/ obj.*//' | cut -c26- | sed 's/h_step.* loop/loop/; s/ms w_step.*/ms/'; }
for cfg in "2 12 32" "2 10 32" "9 16 32" "2 16 48" "2 16 64" "6 14 16" "2 12 16"; do set -- $cfg
  echo "k=$1:$2 R=$3 packed-VALU only"; NMFK_HYB=0 f $1 $2 $3
  echo "k=$1:$2 R=$3 automatic"; f $1 $2 $3
done
