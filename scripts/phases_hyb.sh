#!/bin/bash
# would a two-phase sweep pay at 32 restarts per rank?  phase A: the ranks >= K0 as one mixed-rank group on the
# split-operand MFMA kernel (alone on the GPU), phase B: the other ranks on per-rank packed-VALU launches
f() { python scripts/microbench.py 200 $1 $2 32 | sed 's/ obj.*//' | cut -c26- | sed 's/h_step.* loop/loop/; s/ms w_step.*/ms/'; }
echo "full sweep, default"; NMFK_HYB=0 f 2 16
for K0 in 9 11 12 13; do
  echo "A: k=$K0:16 mixed-rank MFMA group"; NMFK_HYB=1 NMFK_MERGE=1 NMFK_HYB_MINK=$K0 f $K0 16
  echo "A(valu): k=$K0:16 per-rank VALU"; NMFK_HYB=0 f $K0 16
  echo "B: k=2:$((K0-1)) per-rank VALU"; NMFK_HYB=0 f 2 $((K0-1))
done
