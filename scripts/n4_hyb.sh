#!/bin/bash
# one rank's share at 4 GPUs (8 restarts per rank) and 2 GPUs (16): per-rank launches (default) vs merged groups with the
# split-operand MFMA kernel
for R in 8 16; do
f() { python scripts/microbench.py 300 2 16 $R | sed 's/ obj.*//' | cut -c26- | sed 's/h_step.* loop/loop/; s/ms w_step.*/ms/'; }
echo "R=$R default"; NMFK_HYB=0 f
echo "R=$R HYB=0 MERGE=2"; NMFK_HYB=0 NMFK_MERGE=2 f
for cfg in "6 1 1" "6 2 1" "6 2 2" "8 2 2"; do set -- $cfg
  echo "R=$R HYB=1 mink=$1 hyb_groups=$2 merge=$3"; NMFK_HYB=1 NMFK_HYB_MINK=$1 NMFK_HYB_GROUPS=$2 NMFK_MERGE=$3 f
done
done
