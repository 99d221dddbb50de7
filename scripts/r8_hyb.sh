#!/bin/bash
f() { python scripts/microbench.py 300 2 16 8 | sed 's/ obj.*//' | cut -c26- | sed 's/h_step.* loop/loop/; s/ms w_step.*/ms/'; }
echo "default"; f
echo "HYB=0"; NMFK_HYB=0 f
for cfg in "6 1 1" "6 1 2" "9 1 2" "8 1 1" "7 1 1"; do set -- $cfg
  echo "merged: mink=$1 hyb_groups=$2 merge=$3"; NMFK_HYB=1 NMFK_HYB_MINK=$1 NMFK_HYB_GROUPS=$2 NMFK_MERGE=$3 f
done
for K0 in 12 13; do echo "phases K0=$K0"; NMFK_HYB=1 NMFK_HYB_PHASES=1 NMFK_HYB_MINK=$K0 f; done
