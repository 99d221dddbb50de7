#!/bin/bash
f() { python scripts/microbench.py 400 2 16 4 | sed 's/ obj.*//' | cut -c26-; }
export NMFK_HYB=1
for cfg in "5 1 2" "5 2 2" "6 1 1" "6 1 2" "6 2 1" "6 2 2" "6 3 2" "4 1 1" "4 2 1"; do
  set -- $cfg
  echo "mink=$1 hyb_groups=$2 merge=$3"; NMFK_HYB_MINK=$1 NMFK_HYB_GROUPS=$2 NMFK_MERGE=$3 f
done
for t in 256 1024; do echo "mink=6 target_wgs=$t"; NMFK_HYB_MINK=6 NMFK_TARGET_WGS=$t f; done
