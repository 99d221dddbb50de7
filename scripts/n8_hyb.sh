#!/bin/bash
# one rank's share of the bench sweep at 8 GPUs (4 restarts of every rank): merged packed-VALU groups vs mixed-rank
# groups of the split-operand MFMA kernel for the ranks >= MINK
f() { python scripts/microbench.py 400 2 16 4 | sed 's/ obj.*//' | cut -c26-; }
NMFK_HYB=0 f
for mk in ${MINKS:-5 7 9}; do for hg in ${HGS:-1 2}; do
  echo "mink=$mk groups=$hg"; NMFK_HYB=1 NMFK_HYB_MINK=$mk NMFK_HYB_GROUPS=$hg f
done; done
