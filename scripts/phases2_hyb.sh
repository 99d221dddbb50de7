#!/bin/bash
f() { python scripts/microbench.py ${ITERS:-200} 2 16 $1 | sed 's/ obj.*//' | cut -c26- | sed 's/h_step.* loop/loop/; s/ms w_step.*/ms/'; }
for R in ${RS:-32 16 8}; do
echo "R=$R one phase, packed-VALU"; NMFK_HYB=0 f $R
for K0 in ${K0S:-9 10}; do echo "R=$R two phases K0=$K0"; NMFK_HYB=1 NMFK_HYB_PHASES=1 NMFK_HYB_MINK=$K0 f $R; done
echo "R=$R automatic"; f $R
done
