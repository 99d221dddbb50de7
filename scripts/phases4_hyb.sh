#!/bin/bash
f() { python scripts/microbench.py 300 2 16 16 | sed 's/ obj.*//' | cut -c26- | sed 's/h_step.* loop/loop/; s/ms w_step.*/ms/'; }
for rep in 1 2; do
echo "R=16 one phase"; NMFK_HYB=0 f
for K0 in 11 12 13; do echo "R=16 two phases K0=$K0"; NMFK_HYB=1 NMFK_HYB_PHASES=1 NMFK_HYB_MINK=$K0 f; done
done
