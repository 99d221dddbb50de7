#!/bin/bash
f() { python scripts/microbench.py $1 2 16 $2 | sed 's/ obj.*//' | cut -c26- | sed 's/h_step.* loop/loop/; s/ms w_step.*/ms/'; }
for K0 in 8 9 10; do echo "R=32 K0=$K0"; NMFK_HYB=1 NMFK_HYB_PHASES=1 NMFK_HYB_MINK=$K0 f 200 32; done
for K0 in 9 10 11 12 13; do echo "R=16 K0=$K0"; NMFK_HYB=1 NMFK_HYB_PHASES=1 NMFK_HYB_MINK=$K0 f 300 16; done
echo "R=16 one phase"; NMFK_HYB=0 f 300 16
for mk in 5 6 7; do echo "R=8 merged mink=$mk"; NMFK_HYB_MINK=$mk f 300 8; done
for mk in 5 6 7; do echo "R=4 merged mink=$mk"; NMFK_HYB_MINK=$mk f 400 4; done
