#!/bin/bash
f() { python scripts/microbench.py 200 2 16 32 | sed 's/ obj.*//' | cut -c26- | sed 's/h_step.* loop/loop/; s/ms w_step.*/ms/'; }
echo "automatic"; f
echo "K0=8"; NMFK_HYB=1 NMFK_HYB_PHASES=1 NMFK_HYB_MINK=8 f
echo "K0=9 two MFMA groups"; NMFK_HYB=1 NMFK_HYB_PHASES=1 NMFK_HYB_MINK=9 NMFK_HYB_GROUPS=2 f
echo "K0=9 streams=16"; NMFK_HYB=1 NMFK_HYB_PHASES=1 NMFK_HYB_MINK=9 NMFK_STREAMS=16 f
echo "R=16 automatic"; python scripts/microbench.py 200 2 16 16 | sed 's/ obj.*//' | cut -c26- | sed 's/h_step.* loop/loop/; s/ms w_step.*/ms/'
