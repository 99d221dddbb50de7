#!/bin/bash
# ablations of the split-operand MFMA half-step (variant libraries built with -DHYB_DBG_*), k = 16, 256 restarts
for v in "" _NOLOOP _ALL; do
  NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip$v.so python scripts/microbench.py 60 16 16 256 | cut -c1-160
done
