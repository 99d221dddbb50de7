#!/bin/bash
# ablations of the split-operand MFMA half-step (variant libraries built with -DHYB_DBG_*), 256 restarts of one rank
export NMFK_HYB=1
for v in "" ${VARIANTS:-_NOX _NOM1 _NOM2 _NOBAR _NOLOOP _ALL}; do
  NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip$v.so python scripts/microbench.py 60 ${K:-16} ${K:-16} 256 | cut -c34-160
done
