#!/bin/bash
# A/B of the split-operand MFMA half-step against the packed-VALU kernel: saturated single ranks and the sweep
for k in ${KS_LIST:-8 12 16}; do
  NMFK_HYB=0 python scripts/microbench.py 60 $k $k 256 | cut -c1-150
  NMFK_HYB=1 python scripts/microbench.py 60 $k $k 256 | cut -c1-150
  for v in ${VARIANTS}; do NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_$v.so python scripts/microbench.py 60 $k $k 256 | cut -c1-150; done
done
NMFK_HYB=0 python scripts/microbench.py 100 2 16 32 | cut -c1-60
NMFK_HYB=1 python scripts/microbench.py 100 2 16 32 | cut -c1-60
for v in ${VARIANTS}; do NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_$v.so python scripts/microbench.py 100 2 16 32 | cut -c1-60; done
