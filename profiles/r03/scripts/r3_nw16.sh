#!/bin/bash
# Round 3 experiment: 16 waves per workgroup (a workgroup = all 512 columns of a unit in the H half-step) against 8.
cd $(dirname $0)/..
for lib in "" nw16; do
  [ -n "$lib" ] && export NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_$lib.so
  echo "== lib ${lib:-default}"
  NMFK_STREAMS=1 timeout -k 10 100 python scripts/microbench.py 300 2 4 32
  NMFK_STREAMS=1 timeout -k 10 100 python scripts/microbench.py 300 5 8 32
  NMFK_STREAMS=1 timeout -k 10 100 python scripts/microbench.py 300 13 16 32
  NMFK_STREAMS=1 timeout -k 10 100 python scripts/microbench.py 300 9 16 32
  timeout -k 10 100 python scripts/microbench.py 300 2 16 32
done
