#!/bin/bash
# A/B of two builds of the library in one call: nmfk.jl_amd/libnmfk_hip_prev.so (the commit before) against the tree's build
for rep in 1 2; do
  for lib in libnmfk_hip_prev.so libnmfk_hip.so; do
    NMFK_HIP_LIB=$PWD/nmfk.jl_amd/$lib python3 scripts/microbench.py 400 2 16 32
  done
done
for lib in libnmfk_hip_prev.so libnmfk_hip.so; do
  NMFK_HIP_LIB=$PWD/nmfk.jl_amd/$lib python3 scripts/microbench.py 400 13 16 32
  NMFK_HIP_LIB=$PWD/nmfk.jl_amd/$lib python3 scripts/microbench.py 400 2 4 32
done
