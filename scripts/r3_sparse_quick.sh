#!/bin/bash
set -u
for rep in 1 2; do
for lib in libnmfk_hip_pd4.so libnmfk_hip.so libnmfk_hip_pd7.so; do
  echo "$lib [32 16 17]"
  NMFK_HIP_LIB=$PWD/nmfk.jl_amd/$lib timeout -k 10 200 python3 scripts/bench_sparse.py 100 32 16 17 2>&1 | grep -E "units" | cut -c1-100 || exit 1
done
done
