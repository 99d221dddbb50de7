#!/bin/bash
set -u
for lib in libnmfk_hip.so libnmfk_hip_sb4.so libnmfk_hip_sb8.so libnmfk_hip.so libnmfk_hip_sb4.so libnmfk_hip_sb8.so; do
  echo "$lib [32 16 17] single stream"
  NMFK_HIP_LIB=$PWD/nmfk.jl_amd/$lib NMFK_STREAMS=1 timeout -k 10 200 python3 scripts/bench_sparse.py 30 32 16 17 2>&1 | grep -E "step|units" | cut -c1-100 || exit 1
done
