#!/bin/bash
set -u
NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_mink0.so NMFK_SP_BLK=2 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k sparse > gpurun_out/sp_tests_blk.log 2>&1 || { tail -40 gpurun_out/sp_tests_blk.log; exit 1; }
tail -1 gpurun_out/sp_tests_blk.log
for lib in libnmfk_hip.so libnmfk_hip_mink0.so libnmfk_hip.so libnmfk_hip_mink0.so; do
  echo "$lib [8 16 2]"
  NMFK_HIP_LIB=$PWD/nmfk.jl_amd/$lib timeout -k 10 200 python3 scripts/bench_sparse.py 100 8 16 2 2>&1 | grep -E "units" | cut -c1-100 || exit 1
done
for lib in libnmfk_hip.so libnmfk_hip_mink0.so; do
  echo "$lib [32 16 2]"
  NMFK_HIP_LIB=$PWD/nmfk.jl_amd/$lib timeout -k 10 300 python3 scripts/bench_sparse.py 200 32 16 2 2>&1 | grep -E "units" | cut -c1-100 || exit 1
done
