#!/bin/bash
set -u
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "wide or cfg5 or rank" > gpurun_out/t_wide.log 2>&1; tail -2 gpurun_out/t_wide.log
for rep in 1 2; do
for lib in libnmfk_hip_nopipe.so libnmfk_hip.so; do
  for k in 64 32 24; do
  echo -n "$lib "; NMFK_HIP_LIB=$PWD/nmfk.jl_amd/$lib timeout -k 10 300 python3 scripts/microbench_cfg5.py 20 16 $k 2>&1 | tail -1 | cut -c1-200
  done
done
done
