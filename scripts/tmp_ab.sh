#!/bin/bash
set -u
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not sparse and not kmeans" > gpurun_out/t_pipe.log 2>&1; tail -2 gpurun_out/t_pipe.log
for rep in 1 2; do
for lib in libnmfk_hip_noskew.so libnmfk_hip.so; do
  NMFK_HIP_LIB=$PWD/nmfk.jl_amd/$lib timeout -k 10 300 python3 bench.py --maxiter 400 --warmup 1 --steps 2 --no-cpu-baseline --no-kopt-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', 'ms/step', round(d['ms_per_step'],1), 'w_step', round(r['avg_launch_ms'],4), 'h_step', round(r['other_half_step']['h_step<mfma>']['avg_launch_ms'],4), 'loop TF', round(r['whole_mu_loop']['achieved'],2))"
done
done
