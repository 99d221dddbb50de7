#!/bin/bash
# A/B of two builds of the library on the bench sweep itself (default stop rule), alternating: usage r5_ab_bench.sh <other.so>
cd $(dirname $0)/..
OTHER=$PWD/nmfk.jl_amd/$1
ARGS="--steps 2 --warmup 1 --no-kopt-check --no-secondary --no-cpu-baseline"
for rep in 1 2; do
  echo -n "this build: "; python bench.py $ARGS | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['other_half_step'])" | cut -c1-200
  echo -n "$1: "; NMFK_HIP_LIB=$OTHER python bench.py $ARGS | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['other_half_step'])" | cut -c1-200
done
