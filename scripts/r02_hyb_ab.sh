#!/bin/bash
# quick A/B of the split-operand MFMA half-step: parity subset, then timings (saturated k=16, bench group 9..16 x 32)
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "mfma or hyb or phase or few or merged" 2>&1 | tail -3
for lib in "" $EXTRA_LIBS; do
  [ -n "$lib" ] && export NMFK_HIP_LIB=$PWD/nmfk.jl_amd/$lib
  timeout -k 10 300 python scripts/microbench.py 100 16 16 256 2>&1 | tail -1 | cut -c1-200
  timeout -k 10 300 python scripts/microbench.py 200 9 16 32 2>&1 | tail -1 | cut -c1-200
done
