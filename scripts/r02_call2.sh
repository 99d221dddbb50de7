#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "shared_x" > gpurun_out/r02/call2_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -30 gpurun_out/r02/call2_tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
for shx in 0 1; do
  NMFK_SHX=$shx timeout -k 10 300 python scripts/microbench.py 200 2 8 32 2>&1 | tail -1
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done
for k in 2 4 8; do for shx in 0 1; do
  NMFK_SHX=$shx timeout -k 10 300 python scripts/microbench.py 100 $k $k 256 2>&1 | tail -1
done; done
NMFK_SHX=1 timeout -k 10 300 python scripts/microbench.py 200 2 16 32 2>&1 | tail -1
