#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_fullsize.py::test_planted_rank6_same_kopt_at_metric_size > gpurun_out/r02/call3_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r02/call3_tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python scripts/microbench.py 100 16 16 256 2>&1 | tail -1
timeout -k 10 300 python scripts/microbench.py 100 9 9 256 2>&1 | tail -1
timeout -k 10 300 python scripts/microbench.py 200 2 16 32 2>&1 | tail -1
timeout -k 10 300 python scripts/microbench.py 200 9 16 32 2>&1 | tail -1
