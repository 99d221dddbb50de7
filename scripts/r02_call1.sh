#!/bin/bash
# round 2, GPU call 1: new parity tests, planted-matrix kopt in both compute modes, baseline bench
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests/test_gpu_branches.py tests/test_gpu_fullsize.py::test_bench_schedule_fixed_budget_vs_oracle -q -m gpu > gpurun_out/r02/call1_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -30 gpurun_out/r02/call1_tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 python scripts/planted_kopt.py f32 f64 > gpurun_out/r02/planted.log 2>&1
rc=$?; echo "planted rc=$rc"; cat gpurun_out/r02/planted.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02/bench_base.json 2> gpurun_out/r02/bench_base.err
echo "bench rc=$?"; cat gpurun_out/r02/bench_base.json | cut -c1-600
