#!/bin/bash
# profiles/r02/sparse_cfg4.txt: BASELINE configs[3] through the gather kernels (run through gpurun from the repo root)
set -u
mkdir -p gpurun_out/r02
{
echo "scripts/bench_sparse.py ITERS 32 16   (one MI355X, round 2; round 1: 60.39 ms/iter at 50 iterations, 501 GB/s)"
for it in 20 50 120; do echo "--- $it iterations"; python scripts/bench_sparse.py $it 32 16 2>&1 | tail -1; done
} | tee gpurun_out/r02/sparse_cfg4_runs.txt
bash scripts/prof_sparse.sh 20 32 16 2 > gpurun_out/r02/sparse_cfg4_prof.txt 2>&1
