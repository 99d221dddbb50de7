#!/bin/bash
# s_memtime stamps of the streaming chunk (wave 0 of workgroup (0, 0)); needs a library built with -DNMFK_HYB_STAMP=1 as
# nmfk.jl_amd/libnmfk_hip_stamp.so (see profiles/r03/streaming_stamps.txt)
export NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_stamp.so NMFK_STREAMS=1
echo "== k=16 x 256"; timeout -k 10 100 python scripts/microbench.py 4 16 16 256 2>&1 | grep -E "stamp" | tail -3
echo "== k=2:16 x 32 (bench sweep)"; timeout -k 10 100 python scripts/microbench.py 4 2 16 32 2>&1 | grep -E "stamp" | tail -3
echo "== k=4 x 256"; timeout -k 10 100 python scripts/microbench.py 4 4 4 256 2>&1 | grep -E "stamp" | tail -2
echo "== k=8 x 256"; timeout -k 10 100 python scripts/microbench.py 4 8 8 256 2>&1 | grep -E "stamp" | tail -2
