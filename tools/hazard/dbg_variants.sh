#!/bin/bash
# Known hazard: the fp32 merged packed-VALU kernel built several ways (nmfk.jl_amd/libnmfk_hip_v_<name>.so, see the
# make lines in profiles/r02/merged_kernel_hazard.txt), ONE iteration at rank 2 beside the bf16 MFMA burner.
cd $(dirname $0)/../..
export SHOW=0 TAIL=1 REPS=${REPS:-80} SECS=25
for so in nmfk.jl_amd/libnmfk_hip.so nmfk.jl_amd/libnmfk_hip_v_*.so; do
  bash tools/hazard/dbg_first_diff.sh 0 "$(basename $so)" NMFK_HIP_LIB=$PWD/$so NMFK_HYB=0 NMFK_MERGE=1 KS=${KS:-2} ITERS=1
done
