#!/bin/bash
# Known hazard: per-rank packed-VALU kernels whose VGPR allocation is forced upwards (an asm clobber of a high register
# in step_kernel<KP>: make VARIANT='-DNMFK_DBG_TOUCH_VGPR=\"v127\"' ...), beside the bf16 MFMA burner.  One rank at a time.
cd $(dirname $0)/../..
export SHOW=0 TAIL=1 REPS=60 SECS=18
for lib in ${LIBS:-pr127}; do
  for k in ${RANKS:-2 3 5 8 12 16}; do
    bash tools/hazard/dbg_first_diff.sh 0 "per-rank kernel of rank $k, library $lib" NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_v_$lib.so NMFK_HYB=0 KS=$k ITERS=1
  done
done
