#!/bin/bash
# Round 3: where does the time of the small-rank variants of the split-operand MFMA half-step go?  Builds the library with
# one timing ablation each (NMFK_HYB_ABLATE, nmfk_step_hyb.hip; results are WRONG by construction) and times the variants
# alone on the GPU, serial streams.  usage (on the GPU box): bash scripts/r3_ablate.sh   (the libraries are built in the
# build container first: bash scripts/r3_ablate.sh build)
cd $(dirname $0)/..
if [ "${1:-}" = build ]; then
  make -C nmfk.jl_amd/csrc -j8 > /dev/null
  for a in ${ABL:-1 2 4 8 16 3 31}; do  # (only nmfk_step_hyb.hip depends on the macro: the other objects are the product build's)
    B=nmfk.jl_amd/csrc/build_abl$a
    mkdir -p $B && cp nmfk.jl_amd/csrc/build/*.o $B/ && rm -f $B/nmfk_step_hyb.o
    make -C nmfk.jl_amd/csrc NMFK_SKIP_ISA_LINT=1 VARIANT="-DNMFK_HYB_ABLATE=$a" BUILD=build_abl$a OUT=../libnmfk_hip_abl$a.so > /dev/null 2>&1 || echo "build $a failed"
  done
  exit 0
fi
export NMFK_STREAMS=1
for range in ${RANGES:-"2 4" "5 8" "13 16"}; do
  echo "== k $range x 32 restarts, 300 iterations"
  timeout -k 10 100 python scripts/microbench.py 300 $range 32
  for a in ${ABL:-1 2 4 8 16 3 31}; do
    NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_abl$a.so timeout -k 10 100 python scripts/microbench.py 300 $range 32 | sed "s#^.*libnmfk_hip_abl#abl#"
  done
done
