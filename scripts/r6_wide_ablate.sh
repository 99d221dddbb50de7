#!/bin/bash
# Round 6: where the wide-rank half-step's time goes -- builds of nmfk_step_hyb.hip with parts of the chunk body compiled out (NMFK_WIDE_ABL, wrong
# results, same control flow), timed on BASELINE configs[4]'s shape.  Build here (no GPU): bash scripts/r6_wide_ablate.sh build; run through gpurun:
# bash scripts/r6_wide_ablate.sh run
set -u
cd "$(dirname "$0")/.."
MASKS="${MASKS:-1 2 4 8 16 31}"
if [ "${1:-build}" = build ]; then
  for a in $MASKS; do
    (cd nmfk.jl_amd/csrc && mkdir -p build_abl$a && cp build/nmfk_api.o build/nmfk_comm.o build/nmfk_step_f32.o build/nmfk_step_f64.o build/nmfk_cluster.o build/nmfk_kmeans.o build_abl$a/ &&
     touch build_abl$a/*.o && NMFK_SKIP_ISA_LINT=1 make -s BUILD=build_abl$a OUT=../libnmfk_hip_abl$a.so VARIANT=-DNMFK_WIDE_ABL=$a > /dev/null 2>&1 && echo built $a) &
    if [ $a = 2 ] || [ $a = 8 ]; then wait; fi
  done
  wait
else
  for rep in 1 2; do
    python3 scripts/r6_wide_one.py 64 1 60
    for a in $MASKS; do NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_abl$a.so python3 scripts/r6_wide_one.py 64 1 60 | sed "s/^/abl $a: /"; done
  done
fi
