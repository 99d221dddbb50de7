#!/bin/bash
# Round 6: what the X loads and the LDS operand reads cost the RESIDENT half-step (W half-step of the bench): builds with them compiled out of the chunk
# loop (NMFK_HYB_ABL; wrong results, same control flow).  build: here; run: through gpurun.
set -u
cd "$(dirname "$0")/.."
MASKS="${MASKS:-1 2 3}"
if [ "${1:-build}" = build ]; then
  for a in $MASKS; do
    (cd nmfk.jl_amd/csrc && mkdir -p build_rabl$a && cp build/nmfk_api.o build/nmfk_comm.o build/nmfk_step_f32.o build/nmfk_step_f64.o build/nmfk_cluster.o build/nmfk_kmeans.o build_rabl$a/ &&
     touch build_rabl$a/*.o && NMFK_SKIP_ISA_LINT=1 make -s BUILD=build_rabl$a OUT=../libnmfk_hip_rabl$a.so VARIANT=-DNMFK_HYB_ABL=$a > /dev/null 2>&1 && echo built $a) &
  done
  wait
else
  for rep in 1 2; do
    python3 scripts/microbench.py 400 2 16 32
    for a in $MASKS; do NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_rabl$a.so python3 scripts/microbench.py 400 2 16 32; done
  done
fi
