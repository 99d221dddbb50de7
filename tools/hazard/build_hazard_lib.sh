#!/bin/bash
# Builds nmfk.jl_amd/libnmfk_hip_unsafe.so: the library WITHOUT the broadcast-first operand rule of nmfk_step_impl.h
# (-DNMFK_UNSAFE_OPERAND_ORDER: the packed FMAs of step_body take the broadcast operand second, as before the fix), for
# the reproducers of DESIGN.md's "Known hazard":  NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_unsafe.so scripts/dbg_*.sh
cd $(dirname $0)/../..
make -C nmfk.jl_amd/csrc -j6 NMFK_SKIP_ISA_LINT=1 VARIANT="-DNMFK_UNSAFE_OPERAND_ORDER=1" BUILD=build_unsafe OUT=../libnmfk_hip_unsafe.so
