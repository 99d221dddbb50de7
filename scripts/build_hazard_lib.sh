#!/bin/bash
# The fp32 mixed-rank packed-VALU kernel is not part of libnmfk_hip.so (NMFK_WITH_MERGED_F32 = 0).  This builds
# nmfk.jl_amd/libnmfk_hip_merged_f32.so WITH it, for the reproducers of DESIGN.md's "Known hazard"
# (scripts/dbg_first_diff.sh, dbg_burners.sh, dbg_burner_kinds.sh, dbg_cumask.sh, dbg_victims.sh pick it up by themselves).
cd $(dirname $0)/..
make -C nmfk.jl_amd/csrc -j6 VARIANT="-DNMFK_WITH_MERGED_F32=1" BUILD=build_merged_f32 OUT=../libnmfk_hip_merged_f32.so
