export SHOW=0 TAIL=1 REPS=80 SECS=25
bash tools/hazard/dbg_first_diff.sh 0 "per-rank kernel, LDS sized like the merged launcher" NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_v_lds16.so NMFK_HYB=0 KS=2 ITERS=1
bash tools/hazard/dbg_first_diff.sh 0 "per-rank kernel, LDS sized like the merged launcher, 20 iterations, more ranks" NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_v_lds16.so NMFK_HYB=0 KS=2,3,5,8 ITERS=20
bash tools/hazard/dbg_first_diff.sh 0 "merged kernel, LDS sized for rank 2" NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_v_lds2.so NMFK_HYB=0 NMFK_MERGE=1 KS=2 ITERS=1
