export SHOW=2 TAIL=4
bash tools/hazard/dbg_first_diff.sh 0 "per-rank packed-VALU kernels" NMFK_HYB=0 KS=2,3,5,8,12,16 ITERS=20
bash tools/hazard/dbg_first_diff.sh 0 "split-operand MFMA group" KS=9,12,16 ITERS=20 
bash tools/hazard/dbg_first_diff.sh 0 "wide-rank MFMA kernel" KS=20,32 ITERS=20
bash tools/hazard/dbg_first_diff.sh 0 "merged kernel fp64" COMPUTE=f64 NMFK_HYB=0 KS=2,3,5 ITERS=20
bash tools/hazard/dbg_first_diff.sh 0 "merged kernel fp32" NMFK_HYB=0 NMFK_MERGE=1 KS=2,3,5 ITERS=20
bash tools/hazard/dbg_first_diff.sh 1 "merged kernel fp32 beside the fp32 MFMA burner" NMFK_HYB=0 NMFK_MERGE=1 KS=2,3,5 ITERS=20
