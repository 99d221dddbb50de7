set -u
OUT=$PWD/gpurun_out/prof_r05_sp2
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_lds -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/lds.err
cd $REPO
python3 scripts/summarize_pmc.py $OUT 2>&1 | grep -A10 "sp_blk_kernel " 
