#!/bin/bash
# Round 6: where the sparse blocked kernel's time goes -- SQ / LDS / VMEM counters of scripts/secondary.py cfg4 (separate --pmc passes).
set -u
OUT=$PWD/gpurun_out/prof_r06_sp
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_lds -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/lds.err
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES_EQ_64 SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_thr -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/thr.err
cd $REPO
python3 scripts/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name '*.csv' -size +2M -delete
find $OUT -name '*.db' -delete
grep -A10 "sp_blk_kernel " $OUT/summary.txt
tail -3 $OUT/thr.err
