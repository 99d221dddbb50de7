#!/bin/bash
# Round 6: counters of the wide-rank half-step (wide2_step_kernel), numerators on the fp32 pipe (bn = 0) and on the bf16 pipe (bn = 1).
# Run through gpurun from the repo root; separate --pmc passes (MI355X_MICROARCH.md).  Output: gpurun_out/wide_pmc/summary_bn{0,1}.txt
set -u
OUT=$PWD/gpurun_out/wide_pmc
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
K=${1:-64}
cd /tmp
for BN in 0 1; do
  D=$OUT/bn$BN
  mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $REPO/scripts/r6_wide_one.py $K $BN 20 > $D/run.txt 2> $D/trace.err
  find $D/trace -name '*kernel_stats.csv' -exec cp {} $D/kernel_stats.csv \;
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --output-format csv -d $D/pmc_sq -- python3 $REPO/scripts/r6_wide_one.py $K $BN 10 > /dev/null 2> $D/pmc_sq.err
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
    --output-format csv -d $D/pmc_mfma -- python3 $REPO/scripts/r6_wide_one.py $K $BN 10 > /dev/null 2> $D/pmc_mfma.err
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT \
    --output-format csv -d $D/pmc_mem -- python3 $REPO/scripts/r6_wide_one.py $K $BN 10 > /dev/null 2> $D/pmc_mem.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $D/pmc_tcc -- python3 $REPO/scripts/r6_wide_one.py $K $BN 10 > /dev/null 2> $D/pmc_tcc.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 $REPO/scripts/r6_wide_one.py $K $BN 10 > /dev/null 2> $D/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 $REPO/scripts/r6_wide_one.py $K $BN 10 > /dev/null 2> $D/pmc_write.err
  echo "bn $BN done" >&2
  (cd $REPO && python3 scripts/summarize_pmc.py $D > $OUT/summary_bn$BN.txt 2>&1)
  cp $D/kernel_stats.csv $OUT/kernel_stats_bn$BN.csv
  find $D -name '*.csv' -size +2M -delete
  find $D -name '*.db' -delete
done
grep -A9 "wide2_step_kernel" $OUT/summary_bn0.txt | head -120
