#!/bin/bash
# PMC passes for the wide-rank (k > 16) half-step at the configs[4] shape; usage: scripts/pmc_cfg5.sh TAG K
set -u
TAG=${1:-cfg5}
K=${2:-64}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/scripts/microbench_cfg5.py 4 8 $K > $OUT/$name.out 2> $OUT/$name.err
  python3 $REPO/scripts/pmc_per_dispatch.py $OUT/$name mfma_wide 4 > $OUT/$name.txt 2>&1
  python3 $REPO/scripts/pmc_per_dispatch.py $OUT/$name step_kernel 4 >> $OUT/$name.txt 2>&1
  find $OUT/$name -name '*.csv' -size +4M -delete; find $OUT/$name -name '*.db' -delete
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
cat $OUT/*.txt
