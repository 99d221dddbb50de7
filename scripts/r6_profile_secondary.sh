#!/bin/bash
# Round 6, final build: kernel trace of the secondary workloads + FETCH / WRITE / TCC passes of the sparse one (profiles/r06/secondary*.*, traffic_sp_blk.json)
set -u
OUT=$PWD/gpurun_out/prof_r06b
mkdir -p $OUT/sp
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sec -- python3 $REPO/scripts/secondary.py > $OUT/secondary.json 2> $OUT/trace_sec.err
find $OUT/trace_sec -name '*kernel_stats.csv' -exec cp {} $OUT/secondary_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/sp/pmc_fetch -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/sp_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/sp/pmc_write -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/sp_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/sp/pmc_tcc -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/sp_tcc.err
cd $REPO
python3 scripts/make_traffic.py $OUT/sp "sp_blk_kernel" $OUT/traffic_sp_blk.json
find $OUT -name '*.csv' -size +4M -delete
find $OUT -name '*.db' -delete
head -6 $OUT/secondary_kernel_stats.csv | cut -c1-160
