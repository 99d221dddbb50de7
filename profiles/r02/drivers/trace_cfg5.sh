#!/bin/bash
# kernel trace of the wide-rank microbench; usage: scripts/trace_cfg5.sh TAG K [R]
set -u
TAG=${1:-cfg5}; K=${2:-64}; R=${3:-8}
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/scripts/microbench_cfg5.py 10 $R $K > $OUT/out.txt 2> $OUT/err.txt
find $OUT/t -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/t -name '*.csv' -size +4M -delete; find $OUT/t -name '*.db' -delete
cat $OUT/out.txt; cut -c1-200 $OUT/kernel_stats.csv | head -12
