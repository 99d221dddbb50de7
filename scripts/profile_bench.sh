#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench command (the profile the roofline numbers are checked against)
#   2. PMC passes on a short fixed-budget run (separate passes; never combined with tracing domains)
# Outputs go to gpurun_out/prof_<tag>/; copy the summaries to profiles/.
set -u
TAG=${1:-r02}
SHORT="--maxiter ${2:-60} --warmup 0 --no-cpu-baseline --no-kopt-check"
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
REPO=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py ${BENCH_ARGS:-} > $OUT/bench_traced.json 2> $OUT/trace.err
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
python3 $REPO/scripts/trace_overlap.py $OUT/trace > $OUT/trace_overlap.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $SHORT > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcc -- python3 $REPO/bench.py $SHORT > /dev/null 2> $OUT/pmc_tcc.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $SHORT > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $SHORT > /dev/null 2> $OUT/pmc_write.err
cd $REPO
python3 scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
python3 scripts/make_traffic.py $OUT "${TRAFFIC_KERNEL:-hyb_res_kernel<2, false, false, false>}" $OUT/traffic.json > /dev/null 2>&1
# keep only the small summaries (gpurun_out merge is capped at 64 MiB)
find $OUT -name '*.csv' -size +4M -delete
find $OUT -name '*.db' -delete
ls -la $OUT
