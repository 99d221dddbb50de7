#!/bin/bash
# Round 6 evidence (run through gpurun from the repo root; copy the summaries to profiles/r06/):
#   1. rocprofv3 --kernel-trace --stats of the TIMED sweeps of bench.py only (no planted / fp64 / secondary sweeps in the
#      trace: VERDICT r3 hygiene) -> kernel_stats.csv + the JSON line of that run
#   2. separate --pmc passes on a 60-iteration run: SQ / MFMA / TCC / FETCH_SIZE / WRITE_SIZE -> pmc_summary.txt, traffic.json
#   3. rocprofv3 --kernel-trace --stats of the secondary workloads (cfg4 sparse, cfg5 k = 64) -> secondary_kernel_stats.csv,
#      and FETCH_SIZE / WRITE_SIZE / TCC passes of the sparse one
set -u
OUT=$PWD/gpurun_out/prof_r06
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
TIMED="--steps 2 --warmup 1 --no-kopt-check --no-secondary --no-cpu-baseline"
SHORT="--maxiter 60 --steps 1 --warmup 0 --no-kopt-check --no-secondary --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $TIMED > $OUT/bench_traced.json 2> $OUT/trace.err
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
echo "trace done" >&2
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $SHORT > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU \
  --output-format csv -d $OUT/pmc_mfma -- python3 $REPO/bench.py $SHORT > /dev/null 2> $OUT/pmc_mfma.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcc -- python3 $REPO/bench.py $SHORT > /dev/null 2> $OUT/pmc_tcc.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $SHORT > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $SHORT > /dev/null 2> $OUT/pmc_write.err
echo "pmc done" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sec -- python3 $REPO/scripts/secondary.py > $OUT/secondary.json 2> $OUT/trace_sec.err
find $OUT/trace_sec -name '*kernel_stats.csv' -exec cp {} $OUT/secondary_kernel_stats.csv \;
mkdir -p $OUT/sp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/sp/pmc_fetch -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/sp_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/sp/pmc_write -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/sp_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/sp/pmc_tcc -- python3 $REPO/scripts/secondary.py cfg4 > /dev/null 2> $OUT/sp_tcc.err
echo "secondary done" >&2
#   4. the N = 8 share of the bench sweep (60 units, two cohorts on two streams): kernel trace of rank 0's share alone on the GPU
RANK_SIM_FIRST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_n8 -- python3 $REPO/scripts/rank_sim.py 8 > $OUT/rank_sim_n8_traced.txt 2> $OUT/trace_n8.err
find $OUT/trace_n8 -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_n8_share.csv \;
python3 $REPO/scripts/trace_overlap.py $OUT/trace_n8 > $OUT/n8_share_overlap.txt 2>&1
echo "n8 share done" >&2
cd $REPO
python3 scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
python3 scripts/summarize_pmc.py $OUT/sp > $OUT/pmc_summary_sparse.txt 2>&1
python3 scripts/make_traffic.py $OUT "hyb_res_kernel<2, false, false, false>" $OUT/traffic.json > /dev/null 2>&1
python3 scripts/make_traffic.py $OUT "hyb_step_kernel<2, 8, 0>" $OUT/traffic_h_step.json > /dev/null 2>&1
python3 scripts/make_traffic.py $OUT/sp "sp_blk_kernel" $OUT/traffic_sp_blk.json > /dev/null 2>&1
find $OUT -name '*.csv' -size +4M -delete
find $OUT -name '*.db' -delete
ls -la $OUT | head -40
