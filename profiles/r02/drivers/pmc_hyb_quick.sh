#!/bin/bash
# SQ counters of the split-operand MFMA half-step in a 60-iteration bench run (separate passes, no tracing domains)
set -u
OUT=$PWD/gpurun_out/r02/pmc_hyb
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
A="--maxiter 60 --warmup 0 --no-cpu-baseline --no-kopt-check"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2> $OUT/b.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for sub in ('a', 'b'):
    fs = glob.glob(f'gpurun_out/r02/pmc_hyb/{sub}/*/*counter_collection.csv')
    if not fs: print(sub, 'no file'); continue
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if 'hyb_step_kernel<16, 2, 8, false>' not in r['Kernel_Name']: continue
        agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
    for c in sorted(agg): print(sub, c, f"mean per launch {agg[c] / cnt[c]:.5g}  (n={cnt[c]})")
PY
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -delete
