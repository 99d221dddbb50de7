#!/bin/bash
# PMC passes over the sparse bench (separate passes, no tracing domains): gpurun_out/r02/sp_pmc/
set -u
OUT=$PWD/gpurun_out/r02/sp_pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="${1:-10} ${2:-32} ${3:-16} ${4:-32}"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/scripts/bench_sparse.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum \
  --output-format csv -d $OUT/tc -- python3 $GRAFT_REPO_ROOT/scripts/bench_sparse.py $ARGS > $OUT/tc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/scripts/bench_sparse.py $ARGS > $OUT/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for sub in ('sq', 'tc', 'sq2'):
    fs = glob.glob(f'gpurun_out/r02/sp_pmc/{sub}/*/*counter_collection.csv')
    if not fs:
        print(sub, 'no counter file'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        n = r['Kernel_Name']
        if 'sp_' not in n: continue
        key = (n[n.index('sp_'):][:24], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'))
        agg[key][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(key, r['Counter_Name'])] += 1
    for key, cs in sorted(agg.items()):
        print(sub, key, ' '.join(f"{c}={v / cnt[(key, c)]:.4g}" for c, v in sorted(cs.items())))
PY
find $OUT -name '*.csv' -size +2M -delete
find $OUT -name '*.db' -delete
