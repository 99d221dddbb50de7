#!/bin/bash
# rocprofv3 kernel stats of the sparse cfg4 bench (run through gpurun from the repo root): gpurun_out/r03/sp/
set -u
OUT=$PWD/gpurun_out/r03/sp
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/bench_sparse.py ${1:-20} ${2:-32} ${3:-16} ${4:-2} > $OUT/log.txt 2>&1
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name '*.csv' -size +4M -delete
find $OUT -name '*.db' -delete
grep units $OUT/log.txt
cut -d, -f1-6 $OUT/kernel_stats.csv | head -14
python3 - <<'PY'
import csv,collections,glob,os
f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03/sp/trace/*/*_kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'sp_' in n:
        i=n.index('sp_'); d[(n[i:i+18], r['Grid_Size_X'], r['Grid_Size_Y'], r['VGPR_Count'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()): print(k, len(v), f"avg {sum(v)/len(v)/1e3:.0f} us")
PY
