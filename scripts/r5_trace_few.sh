#!/bin/bash
# kernel traces of the few-unit regimes (round 5): the N = 8 and N = 4 shares of the bench sweep and BASELINE configs[1] (one unit, k = 8)
set -u
REPO=$GRAFT_REPO_ROOT
bash $REPO/scripts/trace_rank.sh r5_rank8 8 300 > /dev/null 2>&1
bash $REPO/scripts/trace_rank.sh r5_rank4 4 300 > /dev/null 2>&1
OUT=$REPO/gpurun_out/trace_r5_cfg2
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/scripts/microbench.py 600 8 8 1 > $OUT/out.txt 2> $OUT/err.txt
find $OUT/t -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]) for r in csv.DictReader(open(f)))
mid = len(rows) // 2
out = open(sys.argv[1] + "/timeline.txt", "w")
t0 = rows[mid][0]
for i in range(mid, min(mid + 60, len(rows))):
    r = rows[i]
    print(f"{(r[0]-t0)/1e3:9.1f} -> {(r[1]-t0)/1e3:9.1f} us dur {(r[1]-r[0])/1e3:6.1f} gap_before {(r[0]-rows[i-1][1])/1e3:6.1f}  {r[2]}", file=out)
PY
find $OUT/t -name '*.csv' -size +4M -delete; find $OUT/t -name '*.db' -delete
for t in r5_rank8 r5_rank4; do echo "== $t"; cat $REPO/gpurun_out/trace_$t/out.txt; head -12 $REPO/gpurun_out/trace_$t/overlap.txt; cat $REPO/gpurun_out/trace_$t/queues.txt; done
echo "== cfg2"; cat $OUT/out.txt; head -40 $OUT/timeline.txt
