#!/bin/bash
# Round 3: per-kernel totals of a fixed-budget bench sweep (rocprofv3 --kernel-trace --stats), to see what the check block costs.
set -u
ROOTD=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTD/gpurun_out/${1:-r03}/trace_micro
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOTD/scripts/microbench.py ${IT:-200} 2 16 32 > $OUT/run.log 2>&1
cd $ROOTD
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(f"{'kernel':90s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'%':>6s}")
for r in rows[:25]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>7s} {float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):6.2f}")
PY
tail -2 $OUT/run.log
find $OUT -name '*.csv' -size +3M -delete; find $OUT -name '*.db' -delete
