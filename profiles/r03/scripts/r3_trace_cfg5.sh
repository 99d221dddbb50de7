#!/bin/bash
# per-kernel totals at the BASELINE configs[4] shape (rocprofv3 --kernel-trace --stats)
set -u
ROOTD=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTD/gpurun_out/r03/trace_cfg5
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOTD/scripts/microbench_cfg5.py 50 8 ${1:-64} > $OUT/run.log 2>&1
cd $ROOTD
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(f"{'kernel':90s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'%':>6s}")
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>7s} {float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):6.2f}")
PY
tail -1 $OUT/run.log
find $OUT -name '*.csv' -size +3M -delete; find $OUT -name '*.db' -delete
