#!/bin/bash
# rocprofv3 kernel stats of a fixed-budget bench sweep (k = 2:16, R restarts); usage: scripts/trace_sweep.sh TAG R ITERS
TAG=${1:-sweep}; R=${2:-32}; IT=${3:-200}
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/scripts/microbench.py $IT 2 16 $R > $OUT/out.txt 2> $OUT/err.txt
find $OUT/t -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/t -name '*.csv' -size +4M -delete; find $OUT/t -name '*.db' -delete
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:18]:
    print(r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44].ljust(44), r["Calls"].rjust(7), f'{float(r["AverageNs"])/1e3:9.1f} us', f'{float(r["TotalDurationNs"])/1e6:9.1f} ms', r["Percentage"])
PY
