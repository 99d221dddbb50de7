#!/bin/bash
# kernel trace of one rank's share of the bench sweep at N ranks, fixed budget; usage: scripts/trace_rank.sh TAG N ITERS
set -u
TAG=${1:-rank8}; NR=${2:-8}; IT=${3:-300}
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
R=$((32 / NR))
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/scripts/microbench.py $IT 2 16 $R > $OUT/out.txt 2> $OUT/err.txt
find $OUT/t -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
python3 $REPO/scripts/trace_overlap.py $OUT/t > $OUT/overlap.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
# per queue: chain of kernels in the second half of the trace: durations and gaps
t_mid = rows[len(rows) // 2][0]
byq = collections.defaultdict(list)
for r in rows:
    if r[0] >= t_mid: byq[r[3]].append(r)
out = open(sys.argv[1] + "/queues.txt", "w")
for q, v in sorted(byq.items()):
    dur = sum(b[1] - b[0] for b in v); span = v[-1][1] - v[0][0]
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    names = collections.Counter(b[2].split("(")[0][-40:] for b in v).most_common(3)
    print(f"q={q} n={len(v)} span={span/1e6:.2f}ms busy={dur/1e6:.2f}ms ({100*dur/span:.0f}%) avg_gap={sum(gaps)/max(1,len(gaps))/1e3:.1f}us {names}", file=out)
# union busy time of the GPU
ev = sorted([(r[0], 1) for r in rows if r[0] >= t_mid] + [(r[1], -1) for r in rows if r[0] >= t_mid])
busy = 0; depth = 0; last = None; conc = 0
for t, d in ev:
    if depth > 0: busy += t - last; conc += depth * (t - last)
    depth += d; last = t
span = ev[-1][0] - ev[0][0]
print(f"GPU any-kernel busy {100*busy/span:.0f}% of {span/1e6:.2f} ms, avg concurrency {conc/span:.2f}", file=out)
PY
find $OUT/t -name '*.csv' -size +4M -delete; find $OUT/t -name '*.db' -delete
cat $OUT/out.txt; head -30 $OUT/overlap.txt; cat $OUT/queues.txt
