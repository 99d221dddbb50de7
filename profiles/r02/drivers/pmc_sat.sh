#!/bin/bash
# PMC pass over the saturated single-rank microbench (256 restarts of one rank); usage: scripts/pmc_sat.sh K
set -u
K=${1:-16}
OUT=$PWD/gpurun_out/pmc_sat_$K
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
COUNTERS=${PMC_COUNTERS:-"GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"}
timeout 300 rocprofv3 --pmc $COUNTERS --output-format csv -d $OUT/p -- python3 $REPO/scripts/microbench.py 40 $K $K 256 > $OUT/out.txt 2> $OUT/err.txt
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "step_kernel" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"].split("step_kernel")[1][:12], r["Grid_Size"])
    d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    d[key]["dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key, v in d.items():
    m = {k: sum(x) / len(x) for k, x in v.items()}
    n = len(v["dur"]) // max(1, len(m) - 1)
    # dur is counted once per counter row: same value repeated
    print(key, "n=%d dur=%.1fus clock=%.2fGHz" % (n, m["dur"] / 1e3, m.get("GRBM_GUI_ACTIVE", 0) / 8 / m["dur"]),
          " ".join(f"{k}={x:.4g}" for k, x in m.items() if k != "dur"))
PY
find $OUT/p -name '*.csv' -size +4M -delete; find $OUT/p -name '*.db' -delete
cat $OUT/out.txt | cut -c1-80
