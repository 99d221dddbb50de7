#!/bin/bash
# PMC passes for the wide-rank (k > 16) half-step at the configs[4] shape; usage: scripts/pmc_cfg5.sh TAG K
set -u
TAG=${1:-cfg5}
K=${2:-64}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/scripts/microbench_cfg5.py 4 8 $K > $OUT/$name.out 2> $OUT/$name.err
  python3 $REPO/scripts/pmc_per_dispatch.py $OUT/$name mfma_wide 4 > $OUT/$name.txt 2>&1
  python3 $REPO/scripts/pmc_per_dispatch.py $OUT/$name step_kernel 4 >> $OUT/$name.txt 2>&1
  find $OUT/$name -name '*.csv' -size +4M -delete; find $OUT/$name -name '*.db' -delete
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/sq1/**/*counter_collection.csv", recursive=True)
if f:
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if "mfma_wide" not in n and "mfma_sse" not in n: continue
        key = "mfma_wide_kernel" + n.split("mfma_wide_kernel")[1].split("(")[0] if "mfma_wide_kernel" in n else "mfma_sse_kernel" + n.split("mfma_sse_kernel")[1].split("(")[0]
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[key]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out + "/mfma_utilisation.txt", "w") as fh:
        for k, v in d.items():
            m = {a: sum(b) / len(b) for a, b in v.items()}
            cyc = m["GRBM_GUI_ACTIVE"] / 8                       # kernel cycles (sum over the 8 XCDs / 8)
            util = m["SQ_INSTS_MFMA"] * 32 / (cyc * 1024)         # v_mfma_f32_16x16x4_f32: 32 cycles on one of 1024 SIMDs
            print(f"{k}: launches={len(v['SQ_INSTS_MFMA'])} dur={m['dur_ns']/1e3:.1f}us clock={cyc/m['dur_ns']:.2f}GHz "
                  f"MFMA instructions={m['SQ_INSTS_MFMA']:.4g} VALU instructions={m['SQ_INSTS_VALU']:.4g} "
                  f"matrix-pipe utilisation={100*util:.1f}% (of the cycles the kernel ran)", file=fh)
PY
cat $OUT/*.txt
