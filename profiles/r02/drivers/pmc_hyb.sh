#!/bin/bash
# PMC passes (separate runs) over the saturated single-rank microbench with the split-operand MFMA half-step; usage: scripts/pmc_hyb.sh K
set -u
K=${1:-16}
OUT=$PWD/gpurun_out/pmc_hyb_$K
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp
run() {
  local name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/scripts/microbench.py 20 $K $K 256 > $OUT/$name.out 2> $OUT/$name.err
  python3 - $OUT/$name <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs: print("no csv"); sys.exit(0)
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if "hyb_step_kernel" not in r["Kernel_Name"]: continue
    key = r["Grid_Size"]
    d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    d[key]["dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key, v in d.items():
    m = {k: sum(x) / len(x) for k, x in v.items()}
    print("grid", key, "n=%d dur=%.1fus" % (len(v["dur"]) // max(1, len(m) - 1), m["dur"] / 1e3), " ".join(f"{k}={x:.5g}" for k, x in m.items() if k != "dur"))
PY
  find $OUT/$name -name '*.csv' -size +4M -delete; find $OUT/$name -name '*.db' -delete
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
run sq3 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
