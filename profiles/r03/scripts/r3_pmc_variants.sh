#!/bin/bash
# Round 3: SQ / LDS / TCC counters of the kernel variants of the split-operand MFMA half-step, each variant alone on the GPU
# (serial streams), 60 iterations.  Separate --pmc passes, no tracing domains.  usage (GPU box): bash scripts/r3_pmc_variants.sh [tag]
set -u
TAG=${1:-r03}
ROOTD=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTD/gpurun_out/$TAG/pmc_variants
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp NMFK_STREAMS=1
cd /tmp
for range in ${RANGES:-"2 16"}; do
  t=$(echo $range | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/a_$t -- python3 $ROOTD/scripts/microbench.py 60 $range 32 > /dev/null 2> $OUT/a_$t.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/b_$t -- python3 $ROOTD/scripts/microbench.py 60 $range 32 > /dev/null 2> $OUT/b_$t.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/c_$t -- python3 $ROOTD/scripts/microbench.py 60 $range 32 > /dev/null 2> $OUT/c_$t.err
  rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/d_$t -- python3 $ROOTD/scripts/microbench.py 60 $range 32 > /dev/null 2> $OUT/d_$t.err
done
cd $ROOTD
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, os
out = sys.argv[1]
for sub in sorted(glob.glob(os.path.join(out, "*_*"))):
    if not os.path.isdir(sub): continue
    fs = glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        print(os.path.basename(sub), "no file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        nm = r["Kernel_Name"]
        if "hyb_" not in nm: continue
        short = nm[nm.index("hyb_"):].split("(")[0]
        agg[short][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[short][r["Counter_Name"]] += 1
    for k in sorted(agg):
        for c in sorted(agg[k]): print(os.path.basename(sub), k, c, f"mean per launch {agg[k][c] / cnt[k][c]:.6g} (n={cnt[k][c]})")
PY
find $OUT -name '*.csv' -size +2M -delete; find $OUT -name '*.db' -delete
