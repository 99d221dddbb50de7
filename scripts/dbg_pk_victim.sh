#!/bin/bash
# scratch/pk_victim.hip beside scratch/burner.hip mode 0 (bf16 MFMA 16x16x32 in registers, another process), then alone
cd $(dirname $0)/..
echo "== beside the bf16 MFMA burner"
timeout -k 5 170 ./scratch/burner 0 ${SECS:-100} &
BURN=$!
sleep 3
timeout -k 5 160 ./scratch/pk_victim ${REPS:-200} $ONLY
kill $BURN 2>/dev/null; wait $BURN
echo "== alone"
timeout -k 5 100 ./scratch/pk_victim 60 $ONLY
