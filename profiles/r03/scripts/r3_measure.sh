#!/bin/bash
# Round 3: per-iteration cost of the bench sweep and of its parts under the schedule switches of nmfk_api.hip (read_tuning).
cd $(dirname $0)/..
IT=${IT:-300}
run() { echo "== $1"; shift; env "$@" timeout -k 10 150 python scripts/microbench.py $IT $KR 32; }
KR="2 16"
run "default" A=1
run "serial streams" NMFK_STREAMS=1
run "no resident form" NMFK_HYB_RES=0
[ -n "${QUICK:-}" ] && exit 0
run "round-2 schedule" NMFK_HYB_SMALL=0 NMFK_HYB_RES=0
for KR in "2 4" "5 8" "9 16" "13 16"; do
  run "k $KR alone, serial" NMFK_STREAMS=1
done
