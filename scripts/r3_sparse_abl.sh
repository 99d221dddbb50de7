#!/bin/bash
set -u
for abl in 0 3 11 19 27; do
  echo "NMFK_SPB_ABLATE=$abl"
  NMFK_STREAMS=1 NMFK_SP_UBLK=1 NMFK_SPB_ABLATE=$abl timeout -k 10 200 python3 scripts/bench_sparse.py 30 32 16 17 2>&1 | grep -E "step|units" | cut -c1-100 || exit 1
done
