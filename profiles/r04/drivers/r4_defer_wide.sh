#!/bin/bash
# Deferred check at the BASELINE configs[4] shape (65536 x 2048, 8 restarts, 40 iterations incl. 4 checks): k = 64, 48, 40, 24
for k in 64 48 40 24; do
  for mode in 0 1; do
    echo -n "NMFK_DEFER_OBJ=$mode  "
    NMFK_DEFER_OBJ=$mode python3 scripts/microbench_cfg5.py 40 8 $k
  done
done
