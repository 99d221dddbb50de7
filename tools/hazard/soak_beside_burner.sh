#!/bin/bash
# usage: soak_beside_burner.sh [burner mode, default 0 = v_mfma_f32_16x16x32_bf16] [seconds]
cd $(dirname $0)/../..
[ -x tools/hazard/burner ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/hazard/burner.hip -o tools/hazard/burner 2>/dev/null
hs=/tmp/nmfk_hs_$$; rm -f $hs.ref $hs.go
HANDSHAKE=$hs timeout -k 10 ${LIMIT:-500} python tools/hazard/soak_beside_burner.py > $hs.out 2>&1 &
CHK=$!
for i in $(seq 1 1200); do [ -e $hs.ref ] && break; kill -0 $CHK 2>/dev/null || break; sleep 0.25; done
timeout -k 5 ${LIMIT:-500} ./tools/hazard/burner ${1:-0} ${2:-400} &
BURN=$!
sleep 3; touch $hs.go
wait $CHK; cat $hs.out
kill $BURN 2>/dev/null; wait $BURN 2>/dev/null; rm -f $hs.ref $hs.go $hs.out
