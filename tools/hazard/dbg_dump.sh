#!/bin/bash
cd $(dirname $0)/../..
for b in burner pk_victim cu_map uniform_vload; do [ -x scratch/$b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scratch/$b.hip -o scratch/$b 2>/dev/null; done
hs=/tmp/nmfk_hs_$$; rm -f $hs.ref $hs.go
NMFK_HIP_LIB=$PWD/nmfk.jl_amd/libnmfk_hip_dbg.so NMFK_HYB=0 NMFK_MERGE=1 HANDSHAKE=$hs timeout -k 5 150 python tools/hazard/dbg_dump.py ${REPS:-40} > $hs.out 2>&1 &
CHK=$!
for i in $(seq 1 400); do [ -e $hs.ref ] && break; sleep 0.25; done
timeout -k 5 120 ./tools/hazard/burner ${MODE:-0} ${SECS:-30} > /dev/null &
BURN=$!
sleep 3; touch $hs.go
wait $CHK; cat $hs.out
kill $BURN 2>/dev/null; wait $BURN 2>/dev/null; rm -f $hs.ref $hs.go $hs.out
