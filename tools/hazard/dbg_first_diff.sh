#!/bin/bash
# usage: dbg_first_diff.sh <burner mode> <label> [ENV=... for the checker]   (checker: tools/hazard/dbg_first_diff.py)
# The checker computes its reference alone, then the burner starts (another process) and the repetitions begin.
cd $(dirname $0)/../..
for b in burner pk_victim cu_map uniform_vload; do [ -x scratch/$b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scratch/$b.hip -o scratch/$b 2>/dev/null; done
# (to see the hazard again: build the library from a commit before the broadcast-first operand rule, or with the rule reverted)
mode=$1; label=$2; shift 2
echo "== $label  (burner mode $mode; $*)"
hs=/tmp/nmfk_hs_$$; rm -f $hs.ref $hs.go
env "$@" HANDSHAKE=$hs timeout -k 5 150 python tools/hazard/dbg_first_diff.py ${REPS:-200} ${RCHK:-4} > $hs.out 2>&1 &
CHK=$!
for i in $(seq 1 400); do [ -e $hs.ref ] && break; sleep 0.25; done
timeout -k 5 120 ./tools/hazard/burner $mode ${SECS:-40} > /dev/null &
BURN=$!
sleep 3; touch $hs.go
wait $CHK; tail -${TAIL:-4} $hs.out
kill $BURN 2>/dev/null; wait $BURN 2>/dev/null; rm -f $hs.ref $hs.go $hs.out
