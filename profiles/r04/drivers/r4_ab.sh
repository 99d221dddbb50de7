#!/bin/bash
# A/B of two builds of the library on the bench sweep and its rank classes (scripts/microbench.py): usage r4_ab.sh <other.so> [reps]
cd $(dirname $0)/..
OTHER=$PWD/nmfk.jl_amd/$1
for range in "2 16" "13 16" "5 8" "2 4"; do
  for rep in 1 2; do
    echo -n "k $range x 32 default:   "; timeout -k 10 120 python scripts/microbench.py 300 $range 32 | sed "s/^default *//"
    echo -n "k $range x 32 $1: "; NMFK_HIP_LIB=$OTHER timeout -k 10 120 python scripts/microbench.py 300 $range 32 | sed "s#^/.*\.so *##"
  done
done
