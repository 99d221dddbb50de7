#!/bin/bash
# A/B of two builds of the library (round 5): usage r5_ab.sh <other.so> ; the bench sweep and the shares of 2 / 4 / 8 GPUs, fixed budget
cd $(dirname $0)/..
OTHER=$PWD/nmfk.jl_amd/$1
for R in 32 8 4 1; do
  for rep in 1 2; do
    echo -n "k 2:16 x $R  this build: "; timeout -k 10 200 python scripts/microbench.py 400 2 16 $R | sed "s/^default *//"
    echo -n "k 2:16 x $R  $1: "; NMFK_HIP_LIB=$OTHER timeout -k 10 200 python scripts/microbench.py 400 2 16 $R | sed "s#^/.*\.so *##"
  done
done
