#!/bin/bash
# A/B of library builds in ONE box: tools/ab.sh "lib1.so lib2.so ..." [rounds] [ab_mcm.py arguments]
#   libraries live under gpurun_ab/ (git-ignored, travels with gpurun): make -C vpt_amd/csrc OUT=../../gpurun_ab/x.so EXTRA=-D...
libs=$1; rounds=${2:-2}; shift; shift
for round in $(seq $rounds); do
  for l in $libs; do
    python3 tools/ab_mcm.py --lib gpurun_ab/$l.so --tag $l "$@" || exit 1
  done
done
