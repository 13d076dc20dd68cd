#!/bin/bash
# A/B of the libraries named in $1 (under gpurun_ab/) on the headline frame and on rank 3 of 8's share
mkdir -p gpurun_out/r03
O=gpurun_out/r03/ab_$(date +%H%M%S).txt
for round in 1 2; do for l in $1; do
  python3 tools/ab_mcm.py --lib gpurun_ab/$l.so --tag $l --split 2 >> $O 2>&1
  python3 tools/ab_mcm.py --lib gpurun_ab/$l.so --tag $l --split 2 --shard 3,8,8 --frames 1000 >> $O 2>&1
done; done
for l in $1; do
  python3 tools/ab_mcm.py --lib gpurun_ab/$l.so --tag $l --split 2 --fast 0 >> $O 2>&1
  python3 tools/ab_mcm.py --lib gpurun_ab/$l.so --tag $l --split 2 --fast 0 --shard 3,8,8 --frames 1000 >> $O 2>&1
  python3 tools/ab_mcm.py --lib gpurun_ab/$l.so --tag $l --split 1 --classes 0 >> $O 2>&1
done
grep -v amdgpu.ids $O
