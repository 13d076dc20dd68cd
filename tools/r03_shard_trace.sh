#!/bin/bash
# kernel timeline of one rank's share (3 of 8) of the headline frame, tile classes on two streams
set -o pipefail
OUT=gpurun_out/r03/trace_shard; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 tools/ab_mcm.py --shard 3,8,8 --split ${1:-2} --frames 200 --blocks 2 > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
grep -v amdgpu $OUT/kt.log | tail -2
python3 tools/kernel_timeline.py $OUT/kt 24 k_mcm > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
rm -rf $OUT/kt
