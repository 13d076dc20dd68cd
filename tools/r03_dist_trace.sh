#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03/trace_dist; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --force-dist 1 --gather torch --height 136 --steps 200 --repeats 2 --warmup-seconds 0.05 --other-configs 0 --cpu-baseline 0 --stream-probe 0 --check 0 > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
python3 tools/kernel_timeline.py $OUT/kt 40 "" > $OUT/timeline.txt 2>&1; tail -48 $OUT/timeline.txt | cut -c1-150
rm -rf $OUT/kt
