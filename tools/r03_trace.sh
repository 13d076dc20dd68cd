#!/bin/bash
# kernel-trace of the default bench line (classified, three streams) + timeline of the last frames
set -o pipefail
OUT=gpurun_out/r03/trace_${1:-default}
shift
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --steps 100 --warmup 10 --repeats 2 --warmup-seconds 0.05 "$@" > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
tail -1 $OUT/kt.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['frac'])"
python3 tools/kernel_timeline.py $OUT/kt 30 k_mcm > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
python3 tools/kernel_stats.py $OUT/kt > $OUT/kernel_stats.csv 2>&1; head -8 $OUT/kernel_stats.csv
rm -rf $OUT/kt
