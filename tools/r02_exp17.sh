#!/bin/bash
# which tile-row ranges on which streams: VPT_SPLIT_PLAN experiments (fast variant, headline workload)
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp17; mkdir -p "$out"
export TMPDIR=/tmp
B="timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 --fast-math 1"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us")'
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
cp gpurun_ab/lib_PLAN.so vpt_amd/libvpt_hip.so
for round in 1 2; do
  while read -r k plan; do
    VPT_SPLIT_PLAN="$plan" $B --split-streams $k 2>/dev/null | python3 -c "$P" "K=$k plan=$plan" | tee -a "$out/ab.txt"
  done <<'PLANS'
3 
3 0:0-195,1:195-805,2:805-1000
2 0:195-805,1:0-195,1:805-1000
3 0:0-250,1:250-750,2:750-1000
3 0:0-400,1:400-600,2:600-1000
3 0:0-167,1:167-333,2:333-500,0:500-667,1:667-833,2:833-1000
3 0:0-167,0:167-333,1:333-500,1:500-667,2:667-833,2:833-1000
3 0:0-500,1:500-1000,2:0-0
3 0:0-300,1:300-500,2:500-700,0:700-1000
3 0:0-195,1:195-500,2:500-805,0:805-1000
PLANS
done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
